"""Dev: kernel resource usage of csrc/leap.hip (hipcc -Rpass-analysis=kernel-resource-usage) + where in the ISA the scratch accesses sit.
    python tools/resource_usage.py > profiles/r03_kernel_resource_usage.txt"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir("/tmp")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=on", "-Rpass-analysis=kernel-resource-usage",
                    "-c", os.path.join(ROOT, "magi_v2_amd/csrc/leap.hip"), "-o", "/tmp/leap_ru.o", "--save-temps"], capture_output=True, text=True)
txt = r.stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
print("# hipcc -O3 --offload-arch=gfx950 -ffp-contract=on -Rpass-analysis=kernel-resource-usage magi_v2_amd/csrc/leap.hip  (ROCm 7.2)")
print("# kernel | VGPRs | VGPR spills | SGPR spills | scratch B/lane | occupancy waves/SIMD | LDS B/block")
KEYS = ["    VGPRs", "VGPRs Spill", "SGPRs Spill", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"]
for b in blocks:
    name = b.split(" [")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem); dem = re.sub(r"\(DevProblem.*|\(DevChains.*|\(HIP_vector.*", "", dem); dem = dem.replace("void ", "")
    vals = []
    for k in KEYS:
        m = re.search(k + r": (\d+)", b)
        vals.append(m.group(1) if m else "?")
    print("%-28s | %4s | %3s | %3s | %4s | %2s | %6s" % tuple([dem] + vals))
s = open("/tmp/leap-hip-amdgcn-amd-amdhsa-gfx950.s").read()
def body(sym):
    i = s.index("\n" + sym + ":"); j = s.index(".Lfunc_end", i); return s[i:j].split("\n")
print()
print("# Scratch (ISA of the same compile; line numbers inside each kernel's body).  Round 3: the state-sized boundary passes moved from the decision workgroup to the")
print("# point kernel (leap_point.h: boundary_block) and the stream kernels stopped spilling vector registers (56-116 per kernel before, all on the decision path).")
print("# Round 4: the last 64 B per lane of scratch are gone too -- 32 B were the zero tail of a `LeafPlan p{}` kept in memory by the aggregate copy `*plan_out = p`")
print("# (decide.h: plan_store writes field by field), 16 B the spill slot of a callee-saved VGPR in dual_averaging_eval (now a leaf: its log / exp / pow expand inside it).")
for sym, what in (("_ZN12_GLOBAL__N_112k_stream_sepILi1ELi8EEEv10DevProblem9DevChains13SamplerCfgDevi", "k_stream_sep<SEIR4, 8>"),
                  ("_ZN12_GLOBAL__N_18k_streamILi1ELi1EEEv10DevProblem9DevChains13SamplerCfgDevi", "k_stream<1, SEIR4>")):
    L = body(sym)
    # the streaming body: from the first burst of tile loads (>= 4 global_load_dwordx4 within 14 lines) to 300 lines behind the last burst
    idx = [i for i, l in enumerate(L) if "global_load_dwordx4" in l]
    runs = [i for i in idx if sum(1 for m in idx if i <= m < i + 14) >= 4]
    end = runs[-1] + 300          # (the last burst is followed by its steps' arithmetic and the partial-sum stores)
    sc = [i for i, l in enumerate(L) if "scratch_" in l]
    inside = [i for i in sc if runs[0] <= i <= end]
    print("%s: %d ISA lines; streaming body (first .. last burst of tile loads + 300 lines) = lines %d..%d; %d scratch accesses in the kernel, on lines %d..%d; %d of them inside the streaming body"
          % (what, len(L), runs[0], end, len(sc), sc[0] if sc else -1, sc[-1] if sc else -1, len(inside)))
