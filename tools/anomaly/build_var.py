"""scratch: python build_var.py <name> <opt> [flags...]  ->  var_<name>.so"""
import glob, os, subprocess, sys
name, opt, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
os.makedirs("obj", exist_ok=True)
objs = []
procs = []
for src in sorted(glob.glob("magi_v2_amd/csrc/*.hip")):
    base = os.path.basename(src)
    if base == "leap.hip":
        os.makedirs(f"obj/{name}", exist_ok=True)
        obj = f"obj/{name}/{base}.o"
        cmd = ["/opt/rocm/bin/hipcc", opt, "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function", "-ffp-contract=on", f"--save-temps=obj"] + flags + ["-c", src, "-o", obj]
    else:
        obj = f"obj/base_{base}.o"
        if os.path.exists(obj):
            objs.append(obj); continue
        contract = [] if base == "build.hip" else ["-ffp-contract=on"]
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function"] + contract + ["-c", src, "-o", obj]
    objs.append(obj)
    procs.append(subprocess.Popen(cmd))
for p in procs:
    assert p.wait() == 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", f"var_{name}.so"] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
print(f"var_{name}.so")
