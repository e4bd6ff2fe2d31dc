"""Dev: build a variant library magi_v2_amd/libmagi_hip_<name>.so with an extra compiler flag (default -DMAGI_TAIL_STAMPS).
    python tools/build_stamps.py [-DFLAG] [name]"""
import os, subprocess, sys
sys.path.insert(0, ".")
import magi_v2_amd.build as b
flag = sys.argv[1] if len(sys.argv) > 1 else "-DMAGI_TAIL_STAMPS"
name = sys.argv[2] if len(sys.argv) > 2 else "stamps"
lib = os.path.join(b.HERE, "libmagi_hip_%s.so" % name)
od = os.path.join(b.HERE, "build_stamps", name); os.makedirs(od, exist_ok=True)
objs, procs = [], []
for src in b.sources():
    obj = os.path.join(od, os.path.basename(src) + ".o"); objs.append(obj)
    contract = [] if os.path.basename(src) == "build.hip" else ["-ffp-contract=on"]
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", flag] + contract + ["-c", src, "-o", obj]))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
print(lib)
