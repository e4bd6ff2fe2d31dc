"""Dev: A/B variants of the library -- leap.hip recompiled with extra -D flags, the other objects reused from magi_v2_amd/build.
    python tools/build_variant.py <name> [-DFLAG ...]   ->  build_variants/<name>.so   (select with MAGI_HIP_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magi_v2_amd import build as b
b.build_lib()
name, flags = sys.argv[1], sys.argv[2:]
files = [a[5:] for a in flags if a.startswith("file=")] or ["leap.hip"]
flags = [a for a in flags if not a.startswith("file=")]
out = os.path.join(ROOT, "build_variants")
os.makedirs(out, exist_ok=True)
objs = []
for src in b.sources():
    base = os.path.basename(src)
    if base in files:
        obj = os.path.join(out, f"{name}_{base}.o")
        contract = [] if base == "build.hip" else ["-ffp-contract=on"]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function"] + contract + flags + ["-c", src, "-o", obj])
    else:
        obj = os.path.join(b.HERE, "build", base + ".o")
    objs.append(obj)
lib = os.path.join(out, f"{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
print(lib)
