"""Build libmagi_hip for a traced user drift (callers magi_v2.py:155, 206, 335).

The drift header emitted by magi_v2_amd.drift is written to ``jit_cache/<name>/user_drift.h`` and the library's
HIP sources are compiled with ``-DMAGI_USER_DRIFT_HEADER=...`` (hipcc, gfx950, one process per file).  Such a
build instantiates the sampler's kernels for that drift alone, so it costs about a third of the base build;
files that do not depend on the drift reuse the base build's objects.  Results are cached by content hash."""
import glob
import hashlib
import os
import shutil
import subprocess

from . import build as _build

CACHE = os.path.join(_build.HERE, "jit_cache")
_DRIFT_FREE = ("build.hip", "pack.hip")          # translation units without drift-dependent code


def _source_digest() -> str:
    h = hashlib.sha256()
    for f in sorted(_build.sources() + glob.glob(os.path.join(_build.CSRC, "*.h")) + [os.path.join(_build.HERE, "..", "include", "magi_hip.h")]):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def library_for(drift, verbose: bool = False) -> str:
    """Path of the specialised library for ``drift`` (a user Drift), building it if needed."""
    if drift.header is None:
        raise ValueError("built-in drifts use the base library")
    key = hashlib.sha256((drift.header + _source_digest()).encode()).hexdigest()[:16]
    d = os.path.join(CACHE, f"{drift.name}_{key}")
    lib = os.path.join(d, "libmagi_hip_user.so")
    if os.path.exists(lib):
        return lib
    for old in glob.glob(os.path.join(CACHE, f"{drift.name}_*")):      # builds of this drift against older sources
        if old != d:
            shutil.rmtree(old, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    hdr = os.path.join(d, "user_drift.h")
    with open(hdr, "w") as fh:
        fh.write(drift.header)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    wide = drift.D > 4 or drift.P > 6         # wider per-point lane groups / parameter blocks than the base build's
    objs, procs = [], []
    for src in _build.sources():
        base = os.path.basename(src)
        shared = os.path.join(_build.HERE, "build", base + ".o")
        if base in _DRIFT_FREE and not wide and os.path.exists(shared) and not _build.needs_build():
            objs.append(shared)
            continue
        obj = os.path.join(d, base + ".o")
        objs.append(obj)
        contract = [] if base == "build.hip" else ["-ffp-contract=on"]
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function",
               f'-DMAGI_USER_DRIFT_HEADER="{hdr}"'] + ((["-DMAGI_MAX_D=8"] if drift.D > 4 else []) + (["-DMAGI_MAX_P=8"] if drift.P > 6 else [])) + contract + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for the traced drift:\n" + out.decode(errors="replace")[-4000:])
    tmp = lib + ".tmp"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
    os.replace(tmp, lib)
    return lib
