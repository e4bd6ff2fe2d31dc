"""Build libmagi_hip for a traced user drift (callers magi_v2.py:155, 206, 335).

The drift header emitted by magi_v2_amd.drift is written to ``jit_cache/<name>/user_drift.h`` and the library's
HIP sources are compiled with ``-DMAGI_USER_DRIFT_HEADER=...`` (hipcc, gfx950, one process per file).  Such a
build instantiates the sampler's kernels for that drift alone, so it costs about a third of the base build;
files that do not depend on the drift reuse the base build's objects.  Results are cached by content hash."""
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import tempfile

from . import build as _build

# MAGI_JIT_CACHE relocates the cache (e.g. when the package directory is read-only)
CACHE = os.environ.get("MAGI_JIT_CACHE") or os.path.join(_build.HERE, "jit_cache")
_DRIFT_FREE = ("build.hip", "pack.hip")          # translation units without drift-dependent code


def _hipcc() -> str:
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _source_digest() -> str:
    """Sources + extra flags: what a cached library depends on besides the drift header and the compiler."""
    h = hashlib.sha256()
    for f in sorted(_build.sources() + glob.glob(os.path.join(_build.CSRC, "*.h")) + [os.path.join(_build.HERE, "..", "include", "magi_hip.h")]):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(os.environ.get("MAGI_EXTRA_CFLAGS", "").encode())
    return h.hexdigest()[:12]


def _compiler_version() -> str:
    """``hipcc --version`` ("" when there is no compiler on this machine)."""
    try:
        return subprocess.run([_hipcc(), "--version"], capture_output=True, check=False).stdout.decode(errors="replace")
    except OSError:
        return ""


def _cached_library_usable(d: str, lib: str) -> bool:
    """A library built from these very sources for this very drift header exists: use it when it was built by the compiler found
    here, when there is NO compiler here (a deployment box that received a prebuilt ``jit_cache/``), or when the caller vouches for
    it with MAGI_JIT_CACHE_TRUST=1; otherwise it is rebuilt (same sources, another compiler)."""
    if not os.path.exists(lib):
        return False
    if os.environ.get("MAGI_JIT_CACHE_TRUST") == "1":
        return True
    here = _compiler_version()
    if not here:
        return True
    try:
        with open(os.path.join(d, "compiler.txt")) as fh:
            return fh.read() == here
    except OSError:
        return False


def prune(keep_latest: int = 1) -> None:
    """Explicit maintenance: per drift name keep the ``keep_latest`` most recently built libraries.  (library_for never deletes:
    another process -- an older checkout sharing the cache -- may have a sibling build loaded.)"""
    by_name = {}
    for d in glob.glob(os.path.join(CACHE, "*_*")):
        if os.path.isdir(d):
            by_name.setdefault(os.path.basename(d).rsplit("_", 1)[0], []).append(d)
    for dirs in by_name.values():
        for d in sorted(dirs, key=os.path.getmtime, reverse=True)[keep_latest:]:
            shutil.rmtree(d, ignore_errors=True)


def library_for(drift, verbose: bool = False) -> str:
    """Path of the specialised library for ``drift`` (a user Drift), building it if needed.

    Safe under concurrent callers (torchrun ranks, pytest-xdist workers constructing the same model): the check-build-publish
    sequence holds an exclusive ``flock`` on a per-key lock file, objects are compiled in a private temporary directory and the
    finished library is moved into place with one ``os.replace``."""
    if drift.header is None:
        raise ValueError("built-in drifts use the base library")
    key = hashlib.sha256((drift.header + _source_digest()).encode()).hexdigest()[:16]       # (the compiler is NOT in the key: see _cached_library_usable)
    d = os.path.join(CACHE, f"{drift.name}_{key}")
    lib = os.path.join(d, "libmagi_hip_user.so")
    if _cached_library_usable(d, lib):
        return lib
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if _cached_library_usable(d, lib):            # another process built it while we waited
                return lib
            hdr = os.path.join(d, "user_drift.h")
            tmp_hdr = hdr + f".{os.getpid()}.tmp"
            with open(tmp_hdr, "w") as fh:
                fh.write(drift.header)
            os.replace(tmp_hdr, hdr)
            work = tempfile.mkdtemp(prefix="build_", dir=d)
            try:
                hipcc = _hipcc()
                wide = drift.D > 4 or drift.P > 6         # wider per-point lane groups / parameter blocks than the base build's
                objs, procs = [], []
                for src in _build.sources():
                    base = os.path.basename(src)
                    shared = os.path.join(_build.HERE, "build", base + ".o")
                    if base in _DRIFT_FREE and not wide and os.path.exists(shared) and not _build.needs_build():
                        objs.append(shared)
                        continue
                    obj = os.path.join(work, base + ".o")
                    objs.append(obj)
                    contract = [] if base == "build.hip" else ["-ffp-contract=on"]
                    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function",
                           f'-DMAGI_USER_DRIFT_HEADER="{hdr}"'] + ((["-DMAGI_MAX_D=8"] if drift.D > 4 else []) + (["-DMAGI_MAX_P=8"] if drift.P > 6 else [])) + \
                        contract + os.environ.get("MAGI_EXTRA_CFLAGS", "").split() + ["-c", src, "-o", obj]
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
                for cmd, p in procs:
                    out, _ = p.communicate()
                    if p.returncode != 0:
                        raise RuntimeError("hipcc failed for the traced drift:\n" + out.decode(errors="replace")[-4000:])
                tmp = os.path.join(work, "libmagi_hip_user.so")
                subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
                with open(os.path.join(d, f"compiler.txt.{os.getpid()}.tmp"), "w") as fh:
                    fh.write(_compiler_version())
                os.replace(os.path.join(d, f"compiler.txt.{os.getpid()}.tmp"), os.path.join(d, "compiler.txt"))
                os.replace(tmp, lib)
            finally:
                shutil.rmtree(work, ignore_errors=True)
            return lib
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
