"""Build libmagi_hip for a traced user drift (callers magi_v2.py:155, 206, 335).

The drift header emitted by magi_v2_amd.drift is written to ``jit_cache/<name>/user_drift.h`` and the library's
HIP sources are compiled with ``-DMAGI_USER_DRIFT_HEADER=...`` (hipcc, gfx950, one process per file).  Such a
build instantiates the sampler's kernels for that drift alone, so it costs about a third of the base build;
files that do not depend on the drift reuse the base build's objects.  Results are cached by content hash."""
import fcntl
import glob
import hashlib
import logging
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
    """``hipcc --version`` ("" when the compiler cannot be run)."""
    try:
        return subprocess.run([_hipcc(), "--version"], capture_output=True, check=False).stdout.decode(errors="replace")
    except OSError:
        return ""


def _no_compiler_here() -> bool:
    """A deployment box that received a prebuilt ``jit_cache/``: HIPCC is not set and there is no hipcc at the default place or on PATH.
    (A HIPCC that is set but cannot be run is a configuration error, not "no compiler": nothing is trusted on its account.)"""
    return "HIPCC" not in os.environ and not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None


def _lib_name(compiler_version: str) -> str:
    """The library's file name carries a digest of the compiler that built it: libraries of two compilers live side by side in one
    cache directory (the directory key is compiler-free), nothing beside the library has to agree with it, and ``os.replace`` of the
    finished file is the one atomic publication."""
    return "libmagi_hip_user.%s.so" % hashlib.sha256(compiler_version.encode()).hexdigest()[:10]


def _find_cached_library(d: str):
    """The usable library in cache directory ``d`` (built from these very sources for this very drift header), or None:
    the one built by the compiler found here; with NO compiler here, or when the caller vouches for the cache with
    MAGI_JIT_CACHE_TRUST=1, the most recent one of any compiler (logged: its compiler is not verified)."""
    here = _compiler_version()
    if here:
        lib = os.path.join(d, _lib_name(here))
        if os.path.exists(lib):
            return lib
    trust = os.environ.get("MAGI_JIT_CACHE_TRUST") == "1"
    if trust or (not here and _no_compiler_here()):
        libs = sorted(glob.glob(os.path.join(d, "libmagi_hip_user.*.so")), key=os.path.getmtime, reverse=True)
        if libs:
            logging.getLogger(__name__).warning("using %s without verifying its compiler (%s)", libs[0],
                                                "MAGI_JIT_CACHE_TRUST=1" if trust else "no hipcc on this machine")
            return libs[0]
    return None


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
    key = hashlib.sha256((drift.header + _source_digest()).encode()).hexdigest()[:16]       # (the compiler is NOT in the key: it is in the library's file name)
    d = os.path.join(CACHE, f"{drift.name}_{key}")
    found = _find_cached_library(d)
    if found:
        return found
    version = _compiler_version()
    if not version:
        raise RuntimeError(f"no library for drift '{drift.name}' in {d} and {_hipcc()} cannot be run to build one")
    lib = os.path.join(d, _lib_name(version))
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(lib):                       # another process built it while we waited
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
                objs, jobs = [], []
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
                        contract + os.environ.get("MAGI_EXTRA_CFLAGS", "").split()
                    jobs.append((cmd, src, obj))
                try:
                    _build.compile_checked(jobs, verbose)          # (keeps each unit's ISA and runs the EXEC-prologue check on it, build.py)
                except RuntimeError as e:
                    raise RuntimeError("build for the traced drift failed:\n" + str(e)) from None
                tmp = os.path.join(work, "libmagi_hip_user.so")
                subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-Wl,-rpath,/opt/rocm/lib"])
                os.replace(tmp, lib)                      # (the one publication: the name says which compiler built it)
            finally:
                shutil.rmtree(work, ignore_errors=True)
            return lib
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
