"""Build libmagi_hip.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and by hand:
    python -m magi_v2_amd.build
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmagi_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "magi_hip.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def isa_path(obj):
    """The device ISA hipcc leaves beside an object compiled with --save-temps=obj (<dir>/<stem>-hip-amdgcn-amd-amdhsa-gfx950.s)."""
    stem = os.path.basename(obj)
    stem = stem[:-2] if stem.endswith(".o") else stem           # leap.hip.o -> leap.hip
    stem = stem[:-4] if stem.endswith(".hip") else stem         # -> leap
    return os.path.join(os.path.dirname(obj), stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def compile_checked(jobs, verbose=True):
    """jobs: [(command without -o/--save-temps, source, object)].  Compiles them in parallel keeping each unit's device ISA, runs the
    EXEC-prologue check (isa_check.py: the round-3 miscompile put live-range copies in front of a join block's EXEC restore)
    and recompiles a flagged unit with -mllvm -enable-ipra=false -- one of the switches that removed the pattern -- before giving up."""
    from . import isa_check

    def launch(cmd, src, obj, extra):
        full = cmd + extra + ["--save-temps=obj", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(full), flush=True)
        return full, subprocess.Popen(full, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)

    running = [(job, launch(*job, [])) for job in jobs]
    for (cmd, src, obj), (full, p) in running:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(full) + "\n" + out.decode(errors="replace")[-4000:])
        hits = isa_check.check_file(isa_path(obj)) if os.path.exists(isa_path(obj)) else None
        if hits is None:
            raise RuntimeError(f"no device ISA beside {obj}: the EXEC-prologue check cannot run")
        if hits:
            print(f"[magi build] {os.path.basename(src)}: instructions in front of an EXEC restore -- recompiling with -mllvm -enable-ipra=false\n"
                  + isa_check.report(isa_path(obj), hits), flush=True)
            full, p = launch(cmd, src, obj, ["-mllvm", "-enable-ipra=false"])
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("hipcc failed: " + " ".join(full) + "\n" + out.decode(errors="replace")[-4000:])
            hits = isa_check.check_file(isa_path(obj))
            if hits:
                raise RuntimeError("hipcc placed vector instructions in front of a join block's EXEC restore (they run under the narrowed mask of the "
                                   "skipped region: wrong results, DESIGN.md 4.2), with and without IPRA:\n" + isa_check.report(isa_path(obj), hits))
        stem = os.path.basename(isa_path(obj)).rsplit("-hip-", 1)[0]
        for junk in glob.glob(os.path.join(os.path.dirname(obj), stem + "-*")) + glob.glob(os.path.join(os.path.dirname(obj), stem + ".hip-*")):
            if junk != isa_path(obj):            # keep the device ISA (checked above), drop the other intermediate files of --save-temps
                os.remove(junk)


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "magi_hip.h")]
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(hh) for hh in hdrs)):
            continue
        # -ffp-contract=on: fused multiply-adds are formed per source expression (front end), not by the optimiser, so
        # every instantiation of a kernel (1, 2, 4 chains per matrix pass) rounds a chain's arithmetic identically
        contract = [] if os.path.basename(src) == "build.hip" else ["-ffp-contract=on"]     # (the matrix build keeps the default)
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function"] + contract + \
            os.environ.get("MAGI_EXTRA_CFLAGS", "").split()
        jobs.append((cmd, src, obj))
    compile_checked(jobs, verbose)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
