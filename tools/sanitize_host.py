"""Host-side sanitizer run (SURVEY section 5 "race detection / sanitizers"; GPU AddressSanitizer is not available on this pool, so this
covers the HOST half of the C ABI): csrc/capi.hip -- handle lifetime, argument validation, error strings, option parsing -- compiled
with -fsanitize=address,undefined for the host (device code untouched: -fno-gpu-sanitize), linked with the regular objects of the other
translation units, and driven through every entry point that is reachable WITHOUT a GPU: creation on a machine with no device, every
int-returning function on a NULL handle, the out-parameter helpers.  On a GPU box the same driver goes on through a small build /
log-posterior / sampler round trip (pass --gpu).

    python tools/sanitize_host.py [--gpu]        -> exit code 0 and "sanitizer run clean" when ASan / UBSan reported nothing
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build_variants", "san")
LIB = os.path.join(OUT, "libmagi_hip_san.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SAN = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def build():
    from magi_v2_amd import build as b
    b.build_lib(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(b.CSRC, "capi.hip")
    obj = os.path.join(OUT, "capi.san.o")
    deps = [src, os.path.join(b.CSRC, "magi_internal.h"), os.path.join(ROOT, "include", "magi_hip.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call([HIPCC, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=on", "-Wno-unused-function"] + SAN +
                              ["-c", src, "-o", obj])
        others = [os.path.join(b.HERE, "build", os.path.basename(s) + ".o") for s in b.sources() if not s.endswith("capi.hip")]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-shared-libsan"] + SAN + ["-o", LIB, obj] + others + ["-Wl,-rpath,/opt/rocm/lib"])
    return LIB


def asan_runtime():
    return subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()


def drive(gpu: bool):
    """Runs INSIDE the sanitized process (LD_PRELOAD = the ASan runtime)."""
    from magi_v2_amd import engine as E
    lib = E.load_library(LIB)
    cfg = E.SamplerCfg()
    lib.magi_sampler_cfg_default(C.byref(cfg))
    lib.magi_sampler_cfg_default(None)
    assert cfg.num_results == 1000 and cfg.max_tree_depth == 10
    assert lib.magi_version().startswith(b"magi_hip")
    d, p = C.c_int32(0), C.c_int32(0)
    assert lib.magi_user_drift_info(C.byref(d), C.byref(p)) == 0 and lib.magi_user_drift_info(None, None) == 0
    # every int-returning entry point refuses a NULL handle (MAGI_E_BADARG) without touching anything
    n_null = 0
    for name, (res, args) in E._SYMBOLS.items():
        if res is not C.c_int or not args or args[0] is not C.c_void_p:
            continue
        fn = getattr(lib, name)
        rc = fn(*([None] + [0 if a in (C.c_int, C.c_int64, C.c_uint64, C.c_double) else None for a in args[1:]]))
        assert rc == -1, (name, rc)
        n_null += 1
    lib.magi_destroy(None)
    assert lib.magi_last_error(None) is not None
    h = lib.magi_create(0)
    if not h:
        msg = lib.magi_last_error(None).decode()
        assert "HIP device" in msg or "device" in msg, msg
        assert not lib.magi_create(-1) and not lib.magi_create(10 ** 6)
        print(f"no GPU here: creation refused with '{msg}'; {n_null} entry points checked on a NULL handle")
        return
    lib.magi_destroy(h)
    if not gpu:
        print(f"{n_null} entry points checked on a NULL handle; handle created and destroyed")
        return
    # a GPU is present: argument validation and state errors on a live handle, then a small round trip
    import numpy as np
    from oracle import magi_oracle as orc
    from tests.util import engine_for, load_g4, problem_from_g4
    os.environ["MAGI_HIP_LIB"] = LIB
    g = load_g4("seir4_N81")
    pr = problem_from_g4(g, 20)
    eng = engine_for(pr, 20)
    for bad in (lambda: eng.set_option("no_such_option", 1), lambda: eng.set_option("potrf_panels", 99), lambda: eng.sampler_run(1),
                lambda: eng._check(eng._lib.magi_sampler_profile(eng._h, 1, None, None, None)),
                lambda: eng._check(eng._lib.magi_dense_apply(eng._h, 7, 0, 1, None, None))):
        try:
            bad()
            raise AssertionError("expected an error")
        except E.MagiHipError:
            pass
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
    eng.logpost_grad(X0, s0, t0, 1.0)
    eng.logpost_grad(X0, s0, t0, 1.0, fused=True)
    cfg = eng.default_cfg(num_results=3, num_burnin_steps=5)
    eng.sampler_init(cfg, np.repeat(X0[None], 3, 0), np.repeat(s0[None], 3, 0), np.repeat(t0[None], 3, 0), seed=5, chain_ids=[2, 3, 4])
    eng.sampler_run(4)
    ck = eng.sampler_checkpoint()
    eng.sampler_resume(cfg, ck, seed=5, chain_ids=[2, 3, 4])
    eng.sampler_run(4)
    eng.sampler_samples(); eng.sampler_diag(); eng.sampler_state(); eng.gradient_bytes(3); eng.stream_kernel_name(3)
    eng.build_matrices(g["I"], g["phi1s"], g["phi2s"], 2.01, bandsize=20, want_host=True)
    eng.close()
    print(f"{n_null} entry points checked on a NULL handle; live-handle error paths and a build / log-posterior / sampler / checkpoint round trip done")


if __name__ == "__main__":
    if "--drive" in sys.argv:
        drive("--gpu" in sys.argv)
        print("sanitizer run clean")
        sys.exit(0)
    build()
    env = dict(os.environ, LD_PRELOAD=asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--drive"] + (["--gpu"] if "--gpu" in sys.argv else []), env=env, cwd=ROOT,
                       capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr[-4000:])
    bad = r.returncode != 0 or "ERROR: AddressSanitizer" in r.stderr or "runtime error:" in r.stderr
    sys.exit(1 if bad else 0)
