#!/bin/bash
# Reproduction of the round-3 wrong-energy anomaly (DESIGN.md 4.2 "The round-3 anomaly, diagnosed"; profiles/r04_anomaly_variants.txt).
#   bash tools/anomaly/setup.sh          (from the repo root; writes only under build_variants/anomaly, which is git-ignored)
# Exports commit 1651a8f, puts the k_stream_sep mirror pass of decide_block behind a RUN-TIME condition (macro MAGI_ANOMALY_COND; the committed
# source had turned it into `if constexpr (SEPK)`), adds two debug exports (control block, state vectors) and builds the variants; then, on a GPU box:
#   cd build_variants/anomaly && for v in good bad badO1 badnoipra badnosv badinl; do MAGI_HIP_LIB=$PWD/var_$v.so python anom2.py; done; python anom_cmp.py bad badnoipra badnosv badinl
# and on any box:   python tools/check_exec_prologue.py build_variants/anomaly/obj/<variant>/leap-hip-amdgcn-amd-amdhsa-gfx950.s
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$ROOT/build_variants/anomaly
mkdir -p "$W" && git -C "$ROOT" archive 1651a8f | tar -x -C "$W"
cp "$ROOT"/tools/anomaly/{build_var.py,anom.py,anom2.py,anom_cmp.py} "$W"/
cd "$W"
python - <<'PY'
def sub(path, old, new):
    s = open(path).read(); assert s.count(old) == 1, (path, old[:50]); open(path, "w").write(s.replace(old, new))
sub("magi_v2_amd/csrc/decide.h", "        if constexpr (SEPK) {", "        if (MAGI_ANOMALY_COND) {")
sub("magi_v2_amd/csrc/decide.h", "template <int DRIFT, bool SEPK = false>", "#ifndef MAGI_ANOMALY_COND\n#define MAGI_ANOMALY_COND SEPK\n#endif\ntemplate <int DRIFT, bool SEPK = false>")
for fn in ("m_log", "m_exp", "m_log1p"):
    sub("magi_v2_amd/csrc/magi_internal.h", f"static __device__ __noinline__ double {fn}(double x)", f"static __device__ MAGI_TRANSC_INLINE double {fn}(double x)")
sub("magi_v2_amd/csrc/magi_internal.h", "// scalar helpers.  fp64", "#ifndef MAGI_TRANSC_INLINE\n#define MAGI_TRANSC_INLINE __noinline__\n#endif\n// scalar helpers.  fp64")
sub("magi_v2_amd/csrc/capi.hip", "int magi_debug_par(magi_handle* h, int chain, double* out64) {", '''int magi_debug_vec(magi_handle* h, int chain, int slot, double* out) {
    if (!h || !out || chain < 0 || chain >= h->n_chains || slot < 0 || slot >= V_COUNT) return MAGI_E_BADARG;
    MAGI_HIP_CHECK(h, hipMemcpy(out, h->ch.vec + vec_off(h->pb, chain, slot), sizeof(double) * h->pb.dimp, hipMemcpyDeviceToHost));
    return h->pb.dimp;
}
int magi_debug_ctl(magi_handle* h, int chain, double* out) {
    if (!h || !out || chain < 0 || chain >= h->n_chains) return MAGI_E_BADARG;
    ChainCtl c;
    MAGI_HIP_CHECK(h, hipMemcpy(&c, h->ch.ctl + chain, sizeof(ChainCtl), hipMemcpyDeviceToHost));
    const double v[20] = {c.init_energy, c.cand_L, c.cand_energy, c.L_cur, c.e_sum, c.e_sum_sub, c.sub_L, c.sub_energy, c.eps, c.beta_k, c.beta_cache,
                          (double)c.lf_count, (double)c.is_accepted, c.cand_bfac, c.LL, c.LR, c.bfacL, c.bfacR, (double)c.phase, (double)c.k};
    for (int i = 0; i < 20; ++i) out[i] = v[i];
    return MAGI_OK;
}

int magi_debug_par(magi_handle* h, int chain, double* out64) {''')
PY
B='-DMAGI_ANOMALY_COND=(ch.sep!=0)'
python build_var.py good -O3
python build_var.py bad -O3 "$B"
python build_var.py badO2 -O2 "$B"
python build_var.py badO1 -O1 "$B"
python build_var.py badzero -O3 "$B" -ftrivial-auto-var-init=zero
python build_var.py badpat -O3 "$B" -ftrivial-auto-var-init=pattern
python build_var.py badnoipra -O3 "$B" -mllvm -enable-ipra=false
python build_var.py badnosv -O3 "$B" -mllvm -amdgpu-spill-sgpr-to-vgpr=false
python build_var.py badinl -O3 "$B" -DMAGI_TRANSC_INLINE=__forceinline__
echo "variants built under $W"
