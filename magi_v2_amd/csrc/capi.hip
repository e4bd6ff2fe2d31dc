// extern "C" entry points of libmagi_hip.so (see include/magi_hip.h for the contract).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "magi_internal.h"

thread_local std::string g_magi_last_error;

int magi_fail(magi_handle* h, int code, const std::string& msg) {
    g_magi_last_error = msg;
    if (h) h->err = msg;
    return code;
}

namespace {

// leapfrog slots per captured graph (even: a graph starts at slot parity 0).  MAGI_GRAPH_SLOTS overrides it for experiments.
int graph_slots() {
    static const int n = [] {
        const char* e = getenv("MAGI_GRAPH_SLOTS");
        int v = e ? atoi(e) : 128;     // 32 / 64 / 128 / 256 slots: 107.7 / 108.9 / 109.7 / 109.9 samples/s at config 2 (graph boundaries + control-block snapshot;
                                       // round 4, one box); a run wastes at most one graph of early-exit slots at its end
        v = std::max(2, std::min(v, 4096));
        return v & ~1;
    }();
    return n;
}
#define kGraphSlots graph_slots()

void free_dev(void* p) { if (p) (void)hipFree(p); }

void free_matrices(magi_handle* h) {
    free_dev(h->dCsym); free_dev(h->dM); free_dev(h->dMt); free_dev(h->dKsym); free_dev(h->dYobs);
    free_dev(h->dTiles); free_dev(h->dTasks);
    for (int k = 0; k < 3; ++k) { free_dev(h->dDense[k]); h->dDense[k] = nullptr; }
    h->dense_N = h->dense_D = 0;
    for (int k = 0; k < magi_handle::WS_COUNT; ++k) { free_dev(h->ws[k]); h->ws[k] = nullptr; h->ws_cap[k] = 0; }
    h->dCsym = h->dM = h->dMt = h->dKsym = h->dYobs = nullptr;
    h->dTiles = nullptr; h->dTasks = nullptr;
    h->tiles_cap = h->tasks_cap = 0;
    h->have_matrices = h->have_problem = false;
}

void free_chains(magi_handle* h) {
    DevChains& c = h->ch;
    free_dev(c.vec); free_dev(c.ctl); free_dev(c.par); free_dev(c.plan); free_dev(c.part); free_dev(c.tpart); free_dev(c.xop); free_dev(c.vop); free_dev(c.gctl); free_dev(c.samples);
    free_dev(c.d_step_size); free_dev(c.d_lar); free_dev(c.d_target); free_dev(c.d_energy); free_dev(c.d_beta);
    free_dev(c.d_leapfrogs); free_dev(c.d_depth); free_dev(c.d_flags);
    free_dev(h->d_chain_ids); free_dev(h->d_fin);
    c = DevChains{};
    h->d_chain_ids = nullptr; h->d_fin = nullptr;
    h->cap_chains = 0; h->n_chains = 0; h->samples_cap = 0; h->diag_cap = 0;
    h->sampler_ready = false;
}

void drop_graph(magi_handle* h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->graph_exec = nullptr; h->graph = nullptr; h->graph_valid = false;
}

// host <-> device state layout: host X[N][D] row-major  <->  device comp-major [D][N] | sig | th
void pack_state(const DevProblem& pb, const double* X, const double* sp, const double* tp, double* q) {
    for (int d = 0; d < pb.D; ++d)
        for (int i = 0; i < pb.N; ++i) q[(size_t)d * pb.N + i] = X[(size_t)i * pb.D + d];
    for (int d = 0; d < pb.D; ++d) q[pb.ND + d] = sp[d];
    for (int p = 0; p < pb.P; ++p) q[pb.ND + pb.D + p] = tp[p];
    for (int e = pb.dim; e < pb.dimp; ++e) q[e] = 0.0;
}

void unpack_state(const DevProblem& pb, const double* q, double scale, double* X, double* sp, double* tp) {
    if (X)
        for (int d = 0; d < pb.D; ++d)
            for (int i = 0; i < pb.N; ++i) X[(size_t)i * pb.D + d] = scale * q[(size_t)d * pb.N + i];
    if (sp) for (int d = 0; d < pb.D; ++d) sp[d] = scale * q[pb.ND + d];
    if (tp) for (int p = 0; p < pb.P; ++p) tp[p] = scale * q[pb.ND + pb.D + p];
}

int upload_states(magi_handle* h, int n, const double* X, const double* sp, const double* tp) {
    const DevProblem& pb = h->pb;
    std::vector<double> q((size_t)pb.dimp);
    for (int c = 0; c < n; ++c) {
        pack_state(pb, X + (size_t)c * pb.ND, sp + (size_t)c * pb.D, tp + (size_t)c * pb.P, q.data());
        for (int e = 0; e < pb.dim; ++e)
            if (std::isnan(q[e])) return magi_fail(h, MAGI_E_NAN, "NaN in the initial state of chain " + std::to_string(c));
        MAGI_HIP_CHECK(h, hipMemcpy(h->ch.vec + vec_off(pb, c, V_Q), q.data(), sizeof(double) * pb.dimp, hipMemcpyHostToDevice));
    }
    return MAGI_OK;
}

int build_graph(magi_handle* h) {
    drop_graph(h);
    MAGI_HIP_CHECK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = MAGI_OK;
    for (int s = 0; s < kGraphSlots && rc == MAGI_OK; ++s) {          // kGraphSlots is even: a graph starts at slot parity 0
        rc = magi_launch_stream(h, h->n_chains, s & 1, true, h->stream);
        if (rc == MAGI_OK) rc = magi_launch_point(h, h->n_chains, s & 1, h->stream);
    }
    hipError_t e = hipStreamEndCapture(h->stream, &h->graph);
    if (rc != MAGI_OK) return rc;
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    MAGI_HIP_CHECK(h, hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    h->graph_valid = true;
    h->graph_slots = kGraphSlots;
    return MAGI_OK;
}

}  // namespace

void magi_options_from_env(MagiOptions& o) {
    auto num = [](const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; };
    if (const char* e = getenv("MAGI_STREAM_FAMILY")) o.stream_family = std::string(e) == "mc" ? 1 : std::string(e) == "valu" ? 2 : 0;
    o.sep_pair_min = (int)num("MAGI_SEP_PAIR_MIN", o.sep_pair_min);
    o.fused_parity = num("MAGI_FUSED_PARITY", 0) == 1;
    o.gemm_remap_min = (int)num("MAGI_GEMM_REMAP_MIN", o.gemm_remap_min);
    o.family_chains = (int)num("MAGI_FAMILY_CHAINS", o.family_chains);
    o.potrf_panels = (int)num("MAGI_POTRF_PANELS", o.potrf_panels);
    o.potrf_lookahead_min = (int)num("MAGI_POTRF_LOOKAHEAD_MIN", o.potrf_lookahead_min);
    o.no_graph = getenv("MAGI_NO_GRAPH") != nullptr;
    o.fit_host_loop = getenv("MAGI_FIT_HOST_LOOP") != nullptr;
    o.fit_per_component = getenv("MAGI_FIT_PER_COMPONENT") != nullptr;
    o.build_profile = getenv("MAGI_BUILD_PROFILE") != nullptr;
    o.build_serial = getenv("MAGI_BUILD_SERIAL") != nullptr;
    // (slot_budget_graphs has no variable: a test hook must not be reachable from a job's environment)
}

double* magi_workspace(magi_handle* h, int k, size_t n) {
    if (n <= h->ws_cap[k]) return h->ws[k];
    if (h->ws[k]) (void)hipFree(h->ws[k]);
    h->ws[k] = nullptr; h->ws_cap[k] = 0;
    if (hipMalloc(&h->ws[k], n * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        h->ws[k] = nullptr;
        magi_fail(h, MAGI_E_HIP, "work space allocation failed (" + std::to_string(n * sizeof(double) >> 20) + " MiB)");
        return nullptr;
    }
    h->ws_cap[k] = n;
    return h->ws[k];
}

int magi_ensure_chains(magi_handle* h, int n) {
    if (!h->have_matrices || !h->have_problem) return magi_fail(h, MAGI_E_STATE, "set matrices and problem first");
    if (n <= 0 || n > 4096) return magi_fail(h, MAGI_E_BADARG, "n_chains out of range");
    if (n > h->cap_chains) {
        free_chains(h);
        const size_t vbytes = (size_t)n * V_COUNT * h->pb.dimp * sizeof(double);
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.vec, vbytes));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.vec, 0, vbytes));
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.ctl, sizeof(ChainCtl) * n));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.ctl, 0, sizeof(ChainCtl) * n));
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.par, sizeof(double) * PAR_COUNT * n));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.par, 0, sizeof(double) * PAR_COUNT * n));
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.plan, sizeof(LeafPlan) * 2 * n));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.plan, 0, sizeof(LeafPlan) * 2 * n));
        h->ch.n_wg = magi_leap_wgs(h->pb);
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.part, sizeof(double) * PART_K * h->ch.n_wg * n));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.part, 0, sizeof(double) * PART_K * h->ch.n_wg * n));
        const bool sep = magi_drift_separable(h->pb.drift);
        const size_t tpn = std::max(sep ? magi_sep_tpart_elems(h->pb, n) : (size_t)0, (size_t)n * 4 * h->pb.D * h->pb.nb * h->pb.Np);      // (either layout)
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.tpart, sizeof(double) * tpn));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.tpart, 0, sizeof(double) * tpn));     // slots outside the block band stay zero
        const size_t opn = (size_t)2 * ((n + 15) / 16) * h->pb.D * h->pb.Np * 16;
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.xop, sizeof(double) * opn));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.xop, 0, sizeof(double) * opn));
        h->vop_elems = sep ? magi_sep_vop_elems(h->pb, n) : 16;
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.vop, sizeof(double) * h->vop_elems));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.vop, 0, sizeof(double) * h->vop_elems));
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.gctl, sizeof(GlobalCtl)));
        MAGI_HIP_CHECK(h, hipMemset(h->ch.gctl, 0, sizeof(GlobalCtl)));
        MAGI_HIP_CHECK(h, hipMalloc(&h->d_chain_ids, sizeof(long long) * n));
        MAGI_HIP_CHECK(h, hipMalloc(&h->d_fin, sizeof(double) * 8 * n));
        h->cap_chains = n;
    }
    const bool fam = magi_stream_family_mc(h, n);
    if (h->n_chains != n || fam != h->family_mc) drop_graph(h);
    if (h->n_chains != n && h->ch.vop)       // the mirror's layout depends on the chain count: entries the new layout never writes must read zero
        MAGI_HIP_CHECK(h, hipMemsetAsync(h->ch.vop, 0, sizeof(double) * h->vop_elems, h->stream));
    h->n_chains = n;
    h->ch.n_chains = n;
    h->family_mc = fam;
    const bool sepk = fam && magi_drift_separable(h->pb.drift);
    if ((h->ch.sep != 0) != sepk && h->ch.tpart) {      // the two streaming paths lay tpart out differently: slots the new one never writes must read zero
        const size_t tpn = std::max(magi_drift_separable(h->pb.drift) ? magi_sep_tpart_elems(h->pb, h->cap_chains) : (size_t)0,
                                    (size_t)h->cap_chains * 4 * h->pb.D * h->pb.nb * h->pb.Np);
        MAGI_HIP_CHECK(h, hipMemsetAsync(h->ch.tpart, 0, sizeof(double) * tpn, h->stream));
    }
    h->ch.sep = sepk ? 1 : 0;
    h->ch.mc = (fam && !sepk) ? 1 : 0;
    return MAGI_OK;
}

extern "C" {

const char* magi_version(void) { return "magi_hip 0.1.0 gfx950"; }

int magi_user_drift_info(int* D, int* P) {
#ifdef MAGI_USER_DRIFT_HEADER
    if (D) *D = MAGI_USER_D;
    if (P) *P = MAGI_USER_P;
    return 1;
#else
    (void)D; (void)P;
    return 0;
#endif
}

const char* magi_last_error(const magi_handle* h) { return h ? h->err.c_str() : g_magi_last_error.c_str(); }

magi_handle* magi_create(int device_id) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_magi_last_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_magi_last_error = "device_id out of range";
        return nullptr;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_magi_last_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return nullptr;
    }
    magi_handle* h = new magi_handle();
    h->device = device_id;
    magi_options_from_env(h->opt);
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipHostMalloc((void**)&h->h_gctl, sizeof(GlobalCtl) * 4, hipHostMallocDefault)) != hipSuccess) {
        g_magi_last_error = std::string("handle setup: ") + hipGetErrorString(e);
        delete h;
        return nullptr;
    }
    for (int i = 0; i < 4; ++i) (void)hipEventCreateWithFlags(&h->ev[i], hipEventDisableTiming);
    (void)hipEventCreate(&h->ev_t0);
    (void)hipEventCreate(&h->ev_t1);
    return h;
}

void magi_destroy(magi_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    free_chains(h);
    free_matrices(h);
    for (int i = 0; i < 4; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->ev_t0) (void)hipEventDestroy(h->ev_t0);
    if (h->ev_t1) (void)hipEventDestroy(h->ev_t1);
    if (h->h_gctl) (void)hipHostFree(h->h_gctl);
    if (h->apply_pin) (void)hipHostFree(h->apply_pin);
    for (int i = 0; i < 3; ++i) if (h->ev_la[i]) (void)hipEventDestroy(h->ev_la[i]);
    if (h->stream_chain) (void)hipStreamDestroy(h->stream_chain);
    for (int i = 0; i < 4; ++i) if (h->ev_pw[i]) (void)hipEventDestroy(h->ev_pw[i]);
    if (h->stream_trail) (void)hipStreamDestroy(h->stream_trail);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int magi_set_matrices(magi_handle* h, int N, int D, int bandsize, const double* C_inv, const double* m,
                      const double* K_inv) {
    if (!h) return MAGI_E_BADARG;
    if (!C_inv || !m || !K_inv) return magi_fail(h, MAGI_E_BADARG, "null matrix pointer");
    if (N < 2 || D < 1 || D > MAGI_MAX_D) return magi_fail(h, MAGI_E_BADARG, "need N >= 2 and 1 <= D <= " + std::to_string(MAGI_MAX_D));
    (void)hipSetDevice(h->device);
    h->have_matrices = false;          // the packed operands describe the OLD dense stacks until the pack below has succeeded
    h->sampler_ready = false;
    const size_t bytes = (size_t)D * N * N * sizeof(double);
    int rc = magi_ensure_dense(h, N, D);
    if (rc) return rc;
    MAGI_HIP_CHECK(h, hipMemcpy(h->dDense[0], C_inv, bytes, hipMemcpyHostToDevice));
    MAGI_HIP_CHECK(h, hipMemcpy(h->dDense[1], m, bytes, hipMemcpyHostToDevice));
    MAGI_HIP_CHECK(h, hipMemcpy(h->dDense[2], K_inv, bytes, hipMemcpyHostToDevice));
    rc = magi_pack_matrices(h, N, D, bandsize, h->dDense[0], h->dDense[1], h->dDense[2]);
    (void)hipStreamSynchronize(h->stream);
    return rc;
}

int magi_build_dense(magi_handle* h, const double* I, int N, int D, int n_sel, const int32_t* sel, const double* phi1, const double* phi2, double nu) {
    if (!h) return MAGI_E_BADARG;
    if (!I || !sel || !phi1 || !phi2) return magi_fail(h, MAGI_E_BADARG, "null pointer");
    if (N < 2 || D < 1 || D > MAGI_MAX_D || n_sel < 1 || n_sel > D) return magi_fail(h, MAGI_E_BADARG, "need N >= 2, 1 <= D <= " + std::to_string(MAGI_MAX_D) + " and 1 <= n_sel <= D");
    if (!(nu > 1.0)) return magi_fail(h, MAGI_E_BADARG, "nu must exceed 1 (once-differentiable Matern)");
    (void)hipSetDevice(h->device);
    h->have_matrices = false;          // the packed operands no longer match the dense stacks until magi_pack_resident
    h->sampler_ready = false;
    std::vector<int> s(sel, sel + n_sel);
    return magi_build_dense_device(h, I, N, D, n_sel, s.data(), phi1, phi2, nu);
}

int magi_pack_resident(magi_handle* h, int bandsize) {
    if (!h) return MAGI_E_BADARG;
    if (!h->dDense[0] || h->dense_N <= 0) return magi_fail(h, MAGI_E_STATE, "no resident matrices: build or set them first");
    (void)hipSetDevice(h->device);
    int rc = magi_pack_matrices(h, h->dense_N, h->dense_D, bandsize, h->dDense[0], h->dDense[1], h->dDense[2]);
    (void)hipStreamSynchronize(h->stream);
    return rc;
}

int magi_get_dense(magi_handle* h, int bandsize, double* C_inv, double* m, double* K_inv) {
    if (!h) return MAGI_E_BADARG;
    if (!h->dDense[0] || h->dense_N <= 0) return magi_fail(h, MAGI_E_STATE, "no resident matrices");
    (void)hipSetDevice(h->device);
    const int N = h->dense_N, D = h->dense_D;
    const size_t nn = (size_t)N * N;
    double* outs[3] = {C_inv, m, K_inv};
    for (int k = 0; k < 3; ++k) {
        if (!outs[k]) continue;
        MAGI_HIP_CHECK(h, hipMemcpy(outs[k], h->dDense[k], nn * D * sizeof(double), hipMemcpyDeviceToHost));
        if (bandsize >= 0)           // tf.linalg.band_part(., b, b), magi_v2.py:271-274
            for (int d = 0; d < D; ++d)
                for (int i = 0; i < N; ++i) {
                    double* row = outs[k] + (size_t)d * nn + (size_t)i * N;
                    for (int j = 0; j < std::min(N, i - bandsize); ++j) row[j] = 0.0;
                    for (int j = std::max(0, i + bandsize + 1); j < N; ++j) row[j] = 0.0;
                }
    }
    return MAGI_OK;
}

int magi_dense_apply(magi_handle* h, int which, int transpose, int nv, const double* V, double* Y) {
    if (!h) return MAGI_E_BADARG;
    if (!h->dDense[0] || h->dense_N <= 0) return magi_fail(h, MAGI_E_STATE, "no resident matrices");
    if (which < 0 || which > 2 || nv < 1 || nv > 8 || !V || !Y) return magi_fail(h, MAGI_E_BADARG, "which in 0..2, 1 <= nv <= 8, non-null vectors");
    (void)hipSetDevice(h->device);
    // (called 2 x 10 000 times by the theta initialiser's general branch: device scratch and pinned staging are kept in the handle
    //  -- grow-only -- and the copies ride on the handle's stream: one synchronisation per call, no allocation)
    const size_t n = (size_t)h->dense_D * h->dense_N * nv;
    double* dV = magi_workspace(h, magi_handle::WS_APPLY, 2 * n);
    if (!dV) return MAGI_E_HIP;
    double* dY = dV + n;
    if (2 * n > h->apply_pin_cap) {
        if (h->apply_pin) (void)hipHostFree(h->apply_pin);
        h->apply_pin = nullptr; h->apply_pin_cap = 0;
        MAGI_HIP_CHECK(h, hipHostMalloc((void**)&h->apply_pin, 2 * n * sizeof(double), hipHostMallocDefault));
        h->apply_pin_cap = 2 * n;
    }
    std::memcpy(h->apply_pin, V, n * sizeof(double));
    MAGI_HIP_CHECK(h, hipMemcpyAsync(dV, h->apply_pin, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = magi_dense_apply_device(h, which, transpose, nv, dV, dY);
    if (rc) return rc;
    MAGI_HIP_CHECK(h, hipMemcpyAsync(h->apply_pin + n, dY, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    std::memcpy(Y, h->apply_pin + n, n * sizeof(double));
    return MAGI_OK;
}

int magi_theta_init(magi_handle* h, int drift_id, int P, const double* Xhat, const double* mu, int num_iters, double learning_rate,
                    double* theta, double* loss_trace) {
    if (!h) return MAGI_E_BADARG;
    if (!Xhat || !mu || !theta) return magi_fail(h, MAGI_E_BADARG, "null pointer");
    if (!h->dDense[0] || h->dense_N <= 0) return magi_fail(h, MAGI_E_STATE, "no resident matrices: build or set them first");
    if (num_iters < 0 || !(learning_rate > 0.0)) return magi_fail(h, MAGI_E_BADARG, "num_iters >= 0 and learning_rate > 0");
#ifdef MAGI_USER_DRIFT_HEADER
    if (drift_id != MAGI_DRIFT_USER) return magi_fail(h, MAGI_E_BADARG, "this library is specialised for a traced f_vec: drift id must be MAGI_DRIFT_USER");
#else
    if (drift_id < MAGI_DRIFT_SEIR3 || drift_id > MAGI_DRIFT_SIRW) return magi_fail(h, MAGI_E_BADARG, "unknown drift id (a traced f_vec needs its own library: magi_v2_amd.jit)");
#endif
    for (int i = 0; i < h->dense_N * h->dense_D; ++i)
        if (std::isnan(Xhat[i])) return magi_fail(h, MAGI_E_NAN, "NaN in Xhat");
    (void)hipSetDevice(h->device);
    return magi_theta_init_device(h, drift_id, P, Xhat, mu, num_iters, learning_rate, theta, loss_trace);
}

int magi_build_matrices(magi_handle* h, const double* I, int N, int D, const double* phi1, const double* phi2,
                        double nu, int bandsize, double* C_inv, double* m, double* K_inv) {
    if (!h) return MAGI_E_BADARG;
    if (!I || !phi1 || !phi2) return magi_fail(h, MAGI_E_BADARG, "null pointer");
    if (N < 2 || D < 1 || D > MAGI_MAX_D) return magi_fail(h, MAGI_E_BADARG, "need N >= 2 and 1 <= D <= " + std::to_string(MAGI_MAX_D));
    if (!(nu > 1.0)) return magi_fail(h, MAGI_E_BADARG, "nu must exceed 1 (once-differentiable Matern)");
    (void)hipSetDevice(h->device);
    return magi_build_matrices_device(h, I, N, D, phi1, phi2, nu, bandsize, C_inv, m, K_inv);
}

int magi_matern_blocks(magi_handle* h, const double* I, int N, double phi1, double phi2, double nu, double* Kappa,
                       double* p_Kappa, double* Kappa_pp) {
    if (!h) return MAGI_E_BADARG;
    if (!I || N < 2) return magi_fail(h, MAGI_E_BADARG, "bad grid");
    (void)hipSetDevice(h->device);
    return magi_matern_blocks_device(h, I, N, phi1, phi2, nu, Kappa, p_Kappa, Kappa_pp);
}

int magi_set_problem(magi_handle* h, const double* mu, const double* N_ds, const int64_t* obs_idx, const double* y,
                     int64_t n_obs, double beta, const double* LB, int drift_id, int P) {
    if (!h) return MAGI_E_BADARG;
    if (!h->have_matrices) return magi_fail(h, MAGI_E_STATE, "matrices must be set before the problem");
    if (!mu || !N_ds || !LB || (n_obs > 0 && (!obs_idx || !y))) return magi_fail(h, MAGI_E_BADARG, "null pointer");
    DevProblem& pb = h->pb;
    int needD, needP;
#ifdef MAGI_USER_DRIFT_HEADER
    if (drift_id != MAGI_DRIFT_USER)
        return magi_fail(h, MAGI_E_BADARG, "this library is specialised for a traced f_vec: drift id must be MAGI_DRIFT_USER");
    needD = MAGI_USER_D; needP = MAGI_USER_P;
#else
    switch (drift_id) {
    case MAGI_DRIFT_SEIR3: needD = 3; needP = 3; break;
    case MAGI_DRIFT_SEIR4: needD = 4; needP = 3; break;
    case MAGI_DRIFT_SIRW: needD = 4; needP = 5; break;
    case MAGI_DRIFT_USER: return magi_fail(h, MAGI_E_BADARG, "no user drift compiled into this library (magi_v2_amd.jit builds one)");
    default: return magi_fail(h, MAGI_E_BADARG, "unknown drift id");
    }
#endif
    if (pb.D != needD || P != needP)
        return magi_fail(h, MAGI_E_BADARG, "drift expects D=" + std::to_string(needD) + ", P=" + std::to_string(needP));
    (void)hipSetDevice(h->device);
    const int N = pb.N, D = pb.D;
    std::vector<double> yobs((size_t)N * D, std::nan(""));
    for (int64_t k = 0; k < n_obs; ++k) {
        const int64_t idx = obs_idx[k];
        if (idx < 0 || idx >= (int64_t)N * D) return magi_fail(h, MAGI_E_BADARG, "obs_idx out of range");
        const int i = (int)(idx / D), d = (int)(idx % D);          // flat row-major index into X[N][D]
        yobs[(size_t)d * N + i] = y[k];
    }
    free_dev(h->dYobs);
    h->dYobs = nullptr;
    MAGI_HIP_CHECK(h, hipMalloc(&h->dYobs, sizeof(double) * N * D));
    MAGI_HIP_CHECK(h, hipMemcpy(h->dYobs, yobs.data(), sizeof(double) * N * D, hipMemcpyHostToDevice));
    pb.yobs = h->dYobs;
    pb.P = P;
    pb.drift = drift_id;
    pb.dim = pb.ND + D + P;
    pb.dimp = (pb.dim + 7) & ~7;
    pb.beta_inv = 1.0 / beta;
    for (int d = 0; d < MAGI_MAX_D; ++d) {
        pb.mu[d] = d < D ? mu[d] : 0.0;
        pb.N_ds[d] = d < D ? N_ds[d] : 0.0;
        pb.LB[d] = d < D ? LB[d] : 0.0;
    }
    h->have_problem = true;
    free_chains(h);          // dimp may have changed
    drop_graph(h);
    return MAGI_OK;
}

static int logpost_grad_impl(magi_handle* h, bool fused, int n_chains, const double* X, const double* sig_pre, const double* th_pre,
                             double beta_temp, double* logp, double* gX, double* gsig, double* gth, double* terms) {
    if (!h) return MAGI_E_BADARG;
    if (!X || !sig_pre || !th_pre) return magi_fail(h, MAGI_E_BADARG, "null state pointer");
    (void)hipSetDevice(h->device);
    int rc = magi_ensure_chains(h, n_chains);
    if (rc) return rc;
    h->sampler_ready = false;
    const DevProblem& pb = h->pb;
    if ((rc = upload_states(h, n_chains, X, sig_pre, th_pre))) return rc;
    MAGI_HIP_CHECK(h, hipMemsetAsync(h->ch.gctl, 0, sizeof(GlobalCtl), h->stream));
    if ((rc = magi_launch_prepare(h, n_chains, h->stream))) return rc;
    if (fused) {
        // (option fused_parity = 1: evaluate as an ODD leapfrog slot does -- the streaming kernels walk their blocks backwards there and
        //  read the other halves of the plan ring and of the operand mirrors; tests compare the two)
        const int par = h->opt.fused_parity ? 1 : 0;
        if ((rc = magi_launch_plan_eval(h, n_chains, h->stream, par))) return rc;
        if ((rc = magi_launch_stream(h, n_chains, par, false, h->stream))) return rc;
        if ((rc = magi_launch_point(h, n_chains, par, h->stream))) return rc;
        if ((rc = magi_launch_leap_finalize(h, n_chains, h->d_fin, h->stream, par))) return rc;
    } else {
        if ((rc = magi_launch_gradient(h, n_chains, h->stream))) return rc;
        if ((rc = magi_launch_finalize(h, n_chains, h->d_fin, h->stream))) return rc;
    }
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    std::vector<double> fin((size_t)8 * n_chains), g((size_t)pb.dimp);
    MAGI_HIP_CHECK(h, hipMemcpy(fin.data(), h->d_fin, sizeof(double) * 8 * n_chains, hipMemcpyDeviceToHost));
    for (int c = 0; c < n_chains; ++c) {
        if (logp) logp[c] = beta_temp * fin[(size_t)c * 8];
        if (terms) for (int k = 0; k < 4; ++k) terms[(size_t)c * 4 + k] = fin[(size_t)c * 8 + 1 + k];
        if (gX || gsig || gth) {
            MAGI_HIP_CHECK(h, hipMemcpy(g.data(), h->ch.vec + vec_off(pb, c, V_G), sizeof(double) * pb.dimp, hipMemcpyDeviceToHost));
            unpack_state(pb, g.data(), beta_temp, gX ? gX + (size_t)c * pb.ND : nullptr,
                         gsig ? gsig + (size_t)c * pb.D : nullptr, gth ? gth + (size_t)c * pb.P : nullptr);
        }
    }
    return MAGI_OK;
}

int magi_logpost_grad(magi_handle* h, int n_chains, const double* X, const double* sig_pre, const double* th_pre,
                      double beta_temp, double* logp, double* gX, double* gsig, double* gth, double* terms) {
    return logpost_grad_impl(h, false, n_chains, X, sig_pre, th_pre, beta_temp, logp, gX, gsig, gth, terms);
}

int magi_logpost_grad_fused(magi_handle* h, int n_chains, const double* X, const double* sig_pre, const double* th_pre,
                            double beta_temp, double* logp, double* gX, double* gsig, double* gth, double* terms) {
    return logpost_grad_impl(h, true, n_chains, X, sig_pre, th_pre, beta_temp, logp, gX, gsig, gth, terms);
}

void magi_sampler_cfg_default(magi_sampler_cfg* cfg) {
    if (!cfg) return;
    cfg->num_results = 1000;
    cfg->num_burnin_steps = 1000;
    cfg->num_adaptation_steps = -1;
    cfg->max_tree_depth = 10;
    cfg->mode = MAGI_MODE_NUTS;
    cfg->hmc_leapfrogs = 32;
    cfg->anneal = 1;
    cfg->stale_cache = 1;
    cfg->step_size = 0.1;
    cfg->target_accept_prob = 0.75;
    cfg->max_energy_diff = 1000.0;
    cfg->min_temp = 0.1;
}

int magi_sampler_init(magi_handle* h, const magi_sampler_cfg* cfg, int n_chains, const double* X0, const double* sig_pre0,
                      const double* th_pre0, uint64_t seed, const int64_t* chain_ids) {
    if (!h) return MAGI_E_BADARG;
    if (!cfg || !X0 || !sig_pre0 || !th_pre0) return magi_fail(h, MAGI_E_BADARG, "null pointer");
    if (cfg->num_results < 0 || cfg->num_burnin_steps < 0 || cfg->num_results + cfg->num_burnin_steps <= 0)
        return magi_fail(h, MAGI_E_BADARG, "need num_results + num_burnin_steps > 0");
    if (cfg->max_tree_depth < 1 || cfg->max_tree_depth > MAGI_MAX_DEPTH)
        return magi_fail(h, MAGI_E_BADARG, "max_tree_depth must be in [1, 12]");
    if (cfg->mode != MAGI_MODE_NUTS && cfg->mode != MAGI_MODE_HMC) return magi_fail(h, MAGI_E_BADARG, "unknown sampler mode");
    if (cfg->mode == MAGI_MODE_HMC && cfg->hmc_leapfrogs < 1) return magi_fail(h, MAGI_E_BADARG, "hmc_leapfrogs must be >= 1");
    if (!(cfg->step_size > 0.0)) return magi_fail(h, MAGI_E_BADARG, "step_size must be positive");
    (void)hipSetDevice(h->device);
    int rc = magi_ensure_chains(h, n_chains);
    if (rc) return rc;
    const DevProblem& pb = h->pb;
    SamplerCfgDev& c = h->cfg;
    c.total = cfg->num_results + cfg->num_burnin_steps;
    c.burnin = cfg->num_burnin_steps;
    c.n_adapt = cfg->num_adaptation_steps >= 0 ? cfg->num_adaptation_steps : (int)(0.8 * cfg->num_burnin_steps);
    c.max_depth = cfg->max_tree_depth;
    c.mode = cfg->mode;
    c.hmc_L = cfg->hmc_leapfrogs;
    c.anneal = cfg->anneal;
    c.stale = cfg->stale_cache;
    c.step_size = cfg->step_size;
    c.target_accept = cfg->target_accept_prob;
    c.max_energy_diff = cfg->max_energy_diff;
    c.min_temp = cfg->min_temp;
    c.seed = seed;
    h->num_results = cfg->num_results;

    // output buffers
    const size_t need_s = (size_t)n_chains * std::max(1, cfg->num_results) * pb.dimp;
    if (need_s > h->samples_cap) {
        free_dev(h->ch.samples);
        h->ch.samples = nullptr;
        MAGI_HIP_CHECK(h, hipMalloc(&h->ch.samples, need_s * sizeof(double)));
        h->samples_cap = need_s;
    }
    const size_t need_d = (size_t)n_chains * c.total;
    if (need_d > h->diag_cap) {
        DevChains& d = h->ch;
        free_dev(d.d_step_size); free_dev(d.d_lar); free_dev(d.d_target); free_dev(d.d_energy); free_dev(d.d_beta);
        free_dev(d.d_leapfrogs); free_dev(d.d_depth); free_dev(d.d_flags);
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_step_size, need_d * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_lar, need_d * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_target, need_d * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_energy, need_d * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_beta, need_d * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_leapfrogs, need_d * sizeof(int)));
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_depth, need_d * sizeof(int)));
        MAGI_HIP_CHECK(h, hipMalloc(&d.d_flags, need_d * sizeof(int)));
        h->diag_cap = need_d;
    }
    {   // steps not taken yet read as NaN / -1, not as whatever the allocation held
        DevChains& d = h->ch;
        MAGI_HIP_CHECK(h, hipMemsetAsync(d.samples, 0xFF, need_s * sizeof(double), h->stream));
        for (double* p : {d.d_step_size, d.d_lar, d.d_target, d.d_energy, d.d_beta}) MAGI_HIP_CHECK(h, hipMemsetAsync(p, 0xFF, need_d * sizeof(double), h->stream));
        for (int* p : {d.d_leapfrogs, d.d_depth, d.d_flags}) MAGI_HIP_CHECK(h, hipMemsetAsync(p, 0xFF, need_d * sizeof(int), h->stream));
    }
    drop_graph(h);   // kernel arguments (cfg, buffers) are baked into the captured graph

    if ((rc = upload_states(h, n_chains, X0, sig_pre0, th_pre0))) return rc;
    std::vector<long long> ids(n_chains);
    for (int i = 0; i < n_chains; ++i) ids[i] = chain_ids ? (long long)chain_ids[i] : (long long)i;
    MAGI_HIP_CHECK(h, hipMemcpy(h->d_chain_ids, ids.data(), sizeof(long long) * n_chains, hipMemcpyHostToDevice));
    if ((rc = magi_launch_init_chains(h, h->d_chain_ids, h->stream))) return rc;
    // bootstrap_results: one gradient at the initial state.  Slot 0 evaluates it, the decisions riding in slot 1's
    // stream store it as the proposal and leave the chains idle; a graph launch then starts at slot parity 0 again.
    if ((rc = magi_launch_prepare(h, n_chains, h->stream))) return rc;
    for (int sl = 0; sl < 2; ++sl) {
        if ((rc = magi_launch_stream(h, n_chains, sl, true, h->stream))) return rc;
        if ((rc = magi_launch_point(h, n_chains, sl, h->stream))) return rc;
    }
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    h->epoch = 0;
    h->sampler_ready = true;
    return MAGI_OK;
}

int magi_sampler_steps_done(magi_handle* h, int64_t* steps) {
    if (!h || !steps) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "sampler not initialised");
    (void)hipSetDevice(h->device);
    std::vector<ChainCtl> ctl(h->n_chains);
    MAGI_HIP_CHECK(h, hipMemcpy(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains, hipMemcpyDeviceToHost));
    for (int i = 0; i < h->n_chains; ++i) steps[i] = ctl[i].k;
    return MAGI_OK;
}

int magi_sampler_run(magi_handle* h, int n_steps, int64_t* leapfrogs_done, double* kernel_ms) {
    if (!h) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "sampler not initialised");
    if (n_steps <= 0) return magi_fail(h, MAGI_E_BADARG, "n_steps must be positive");
    (void)hipSetDevice(h->device);
    int rc;
    // option no_graph (MAGI_NO_GRAPH=1 at handle creation): launch the leapfrog slots directly (debugging / rocprofv3 kernel traces of
    // long runs: the profiler's graph-node bookkeeping does not survive ~10^5 replayed nodes)
    const bool use_graph = !h->opt.no_graph;
    if (use_graph && !h->graph_valid && (rc = build_graph(h))) return rc;

    // (copies of this call ride on the handle's own non-blocking stream: a blocking hipMemcpy on the legacy stream is refused by HIP while ANOTHER
    //  handle, driven from another host thread, is capturing its graph -- "would make the legacy stream depend on a capturing stream")
    auto fetch = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream);
        return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
    };
    std::vector<ChainCtl> ctl(h->n_chains);
    MAGI_HIP_CHECK(h, fetch(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains));
    long long lf0 = 0;
    int kmin = h->cfg.total;
    for (auto& c : ctl) { lf0 += c.total_leapfrogs; kmin = std::min(kmin, c.k); }

    GlobalCtl g{};
    g.done_chains = 0;
    g.n_chains = h->n_chains;
    g.all_done = 0;
    g.stop_k = std::min(h->cfg.total, kmin + n_steps);
    g.epoch = ++h->epoch;
    MAGI_HIP_CHECK(h, hipMemcpyAsync(h->ch.gctl, &g, sizeof(GlobalCtl), hipMemcpyHostToDevice, h->stream));
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // &g is pageable stack memory
    MAGI_HIP_CHECK(h, hipEventRecord(h->ev_t0, h->stream));

    // pump: keep two graph launches in flight; after each, snapshot the control block
    const int depth = 2;
    long long issued = 0, retired = 0;
    bool done = false;
    // every slot advances every unfinished chain by a leaf or a set-up step, so a transition takes at most 2^max_depth - 1 leaves
    // + one set-up slot per doubling + start / end: a pump that outlives that budget means a corrupted control block
    // (+ one more set-up slot per doubling and two per transition end when a batch of NUTS chains spreads those passes: decide.h)
    const long long per_transition = (h->cfg.mode == MAGI_MODE_HMC ? (long long)h->cfg.hmc_L : (1ll << (h->cfg.max_depth + 1))) + 2 * h->cfg.max_depth + 8;
    long long max_graphs = ((long long)(g.stop_k - kmin) * per_transition + 8) / kGraphSlots + 4;
    if (h->opt.slot_budget_graphs > 0) max_graphs = std::max(1ll, std::min(max_graphs, h->opt.slot_budget_graphs));    // (test hook, magi_set_option only: force the budget exit)
    while (!done) {
        if (issued >= max_graphs && issued == retired) {
            (void)hipStreamSynchronize(h->stream);
            (void)fetch(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains);
            std::string where;
            for (int i = 0; i < h->n_chains; ++i)
                if (ctl[i].phase != PH_IDLE || ctl[i].k < g.stop_k) { where = " (chain " + std::to_string(i) + ": k=" + std::to_string(ctl[i].k) + ", phase=" + std::to_string(ctl[i].phase) + ", depth=" + std::to_string(ctl[i].depth) + ")"; break; }
            h->sampler_ready = false;
            return magi_fail(h, MAGI_E_STATE, "sampler did not finish within its slot budget" + where);
        }
        while (issued - retired < depth && issued < max_graphs) {
            const int slot = (int)(issued % 4);
            if (use_graph) {
                MAGI_HIP_CHECK(h, hipGraphLaunch(h->graph_exec, h->stream));
            } else {
                for (int sl = 0; sl < kGraphSlots; ++sl) {
                    if ((rc = magi_launch_stream(h, h->n_chains, sl & 1, true, h->stream))) return rc;
                    if ((rc = magi_launch_point(h, h->n_chains, sl & 1, h->stream))) return rc;
                }
            }
            MAGI_HIP_CHECK(h, hipMemcpyAsync(&h->h_gctl[slot], h->ch.gctl, sizeof(GlobalCtl), hipMemcpyDeviceToHost, h->stream));
            MAGI_HIP_CHECK(h, hipEventRecord(h->ev[slot], h->stream));
            ++issued;
        }
        if (issued == retired) continue;
        const int slot = (int)(retired % 4);
        MAGI_HIP_CHECK(h, hipEventSynchronize(h->ev[slot]));
        if (h->h_gctl[slot].all_done) done = true;
        ++retired;
    }
    MAGI_HIP_CHECK(h, hipEventRecord(h->ev_t1, h->stream));
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    {
        GlobalCtl gend{};
        MAGI_HIP_CHECK(h, fetch(&gend, h->ch.gctl, sizeof(GlobalCtl)));
        h->last_slots = gend.slots;
        h->last_graphs = issued;
    }
    if (kernel_ms) {
        float ms = 0.f;
        MAGI_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
        *kernel_ms = ms;
    }
    if (leapfrogs_done) {
        MAGI_HIP_CHECK(h, fetch(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains));
        long long lf1 = 0;
        for (auto& c : ctl) lf1 += c.total_leapfrogs;
        *leapfrogs_done = lf1 - lf0;
    }
    return MAGI_OK;
}

int magi_sampler_run_stats(magi_handle* h, int64_t* slots_issued, int64_t* graphs_launched) {
    if (!h) return MAGI_E_BADARG;
    if (slots_issued) *slots_issued = h->last_slots;
    if (graphs_launched) *graphs_launched = h->last_graphs;
    return MAGI_OK;
}

int magi_sampler_profile(magi_handle* h, int n_slots, double* stream_us, double* point_us, int64_t* leapfrogs_done) {
    if (!h) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "sampler not initialised");
    if (n_slots < 2 || n_slots > 65536) return magi_fail(h, MAGI_E_BADARG, "n_slots in [2, 65536]");
    (void)hipSetDevice(h->device);
    n_slots &= ~1;                       // slot parity 0 first, as a graph launch
    std::vector<ChainCtl> ctl(h->n_chains);
    MAGI_HIP_CHECK(h, hipMemcpy(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains, hipMemcpyDeviceToHost));
    long long lf0 = 0;
    for (auto& c : ctl) lf0 += c.total_leapfrogs;
    GlobalCtl g{};
    g.n_chains = h->n_chains;
    g.stop_k = h->cfg.total;             // run on: the chains are paused again below
    g.epoch = ++h->epoch;
    MAGI_HIP_CHECK(h, hipMemcpy(h->ch.gctl, &g, sizeof(GlobalCtl), hipMemcpyHostToDevice));
    std::vector<hipEvent_t> ev((size_t)4 * n_slots, nullptr);
    int rc = MAGI_OK;
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, "hipEventCreate");
    for (int sl = 0; sl < n_slots && rc == MAGI_OK; ++sl) {
        h->prof_e0 = ev[(size_t)4 * sl]; h->prof_e1 = ev[(size_t)4 * sl + 1];
        rc = magi_launch_stream(h, h->n_chains, sl & 1, true, h->stream);
        h->prof_e0 = ev[(size_t)4 * sl + 2]; h->prof_e1 = ev[(size_t)4 * sl + 3];
        if (rc == MAGI_OK) rc = magi_launch_point(h, h->n_chains, sl & 1, h->stream);
    }
    h->prof_e0 = h->prof_e1 = nullptr;
    (void)hipStreamSynchronize(h->stream);
    double ts = 0.0, tp = 0.0;
    int ns = 0, np_ = 0;
    for (int sl = 0; sl < n_slots && rc == MAGI_OK; ++sl) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, ev[(size_t)4 * sl], ev[(size_t)4 * sl + 1]) == hipSuccess) { ts += a; ++ns; }
        if (hipEventElapsedTime(&b, ev[(size_t)4 * sl + 2], ev[(size_t)4 * sl + 3]) == hipSuccess) { tp += b; ++np_; }
    }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    (void)hipGetLastError();
    if (rc) return rc;
    if (stream_us) *stream_us = ns ? ts / ns * 1e3 : 0.0;
    if (point_us) *point_us = np_ ? tp / np_ * 1e3 : 0.0;
    MAGI_HIP_CHECK(h, hipMemcpy(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains, hipMemcpyDeviceToHost));
    long long lf1 = 0;
    for (auto& c : ctl) lf1 += c.total_leapfrogs;
    if (leapfrogs_done) *leapfrogs_done = lf1 - lf0;
    // the chains stand somewhere inside a transition: the sampler state is no longer at a transition boundary a later
    // magi_sampler_run could be compared against anything from -- the handle must be re-initialised
    h->sampler_ready = false;
    return MAGI_OK;
}

int magi_sampler_get_samples(magi_handle* h, double* X, double* sig_pre, double* th_pre) {
    if (!h) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "sampler not initialised");
    (void)hipSetDevice(h->device);
    const DevProblem& pb = h->pb;
    const int R = h->num_results;
    std::vector<double> buf((size_t)R * pb.dimp);
    for (int c = 0; c < h->n_chains; ++c) {
        if (R == 0) break;
        MAGI_HIP_CHECK(h, hipMemcpy(buf.data(), h->ch.samples + (size_t)c * R * pb.dimp, sizeof(double) * R * pb.dimp, hipMemcpyDeviceToHost));
        for (int s = 0; s < R; ++s) {
            const size_t o = (size_t)c * R + s;
            unpack_state(pb, buf.data() + (size_t)s * pb.dimp, 1.0, X ? X + o * pb.ND : nullptr,
                         sig_pre ? sig_pre + o * pb.D : nullptr, th_pre ? th_pre + o * pb.P : nullptr);
        }
    }
    return MAGI_OK;
}

int magi_sampler_get_diag(magi_handle* h, double* step_size, double* log_accept_ratio, int32_t* leapfrogs_taken,
                          int32_t* tree_depth, int32_t* has_divergence, int32_t* reach_max_depth, int32_t* is_accepted,
                          double* target_log_prob, double* energy, double* beta_temp) {
    if (!h) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "sampler not initialised");
    (void)hipSetDevice(h->device);
    const size_t n = (size_t)h->n_chains * h->cfg.total;
    const DevChains& d = h->ch;
    if (step_size) MAGI_HIP_CHECK(h, hipMemcpy(step_size, d.d_step_size, n * sizeof(double), hipMemcpyDeviceToHost));
    if (log_accept_ratio) MAGI_HIP_CHECK(h, hipMemcpy(log_accept_ratio, d.d_lar, n * sizeof(double), hipMemcpyDeviceToHost));
    if (target_log_prob) MAGI_HIP_CHECK(h, hipMemcpy(target_log_prob, d.d_target, n * sizeof(double), hipMemcpyDeviceToHost));
    if (energy) MAGI_HIP_CHECK(h, hipMemcpy(energy, d.d_energy, n * sizeof(double), hipMemcpyDeviceToHost));
    if (beta_temp) MAGI_HIP_CHECK(h, hipMemcpy(beta_temp, d.d_beta, n * sizeof(double), hipMemcpyDeviceToHost));
    if (leapfrogs_taken) MAGI_HIP_CHECK(h, hipMemcpy(leapfrogs_taken, d.d_leapfrogs, n * sizeof(int), hipMemcpyDeviceToHost));
    if (tree_depth) MAGI_HIP_CHECK(h, hipMemcpy(tree_depth, d.d_depth, n * sizeof(int), hipMemcpyDeviceToHost));
    if (has_divergence || reach_max_depth || is_accepted) {
        std::vector<int> f(n);
        MAGI_HIP_CHECK(h, hipMemcpy(f.data(), d.d_flags, n * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            if (has_divergence) has_divergence[i] = f[i] & 1;
            if (reach_max_depth) reach_max_depth[i] = (f[i] >> 1) & 1;
            if (is_accepted) is_accepted[i] = (f[i] >> 2) & 1;
        }
    }
    return MAGI_OK;
}

int magi_sampler_get_state(magi_handle* h, double* X, double* sig_pre, double* th_pre, double* step_size, double* beta_cache) {
    if (!h) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "sampler not initialised");
    (void)hipSetDevice(h->device);
    const DevProblem& pb = h->pb;
    std::vector<double> q((size_t)pb.dimp);
    std::vector<ChainCtl> ctl(h->n_chains);
    MAGI_HIP_CHECK(h, hipMemcpy(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains, hipMemcpyDeviceToHost));
    for (int c = 0; c < h->n_chains; ++c) {
        MAGI_HIP_CHECK(h, hipMemcpy(q.data(), h->ch.vec + vec_off(pb, c, V_CANDQ), sizeof(double) * pb.dimp, hipMemcpyDeviceToHost));
        unpack_state(pb, q.data(), 1.0, X ? X + (size_t)c * pb.ND : nullptr, sig_pre ? sig_pre + (size_t)c * pb.D : nullptr,
                     th_pre ? th_pre + (size_t)c * pb.P : nullptr);
        if (step_size) step_size[c] = ctl[c].da_step_size;
        if (beta_cache) beta_cache[c] = ctl[c].beta_cache;
    }
    return MAGI_OK;
}

// per-chain scalars of a checkpoint (magi_sampler_get_checkpoint / magi_sampler_set_checkpoint)
enum { CKPT_K = 0, CKPT_DA_STEP, CKPT_STEP_SIZE, CKPT_ERR_SUM, CKPT_LOG_AVG, CKPT_LOG_SHRINK, CKPT_BETA_CACHE, CKPT_TOTAL_LF, CKPT_TAG, CKPT_COUNT };

// What a checkpoint belongs to: the sampler configuration, the seed, the chain's Philox id and the state size, folded into 52 bits (exact in a
// double).  A resume promises the uninterrupted run bit for bit, which only holds under the same configuration: set_checkpoint compares.
static double ckpt_tag(const magi_handle* h, long long chain_id) {
    unsigned long long x = 1469598103934665603ull;
    auto mix = [&x](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { x ^= b[i]; x *= 1099511628211ull; } };
    const SamplerCfgDev& c = h->cfg;
    const int ints[9] = {c.total, c.burnin, c.n_adapt, c.max_depth, c.mode, c.mode == MAGI_MODE_HMC ? c.hmc_L : 0, c.anneal, c.stale, h->pb.dimp};
    const double dbl[4] = {c.step_size, c.target_accept, c.max_energy_diff, c.min_temp};
    mix(ints, sizeof(ints)); mix(dbl, sizeof(dbl)); mix(&c.seed, sizeof(c.seed)); mix(&chain_id, sizeof(chain_id));
    return (double)(x & ((1ull << 52) - 1));
}

int magi_sampler_get_checkpoint(magi_handle* h, double* scalars) {
    if (!h || !scalars) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "sampler not initialised");
    (void)hipSetDevice(h->device);
    std::vector<ChainCtl> ctl(h->n_chains);
    MAGI_HIP_CHECK(h, hipMemcpy(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains, hipMemcpyDeviceToHost));
    for (int c = 0; c < h->n_chains; ++c) {
        if (ctl[c].phase != PH_IDLE) return magi_fail(h, MAGI_E_STATE, "chain " + std::to_string(c) + " is inside a transition");
        double* o = scalars + (size_t)c * MAGI_CKPT_SCALARS;
        for (int k = 0; k < MAGI_CKPT_SCALARS; ++k) o[k] = 0.0;
        o[CKPT_K] = ctl[c].k; o[CKPT_DA_STEP] = ctl[c].da_step; o[CKPT_STEP_SIZE] = ctl[c].da_step_size;
        o[CKPT_ERR_SUM] = ctl[c].da_error_sum; o[CKPT_LOG_AVG] = ctl[c].da_log_avg; o[CKPT_LOG_SHRINK] = ctl[c].da_log_shrink;
        o[CKPT_BETA_CACHE] = ctl[c].beta_cache; o[CKPT_TOTAL_LF] = (double)ctl[c].total_leapfrogs;
        o[CKPT_TAG] = ckpt_tag(h, ctl[c].chain_id);
    }
    return MAGI_OK;
}

int magi_sampler_set_checkpoint(magi_handle* h, const double* scalars) {
    if (!h || !scalars) return MAGI_E_BADARG;
    if (!h->sampler_ready) return magi_fail(h, MAGI_E_STATE, "call magi_sampler_init with the checkpointed states first");
    (void)hipSetDevice(h->device);
    static_assert(CKPT_COUNT <= MAGI_CKPT_SCALARS, "checkpoint layout");
    std::vector<ChainCtl> ctl(h->n_chains);
    MAGI_HIP_CHECK(h, hipMemcpy(ctl.data(), h->ch.ctl, sizeof(ChainCtl) * h->n_chains, hipMemcpyDeviceToHost));
    for (int c = 0; c < h->n_chains; ++c) {
        const double* o = scalars + (size_t)c * MAGI_CKPT_SCALARS;
        if (ctl[c].phase != PH_IDLE || ctl[c].k != 0) return magi_fail(h, MAGI_E_STATE, "set the checkpoint right after magi_sampler_init");
        auto whole = [](double v, double hi) { return std::isfinite(v) && v >= 0.0 && v <= hi && v == std::floor(v); };
        const std::string who = "checkpoint of chain " + std::to_string(c);
        if (!whole(o[CKPT_K], (double)h->cfg.total) || !whole(o[CKPT_DA_STEP], 2147483647.0) || !whole(o[CKPT_TOTAL_LF], 9007199254740992.0))
            return magi_fail(h, MAGI_E_BADARG, who + ": transition index / adaptation step / leapfrog count must be whole numbers inside this configuration");
        if (!(o[CKPT_STEP_SIZE] > 0.0) || !std::isfinite(o[CKPT_STEP_SIZE]) || !std::isfinite(o[CKPT_ERR_SUM]) || !std::isfinite(o[CKPT_LOG_AVG]) ||
            !std::isfinite(o[CKPT_LOG_SHRINK]) || !(o[CKPT_BETA_CACHE] > 0.0) || !(o[CKPT_BETA_CACHE] <= 1.4426950408889636))      // (the schedule starts at 1 / ln 2, magi_v2.py:833-835)
            return magi_fail(h, MAGI_E_BADARG, who + ": non-finite dual-averaging state, step size <= 0 or cached temperature outside (0, 1 / ln 2]");
        if (o[CKPT_TAG] != ckpt_tag(h, ctl[c].chain_id))
            return magi_fail(h, MAGI_E_BADARG, who + " was taken under another sampler configuration, seed, chain id or problem size: a resume continues "
                                               "the SAME run (magi_hip.h)");
        ctl[c].k = (int)o[CKPT_K]; ctl[c].da_step = (int)o[CKPT_DA_STEP]; ctl[c].da_step_size = o[CKPT_STEP_SIZE];
        ctl[c].da_error_sum = o[CKPT_ERR_SUM]; ctl[c].da_log_avg = o[CKPT_LOG_AVG]; ctl[c].da_log_shrink = o[CKPT_LOG_SHRINK];
        ctl[c].beta_cache = o[CKPT_BETA_CACHE]; ctl[c].total_leapfrogs = (long long)o[CKPT_TOTAL_LF];
    }
    MAGI_HIP_CHECK(h, hipMemcpy(h->ch.ctl, ctl.data(), sizeof(ChainCtl) * h->n_chains, hipMemcpyHostToDevice));
    return MAGI_OK;
}

int magi_sample(magi_handle* h, const magi_sampler_cfg* cfg, int n_chains, const double* X0, const double* sig_pre0,
                const double* th_pre0, uint64_t seed, const int64_t* chain_ids, double* X_samps, double* sig_pre_samps,
                double* th_pre_samps) {
    int rc = magi_sampler_init(h, cfg, n_chains, X0, sig_pre0, th_pre0, seed, chain_ids);
    if (rc) return rc;
    if ((rc = magi_sampler_run(h, cfg->num_results + cfg->num_burnin_steps, nullptr, nullptr))) return rc;
    return magi_sampler_get_samples(h, X_samps, sig_pre_samps, th_pre_samps);
}

int magi_fit_hparams(magi_handle* h, const double* I, int N, int D, const double* X_filled, const double* mu, const double* mu_phi2,
                     const double* sd_phi2, const double* sigma_sq_loc, double nu, int num_iters, double learning_rate, double jitter,
                     double* phi1, double* phi2, double* sigma_sq, double* loss_trace) {
    if (!h) return MAGI_E_BADARG;
    if (!I || !X_filled || !mu || !mu_phi2 || !sd_phi2 || !sigma_sq_loc || !phi1 || !phi2 || !sigma_sq)
        return magi_fail(h, MAGI_E_BADARG, "null pointer");
    if (N < 2 || D < 1 || num_iters < 0 || !(nu > 1.0)) return magi_fail(h, MAGI_E_BADARG, "bad N, D, num_iters or nu");
    for (int d = 0; d < D; ++d)
        if (!(phi1[d] > 0.0) || !(phi2[d] > 0.0) || !(sigma_sq[d] > 0.0)) return magi_fail(h, MAGI_E_BADARG, "initial values must be positive");
    (void)hipSetDevice(h->device);
    return magi_fit_hparams_device(h, I, N, D, X_filled, mu, mu_phi2, sd_phi2, sigma_sq_loc, nu, num_iters, learning_rate, jitter,
                                   phi1, phi2, sigma_sq, loss_trace);
}

int magi_set_option(magi_handle* h, const char* name, int64_t value) {
    if (!h || !name) return MAGI_E_BADARG;
    const std::string k(name);
    MagiOptions& o = h->opt;
    if (k == "stream_family") { if (value < 0 || value > 2) return magi_fail(h, MAGI_E_BADARG, "stream_family: 0 auto, 1 mc, 2 valu"); o.stream_family = (int)value; }
    else if (k == "sep_pair_min") o.sep_pair_min = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (k == "family_chains") { if (value < 0 || value > 4096) return magi_fail(h, MAGI_E_BADARG, "family_chains in [0, 4096]"); o.family_chains = (int)value; }
    else if (k == "fused_parity") o.fused_parity = value == 1;
    else if (k == "gemm_remap_min") o.gemm_remap_min = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (k == "potrf_lookahead_min") { if (value < 0) return magi_fail(h, MAGI_E_BADARG, "potrf_lookahead_min >= 0"); o.potrf_lookahead_min = (int)value; }
    else if (k == "potrf_panels") { if (value < 1 || value > 16) return magi_fail(h, MAGI_E_BADARG, "potrf_panels in [1, 16]"); o.potrf_panels = (int)value; }
    else if (k == "slot_budget_graphs") o.slot_budget_graphs = std::max<int64_t>(0, value);
    else if (k == "no_graph") o.no_graph = value != 0;
    else if (k == "fit_host_loop") o.fit_host_loop = value != 0;
    else if (k == "fit_per_component") o.fit_per_component = value != 0;
    else if (k == "build_profile") o.build_profile = value != 0;
    else if (k == "build_serial") o.build_serial = value != 0;
    else return magi_fail(h, MAGI_E_BADARG, "unknown option '" + k + "'");
    return MAGI_OK;
}

int magi_build_profile(magi_handle* h, double* flops, double* ms, int64_t* calls) {
    if (!h || !flops || !ms || !calls) return MAGI_E_BADARG;
    long c[16];
    const int n = magi_build_profile_get(h, flops, ms, c);
    for (int i = 0; i < n; ++i) calls[i] = c[i];
    return n;
}

int magi_debug_par(magi_handle* h, int chain, double* out64) {
    if (!h || !out64 || chain < 0 || chain >= h->n_chains) return MAGI_E_BADARG;
    (void)hipSetDevice(h->device);
    MAGI_HIP_CHECK(h, hipMemcpy(out64, h->ch.par + (size_t)chain * PAR_COUNT, sizeof(double) * PAR_COUNT, hipMemcpyDeviceToHost));
    return MAGI_OK;
}

int magi_gradient_bytes(magi_handle* h, int n_chains, double* phase_bytes) {
    if (!h || !phase_bytes) return MAGI_E_BADARG;
    if (!h->have_matrices) return magi_fail(h, MAGI_E_STATE, "no matrices");
    const DevProblem& pb = h->pb;
    const double W = pb.band < 0 ? (double)pb.N : (double)(2 * pb.band + 1);
    const double mat = (double)pb.D * pb.N * W * 8.0;      // one matrix stack, algorithmic (unpadded) bytes
    const double vec = (double)n_chains * pb.N * pb.D * 8.0;
    phase_bytes[0] = 2.0 * mat + 3.0 * vec;   // Csym, M ; read X, write Cx, r
    phase_bytes[1] = 1.0 * mat + 2.0 * vec;   // Ksym    ; read r, write Kr
    phase_bytes[2] = 1.0 * mat + 5.0 * vec;   // Mt      ; read Kr, X, Cx, yobs, write gX
    phase_bytes[3] = 5.0 * vec;               // reduce  ; read X, Cx, r, Kr, yobs
    const double nslot = (double)std::min(pb.nb, 2 * pb.wb + 1);
    const double tiles = (double)pb.n_tasks * MAGI_TB * MAGI_TB * 8.0;      // packed blocks of FH, FK (lower block triangle), FE
    phase_bytes[4] = tiles + 2.0 * (double)n_chains * pb.n_tasks * MAGI_TB * 8.0;   // k_stream / k_stream_mc: + 2 TB partials per block and chain
    phase_bytes[5] = (double)n_chains * magi_leap_wgs(pb) * PART_K * 8.0; // tail: the workgroup partials
    phase_bytes[6] = 4.0 * vec * nslot + 10.0 * vec;                       // k_point: block partials of the four products; X, yobs, p, rho, g, p_leaf, p', x'
    if (h->have_problem && magi_stream_family_mc(h, n_chains) && magi_drift_separable(pb.drift)) {
        // k_stream_sep writes a block vector per (task, product, matrix-core column in use) and reads its operands from the mirror planes
        // (once per XCD at the fabric); its point kernel re-reads the product slots and writes the next slot's mirror
        double st = 0.0, op = 0.0, pr = 0.0, mw = 0.0;
        magi_sep_traffic(pb, n_chains, &st, &op, &pr, &mw);
        phase_bytes[4] = tiles + op + st;
        phase_bytes[6] = pr + 10.0 * vec + mw;
    }
    phase_bytes[7] = 3.0 * (double)pb.D * pb.N * W * 8.0 + 10.0 * vec;    // SURVEY 8d algorithmic bytes of one gradient evaluation
    return MAGI_OK;
}

int magi_stream_kernel_name(magi_handle* h, int n_chains, char* buf, int cap) {
    if (!h || !buf || cap < 2) return MAGI_E_BADARG;
    if (!h->have_matrices || !h->have_problem) return magi_fail(h, MAGI_E_STATE, "set matrices and problem first");
    std::string s;
    if (magi_stream_family_mc(h, n_chains))
        s = magi_drift_separable(h->pb.drift) ? (n_chains <= 8 ? "k_stream_sep<CW=8>" : "k_stream_sep<CW=16>") : "k_stream_mc";
    else
        s = n_chains >= 2 ? "k_stream<2>" : "k_stream<1>";
    std::snprintf(buf, (size_t)cap, "%s", s.c_str());
    return MAGI_OK;
}

int magi_time_gradient(magi_handle* h, int n_chains, int reps, double* total_ms_per_eval, double* phase_ms) {
    if (!h) return MAGI_E_BADARG;
    if (reps <= 0) return magi_fail(h, MAGI_E_BADARG, "reps must be positive");
    (void)hipSetDevice(h->device);
    int rc = magi_ensure_chains(h, n_chains);
    if (rc) return rc;
    MAGI_HIP_CHECK(h, hipMemsetAsync(h->ch.gctl, 0, sizeof(GlobalCtl), h->stream));
    if ((rc = magi_launch_prepare(h, n_chains, h->stream))) return rc;
    if ((rc = magi_launch_plan_eval(h, n_chains, h->stream))) return rc;
    h->sampler_ready = false;          // the chain state is clobbered by the timing launches
    float ms = 0.f;
    // warm
    for (int i = 0; i < 3; ++i) {
        if ((rc = magi_launch_gradient(h, n_chains, h->stream))) return rc;
        if ((rc = magi_launch_finalize(h, n_chains, h->d_fin, h->stream))) return rc;
    }
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    MAGI_HIP_CHECK(h, hipEventRecord(h->ev_t0, h->stream));
    for (int i = 0; i < reps; ++i) {
        if ((rc = magi_launch_gradient(h, n_chains, h->stream))) return rc;
        if ((rc = magi_launch_finalize(h, n_chains, h->d_fin, h->stream))) return rc;
    }
    MAGI_HIP_CHECK(h, hipEventRecord(h->ev_t1, h->stream));
    MAGI_HIP_CHECK(h, hipEventSynchronize(h->ev_t1));
    MAGI_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
    if (total_ms_per_eval) *total_ms_per_eval = ms / reps;
    if (phase_ms) {
        for (int ph = 1; ph <= 8; ++ph) {
            MAGI_HIP_CHECK(h, hipEventRecord(h->ev_t0, h->stream));
            for (int i = 0; i < reps; ++i) {
                if (ph <= 3) rc = magi_launch_phase(h, ph, n_chains, h->stream);
                else if (ph == 4) rc = magi_launch_finalize(h, n_chains, h->d_fin, h->stream);
                else if (ph == 5) rc = magi_launch_stream(h, n_chains, i & 1, false, h->stream);       // slot parity alternates as in the sampler (the block walk reverses)
                else if (ph == 6) rc = magi_launch_leap_finalize(h, n_chains, h->d_fin, h->stream);
                else if (ph == 7) rc = magi_launch_point(h, n_chains, 0, h->stream);
                else rc = magi_launch_read_tiles(h, i & 1, h->stream);
                if (rc) return rc;
            }
            MAGI_HIP_CHECK(h, hipEventRecord(h->ev_t1, h->stream));
            MAGI_HIP_CHECK(h, hipEventSynchronize(h->ev_t1));
            MAGI_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
            phase_ms[ph - 1] = ms / reps;
        }
    }
    return MAGI_OK;
}

}  // extern "C"
