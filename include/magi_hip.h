/*
 * magi_hip.h -- C ABI of libmagi_hip.so, the MI355X (gfx950) engine for the MAGI hot path.
 *
 * The reference (sophiaxxiao/magi_v2) has no FFI: its boundary is the Python class surface
 * of magi_v2.py.  Each entry point below names the reference computation it replaces
 * (file:line relative to the reference tree).  The host-side mirror of that class lives in
 * magi_v2_amd/api.py and binds these symbols with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every array is C-contiguous fp64 (or int64 where stated) and is
 *     owned by the caller; no pointer outlives the call except the opaque handle.
 *   - host layout of trajectories follows the reference: X[chain][N][D] (row-major [N,D]).
 *   - every int-returning function returns 0 on success and a negative MAGI_E_* code on
 *     failure; magi_last_error() gives the message (pass NULL for creation failures).
 *   - one handle per GPU; a handle is not thread-safe; different handles may be driven from
 *     different host threads.
 */
#ifndef MAGI_HIP_H
#define MAGI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct magi_handle magi_handle;

enum {
    MAGI_OK = 0,
    MAGI_E_BADARG = -1,   /* shape / pointer / state error                              */
    MAGI_E_HIP = -2,      /* a HIP runtime call failed                                  */
    MAGI_E_NOTSPD = -3,   /* Cholesky met a non-positive pivot (index in the message)   */
    MAGI_E_NAN = -4,      /* NaN in an initial state (reference asserts magi_v2.py:289-291) */
    MAGI_E_STATE = -5     /* call order violated (e.g. sampling before set_problem)     */
};

/* compiled-in ODE drifts f(t, X, theta) with analytic Jacobians:
 *   SEIR3: vignette.ipynb cell 3 (S implicit; theta = beta, gamma, sigma)
 *   SEIR4: the four data columns with S explicit (data/SEIR_seed=0.csv:1)
 *   SIRW : test_magi_script.py:19-45 (theta = beta, phi, xi, chi, kappa)          */
enum { MAGI_DRIFT_SEIR3 = 0, MAGI_DRIFT_SEIR4 = 1, MAGI_DRIFT_SIRW = 2,
       MAGI_DRIFT_USER = 3 /* only in a library specialised for a traced f_vec, see magi_user_drift_info */ };

enum { MAGI_MODE_NUTS = 0, MAGI_MODE_HMC = 1 };

/* Generic ODEs (callers magi_v2.py:155, 206, 335 pass an arbitrary f_vec): magi_v2_amd.drift traces the callable,
 * emits DriftT<MAGI_DRIFT_USER> and magi_v2_amd.jit compiles this library's kernels for it.  Returns 1 and fills
 * D, P when this library was built that way (it then accepts ONLY drift = MAGI_DRIFT_USER), else 0. */
int magi_user_drift_info(int* D, int* P);

/* ---- lifetime --------------------------------------------------------------------------- */

/* Binds device `device_id`, creates the stream.  NULL on failure (see magi_last_error(NULL)). */
magi_handle* magi_create(int device_id);
void magi_destroy(magi_handle* h);
const char* magi_last_error(const magi_handle* h);
/* "magi_hip <version> gfx950" */
const char* magi_version(void);

/* ---- kernel matrices ---------------------------------------------------------------------- */

/* Replaces MAGI_v2._build_matrices (magi_v2.py:774-823), the pinv sites (magi_v2.py:126,128 /
 * 266,268 / 449,451) and the band approximation (magi_v2.py:271-274) for all D components:
 * Matern(nu) Kappa / p_Kappa / Kappa_pp assembly on the grid I[N], fp64 blocked Cholesky,
 * m = p_Kappa Kappa^-1, K = Kappa_pp - p_Kappa Kappa^-1 Kappa_p, C^-1, K^-1, band mask.
 * bandsize < 0 means dense.  The results stay resident on the device for the log-posterior
 * and sampler calls; C_inv / m / K_inv ([D][N][N] row-major host buffers) may each be NULL
 * when the caller does not want a host copy. */
int magi_build_matrices(magi_handle* h, const double* I, int N, int D,
                        const double* phi1, const double* phi2, double nu, int bandsize,
                        double* C_inv, double* m, double* K_inv);

/* GP hyper-parameter fit: replaces MAGI_v2._fit_kernel_hparams (magi_v2.py:538-691) -- Adam (tf_keras defaults,
 * `learning_rate`, `num_iters` steps) on the softplus-reparameterised (phi1, phi2, sigma^2) of every component,
 * maximising  D * sum_d [ log N(x_d ; mu_d, phi1_d R_nu(phi2_d) + (sigma_d^2 + jitter) I) + TruncatedNormal priors ]
 * with the reference's priors (magi_v2.py:611-627: loc 1e-4 / sigma_sq_loc / mu_phi2, scales 1000 sqrt(D) /
 * 1000 sqrt(D) / sd_phi2 sqrt(D)).  X_filled[N][D] row-major (no NaN), mu[D] the GP means (magi_v2.py:559).
 * phi1 / phi2 / sigma_sq hold the starting values on entry (magi_v2.py:631-639) and the fitted values on return.
 * Every step is Matern assembly + Cholesky + inverse + trace terms on the GPU; loss_trace[num_iters] may be NULL.
 * jitter: TFP's GaussianProcess default 1e-6. */
int magi_fit_hparams(magi_handle* h, const double* I, int N, int D, const double* X_filled, const double* mu,
                     const double* mu_phi2, const double* sd_phi2, const double* sigma_sq_loc, double nu,
                     int num_iters, double learning_rate, double jitter,
                     double* phi1, double* phi2, double* sigma_sq, double* loss_trace);

/* The three Matern blocks alone (magi_v2.py:781-815) for one component; host outputs [N][N]. */
int magi_matern_blocks(magi_handle* h, const double* I, int N, double phi1, double phi2, double nu,
                       double* Kappa, double* p_Kappa, double* Kappa_pp);

/* User-overwritten or golden matrices (the reference lets users overwrite C_d_invs, m_ds,
 * K_d_invs between fit and predict, magi_v2.py:77-80).  [D][N][N] row-major.  With
 * bandsize >= 0 entries with |i-j| > bandsize are dropped exactly as tf.linalg.band_part
 * does (magi_v2.py:271-274) and the matrices are kept in banded storage N x (2b+1). */
int magi_set_matrices(magi_handle* h, int N, int D, int bandsize,
                      const double* C_inv, const double* m, const double* K_inv);

/* The three entry points above keep dense device copies of C^-1, m, K^-1 ([D][N][N], before the band mask) resident in the
 * handle; the calls below let a caller stage its work on them without moving N x N matrices over PCIe
 * (MAGI_v2.initial_fit builds the observed components, initialises theta against the UNbanded m / K^-1, builds the
 * unobserved components later and only then applies the band: magi_v2.py:122-128, 133-179, 262-274).
 *
 * magi_build_dense: Eqn.-6 matrices of the components sel[0 .. n_sel) (phi1 / phi2 indexed like sel) into the resident
 *   stacks of a D-component problem (stacks are allocated, and zero for components not built yet, when N or D change).
 *   Invalidates the packed operands until magi_pack_resident.
 * magi_pack_resident: band mask + packing of the resident stacks (what magi_build_matrices / magi_set_matrices do last).
 * magi_get_dense: host copies of the resident stacks with tf.linalg.band_part(., b, b) applied (bandsize < 0: as built);
 *   any pointer may be NULL.
 * magi_dense_apply: Y[d] = A_d V[d] (transpose = 0) or A_d^T V[d] for the resident stack `which` (0 = C^-1, 1 = m,
 *   2 = K^-1); V, Y host [D][N][nv], nv <= 8 -- the N x N products of the theta initialiser (magi_v2.py:142, 157-158). */
int magi_build_dense(magi_handle* h, const double* I, int N, int D, int n_sel, const int32_t* sel,
                     const double* phi1, const double* phi2, double nu);
int magi_pack_resident(magi_handle* h, int bandsize);
int magi_get_dense(magi_handle* h, int bandsize, double* C_inv, double* m, double* K_inv);
int magi_dense_apply(magi_handle* h, int which, int transpose, int nv, const double* V, double* Y);

/* ---- log posterior ------------------------------------------------------------------------ */

/* Constants captured by the reference's log-posterior closure (magi_v2.py:294-300):
 * mu[D] (magi_v2.py:114,259), N_ds[D] (:53), obs_idx[n_obs] = flat row-major indices into
 * X[N][D] of the non-NaN observations (:96), y[n_obs] (:100), beta = D|I|/sum N_d (:89),
 * LB[D] (:300), the drift and its parameter count P.  Requires matrices (for N, D). */
int magi_set_problem(magi_handle* h, const double* mu, const double* N_ds,
                     const int64_t* obs_idx, const double* y, int64_t n_obs,
                     double beta, const double* LB, int drift_id, int P);

/* unnormalized_log_prob (magi_v2.py:308-348) and its gradient (TF autodiff in the reference,
 * induced by magi_v2.py:360-364) for n_chains independent states.
 * X[n_chains][N][D], sig_pre[n_chains][D], th_pre[n_chains][P] -> logp[n_chains],
 * gX[n_chains][N][D], gsig[n_chains][D], gth[n_chains][P] (any output may be NULL).
 * terms, if not NULL, receives t1..t4 per chain ([n_chains][4], magi_v2.py:332-345). */
int magi_logpost_grad(magi_handle* h, int n_chains, const double* X, const double* sig_pre,
                      const double* th_pre, double beta_temp,
                      double* logp, double* gX, double* gsig, double* gth, double* terms);

/* Same quantity through the sampler's single-phase formulation: operators
 * FH = Csym + m^T Ksym m, FE = Ksym m, FE^T, Ksym are applied to xc and f(X, theta) in ONE streaming
 * kernel.  Exposed so the formulation the sampler runs can be validated against the reference
 * op order above; terms = {t1 + t2, 0, t3, t4}. */
int magi_logpost_grad_fused(magi_handle* h, int n_chains, const double* X, const double* sig_pre,
                            const double* th_pre, double beta_temp,
                            double* logp, double* gX, double* gsig, double* gth, double* terms);

/* ---- sampler ------------------------------------------------------------------------------ */

/* Replaces the TFP wiring of predict (magi_v2.py:357-396) and LogAnnealedNUTS
 * (magi_v2.py:838-889).  Defaults via magi_sampler_cfg_default() are the reference's. */
typedef struct magi_sampler_cfg {
    int32_t num_results;          /* magi_v2.py:390                                         */
    int32_t num_burnin_steps;     /* magi_v2.py:391                                         */
    int32_t num_adaptation_steps; /* < 0 -> int(0.8 * num_burnin_steps), magi_v2.py:365     */
    int32_t max_tree_depth;       /* TFP default 10                                         */
    int32_t mode;                 /* MAGI_MODE_NUTS (reference) or MAGI_MODE_HMC (fixed L)  */
    int32_t hmc_leapfrogs;        /* L for MAGI_MODE_HMC                                    */
    int32_t anneal;               /* 1: beta_temp(k) = max(1/ln(k+2), min_temp), :833-835   */
    int32_t stale_cache;          /* 1: reuse the previous step's cached target/grad as TFP
                                     does (computed at the previous temperature)            */
    double step_size;             /* 0.1, magi_v2.py:364                                    */
    double target_accept_prob;    /* 0.75, magi_v2.py:366                                   */
    double max_energy_diff;       /* TFP default 1000                                       */
    double min_temp;              /* 0.1, magi_v2.py:357                                    */
} magi_sampler_cfg;

void magi_sampler_cfg_default(magi_sampler_cfg* cfg);

/* Start n_chains independent chains from pre-transformed states (the softplus-inverse inits
 * of magi_v2.py:374-383 are formed by the host layer).  chain_ids[n_chains] (may be NULL =
 * 0..n-1) select the Philox stream of each chain so results do not depend on how chains are
 * sharded over GPUs.  Fails with MAGI_E_NAN when a state holds NaN (magi_v2.py:289-291). */
int magi_sampler_init(magi_handle* h, const magi_sampler_cfg* cfg, int n_chains,
                      const double* X0, const double* sig_pre0, const double* th_pre0,
                      uint64_t seed, const int64_t* chain_ids);

/* Advance every chain by up to n_steps transitions (burn-in steps count); blocks until done.
 * leapfrogs_done, if not NULL, receives the gradient evaluations taken in this call (sum over
 * chains); kernel_ms the device time between the first and last launch of the call. */
int magi_sampler_run(magi_handle* h, int n_steps, int64_t* leapfrogs_done, double* kernel_ms);

/* Steps completed so far per chain (burn-in included). */
int magi_sampler_steps_done(magi_handle* h, int64_t* steps /* [n_chains] */);

/* Post-burn-in samples, pre-transform (sample_results of magi_v2.py:421):
 * X[n_chains][num_results][N][D], sig_pre[n_chains][num_results][D], th_pre[..][P]. */
int magi_sampler_get_samples(magi_handle* h, double* X, double* sig_pre, double* th_pre);

/* Per-step diagnostics for ALL steps taken so far (burn-in included), each [n_chains][n_total]
 * with n_total = num_burnin_steps + num_results; any pointer may be NULL.  These are the
 * NUTSKernelResults fields the reference traces (magi_v2.py:394). */
int magi_sampler_get_diag(magi_handle* h, double* step_size, double* log_accept_ratio,
                          int32_t* leapfrogs_taken, int32_t* tree_depth, int32_t* has_divergence,
                          int32_t* reach_max_depth, int32_t* is_accepted,
                          double* target_log_prob, double* energy, double* beta_temp);

/* Current state of every chain (checkpoint / hand-over to the CPU oracle in tests):
 * X[n_chains][N][D], sig_pre, th_pre, step_size[n_chains] (the dual-averaging "new step
 * size"), beta_cache[n_chains] (temperature at which the cached target was computed). */
int magi_sampler_get_state(magi_handle* h, double* X, double* sig_pre, double* th_pre,
                           double* step_size, double* beta_cache);

/* Checkpoint / resume (the reference has none: a run is all-or-nothing, magi_v2.py:386-425).  Between two magi_sampler_run
 * calls every chain stands at a transition boundary; its state is (X, sig_pre, th_pre) from magi_sampler_get_state plus
 * MAGI_CKPT_SCALARS doubles per chain from magi_sampler_get_checkpoint (transition index, dual-averaging state, cached
 * temperature, leapfrog count).  To resume -- in this or another process / handle: magi_sampler_init with the SAME cfg, seed
 * and chain_ids and the checkpointed states, then magi_sampler_set_checkpoint(scalars), then magi_sampler_run: the remaining
 * transitions are those the uninterrupted run would have made, bit for bit (the Philox streams are keyed by transition index
 * and chain id; target and gradient of the restored state are re-evaluated by the same kernels).  Samples and diagnostics of
 * the steps taken before the checkpoint belong to the run that took them.  The scalars carry a tag of (cfg, seed, chain id,
 * state size): magi_sampler_set_checkpoint rejects (MAGI_E_BADARG) a checkpoint taken under anything else, and any scalar that
 * is not finite / not a whole number where one is expected / outside its range. */
#define MAGI_CKPT_SCALARS 16
int magi_sampler_get_checkpoint(magi_handle* h, double* scalars /* [n_chains][MAGI_CKPT_SCALARS] */);
int magi_sampler_set_checkpoint(magi_handle* h, const double* scalars /* [n_chains][MAGI_CKPT_SCALARS] */);

/* One-call form: init + run(num_burnin + num_results) + get_samples. */
int magi_sample(magi_handle* h, const magi_sampler_cfg* cfg, int n_chains,
                const double* X0, const double* sig_pre0, const double* th_pre0,
                uint64_t seed, const int64_t* chain_ids,
                double* X_samps, double* sig_pre_samps, double* th_pre_samps);

/* theta initialiser of initial_fit (magi_v2.py:133-179): Adam(learning_rate) x num_iters, from theta[P] as passed
 * in (the reference starts at 1), on theta_objective = sum_d toNorm_d^T K_d^-1 toNorm_d with toNorm = reshape(f(Xhat, theta),
 * [D, N, 1]) - m_d (Xhat - mu)_d -- including the reference's reshape (:155-156) -- evaluated with the device-resident UNbanded
 * m and K^-1 stacks (the initialiser runs before the band approximation).  The whole loop stays on the device: one captured graph
 * per Adam step (drift values + theta-Jacobian, K^-1 r, K^-T r, reduction + update), the host waits once.  Xhat[N][D] row-major,
 * mu[D]; theta[P] in/out; loss_trace[num_iters] (optional) receives the objective at every step.  tf_keras Adam restated from its
 * documented defaults: parity unpinned. */
int magi_theta_init(magi_handle* h, int drift_id, int P, const double* Xhat, const double* mu, int num_iters, double learning_rate,
                    double* theta, double* loss_trace);

/* ---- multi-GPU ---------------------------------------------------------------------------------
 * There is deliberately NO magi_gather in this ABI (SURVEY 8b had listed one).  The path shards by independent (dataset, chain) units with
 * no exchange while sampling (one handle per GPU, one process per GPU); its only collective is ONE gather of the post-burn-in samples at
 * the end of a job, and that one lives where the job's process group already lives: magi_v2_amd/shard.py calls torch.distributed.gather
 * (backend "nccl" = RCCL over xGMI) on the arrays magi_sampler_get_samples returned.  A C-side gather would have to bootstrap a second
 * RCCL communicator (ncclGetUniqueId handed through the caller, ncclCommInitRank per handle) beside the one the host framework owns, for
 * 52 MB per GPU once per job (< 1 ms over xGMI).  A binding in another host language gathers the same host arrays with its own collective. */

/* ---- instrumentation (bench.py roofline leg) ----------------------------------------------- */

/* Launch one gradient evaluation (the 3 mat-vec phases + reduce) `reps` times on the handle's
 * stream, bracketed by HIP events on that stream; phase_ms[8] receives the average device
 * time per launch of phase 1, 2, 3, the reduce kernel, the sampler's streaming kernel (single-phase
 * block mat-vecs, k_stream), its reduce (k_leap_finalize), its point kernel (k_point) and, in [7], a
 * load-only pass over the same packed operator blocks with the same access pattern (the ceiling the
 * memory system offers k_stream for this layout and working set), measured in separate event-bracketed loops.  Uses the states currently on the
 * device (n_chains as last set). */
int magi_time_gradient(magi_handle* h, int n_chains, int reps, double* total_ms_per_eval, double* phase_ms);

/* Of the last magi_sampler_run: leapfrog slots (kernel pairs [k_stream, k_point]) issued before every chain had finished --
 * counted on the device -- and graph launches (of 128 slots each; MAGI_GRAPH_SLOTS) the host made.  A slot advances every unfinished chain by one
 * leaf or one set-up step, so slots_issued >= the leapfrogs of the busiest chain. */
int magi_sampler_run_stats(magi_handle* h, int64_t* slots_issued, int64_t* graphs_launched);

/* In-sampler kernel durations: continues the chains for n_slots leapfrog slots launched one by one (no graph) with HIP events
 * attached to every launch (hipExtLaunchKernel start / stop events = the kernel's own begin / end time stamps, the quantity
 * rocprofv3 --kernel-trace reports), decisions riding along as in production.  stream_us / point_us = mean device time per
 * launch of the streaming kernel and of k_point; leapfrogs_done = gradient evaluations taken (sum over chains).  The chains are
 * left inside a transition: the sampler must be re-initialised afterwards (MAGI_E_STATE otherwise). */
int magi_sampler_profile(magi_handle* h, int n_slots, double* stream_us, double* point_us, int64_t* leapfrogs_done);

/* phase_bytes[8]: bytes each of those seven kernels must move per launch for the current matrices
 * and n_chains, for the kernel family that serves a batch of n_chains (magi_stream_kernel_name):
 * [4] = the streaming kernel: packed operator blocks + what it stores (+ for k_stream_sep its operand planes, once per XCD);
 * [6] = k_point; [7] = the algorithmic bytes of one gradient evaluation as SURVEY 8d counts them (3 D N W 8 + C 10 N D 8). */
int magi_gradient_bytes(magi_handle* h, int n_chains, double* phase_bytes);

/* Name of the streaming kernel a batch of n_chains runs on with the current matrices, problem and options:
 * "k_stream<1>", "k_stream<2>" (VALU), "k_stream_sep<CW=8|16>" (matrix cores, separable drift), "k_stream_mc". */
int magi_stream_kernel_name(magi_handle* h, int n_chains, char* buf, int cap);

/* Tuning / test switches of a handle (csrc/magi_internal.h: MagiOptions).  The environment variables MAGI_STREAM_FAMILY,
 * MAGI_FAMILY_CHAINS, MAGI_SEP_PAIR_MIN, MAGI_FUSED_PARITY, MAGI_GEMM_REMAP_MIN, MAGI_POTRF_PANELS, MAGI_NO_GRAPH, MAGI_FIT_HOST_LOOP,
 * MAGI_FIT_PER_COMPONENT, MAGI_BUILD_PROFILE, MAGI_BUILD_SERIAL are read ONCE, by magi_create; afterwards only this call
 * changes an option (no getenv on a compute path).  Names: "stream_family" (0 auto, 1 mc, 2 valu; takes effect at the next
 * magi_sampler_init / log-posterior call), "family_chains" (> 0: "auto" chooses the kernel family as if the batch had this many chains -- a
 * sharded job passes its largest per-GPU share on every rank, so that a chain's samples do not depend on how many chains share its GPU),
 * "sep_pair_min" (next packing), "fused_parity", "gemm_remap_min", "potrf_panels", "potrf_lookahead_min",
 * "no_graph", "fit_host_loop", "fit_per_component", "build_profile", "build_serial", and the test hook
 * "slot_budget_graphs" (cap on the graph launches of one magi_sampler_run; 0 = the computed bound; no environment variable). */
int magi_set_option(magi_handle* h, const char* name, int64_t value);

/* Diagnostics: per-class device time of the last magi_build_matrices run with option "build_profile" set
 * (HIP events around every launch; the build is serialised while profiling).
 * Classes, in order: matern, diag-block Cholesky+inverse, potrf panel, potrf trailing SYRK, trtri,
 * T^T T, m / K products, single-phase operators.  flops = fp64 operations actually issued.  One more row follows the
 * classes and is filled by EVERY dense build of the handle, profiled or not: "potrf_wall" = the two blocked Cholesky
 * factorisations as a whole on the device clock (ms; flops = N^3 / 3 per matrix) -- the only figure that shows the
 * look-ahead of the factorisation (option "potrf_lookahead_min": grids from that size on fork their rank-k updates
 * to a second, CU-masked stream; 0 = never), which the serialising per-launch profile switches off.  Returns the
 * number of rows (9); arrays must hold at least 16 entries. */
int magi_build_profile(magi_handle* h, double* flops, double* ms, int64_t* calls);

/* Diagnostics: the 64-double transformed-parameter block of a chain (softplus / sigmoid / log
 * terms of the state in flight; in a -DMAGI_TAIL_STAMPS build entries 40.. hold timing stamps). */
int magi_debug_par(magi_handle* h, int chain, double* out64);

#ifdef __cplusplus
}
#endif
#endif /* MAGI_HIP_H */
