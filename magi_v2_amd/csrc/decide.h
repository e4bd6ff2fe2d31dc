// Decisions of one leapfrog slot: the device-resident NUTS / dual-averaging / annealing state machine.
//
// Replaces the TFP wiring of MAGI_v2.predict (magi_v2.py:357-396): NoUTurnSampler(step_size=0.1)
// inside DualAveragingStepSizeAdaptation(int(0.8*burnin), 0.75), wrapped by LogAnnealedNUTS
// (magi_v2.py:838-889, temperature schedule :833-835).  TFP itself is not in the reference tree;
// its published algorithm (iterative tree doubling with multinomial sampling and the generalised
// U-turn test, max_tree_depth 10, max_energy_diff 1000; Nesterov dual averaging) is restated in
// oracle/magi_oracle.py and mirrored here decision for decision, with a shared Philox4x32-10 RNG.
// Energies follow TFP: energy = target - 0.5 p.p  (minus the Hamiltonian).
//
// decide_block runs as ONE extra workgroup per chain inside k_stream (leap.hip), i.e. concurrently with
// the block mat-vecs of the NEXT slot: it adds the point phase's partial sums, finishes the D + P
// parameter entries, does all tree bookkeeping (multinomial proposal, checkpointed U-turn tests,
// doubling, merge, transition end, dual averaging, temperature, next momentum draw) and publishes the
// plan the next point phase executes.  The stream next to it has already assumed "same subtree, next
// leaf" (it derives theta' itself, leap_reduce.h); when the decisions end the subtree instead they set up
// the new state and mark the slot `skip`.
//
// It runs once per launch on one CU, on a cold instruction cache: its speed is set by code size along
// the executed path.  Hence one instantiation per drift (compile-time D, P), single call sites for the
// big helpers, noinline fp64 transcendentals, and the rare transition-end work behind one branch.
#pragma once
#include "magi_internal.h"
#include "leap_reduce.h"
#include "leap_point.h"

struct TailVecs {
    double *q, *p, *g, *pL, *qL, *gL, *pR, *qR, *gR, *candq, *candg, *subq, *subg, *rho, *rhosub, *ckp, *ckrho;
};

__device__ __forceinline__ TailVecs tail_vecs(const DevProblem& pb, double* vb) {
    const size_t s = pb.dimp;
    TailVecs v;
    v.q = vb + V_Q * s; v.p = vb + V_P * s; v.g = vb + V_G * s;
    v.pL = vb + V_PL * s; v.qL = vb + V_QL * s; v.gL = vb + V_GL * s;
    v.pR = vb + V_PR * s; v.qR = vb + V_QR * s; v.gR = vb + V_GR * s;
    v.candq = vb + V_CANDQ * s; v.candg = vb + V_CANDG * s;
    v.subq = vb + V_SUBQ * s; v.subg = vb + V_SUBG * s;
    v.rho = vb + V_RHO * s; v.rhosub = vb + V_RHOSUB * s;
    v.ckp = vb + V_CKP0 * s; v.ckrho = vb + V_CKRHO0 * s;
    return v;
}

// State-sized loops of the rare paths (subtree ends, transition ends), one workgroup over `dim` entries: written element by
// element (`dst[e] = src[e]`) every iteration is its own memory round trip -- DB elements' loads are issued together instead.
constexpr int DB = 4;        // (loops that keep seven vectors' elements in registers)
constexpr int DBC = 6;       // copies, sample set-up, doubling set-up: fewer vectors, more elements in flight
__device__ __forceinline__ void copy2_batched(double* d0, const double* s0, double* d1, const double* s1, int dim) {
    for (int e0 = threadIdx.x; e0 < dim; e0 += DBC * (int)blockDim.x) {
        double a[DBC], b[DBC];
#pragma unroll
        for (int u = 0; u < DBC; ++u) {
            const int e = e0 + u * (int)blockDim.x;
            if (e < dim) { a[u] = s0[e]; if (s1) b[u] = s1[e]; }
        }
#pragma unroll
        for (int u = 0; u < DBC; ++u) {
            const int e = e0 + u * (int)blockDim.x;
            if (e < dim) { d0[e] = a[u]; if (d1) d1[e] = b[u]; }
        }
    }
}

// DualAveragingStepSizeAdaptation.one_step after the inner NUTS step (oracle: dual_averaging_update).
// Evaluated by one thread; results returned through out[0..3].
__device__ __noinline__ void dual_averaging_eval(double target_accept, int n_adapt, int prev, double da_step_size, double da_error_sum,
                                                double da_log_avg, double da_log_shrink, double e_sum, int lf_count, double* out) {
    const double log_accept_ratio = m_log(e_sum / (double)lf_count);
    double lap = isfinite(log_accept_ratio) ? log_accept_ratio : -INFINITY;
    lap = fmin(lap, 0.0);
    const double accept = (lap > -INFINITY) ? m_exp(lap) : 0.0;
    const double t = (double)(prev + 1);
    double new_err = da_error_sum + target_accept - accept;
    const double soft_t = 10.0 + t;                              // step_count_smoothing
    const double new_log_step = da_log_shrink - (new_err * sqrt(t)) / (soft_t * 0.05);   // exploration_shrinkage
    const double eta = pow(t, -0.75);                            // decay_rate
    double new_log_avg = eta * new_log_step + (1.0 - eta) * da_log_avg;
    double new_ss;
    if (prev < n_adapt) new_ss = m_exp(new_log_step);
    else if (prev > n_adapt) new_ss = da_step_size;
    else new_ss = m_exp(new_log_avg);
    if (prev > n_adapt) { new_err = da_error_sum; new_log_avg = da_log_avg; }
    out[0] = log_accept_ratio;
    out[1] = new_ss;
    out[2] = new_err;
    out[3] = new_log_avg;
}

// plan of the leaf with index c.it of the current subtree (evaluated from buffer c.cur)
__device__ __forceinline__ LeafPlan make_leaf_plan(const ChainCtl& c, unsigned long long seed, bool hmc) {
    LeafPlan p{};
    p.active = 1;
    p.leaf = 1;
    p.cur = c.cur;
    const int it = c.it;
    p.even = (!hmc && (it & 1) == 0) ? 1 : 0;
    p.ck_slot = __popc((unsigned)it);
    int nk = 0;
    if (!hmc && (it & 1) != 0)
        for (int kk = 1; ((it + 1) & ((1 << kk) - 1)) == 0 && (1 << kk) <= c.nsteps; ++kk) nk = kk;
    p.nchk = min(nk, 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) p.chk_slot[k] = (k < p.nchk) ? __popc((unsigned)(it + 1 - (2 << k))) : 0;
    p.eps = c.dir * c.eps;
    p.hs = 0.5 * p.eps * c.beta_k;
    p.leaf_ctr = (unsigned)c.leaf_ctr;
    p.depth = (unsigned)c.depth;
    p.step_k = (unsigned)c.k;
    p.chain_id = (unsigned)c.chain_id;
    p.seed = seed;
    return p;
}

// All threads of a 256-thread workgroup call it; `parity` = slot whose stream is running next to these decisions
// (they complete slot parity ^ 1 ... i.e. the slot the point phase executed last, and write plan[parity]).
// SEPK: the kernel these decisions ride in is k_stream_sep (they then also write the operand mirror of a state they set up).  A template
// parameter, not a test of ch.sep: with the mirror pass merely COMPILED into the VALU kernels (never executed there) the one-chain SIRW
// instantiation sampled wrong energies (tools/exp_family_hmc.py; round 3) -- the decision path is at the edge of what the register
// allocator handles, and every instantiation is therefore held to an oracle run with deep trees (tests/test_sampler_gpu.py).
template <int DRIFT, bool SEPK = false>
__device__ __forceinline__ void decide_block(const DevProblem& pb, const DevChains& ch, const SamplerCfgDev& cfg, int chain, int parity, int all_done,
                                             double* sh /* 25*16 */, double* shs /* 24 */, ChainCtl* s_ctl, int* s_g, double* s_par /* PAR_COUNT */, double* s_ops /* OPS_COUNT * OPS_W */, double* s_cst /* 3 * MAGI_MAX_D: N_ds, LB, mu */) {
    const int tid = threadIdx.x;
    MAGI_STAMP(ch.par, 8);
    // ONE round of loads with no dependence on anything: control state, the previous plan, the state's parameter block
    // and the point phase's partial sums.  These decisions run next to a stream that saturates the memory system, where
    // every dependent round trip costs several microseconds -- their number, not the instruction count, is what matters.
    static_assert(sizeof(ChainCtl) % 4 == 0 && sizeof(ChainCtl) / 4 <= PT_THREADS - 2, "ChainCtl staging");
    // (all loads into registers first, the LDS stores after the last of them: a load followed by its own store is a round trip)
    int ctl_w = 0;
    if (tid < (int)(sizeof(ChainCtl) / 4)) ctl_w = reinterpret_cast<const int*>(ch.ctl + chain)[tid];
    else if (tid == PT_THREADS - 2) ctl_w = ch.gctl->stop_k;
    else if (tid == PT_THREADS - 1) ctl_w = ch.gctl->epoch;
    LeafPlan* plan_out = ch.plan + (size_t)parity * ch.n_chains + chain;
    const LeafPlan lp = ch.plan[(size_t)(parity ^ 1) * ch.n_chains + chain];     // what the point phase executed last for this chain
    const double par_v = (tid >= 64 && tid < 64 + PAR_COUNT) ? ch.par[(size_t)chain * PAR_COUNT + (tid - 64)] : 0.0;
    double pre[RedLayout<DRIFT>::PER_WAVE];
    leap_reduce_issue<DRIFT>(ch, chain, pre);
    static_assert(OPS_W >= MAGI_MAX_D + MAGI_MAX_P, "operand prefetch");
    constexpr int OPS_PER = (OPS_COUNT * OPS_W + PT_THREADS - 1) / PT_THREADS;
    double ops_v[OPS_PER];
    reduce_prefetch_ops_load<OPS_PER>(pb, ch.vec + vec_off(pb, chain, 0), DriftT<DRIFT>::D + DriftT<DRIFT>::P, ops_v);
    __builtin_amdgcn_sched_barrier(0);
    if (all_done) return;
    if (chain == 0 && tid == PT_THREADS - 4) __hip_atomic_fetch_add(&ch.gctl->slots, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (no return value: fire and forget)
    if (tid < (int)(sizeof(ChainCtl) / 4)) reinterpret_cast<int*>(s_ctl)[tid] = ctl_w;
    else if (tid == PT_THREADS - 2) s_g[0] = ctl_w;
    else if (tid == PT_THREADS - 1) s_g[1] = ctl_w;
    if (tid >= 64 && tid < 64 + PAR_COUNT) s_par[tid - 64] = par_v;
    if (tid == PT_THREADS - 3) {
#pragma unroll
        for (int k = 0; k < MAGI_MAX_D; ++k) { s_cst[k] = pb.N_ds[k]; s_cst[MAGI_MAX_D + k] = pb.LB[k]; s_cst[2 * MAGI_MAX_D + k] = pb.mu[k]; }      // (static indices)
    }
    reduce_prefetch_ops_store<OPS_PER>(ops_v, s_ops);
    MAGI_STAMP(ch.par, 2);
    __syncthreads();
    MAGI_STAMP(ch.par + (size_t)chain * PAR_COUNT, 0);
    ChainCtl c = *s_ctl;
    const int dim = pb.dim;
    const int stop_k = min(s_g[0], cfg.total);
    const int epoch = s_g[1];
    double* vb = ch.vec + vec_off(pb, chain, 0);
    double* par = ch.par + (size_t)chain * PAR_COUNT;
    const TailVecs v = tail_vecs(pb, vb);
    const size_t sv = pb.dimp;

    bool do_sample = false, do_doubling = false;     // what to set up before returning
    const bool hmc = cfg.mode == MAGI_MODE_HMC;       // fixed-L HMC: one forward "subtree" of L leaves, Metropolis at its end

    if (lp.active && lp.skip) {
        // the state set up by the previous decisions is only now being evaluated by the stream next to us: publish
        // what the point phase has to do with it
        if (tid == 0) {
            LeafPlan p{};
            if (c.phase == PH_LEAF) p = make_leaf_plan(c, cfg.seed, hmc);
            else if (c.phase == PH_INIT) { p.active = 1; p.cur = c.cur; }       // bootstrap gradient, no leapfrog
            *plan_out = p;
        }
        return;
    }
    // Batches of NUTS chains (three or more per GPU): a subtree end (merge pass, then the set-up pass of the next doubling) or a
    // transition end (merge, sample, momentum draw, set-up) keeps ONE workgroup busy for 25-100 us while every other chain's slot
    // waits for the kernel to end -- with eight chains a quarter of the slots carried such a tail (k_stream_mc 26.2 us in the sampler
    // against 22.7 us by itself).  There the passes go into consecutive slots, each short enough to hide under the stream: the chain
    // publishes an idle plan in between (its point phase and the stream's output for it are off), at the price of one or two more
    // set-up slots per subtree for that chain.  Fixed-L HMC chains end their transitions in step: nothing to gain, left alone.
    const bool spread = !hmc && ch.n_chains >= 3;
    const int phase_in = c.phase;
    if (c.phase == PH_PEND_DOUBLE) {
        do_doubling = true;
    } else if (c.phase == PH_PEND_SAMPLE) {
        do_sample = true;
    } else if (c.phase == PH_IDLE) {
        if (c.k < stop_k) {
            do_sample = true;                        // resumed by a later magi_sampler_run
        } else {
            if (tid == 0) { LeafPlan off{}; *plan_out = off; }       // (both ring entries must read "idle")
            if (c.done_epoch != epoch) {
                c.done_epoch = epoch;
                if (tid == 0) {
                    ch.ctl[chain] = c;
                    const int done = atomicAdd(&ch.gctl->done_chains, 1) + 1;
                    if (done >= ch.gctl->n_chains) ch.gctl->all_done = 1;
                }
            }
            return;
        }
    } else {
        const bool leaf = lp.leaf != 0;
        MAGI_STAMP(par, 1);
        // the leaf's two uniform draws (off the critical path here: these decisions run next to the following stream)
        if (leaf && tid == 128) shs[21] = m_log1p(-rng_uniform(lp.leaf_ctr, lp.step_k, lp.chain_id, STREAM_LEAF, lp.seed));
        if (leaf && tid == 192) shs[22] = m_log1p(-rng_uniform(lp.depth, lp.step_k, lp.chain_id, STREAM_MERGE, lp.seed));
        if (!leaf && tid == 64) shs[16] = cfg.anneal ? temperature(0, cfg.min_temp) : 1.0;
        // ---- add the streaming kernel's partial sums, finish the parameter entries -----------------------
        const ReduceOut ro = leap_reduce<DRIFT>(pb, ch, chain, vb, par, s_par, lp, pre, sh, shs, s_ops, s_cst);
        MAGI_STAMP(par, 4);
        const double L = ro.L;
        const double u_leaf = shs[21], u_merge = shs[22];        // (published by the barriers inside leap_reduce)
        double* qcur = vb + (size_t)(V_Q + lp.cur) * sv;      // the state that was evaluated
        double* pleaf = vb + (size_t)V_PLEAF * sv;

        if (!leaf) {
            // bootstrap_results: target / gradient at the initial state, cached at beta_temp(0)
            copy2_batched(v.candq, qcur, v.candg, v.g, dim);
            c.cand_L = L;
            c.beta_cache = shs[16];
            if (c.k < stop_k) do_sample = true;
            else c.phase = PH_IDLE;
        } else {
            c.L_cur = L;
            c.total_leapfrogs += 1;
            const int it = c.it;
            int nk = 0;
            if (!hmc && (it & 1) != 0)
                for (int kk = 1; ((it + 1) & ((1 << kk) - 1)) == 0 && (1 << kk) <= c.nsteps; ++kk) nk = kk;
            bool no_u = true;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nk) no_u = no_u && (ro.dA[k] > 0.0) && (ro.dB[k] > 0.0);
            for (int kk = 5; kk <= nk; ++kk) {       // one leaf in 32 gets here
                const int left = it + 1 - (1 << kk);
                const double* cp = v.ckp + (size_t)__popc((unsigned)left) * sv;
                const double* cr = v.ckrho + (size_t)__popc((unsigned)left) * sv;
                double dots[2] = {0.0, 0.0};
                constexpr int DBU = 8;           // (no stores: eight elements' loads of the four vectors in flight together -- two round trips at dim = 4 103)
                for (int e0 = tid; e0 < dim; e0 += DBU * (int)blockDim.x) {
                    double rs[DBU], crv[DBU], cpv[DBU], plv[DBU];
#pragma unroll
                    for (int u = 0; u < DBU; ++u) {
                        const int e = min(e0 + u * (int)blockDim.x, dim - 1);
                        rs[u] = v.rhosub[e]; crv[u] = cr[e]; cpv[u] = cp[e]; plv[u] = pleaf[e];
                    }
#pragma unroll
                    for (int u = 0; u < DBU; ++u) {
                        if (e0 + u * (int)blockDim.x < dim) {
                            const double df = rs[u] - crv[u];
                            dots[0] = fma(df, cpv[u], dots[0]);
                            dots[1] = fma(df, plv[u], dots[1]);
                        }
                    }
                }
                block_sum<2>(dots, sh);
                no_u = no_u && (dots[0] > 0.0) && (dots[1] > 0.0);
            }

            // ---- energy, multinomial proposal inside the subtree --------------------------------------
            double energy = c.beta_k * L - 0.5 * ro.pp;
            if (isnan(energy)) energy = -INFINITY;
            const double ediff = energy - c.init_energy;
            const bool not_divergent = (-ediff < cfg.max_energy_diff);
            if (tid == 0) shs[18] = logaddexp(c.sub_weight, ediff);
            if (tid == 192) shs[19] = m_exp(fmin(ediff, 0.0));
            __syncthreads();
            MAGI_STAMP(par, 5);
            const double wsum_leaf = shs[18];
            const bool accept_leaf = !hmc && (u_leaf <= ediff - wsum_leaf);
            c.leaf_ctr += 1;
            if (accept_leaf) { c.sub_L = L; c.sub_energy = energy; }
            c.sub_weight = wsum_leaf;
            const bool cont_tree = hmc || (not_divergent && (c.cont != 0));
            c.cont = (no_u && cont_tree) ? 1 : 0;
            c.nd = (c.nd && not_divergent) ? 1 : 0;
            if (cont_tree) c.e_sum_sub += shs[19];
            c.sub_lf += 1;
            c.it = it + 1;

            if (c.it < c.nsteps && c.cont) {
                // ---- the speculative next leaf stands: flip buffers, publish its plan -------------------------
                if (accept_leaf) {                   // proposal copy (expected O(log n) times per subtree)
                    copy2_batched(v.subq, qcur, v.subg, v.g, dim);
                }
                c.cur = lp.cur ^ 1;
                if (tid == 0) {
                    *plan_out = make_leaf_plan(c, cfg.seed, hmc);
                    ch.ctl[chain] = c;
                }
                MAGI_STAMP(par, 6);
                MAGI_STAMP_FLUSH(par);
                return;
            }

            // ---- subtree finished: merge into the trajectory (biased progressive sampling); for HMC the
            //      "subtree" is the whole trajectory and the merge is the Metropolis test on its last state ----
            const double tree_weight = c.cont ? c.sub_weight : -INFINITY;
            if (tid == 0) shs[20] = hmc ? 0.0 : logaddexp(tree_weight, c.cand_weight);
            const double thresh = hmc ? ediff : tree_weight - c.cand_weight;
            const bool choose = (u_merge <= thresh) && (hmc ? not_divergent : (c.cont != 0));
            double* pe = (c.dir > 0) ? v.pR : v.pL;
            double* qe = (c.dir > 0) ? v.qR : v.qL;
            double* ge = (c.dir > 0) ? v.gR : v.gL;
            const double* po_ = (c.dir > 0) ? v.pL : v.pR;    // the other end
            double dots[2] = {0.0, 0.0};
            const bool take_leaf = accept_leaf || hmc;
            constexpr int DBM = 4;
            for (int e0 = tid; e0 < dim; e0 += DBM * (int)blockDim.x) {       // DBM elements' loads in flight, then their stores
                double qv[DBM], gv[DBM], sq[DBM], sg[DBM], pn[DBM], rr[DBM], po[DBM];
#pragma unroll
                for (int u = 0; u < DBM; ++u) {
                    const int e = e0 + u * (int)blockDim.x;
                    if (e < dim) {
                        qv[u] = qcur[e]; gv[u] = v.g[e];
                        // the subtree proposal is this leaf if it was just accepted, else what V_SUB holds
                        sq[u] = take_leaf ? qv[u] : v.subq[e]; sg[u] = take_leaf ? gv[u] : v.subg[e];
                        pn[u] = pleaf[e];
                        rr[u] = v.rho[e] + v.rhosub[e];
                        po[u] = po_[e];
                    }
                }
#pragma unroll
                for (int u = 0; u < DBM; ++u) {
                    const int e = e0 + u * (int)blockDim.x;
                    if (e < dim) {
                        if (choose) { v.candq[e] = sq[u]; v.candg[e] = sg[u]; }
                        pe[e] = pn[u]; qe[e] = qv[u]; ge[e] = gv[u];
                        v.rho[e] = rr[u];
                        dots[0] = fma(rr[u], po[u], dots[0]);
                        dots[1] = fma(rr[u], pn[u], dots[1]);
                    }
                }
            }
            block_sum<2>(dots, sh);        // (its barriers publish shs[20])
            if (hmc) { c.sub_L = L; c.sub_energy = energy; }
            if (choose) { c.cand_L = c.sub_L; c.cand_energy = c.sub_energy; c.cand_bfac = c.beta_k; c.is_accepted = 1; }
            c.cand_weight = shs[20];
            if (c.dir > 0) { c.LR = c.L_cur; c.bfacR = c.beta_k; } else { c.LL = c.L_cur; c.bfacL = c.beta_k; }
            const bool no_u_traj = (dots[0] > 0.0) && (dots[1] > 0.0);
            c.e_sum += c.e_sum_sub;
            c.lf_count += c.sub_lf;
            c.not_div = c.nd;
            c.depth += 1;
            const bool continue_tree = !hmc && (c.cont != 0) && no_u_traj;
            if (hmc) { c.e_sum = shs[19]; }          // acceptance statistic of HMC: min(1, exp(energy difference))
            if (c.depth < cfg.max_depth && continue_tree) {
                if (spread) {
                    c.phase = PH_PEND_DOUBLE;
                    if (tid == 0) { LeafPlan off{}; *plan_out = off; ch.ctl[chain] = c; }
                    return;
                }
                do_doubling = true;
            } else {
                // ---- transition finished ------------------------------------------------------------------
                if (tid == 0)
                    dual_averaging_eval(cfg.target_accept, cfg.n_adapt, c.da_step, c.da_step_size, c.da_error_sum, c.da_log_avg,
                                        c.da_log_shrink, c.e_sum, hmc ? 1 : c.lf_count, &shs[0]);
                __syncthreads();
                const int k = c.k;
                if (tid == 0) {
                    const size_t o = (size_t)chain * cfg.total + k;
                    ch.d_step_size[o] = c.eps;
                    ch.d_lar[o] = shs[0];
                    ch.d_target[o] = c.cand_bfac * c.cand_L;
                    ch.d_energy[o] = c.cand_energy;
                    ch.d_beta[o] = c.beta_k;
                    ch.d_leapfrogs[o] = c.lf_count;
                    ch.d_depth[o] = c.depth;
                    ch.d_flags[o] = (c.not_div ? 0 : 1) | (continue_tree ? 2 : 0) | (c.is_accepted ? 4 : 0);
                }
                if (k >= cfg.burnin) {
                    double* out = ch.samples + ((size_t)chain * (cfg.total - cfg.burnin) + (k - cfg.burnin)) * pb.dimp;
                    copy2_batched(out, v.candq, nullptr, nullptr, dim);
                }
                if (c.is_accepted) c.beta_cache = c.beta_k;
                c.da_step_size = shs[1];
                c.da_error_sum = shs[2];
                c.da_log_avg = shs[3];
                c.da_step += 1;
                c.k = k + 1;
                if (c.k >= stop_k) {
                    c.phase = PH_IDLE;
                    c.done_epoch = epoch;
                    if (tid == 0) {
                        LeafPlan off{};
                        *plan_out = off;
                        ch.ctl[chain] = c;
                        const int done = atomicAdd(&ch.gctl->done_chains, 1) + 1;
                        if (done >= ch.gctl->n_chains) ch.gctl->all_done = 1;
                    }
                    return;
                }
                if (spread) {
                    c.phase = PH_PEND_SAMPLE;
                    if (tid == 0) { LeafPlan off{}; *plan_out = off; ch.ctl[chain] = c; }
                    return;
                }
                do_sample = true;
            }
        }
    }

    if (do_sample) {
        // ---- start transition k: temperature, momentum draw, both ends = current proposal ----------------
        __syncthreads();
        if (tid == 0) shs[4] = cfg.anneal ? temperature(c.k, cfg.min_temp) : 1.0;
        double pp0[1] = {0.0};
        // (pair j = elements 2 j, 2 j + 1 of the state: one Philox block, one log / sqrt per pair; 16-B accesses -- every vector is
        //  16-B aligned and padded to an even length)
        constexpr int DBP = 4;
        const int npair = (dim + 1) >> 1;
        for (int j0 = tid; j0 < npair; j0 += DBP * (int)blockDim.x) {
            double2 qq[DBP], gg[DBP];
#pragma unroll
            for (int u = 0; u < DBP; ++u) {
                const int j = j0 + u * (int)blockDim.x;
                if (j < npair) { qq[u] = reinterpret_cast<const double2*>(v.candq)[j]; gg[u] = reinterpret_cast<const double2*>(v.candg)[j]; }
            }
#pragma unroll
            for (int u = 0; u < DBP; ++u) {
                const int j = j0 + u * (int)blockDim.x;
                if (j < npair) {
                    double2 z;
                    rng_normal_pair((unsigned)j, (unsigned)c.k, (unsigned)c.chain_id, cfg.seed, z.x, z.y);
                    if (2 * j + 1 >= dim) z.y = 0.0;                 // (odd dim: the pad entry stays zero)
                    pp0[0] = fma(z.x, z.x, pp0[0]);
                    pp0[0] = fma(z.y, z.y, pp0[0]);
                    reinterpret_cast<double2*>(v.pL)[j] = z; reinterpret_cast<double2*>(v.pR)[j] = z; reinterpret_cast<double2*>(v.rho)[j] = z;
                    reinterpret_cast<double2*>(v.qL)[j] = qq[u]; reinterpret_cast<double2*>(v.qR)[j] = qq[u];
                    reinterpret_cast<double2*>(v.gL)[j] = gg[u]; reinterpret_cast<double2*>(v.gR)[j] = gg[u];
                }
            }
        }
        block_sum<1>(pp0, sh);              // (its barriers also publish shs[4])
        c.beta_k = shs[4];
        const double bc = cfg.stale ? c.beta_cache : c.beta_k;
        c.eps = c.da_step_size;
        c.init_energy = bc * c.cand_L - 0.5 * pp0[0];
        c.LL = c.LR = c.cand_L;
        c.bfacL = c.bfacR = bc;
        c.cand_bfac = bc;
        c.cand_energy = c.init_energy;
        c.cand_weight = 0.0;
        c.e_sum = 0.0;
        c.lf_count = 0;
        c.not_div = 1;
        c.is_accepted = 0;
        c.depth = 0;
        c.leaf_ctr = 0;
        if (spread && phase_in == PH_PEND_SAMPLE) {
            c.phase = PH_PEND_DOUBLE;
            if (tid == 0) { LeafPlan off{}; *plan_out = off; ch.ctl[chain] = c; }
            return;
        }
        do_doubling = true;
    }

    if (do_doubling) {
        // ---- start a doubling from the end selected by the direction bit; first half/full step ----------
        // (leapfrog with identity mass: p_half = p + eps/2 * grad ; q' = q + eps * p_half)
        Philox4 r = philox4x32_10((unsigned)c.depth, (unsigned)c.k, (unsigned)c.chain_id, STREAM_DIRECTION, cfg.seed);
        const bool fwd = hmc || (r.x & 1u) != 0;
        c.dir = fwd ? 1 : -1;
        const double* pe = fwd ? v.pR : v.pL;
        const double* qe = fwd ? v.qR : v.qL;
        const double* ge = fwd ? v.gR : v.gL;
        const double bf = fwd ? c.bfacR : c.bfacL;
        const double eps = c.dir * c.eps;
        const double hs = 0.5 * eps * bf;
        c.cur ^= 1;
        double* qw = vb + (size_t)(V_Q + c.cur) * sv;
        double* pw = vb + (size_t)(V_P + c.cur) * sv;
        for (int e0 = tid; e0 < dim; e0 += DBC * (int)blockDim.x) {
            double p0[DBC], g0[DBC], q0[DBC];
#pragma unroll
            for (int u = 0; u < DBC; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                if (e < dim) { p0[u] = pe[e]; g0[u] = ge[e]; q0[u] = qe[e]; }
            }
#pragma unroll
            for (int u = 0; u < DBC; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                if (e < dim) {
                    const double ph = p0[u] + hs * g0[u];
                    pw[e] = ph;
                    const double qn = q0[u] + eps * ph;
                    qw[e] = qn;
                    if (ch.mc && e < pb.ND) { const int dd = e / pb.N; ch.xop[xop_off(pb, ch.n_chains, parity ^ 1, chain, dd, e - dd * pb.N)] = qn; }   // (read by the next slot's stream)
                    v.rhosub[e] = 0.0;
                    if (e >= pb.ND) compute_par_entry(pb, e - pb.ND, qn, par, true, s_cst + MAGI_MAX_D);
                }
            }
        }
        if constexpr (SEPK) {
            // operand mirror of the new state for the next slot's stream (k_stream_sep): xc and the basis values phi_{d,k} need ALL
            // components of a grid point, so the positions are formed again per point (same expressions, same rounding as above)
            using DR = DriftT<DRIFT>;
            constexpr int DD = DR::D, NBM = DR::NBMAX;
            const int cw = xop_width(ch.n_chains), groups = (ch.n_chains + 15) >> 4, cl = chain & 15;
            const int planes = 1 + (NBM * cw + 15) / 16;
            for (int i = tid; i < pb.N; i += (int)blockDim.x) {
                double xq[DD], ph[DD][NBM];
#pragma unroll
                for (int dd = 0; dd < DD; ++dd) {
                    const int e = dd * pb.N + i;
                    const double phh = pe[e] + hs * ge[e];
                    xq[dd] = qe[e] + eps * phh;
                }
                DR::basis(xq, ph);
#pragma unroll
                for (int dd = 0; dd < DD; ++dd) {
                    double* m0 = ch.vop + vop_off(DD, planes, pb.Np, groups, parity ^ 1, chain >> 4, dd, 0, i);
                    m0[cl] = xq[dd] - s_cst[2 * MAGI_MAX_D + dd];
#pragma unroll
                    for (int k = 0; k < NBM; ++k)
                        if (k < DR::nbasis(dd))
                            m0[(size_t)(cw == 8 ? 1 + (k >> 1) : 1 + k) * pb.Np * 16 + (cw == 8 ? (k & 1) * 8 + cl : cl)] = ph[dd][k];
                }
            }
        }
        c.nsteps = hmc ? cfg.hmc_L : (1 << c.depth);
        c.it = 0;
        c.sub_weight = -INFINITY;
        c.e_sum_sub = 0.0;
        c.sub_lf = 0;
        c.cont = 1;
        c.nd = c.not_div;
        c.phase = PH_LEAF;
        if (tid == 0) {
            LeafPlan p{};                // the stream next to us is not looking at this state: skip slot, then evaluate it as is
            p.active = 1; p.skip = 1; p.cur = c.cur;
            *plan_out = p;
        }
    } else if (tid == 0) {
        LeafPlan off{};                  // idle after the bootstrap gradient
        *plan_out = off;
    }
    if (tid == 0) ch.ctl[chain] = c;
}

