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
// doubling, merge, transition end, dual averaging, temperature) and publishes the plan the next point
// phase executes.  The stream next to it has already assumed "same subtree, next leaf" (it derives
// theta' itself, leap_reduce.h); when the decisions end the subtree instead, the plan orders a BOUNDARY OP
// (LeafPlan::vop; leap_point.h: boundary_block): every state-sized pass of a subtree's or a transition's
// end -- merge of the ends, rho, proposal copy, sample row, momentum draw, the next doubling's first state --
// is the point kernel's, the sums it leaves (U-turn, p.p) arrive here with the next slot (PH_MERGED, PH_DRAWN).
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

// The state-sized loop left here: the checkpoint tests of levels >= 5 (one leaf in 32), one workgroup over `dim` entries.

// DualAveragingStepSizeAdaptation.one_step after the inner NUTS step (oracle: dual_averaging_update).
// Evaluated by one thread; results returned through out[0..3].  A function of its own (its five transcendentals stay out of the kernels'
// bodies) and a LEAF (log / exp / pow are spelled directly here, not as the shared noinline m_log / m_exp: they expand inside it) -- with calls of its own it kept a value in a callee-saved VGPR, whose
// spill slot was the last scratch reservation of the stream kernels (16 B per lane).
__device__ __noinline__ void dual_averaging_eval(double target_accept, int n_adapt, int prev, double da_step_size, double da_error_sum,
                                                double da_log_avg, double da_log_shrink, double e_sum, int lf_count, double* out) {
    const double log_accept_ratio = log(e_sum / (double)lf_count);
    double lap = isfinite(log_accept_ratio) ? log_accept_ratio : -INFINITY;
    lap = fmin(lap, 0.0);
    const double accept = (lap > -INFINITY) ? exp(lap) : 0.0;
    const double t = (double)(prev + 1);
    double new_err = da_error_sum + target_accept - accept;
    const double soft_t = 10.0 + t;                              // step_count_smoothing
    const double new_log_step = da_log_shrink - (new_err * sqrt(t)) / (soft_t * 0.05);   // exploration_shrinkage
    const double eta = pow(t, -0.75);                            // decay_rate
    double new_log_avg = eta * new_log_step + (1.0 - eta) * da_log_avg;
    double new_ss;
    if (prev < n_adapt) new_ss = exp(new_log_step);
    else if (prev > n_adapt) new_ss = da_step_size;
    else new_ss = exp(new_log_avg);
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

// Field by field: `*out = p` is an aggregate copy, which keeps the fields a caller never touched (the zero tail of a `LeafPlan p{}`) in
// memory -- 32 bytes of scratch per lane in every stream kernel, two stores and two loads on the decision path (round 3's resource table).
__device__ __forceinline__ void plan_store(LeafPlan* o, const LeafPlan& p) {
    static_assert(sizeof(LeafPlan) == 120, "plan_store copies every field: extend it with the struct");
    o->active = p.active; o->skip = p.skip; o->leaf = p.leaf; o->cur = p.cur; o->even = p.even; o->ck_slot = p.ck_slot; o->nchk = p.nchk;
    o->chk_slot[0] = p.chk_slot[0]; o->chk_slot[1] = p.chk_slot[1]; o->chk_slot[2] = p.chk_slot[2]; o->chk_slot[3] = p.chk_slot[3];
    o->leaf_ctr = p.leaf_ctr; o->depth = p.depth; o->step_k = p.step_k; o->chain_id = p.chain_id; o->hs = p.hs; o->eps = p.eps; o->seed = p.seed;
    o->vop = p.vop; o->vleaf = p.vleaf; o->vdir = p.vdir; o->ndir = p.ndir; o->vout = p.vout; o->sub_copy = p.sub_copy;
}

// All threads of a 256-thread workgroup call it; `parity` = slot whose stream is running next to these decisions
// (they complete slot parity ^ 1 ... i.e. the slot the point phase executed last, and write plan[parity]).
// (Round 3: with a pass for another kernel family merely COMPILED into the one-chain SIRW instantiation -- never executed -- it sampled wrong
// energies (tools/exp_family_hmc.py).  Diagnosed in round 4: a hipcc (ROCm 7.2) defect, not undefined behaviour of this source.  Behind
// `if (tid == 0) shs[4] = temperature(..)` the register allocator's live-range split copies of three wave-uniform values kept in vector
// registers -- the chain's cached temperature among them -- were placed IN FRONT of the join block's `s_or_b64 exec, exec, sN`, i.e. they ran
// for lane 0 only; lanes 1..255 then computed the doubling's first half step with hs = 0.5 eps beta = 0.  The copies appear only where
// calls to the noinline transcendentals, inter-procedural register allocation and SGPR spills into VGPR lanes meet (each of
// -mllvm -enable-ipra=false, -mllvm -amdgpu-spill-sgpr-to-vgpr=false, inlined m_exp / m_log removes them).  Guard: magi_v2_amd/isa_check.py
// scans the ISA of every translation unit at build time for vector instructions in front of a join block's EXEC restore; a flagged unit is
// rebuilt without IPRA, and the build fails if the shape persists.  Every instantiation is still held to an oracle run with deep trees.)
template <int DRIFT>
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

    const bool hmc = cfg.mode == MAGI_MODE_HMC;       // fixed-L HMC: one forward "subtree" of L leaves, Metropolis at its end
    // What this pass orders from the slot's point phase (boundary_block, leap_point.h): every state-sized pass of a subtree's or a
    // transition's end is element by element and runs there, on all workgroups of the point kernel -- here it was a loop of dependent
    // round trips on ONE workgroup next to the saturating stream (25-100 us per pass while every chain's slot waited for the kernel to
    // end; round 2 spread the passes of chain batches over extra slots).  These decisions keep the scalars.
    int vop = 0;
    bool finish = false;          // transition c.k ends with this pass
    bool start = false;           // a transition starts (temperature, momentum draw, first doubling)
    bool doubling = false;        // a doubling starts
    bool continue_tree = false;
    int vleaf = lp.cur, vdir = 0;

    if (lp.vop != 0 && (c.phase == PH_DRAWN || c.phase == PH_MERGED)) {
        // ---- sums of the boundary op the point phase has just executed (fixed order: lane = workgroup, butterfly) ----
        constexpr int K0 = RedLayout<DRIFT>::K0;
        {
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int i = 0; i < RedLayout<DRIFT>::PER_WAVE; ++i) {
                const int k = wave + RED_WAVES * i;
                const double vsum = wave_sum(pre[i]);
                if (lane == 0 && k < K0 + 8) sh[k] = vsum;
            }
            __syncthreads();
        }
        const double pp0 = sh[K0 - 1], dot0 = sh[K0], dot1 = sh[K0 + 1];
        if (c.phase == PH_DRAWN) {
            // the momentum of transition c.k is drawn and its first state set up (the stream next to us is evaluating it)
            c.init_energy = c.cand_bfac * c.cand_L - 0.5 * pp0;
            c.cand_energy = c.init_energy;
            c.phase = PH_LEAF;
            if (tid == 0) { plan_store(plan_out, make_leaf_plan(c, cfg.seed, hmc)); ch.ctl[chain] = c; }
            return;
        }
        // PH_MERGED: the subtree is merged; does the trajectory go on?
        continue_tree = (dot0 > 0.0) && (dot1 > 0.0);          // (c.cont != 0 and not HMC, or this phase would not have been entered)
        if (continue_tree && c.depth < cfg.max_depth) {
            c.phase = PH_LEAF;                                   // the doubling set up speculatively with the merge stands
            if (tid == 0) { plan_store(plan_out, make_leaf_plan(c, cfg.seed, hmc)); ch.ctl[chain] = c; }
            return;
        }
        finish = true;                                           // (the proposal was merged with the subtree: nothing of VOP_CAND left)
    } else if (c.phase == PH_IDLE) {
        if (c.k < stop_k) {
            start = true;                            // resumed by a later magi_sampler_run
        } else {
            if (tid == 0) { LeafPlan off{}; plan_store(plan_out, off); }       // (both ring entries must read "idle")
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
    } else if (lp.active && lp.skip) {
        // (slot 0 of a fresh sampler: the stream next to us evaluates the initial state, the point phase gets the bootstrap plan)
        if (tid == 0) {
            LeafPlan p{};
            if (c.phase == PH_LEAF) p = make_leaf_plan(c, cfg.seed, hmc);
            else if (c.phase == PH_INIT) { p.active = 1; p.cur = c.cur; }       // bootstrap gradient, no leapfrog
            plan_store(plan_out, p);
        }
        return;
    } else {
        const bool leaf = lp.leaf != 0;
        MAGI_STAMP(par, 1);
        if (!leaf && tid == 64) shs[16] = cfg.anneal ? temperature(0, cfg.min_temp) : 1.0;
        // ---- add the streaming kernel's partial sums, finish the parameter entries -----------------------
        const ReduceOut ro = leap_reduce<DRIFT>(pb, ch, chain, vb, par, s_par, lp, pre, sh, shs, s_ops, s_cst);
        MAGI_STAMP(par, 4);
        const double L = ro.L;
        double* qcur = vb + (size_t)(V_Q + lp.cur) * sv;      // the state that was evaluated
        double* pleaf = vb + (size_t)V_PLEAF * sv;

        if (!leaf) {
            // bootstrap_results: target / gradient at the initial state, cached at beta_temp(0); the proposal <- the initial state
            vop |= VOP_CAND | VOP_TAKE_LEAF;
            c.cand_L = L;
            c.beta_cache = shs[16];
            if (c.k < stop_k) start = true;
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
            // The transcendental work of a leaf, one piece per wave, between two barriers (this workgroup is the longest of the kernel it
            // rides in whenever the grid is small -- device stamps: its time is its chain of dependent exp / log calls):
            //   wave 0: log-sum-exp of the subtree weight | wave 1: transform of the next state's parameter entries (exp -> log)
            //   wave 2: the leaf's two uniform draws (Philox, log1p; lanes 128, 129) | wave 3: exp of the energy difference
            if (tid == 0) shs[18] = logaddexp(c.sub_weight, ediff);
            if (tid == 192) shs[19] = m_exp(fmin(ediff, 0.0));
            if (tid == 128 || tid == 129)
                shs[21 + (tid - 128)] = m_log1p(-rng_uniform(tid == 128 ? lp.leaf_ctr : lp.depth, lp.step_k, lp.chain_id, tid == 128 ? STREAM_LEAF : STREAM_MERGE, lp.seed));
            leap_next_par<DRIFT>(pb, par, sh, s_cst + MAGI_MAX_D);
            __syncthreads();
            MAGI_STAMP(par, 5);
            const double u_leaf = shs[21], u_merge = shs[22];
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
                // an accepted leaf becomes the subtree's proposal (expected O(log n) times per subtree): the NEXT leaf's point phase copies its
                // position and gradient (LeafPlan::sub_copy) before overwriting them; the D + P parameter entries are copied here
                if (accept_leaf && tid < DriftT<DRIFT>::D + DriftT<DRIFT>::P) {
                    v.subq[pb.ND + tid] = qcur[pb.ND + tid];
                    v.subg[pb.ND + tid] = v.g[pb.ND + tid];
                }
                c.cur = lp.cur ^ 1;
                if (tid == 0) {
                    LeafPlan pn = make_leaf_plan(c, cfg.seed, hmc);
                    pn.sub_copy = accept_leaf ? 1 : 0;
                    plan_store(plan_out, pn);
                    ch.ctl[chain] = c;
                }
                MAGI_STAMP(par, 6);
                MAGI_STAMP_FLUSH(par);
                return;
            }

            // ---- subtree finished: merge into the trajectory (biased progressive sampling); for HMC the
            //      "subtree" is the whole trajectory and the merge is the Metropolis test on its last state ----
            const double tree_weight = c.cont ? c.sub_weight : -INFINITY;
            const double new_weight = hmc ? 0.0 : logaddexp(tree_weight, c.cand_weight);
            const double thresh = hmc ? ediff : tree_weight - c.cand_weight;
            const bool choose = (u_merge <= thresh) && (hmc ? not_divergent : (c.cont != 0));
            const bool take_leaf = accept_leaf || hmc;         // the subtree proposal is this leaf if it was just accepted, else what V_SUB holds
            if (choose) vop |= VOP_CAND | (take_leaf ? VOP_TAKE_LEAF : 0);
            if (hmc) { c.sub_L = L; c.sub_energy = energy; }
            if (choose) { c.cand_L = c.sub_L; c.cand_energy = c.sub_energy; c.cand_bfac = c.beta_k; c.is_accepted = 1; }
            c.cand_weight = new_weight;
            if (c.dir > 0) { c.LR = c.L_cur; c.bfacR = c.beta_k; } else { c.LL = c.L_cur; c.bfacL = c.beta_k; }
            c.e_sum += c.e_sum_sub;
            c.lf_count += c.sub_lf;
            c.not_div = c.nd;
            c.depth += 1;
            if (hmc) { c.e_sum = shs[19]; }          // acceptance statistic of HMC: min(1, exp(energy difference))
            if (hmc || c.cont == 0) {
                finish = true;                         // (a rejected subtree ends the transition whatever the trajectory-level U-turn test says)
            } else {
                // the trajectory-level U-turn test needs rho . p over the merged state: the point phase merges (ends, rho, proposal) and
                // leaves the two sums; the next doubling is set up with it, speculatively (the stream of the next slot evaluates its first
                // state; if the sums say "stop", that evaluation is dropped and the transition ends one slot later)
                vop |= VOP_ENDS;
                vdir = c.dir;
                c.phase = PH_MERGED;
                if (c.depth < cfg.max_depth) doubling = true;
            }
        }
    }

    if (finish) {
        // ---- transition finished ------------------------------------------------------------------
        __syncthreads();
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
        if (k >= cfg.burnin) vop |= VOP_OUT;
        if (c.is_accepted) c.beta_cache = c.beta_k;
        c.da_step_size = shs[1];
        c.da_error_sum = shs[2];
        c.da_log_avg = shs[3];
        c.da_step += 1;
        c.k = k + 1;
        if (c.k >= stop_k) c.phase = PH_IDLE;        // (reported idle by the next slot's decisions: this slot's point phase still has the op below to run)
        else start = true;
    }
    const long long vout = (long long)chain * (cfg.total - cfg.burnin) + (c.k - 1 - cfg.burnin);      // (used with VOP_OUT only: c.k was just advanced)

    if (start) {
        // ---- start transition c.k: temperature, momentum draw (point phase), both ends = current proposal ----------------
        __syncthreads();
        if (tid == 0) shs[4] = cfg.anneal ? temperature(c.k, cfg.min_temp) : 1.0;
        __syncthreads();
        c.beta_k = shs[4];
        const double bc = cfg.stale ? c.beta_cache : c.beta_k;
        c.eps = c.da_step_size;
        c.LL = c.LR = c.cand_L;
        c.bfacL = c.bfacR = bc;
        c.cand_bfac = bc;                 // (init_energy = bc cand_L - p.p / 2 follows with the draw's sum: PH_DRAWN)
        c.cand_weight = 0.0;
        c.e_sum = 0.0;
        c.lf_count = 0;
        c.not_div = 1;
        c.is_accepted = 0;
        c.depth = 0;
        c.leaf_ctr = 0;
        vop |= VOP_DRAW;
        doubling = true;
    }

    double d_eps = 0.0, d_hs = 0.0;
    if (doubling) {
        // ---- a doubling from the end selected by the direction bit: the point phase takes the first half / full step ----------
        Philox4 r = philox4x32_10((unsigned)c.depth, (unsigned)c.k, (unsigned)c.chain_id, STREAM_DIRECTION, cfg.seed);
        const bool fwd = hmc || (r.x & 1u) != 0;
        c.dir = fwd ? 1 : -1;
        const double bf = fwd ? c.bfacR : c.bfacL;
        d_eps = c.dir * c.eps;
        d_hs = 0.5 * d_eps * bf;
        c.cur ^= 1;
        c.nsteps = hmc ? cfg.hmc_L : (1 << c.depth);
        c.it = 0;
        c.sub_weight = -INFINITY;
        c.e_sum_sub = 0.0;
        c.sub_lf = 0;
        c.cont = 1;
        c.nd = c.not_div;
        vop |= VOP_DOUBLE;
        if (start) c.phase = PH_DRAWN;          // (else PH_MERGED, set with the merge)
    }
    if (tid == 0) {
        LeafPlan p{};
        p.vop = vop;
        p.active = doubling ? 1 : 0;             // with a doubling: the stream of the NEXT slot evaluates buffer `cur` as is
        p.skip = doubling ? 1 : 0;
        p.cur = c.cur;
        p.eps = d_eps; p.hs = d_hs;
        p.vleaf = vleaf; p.vdir = vdir; p.ndir = c.dir; p.vout = vout;
        p.step_k = (unsigned)c.k; p.chain_id = (unsigned)c.chain_id; p.seed = cfg.seed;
        plan_store(plan_out, p);
        ch.ctl[chain] = c;
    }
}

