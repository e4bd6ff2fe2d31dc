// Chain initialisation of the device-resident sampler (the state machine itself is decide.h, run inside k_stream).
#include "magi_internal.h"

namespace {

__global__ void k_init_chains(DevChains ch, SamplerCfgDev cfg, const long long* chain_ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        ch.gctl->done_chains = 0;
        ch.gctl->n_chains = ch.n_chains;
        ch.gctl->all_done = 0;
        ch.gctl->stop_k = 0;
        ch.gctl->epoch = 0;
    }
    if (i >= ch.n_chains) return;
    ChainCtl c{};
    c.phase = PH_INIT;
    c.cur = 0;
    // slot 0: the stream evaluates buffer 0 as is (ring entry 1 = "previous" plan, skip-type), the point phase runs
    // the bootstrap plan the decisions publish for it (PH_INIT)
    LeafPlan p{};
    p.active = 1; p.skip = 1; p.cur = 0;
    ch.plan[(size_t)ch.n_chains + i] = p;
    LeafPlan off{};
    ch.plan[i] = off;
    c.chain_id = chain_ids ? chain_ids[i] : (long long)i;
    c.da_step_size = cfg.step_size;
    c.da_log_shrink = log(10.0 * cfg.step_size);
    c.beta_cache = 1.0;
    c.done_epoch = -1;
    ch.ctl[i] = c;
}

}  // namespace

int magi_launch_init_chains(magi_handle* h, const long long* d_chain_ids, hipStream_t s) {
    const int n = h->n_chains;
    hipLaunchKernelGGL(k_init_chains, dim3((n + 63) / 64), dim3(64), 0, s, h->ch, h->cfg, d_chain_ids);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("init launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}
