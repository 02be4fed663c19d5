// Work-list forward for launches that under-fill or barely fill the chip (short sequences, few groups): the reference's own use of the op -
// nanoGPT at block 1024 through the SDPA override (reference demo/train.py:58, scaled_dot_product_attention.py:52-53) - is such a launch.
//
// The kernel IS fa_fwd_dma_body (fa_fwd_dma.hip: 4 waves x 32 query rows, 64-key tiles staged by LDS-DMA, skip-max arithmetic with the exact
// fallback pass) instantiated with WL = true; what changes is who decides which tiles a workgroup runs:
//   * the grid is exactly the chip's resident workgroups (CUs x workgroups per CU); each workgroup PULLS units (group, 128-row q tile, kv part)
//     from its XCD's list - heaviest first - with one agent-scope fetch-add per unit, drawn while the previous unit's last tiles are computed;
//   * the next unit's Q rows and first K/V tile are requested right after the current unit's loop and land under its epilogue: a workgroup pays
//     the 2.6-2.8 us first-touch prologue once per launch, not once per q tile (26 % of a workgroup's life at L = 1024);
//   * a q tile whose kv loop is longer than the chain cap the host chose (fa_capi.cpp wl_plan: about what one slot's fair share of the launch is)
//     is cut into kv parts that run as independent units on different workgroups; the parts meet in the caller's workspace (fp32 (m, l, O) per
//     part, write-through stores, one arrival ticket per q tile) and the part that arrives last folds them in part order - the launch no longer
//     lasts as long as its heaviest q tile's serial kv loop (BASELINE config 4's attention call: 16 tiles, 22 of its 23 us);
//   * a list's groups are the ones whose index is the XCD's number mod 8, so a group's K/V stay in one XCD's L2; a workgroup whose list is empty
//     steals from the nearest list that is not (one load of the eight heads, one fetch-add).
// Nothing waits on another workgroup anywhere (no spin, no grid barrier): a workgroup's progress never depends on one that has not started.
#define FA_DMA_BODY_ONLY 1
#include "fa_fwd_dma.hip"

namespace fa {

// (128 registers = 4 workgroups per CU at head_dim 64, as the unpaired kernel; head_dim 128: 66 KB of LDS = 2 per CU, 256 registers)
template <typename T, int D, bool SKIP>
__global__ __launch_bounds__(256, D == 64 ? 4 : 2) void fa_fwd_wl_kernel(const FwdParams p, const WlParams w) {
    fa_fwd_dma_body<T, D, 4, SKIP, 1, 1, false, false, false, true>(p, &w);
}

template <typename T, int D, bool SKIP>
static hipError_t launch_wl_t(const FwdParams& p, const WlParams& w, int grid, hipStream_t stream) {
    FwdParams pa = p;
    WlParams wa = w;
    void* args[] = {&pa, &wa};
    return hipLaunchKernel((const void*)(fa_fwd_wl_kernel<T, D, SKIP>), dim3((unsigned)grid), dim3(256), args, 0, stream);
}

// grid = persistent workgroups (the host caps it at the unit count and at the chip's slots)
hipError_t launch_fwd_wl(const FwdParams& p, const WlParams& w, int dtype, int grid, hipStream_t stream) {
    if (dtype == 1) return p.D == 64 ? launch_wl_t<__bf16, 64, true>(p, w, grid, stream) : launch_wl_t<__bf16, 128, true>(p, w, grid, stream);
    if (p.skipmax) return p.D == 64 ? launch_wl_t<_Float16, 64, true>(p, w, grid, stream) : launch_wl_t<_Float16, 128, true>(p, w, grid, stream);
    return p.D == 64 ? launch_wl_t<_Float16, 64, false>(p, w, grid, stream) : launch_wl_t<_Float16, 128, false>(p, w, grid, stream);
}

}  // namespace fa
