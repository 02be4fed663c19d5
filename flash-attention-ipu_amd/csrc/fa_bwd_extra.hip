// Backward of a forward that ran with an attention mask / additive bias and / or dropout (fa_extras.h).  The kernels
// ARE fa_bwd_dq_body / fa_bwd_dkdv_body (fa_bwd_kernels.hip) instantiated with EXTRA = true: the mask terms go onto the recomputed
// scores, the dropout decisions are regenerated from (seed, offset, g, row, col) - the same pure function the forward used.
// Deterministic like the plain backward: no gradient is summed across workgroups.  No gradient flows to a bias.
#define FA_BWD_BODY_ONLY 1
#include "fa_bwd_kernels.hip"

namespace fa {

// MODE 1: dropout alone, or a row-broadcast mask (row stride 0: key padding, a per-key bias) - its terms depend on the key alone;
// MODE 2 | 3: a boolean mask | a bias that differs from row to row ([.., L, L]): its tiles travel to LDS by DMA with the operand tiles
// (fa_bwd_kernels.hip).  B = 8, H = 16, L = 4096, D = 64, forward + backward: bias [1, H, L, L] 9.2 -> 4.8 ms, boolean [B, H, L, L]
// 7.9 -> 4.4 ms, key padding 5.2 -> 3.1 ms against the in-place loads of the rounds before (profiles/r05_mask).
template <typename T, int D, int MODE>
__global__ __launch_bounds__(256) void fa_bwd_dq_extra_kernel(const BwdParams p) {
    fa_bwd_dq_body<T, D, 4, MODE>(p);
}
// PASS 0: dK and dV in one sweep; 1 | 2: dV alone | dK alone (fa_bwd_kernels.hip) - head_dim 256, where the single sweep has no register left for the mask terms
template <typename T, int D, int MODE, int PASS>
__global__ __launch_bounds__(256) void fa_bwd_dkdv_extra_kernel(const BwdParams p) {
    fa_bwd_dkdv_body<T, D, 4, MODE, PASS>(p);
}

template <typename T, int D, int MODE>
static hipError_t launch_bwd_extra_m(const BwdParams& p, hipStream_t stream) {
    const dim3 kgrid((unsigned)((int64_t)p.G8kv * p.nk_tiles));
    hipError_t e = launch1((const void*)(fa_bwd_dq_extra_kernel<T, D, MODE>), dim3((unsigned)((int64_t)p.G8 * p.nq_tiles)), dim3(256), 0, stream, p);
    if constexpr (D == 256) {
        if (e == hipSuccess) e = launch1((const void*)(fa_bwd_dkdv_extra_kernel<T, D, MODE, 1>), kgrid, dim3(256), 0, stream, p);
        if (e == hipSuccess) e = launch1((const void*)(fa_bwd_dkdv_extra_kernel<T, D, MODE, 2>), kgrid, dim3(256), 0, stream, p);
    } else {
        if (e == hipSuccess) e = launch1((const void*)(fa_bwd_dkdv_extra_kernel<T, D, MODE, 0>), kgrid, dim3(256), 0, stream, p);
    }
    return e;
}
template <typename T, int D>
static hipError_t launch_bwd_extra_t(const BwdParams& p, hipStream_t stream) {
    if (p.ex.mask == nullptr || p.ex.m_sl == 0) return launch_bwd_extra_m<T, D, 1>(p, stream);
    return p.ex.mask_kind == 2 ? launch_bwd_extra_m<T, D, 3>(p, stream) : launch_bwd_extra_m<T, D, 2>(p, stream);
}

// head dims: 64, 128, 192 and 256 (one workgroup per CU from 192 on).  At 256 dK/dV runs as two passes - dV, then dK - of the same body: in
// one sweep it fills all 512 registers without the mask terms (with them it spilled, and one instantiation crashed the compiler).
template <typename T>
static hipError_t launch_bwd_extra_d(const BwdParams& p, hipStream_t stream) {
    switch (p.D) {
        case 64: return launch_bwd_extra_t<T, 64>(p, stream);
        case 128: return launch_bwd_extra_t<T, 128>(p, stream);
        case 192: return launch_bwd_extra_t<T, 192>(p, stream);
        case 256: return launch_bwd_extra_t<T, 256>(p, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_bwd_extra(const BwdParams& p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_bwd_extra_d<__bf16>(p, stream) : launch_bwd_extra_d<_Float16>(p, stream);
}

}  // namespace fa
