// Forward with an attention mask / additive bias and / or dropout (fa_extras.h; SURVEY 8f-4: the modes
// scaled_dot_product_attention.py:21-28 of the reference rejects).  The kernel IS fa_fwd_dma_body (fa_fwd_dma.hip: 4 waves x 32
// query rows, 64-key tiles staged by LDS-DMA) instantiated with EXTRA = true and the exact-max arithmetic: the mask terms are
// added to the score registers of every tile, the dropout decisions are taken on the P registers between the row sums and the
// PV product.  16-bit operands, head_dim 64 / 128 / 192 / 256 (and what runs zero-extended at them); causal or unmasked underneath.
#define FA_DMA_BODY_ONLY 1
#include "fa_fwd_dma.hip"

namespace fa {

// MM: how the mask reaches the scores (fa_fwd_dma.hip) - 0: none, or a row-broadcast one; 2 | 3: a boolean mask | a bias that differs from row to row
template <typename T, int D, int MM>
__global__ __launch_bounds__(256) void fa_fwd_extra_kernel(const FwdParams p) {
    fa_fwd_dma_body<T, D, 4, false, 1, 1, false, true, MM>(p);
}

template <typename T, int D>
static hipError_t launch_fwd_extra_t(const FwdParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)((int64_t)p.G8 * p.nq_tiles)), block(256);
    if (p.ex.mask == nullptr || p.ex.m_sl == 0) return launch1((const void*)(fa_fwd_extra_kernel<T, D, 0>), grid, block, 0, stream, p);
    return p.ex.mask_kind == 2 ? launch1((const void*)(fa_fwd_extra_kernel<T, D, 3>), grid, block, 0, stream, p)
                               : launch1((const void*)(fa_fwd_extra_kernel<T, D, 2>), grid, block, 0, stream, p);
}

template <typename T>
static hipError_t launch_fwd_extra_d(const FwdParams& p, hipStream_t stream) {
    switch (p.D) {
        case 64: return launch_fwd_extra_t<T, 64>(p, stream);
        case 128: return launch_fwd_extra_t<T, 128>(p, stream);
        case 192: return launch_fwd_extra_t<T, 192>(p, stream);   // (192 / 256: one workgroup per CU - 100 / 133 KB of K / V tiles plus the mask images)
        case 256: return launch_fwd_extra_t<T, 256>(p, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_fwd_extra(const FwdParams& p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_fwd_extra_d<__bf16>(p, stream) : launch_fwd_extra_d<_Float16>(p, stream);
}

}  // namespace fa
