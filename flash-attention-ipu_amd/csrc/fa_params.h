// Kernel argument block shared by the launchers (fa_fwd_kernels.hip) and the C ABI (fa_capi.cpp).
#ifndef FA_PARAMS_H
#define FA_PARAMS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;            // optional [G, L] fp32
    // strides in elements: batch (outer group), head (inner group), row
    int64_t q_sb, q_sh, q_sl;
    int64_t k_sb, k_sh, k_sl;
    int64_t v_sb, v_sh, v_sl;
    int64_t o_sb, o_sh, o_sl;
    int32_t G;             // groups = batch * heads
    int32_t G8;            // G rounded up to a multiple of 8 (XCD-aware grid)
    int32_t H;             // inner group count (heads) for the 2-level group stride
    int32_t L;
    int32_t D;
    int32_t nq_tiles;      // q tiles per group
    int32_t sched;         // groups per XCD bunch (>= 1): heavy-first order inside a bunch
    float scale;           // softmax scale
    float scale_log2;      // scale * log2(e)
};

hipError_t launch_fwd_tile(const FwdParams& p, int dtype, int waves, hipStream_t stream);
hipError_t launch_fwd_dma(const FwdParams& p, int dtype, int waves, hipStream_t stream);
hipError_t launch_fwd_generic(const FwdParams& p, int dtype, hipStream_t stream);

}  // namespace fa
#endif
