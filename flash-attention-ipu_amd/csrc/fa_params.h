// Kernel argument block shared by the launchers (fa_fwd_kernels.hip) and the C ABI (fa_capi.cpp).
#ifndef FA_PARAMS_H
#define FA_PARAMS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

// attention mask / bias and dropout (fa_extras.h); mask == nullptr and drop_thr == 0 in ordinary launches
struct ExtraParams {
    const void* mask;      // nullptr = none
    int64_t m_sb, m_sh, m_sl;   // mask strides in elements: batch (outer group), head (inner group), row; unit key stride
    int32_t mask_kind;     // 0 none, 1 boolean bytes (non-zero = keep), 2 additive bias in the operand dtype
    int32_t mask_vec;      // bit 0: four consecutive keys may be fetched with one aligned load; bit 1: and a row is shorter than 2^24 bytes (LDS-DMA of mask tiles)
    float inv_scale;       // 1 / softmax_scale (a bias is added to the RAW scores as bias / scale)
    uint32_t drop_key0;    // seed and offset folded (fa_capi.cpp); ignored when drop_state is set
    const unsigned long long* drop_state;   // device {seed, offset} (ABI 5: graph-safe dropout), nullptr = drop_key0 holds the folded immediates
    uint32_t drop_thr;     // floor(p * 2^32); 0 = no dropout
    float drop_rp;         // 1 / (1 - p)
};

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
    int32_t kv_div;        // query heads per key/value head (1 = multi-head attention): K/V head = head / kv_div
    int32_t causal;        // 1 causal mask, 0 none
    int32_t L;
    int32_t D;             // head dim the kernel runs at (64 | 128 for the MFMA kernels)
    int32_t Dr;            // the operands' true head dim (<= D, a multiple of 8): fa_fwd_dma_kernel reads the columns Dr .. D - 1 as zeros
                           // (out-of-range LDS-DMA chunks, predicated Q loads) and stores Dr columns - rows are Dr elements wide in memory
    int32_t nq_tiles;      // q tiles per group of the block map (with xs > 1: per (group, subset) pair = ceil(nq_real / xs))
    int32_t xs;            // q-tile subsets per group (>= 1).  Groups go to XCDs by index mod 8; a group count that leaves XCDs idle or
                           // unevenly loaded (G = 2: six of eight idle; G = 12: four XCDs with two groups, four with one) is served as
                           // G xs (group, subset) pairs - xs the smallest of 2, 4, 8 that wastes under 9 % of the slots, 1 when the L2 budget
                           // or the tile count forbids subsets (fa_capi.cpp subsets_for): G, G8 and nq_tiles then describe the PAIRS, subset s
                           // of a group holds its q tiles of depth s, s + xs, ... (depth 0 = the last, heaviest tile), nq_real its tile count
    int32_t nq_real;
    int32_t sched;         // groups per XCD bunch (>= 1): heavy-first order inside a bunch
    int32_t skipmax;       // f16 only: 1 = reference max from the first tile + overflow-triggered redo
    float scale;           // softmax scale
    float scale_log2;      // scale * log2(e)
    ExtraParams ex;
    int32_t perm_n;        // > 0: the first perm_n slots of every XCD run the work item perm[slot] (fa_fwd_dma.hip; fa_capi.cpp balanced_order)
    uint8_t perm[128];
    unsigned long long* stamps;   // measurement builds only (-DFA_STAMPS, scripts/phase_times.py): 16 u64 per wave, else null
};

// backward (fa_bwd_kernels.hip)
struct BwdParams {
    const void* q; const void* k; const void* v; const void* o; const void* dout;
    const float* lse;      // [G, L] from the forward
    float* dsum;           // [G, L] workspace: rowsum(dO o O)
    void* dq; void* dk; void* dv;
    int64_t q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, o_sl, do_sb, do_sh, do_sl;
    int64_t dq_sb, dq_sh, dq_sl, dk_sb, dk_sh, dk_sl, dv_sb, dv_sh, dv_sl;
    int32_t G, G8, H, L, D;
    int32_t Dr;            // true head dim (<= D, multiple of 8), as FwdParams::Dr
    int32_t Hkv, kv_div;   // K/V heads per batch entry, query heads per K/V head (grouped-query attention: H = Hkv * kv_div)
    int32_t Gkv, G8kv;     // K/V groups (G / kv_div) and that count rounded up to a multiple of 8: the dK/dV grid
    int32_t causal;        // 1 causal mask, 0 none
    int32_t nq_tiles;      // workgroup row tiles per group (32 rows x waves x query blocks per wave each)
    int32_t nk_tiles;      // 128-key blocks per group
    int32_t sched;         // query groups per XCD bunch (dQ grid)
    int32_t sched_kv;      // K/V groups per XCD bunch (dK/dV grid)
    // block subsets, as FwdParams::xs: with xs_q > 1 the dQ grid runs over G xs_q (group, q-tile subset) pairs - G, G8 and nq_tiles describe the
    // pairs, nq_real is the group's tile count; xs_kv / nk_real the same for the dK/dV grid over Gkv, G8kv, nk_tiles
    int32_t xs_q, nq_real, xs_kv, nk_real;
    float scale, scale_log2;
    ExtraParams ex;
    // split groups inside a workgroup (fa_bwd_kernels.hip QS; 1 = none): qs_q kv split groups of a dQ workgroup, qs_kv query-sweep split groups
    // of a dK/dV workgroup - for grids of at most one workgroup per CU, where the launch lasts as long as its heaviest block's chain.  (Read by
    // the host-side launcher only; last, so that no field the kernels read moved.)
    int32_t qs_q, qs_kv;
    // balanced dQ units (fa_bwd_kernels.hip BAL): with bal = 1 the dQ grid runs over units - nq_real / nq_tiles count UNITS (1.5 per row block);
    // the heavy row blocks' fp32 partial sums are parked in dk / dv until fa_bwd_dq_reduce_kernel has added them.  (Host-side launcher only.)
    int32_t bal;
};
// One launch, reporting ITS OWN status: hipLaunchKernel returns this launch's result, whereas hipGetLastError()
// after a <<<>>> launch also returns (and clears) a stale error left on the thread by an unrelated earlier HIP
// call of the host framework.
template <typename P>
inline hipError_t launch1(const void* kernel, dim3 grid, dim3 block, size_t dyn_lds, hipStream_t stream, const P& p) {
    P arg = p;
    void* args[] = {&arg};
    return hipLaunchKernel(kernel, grid, block, args, dyn_lds, stream);
}

hipError_t launch_bwd(const BwdParams& p, int dtype, hipStream_t stream);
hipError_t launch_bwd_generic(const BwdParams& p, int dtype, hipStream_t stream);
hipError_t launch_bwd_f32(const BwdParams& p, hipStream_t stream);   // fa_bwd_f32.hip: fp32, D in {64, 128}

hipError_t launch_fwd_dma(const FwdParams& p, int dtype, int waves, int qb, int split, int pair, hipStream_t stream);
hipError_t launch_fwd_extra(const FwdParams& p, int dtype, hipStream_t stream);   // fa_fwd_extra.hip: mask / bias / dropout (16-bit, D in {64, 128})
hipError_t launch_bwd_extra(const BwdParams& p, int dtype, hipStream_t stream);   // fa_bwd_extra.hip
hipError_t launch_fwd_generic(const FwdParams& p, int dtype, hipStream_t stream);
hipError_t launch_fwd_f32(const FwdParams& p, hipStream_t stream);   // fa_fwd_f32.hip

hipError_t launch_fwd_pipe(const FwdParams& p, int dtype, int waves, int qb, hipStream_t stream);   // fa_fwd_pipe.hip
hipError_t launch_fwd_pipe16(const FwdParams& p, int dtype, int waves, int qb, hipStream_t stream); // fa_fwd_pipe16.hip
hipError_t launch_fwd_kh(const FwdParams& p, int dtype, int split, hipStream_t stream);             // fa_fwd_kh.hip
double probe_mfma_peak(int dtype, int iters, int random_data);
hipError_t launch_clock_sampler(unsigned long long* out, int spins, hipStream_t stream);   // fa_probe.hip

}  // namespace fa
#endif
