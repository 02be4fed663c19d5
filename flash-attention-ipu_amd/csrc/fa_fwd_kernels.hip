// FlashAttention forward for MI355X (gfx950 / CDNA4) - hand-written HIP, no compatibility layers.
//
// Replaces the whole L2-L4 stack of the reference forward (the Poplar program built by
// flashAttentionQKVPackedWithStash, flash_attention_ipu/cpp/flash_attention_qkv_packed.cpp:175-334:
// dynamicSlice + matMulGrouped + ~12 elementwise/reduce programs + matMulGroupedAcc per block)
// with ONE fused kernel:
//
//   * one workgroup = NWAVES wavefronts = one (group g, q-tile) of BQ = 32*NWAVES query rows;
//     each wave owns 32 query rows and keeps Q^T fragments, the O^T accumulator (fp32), the running
//     max m and running sum l in VGPRs for the whole kv sweep;
//   * K/V tiles of 64 keys stream HBM -> LDS by LDS-DMA (double buffered, the next tile requested before the
//     current tile is computed: one barrier per tile);
//   * S^T = K Q^T and O^T += V^T P^T run on v_mfma_f32_32x32x16_{bf16,f16}.  Computing the
//     TRANSPOSED products puts a query row on a lane: the 32x32 accumulator of S^T holds 16 keys of
//     one query per lane, so the row max/sum are in-lane plus ONE v_permlane32_swap, the O rescale
//     is a per-lane scalar, and P^T is already in B-operand layout for the PV MFMA (no LDS round
//     trip for P);
//   * K is read from an XOR-swizzled row-major LDS image with ds_read_b128 (conflict-free), V from
//     an [d/32][key/8][8][32] image with ds_read_b64_tr_b16 (hardware transpose, conflict-free);
//   * the causal mask is an index predicate applied only on diagonal tiles; tiles entirely above the
//     diagonal are never visited (the reference's doBlock, FA.cpp:244) and the reference's cyclic
//     mask-block machinery (FA.cpp:47-173) is not needed;
//   * softmax uses exp2 with the scale folded (p = 2^(s*c - m*c), c = scale*log2 e).
//
// The MFMA kernels live in fa_fwd_dma.hip (64-key tiles staged by LDS-DMA: this tiling, read it next), fa_fwd_pipe.hip /
// fa_fwd_pipe16.hip (software-pipelined 32-key steps), fa_fwd_kh.hip (key halves) and fa_fwd_f32.hip (fp32 inputs); the
// register-staged first form of the kernel (HBM -> VGPR -> ds_write -> LDS: round 1's A/B baseline) left the tree in round 5
// (docs/HISTORY.md).  This file holds what covers every other shape / dtype (head dims the MFMA kernels do not take,
// unaligned rows): the generic kernel - one workgroup per query row, fp32 arithmetic.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_device.h"
#include "fa_extras.h"


namespace fa {


// ---------------------------------------------------------------------------------------------
// Generic kernel: any D, any L, fp32/bf16/f16 I/O, fp32 arithmetic.  One workgroup per query row.
// Keys are processed in chunks of GEN_CHUNK: scores -> chunk max -> online rescale -> P in LDS ->
// threads over d accumulate O.  Correctness path for odd shapes (reference tests use L=6, D=2,
// tests/cpp/test_flash_attention.cpp:245-251), not a performance path.
// ---------------------------------------------------------------------------------------------
constexpr int GEN_THREADS = 256;
constexpr int GEN_CHUNK = 1024;
constexpr int GEN_MAXD_PER_THREAD = 4;   // D <= 1024

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }

__device__ __forceinline__ float block_reduce(float x, float* red, bool is_max) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float y = __shfl_xor(x, off, 64);
        x = is_max ? fmaxf(x, y) : x + y;
    }
    __syncthreads();   // red[] reuse
    if (lane == 0) red[w] = x;
    __syncthreads();
    float rcc = red[0];
#pragma unroll
    for (int i = 1; i < GEN_THREADS / 64; ++i) rcc = is_max ? fmaxf(rcc, red[i]) : rcc + red[i];
    return rcc;
}

template <typename T>
__global__ __launch_bounds__(GEN_THREADS) void fa_fwd_generic_kernel(const FwdParams p) {
    __shared__ float pbuf[GEN_CHUNK];
    __shared__ float red[GEN_THREADS / 64];
    extern __shared__ float qrow_s[];   // D floats

    const int L = p.L, D = p.D;
    const int g = blockIdx.x / L;
    const int row = blockIdx.x - g * L;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh + (int64_t)row * p.q_sl;
    const T* kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;
    const T* vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;
    const int tid = threadIdx.x;

    for (int d = tid; d < D; d += GEN_THREADS) qrow_s[d] = to_f32(qg[d]);
    __syncthreads();

    float oacc[GEN_MAXD_PER_THREAD];
#pragma unroll
    for (int i = 0; i < GEN_MAXD_PER_THREAD; ++i) oacc[i] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int nkeys = p.causal ? row + 1 : L;   // causal: keys 0..row
    // attention mask / bias and dropout (fa_extras.h; fa_fwd_ex on operands the MFMA kernels do not take): the mask term is added
    // to the SCALED logit, the dropout decision zeroes the weight the PV product sees while the denominator keeps it
    const bool has_mask = p.ex.mask != nullptr, dropping = p.ex.drop_thr != 0u;
    const char* const mbase = has_mask ? (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)hh * p.ex.m_sh) * (p.ex.mask_kind == 2 ? (int)sizeof(T) : 1) : nullptr;
    const unsigned drop_kg = drop_group_key(drop_key0_of(p.ex), (unsigned)g);

    for (int k0 = 0; k0 < nkeys; k0 += GEN_CHUNK) {
        const int kn = nkeys - k0 < GEN_CHUNK ? nkeys - k0 : GEN_CHUNK;
        float mx = -INFINITY;
        for (int cidx = tid; cidx < kn; cidx += GEN_THREADS) {
            const T* kp = kg + (int64_t)(k0 + cidx) * p.k_sl;
            float acc = 0.f;
            for (int d = 0; d < D; ++d) acc = __builtin_fmaf(qrow_s[d], to_f32(kp[d]), acc);
            acc *= p.scale;
            if (has_mask) acc += mask_load1<T>(p.ex.mask_kind, mbase, (int64_t)row * p.ex.m_sl, k0 + cidx, 1.0f);
            pbuf[cidx] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = block_reduce(mx, red, true);
        const float m_new = fmaxf(m, mx);
        // exp(-inf) = 0 on the first chunk; a row that has met no admitted key yet (m_new = -inf under a mask) keeps alpha = 1, P = 0
        const float alpha = m_new == -INFINITY ? 1.0f : __expf(m - m_new);
        m = m_new;
        float ls = 0.f;
        for (int cidx = tid; cidx < kn; cidx += GEN_THREADS) {
            const float e = m_new == -INFINITY ? 0.0f : __expf(pbuf[cidx] - m_new);
            ls += e;
            const unsigned col = (unsigned)(k0 + cidx);
            const bool keep = !dropping || mix32(drop_tile_key(drop_kg, (unsigned)row, col) ^ ((((unsigned)row & 0xffffu) << 16) | (col & 0xffffu))) >= p.ex.drop_thr;
            pbuf[cidx] = keep ? e : 0.0f;
        }
        ls = block_reduce(ls, red, false);   // barriers inside also publish pbuf
        l = l * alpha + ls;
#pragma unroll
        for (int i = 0; i < GEN_MAXD_PER_THREAD; ++i) {
            const int d = tid + i * GEN_THREADS;
            if (d < D) {
                float acc = oacc[i] * alpha;
                const T* vp = vg + (int64_t)k0 * p.v_sl + d;
                for (int cidx = 0; cidx < kn; ++cidx) acc = __builtin_fmaf(pbuf[cidx], to_f32(vp[(int64_t)cidx * p.v_sl]), acc);
                oacc[i] = acc;
            }
        }
        __syncthreads();   // pbuf is rewritten by the next chunk
    }

    T* og = (T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)row * p.o_sl;
    const float inv = (dropping ? p.ex.drop_rp : 1.0f) / l;
#pragma unroll
    for (int i = 0; i < GEN_MAXD_PER_THREAD; ++i) {
        const int d = tid + i * GEN_THREADS;
        if (d < D) og[d] = (T)(oacc[i] * inv);
    }
    if (p.lse != nullptr && tid == 0) p.lse[(int64_t)g * L + row] = m + __logf(l);
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_fwd_generic(const FwdParams& p, int dtype, hipStream_t stream) {
    dim3 grid((unsigned)((int64_t)p.L * p.G));
    const size_t dyn = (size_t)p.D * sizeof(float);
    hipError_t e;
    if (dtype == 1) e = launch1((const void*)(fa_fwd_generic_kernel<__bf16>), grid, dim3(GEN_THREADS), dyn, stream, p);
    else if (dtype == 0) e = launch1((const void*)(fa_fwd_generic_kernel<_Float16>), grid, dim3(GEN_THREADS), dyn, stream, p);
    else e = launch1((const void*)(fa_fwd_generic_kernel<float>), grid, dim3(GEN_THREADS), dyn, stream, p);
    return e;
}

}  // namespace fa
