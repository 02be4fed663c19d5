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
//   * K/V tiles of 64 keys stream HBM -> VGPR -> LDS (double buffered, loads issued before the
//     tile's compute and committed after it: one barrier per tile);
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
// This file holds the register-staged form of the kernel (HBM -> VGPR -> ds_write -> LDS).  The
// shipped default is its LDS-DMA form in fa_fwd_dma.hip (same tiling and arithmetic, bit-identical
// results); this one stays selectable (FA_AMD_KERNEL=tile) as the A/B baseline.
//
// A second, generic kernel covers every other shape/dtype (fp32, D not in {64,128}, unaligned
// rows): one workgroup per query row, fp32 arithmetic.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_device.h"
#include "fa_extras.h"

#ifndef FA_EXPERIMENTAL
#define FA_EXPERIMENTAL 0
#endif
#ifndef FA_CHAIN_SPLIT
#define FA_CHAIN_SPLIT 0   // experiment knob: split the max / row-sum dependency chains
#endif

namespace fa {


template <typename T, int D, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void fa_fwd_tile_kernel(const FwdParams p) {
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int NT = NWAVES * 64;
    constexpr int BQ = NWAVES * 32;
    constexpr int BK = 64;
    constexpr int ROWB = D * 2;                 // bytes per K row in LDS
    constexpr int KTILE_B = BK * ROWB;          // bytes per K tile
    constexpr int VBLK_B = BK * 64 + 64;        // bytes per 32-column d-block of the V image (+64 B pad:
                                                // the two 64-B halves of a staged 128-B row hit distinct banks)
    constexpr int VTILE_B = (D / 32) * VBLK_B;  // bytes per V tile
    constexpr int CHUNKS = KTILE_B / 16;        // 16-byte chunks per tile
    constexpr int CPT = CHUNKS / NT;            // chunks per thread per tile
    constexpr int CPR = D / 8;                  // chunks per row
    constexpr int KS = D / 16;                  // k-steps of the QK^T contraction
    constexpr int DT = D / 32;                  // 32-wide d tiles of O^T
    static_assert(CHUNKS % NT == 0, "tile must split evenly over the workgroup");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KTILE_B + 2 * VTILE_B];  // K0 K1 V0 V1
    unsigned char* const smem_v = smem + 2 * KTILE_B;

    // ---- which (group, q-tile) is this workgroup?  Blocks b and b+8 share an XCD (round-robin
    // dispatch).  Each XCD owns the groups g = 8*gl + xcd and walks them in bunches of `sched`
    // groups; inside a bunch the q-tiles go heaviest first across all its groups.  A bunch keeps
    // its K/V (sched x 2*L*D*2 bytes) in that XCD's 4 MiB L2 while longest-processing-time-first
    // ordering evens out the causal imbalance (one group at a time leaves 10-16 % of the slots idle).
    const int b = blockIdx.x;
    const int nq = p.nq_tiles;
    int g, qi;
    {
        const int xcd = b & 7, slot = b >> 3;
        const int ng = p.G8 >> 3;                      // groups per XCD
        const int per_bunch = p.sched * nq;
        const int bunch = slot / per_bunch;
        const int rem = slot - bunch * per_bunch;
        const int g0 = bunch * p.sched;
        const int bc = (ng - g0) < p.sched ? (ng - g0) : p.sched;   // groups in this bunch
        const int qd = rem / bc;
        qi = nq - 1 - qd;
        g = (g0 + (rem - qd * bc)) * 8 + xcd;
    }
    if (g >= p.G) return;

    const int L = p.L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* __restrict__ qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;
    const T* __restrict__ vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;

    const int q_row0 = qi * BQ + wave * 32;   // first query row of this wave
    const int qrow = q_row0 + r;              // this lane's query row (both halves hold the same row)
    const bool wave_active = q_row0 < L;

    // ---- Q^T fragments (B operand of S^T = K Q^T): lane (r,h) holds Q[qrow][16ks + 8h + 0..7]
    vec8 qf[KS];
    {
        const int qr = qrow < L ? qrow : L - 1;
        const T* qp = qg + (int64_t)qr * p.q_sl + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const vec8*)(qp + 16 * ks);
    }

    // ---- per-thread staging geometry.  Global side: 32-bit element offsets from the group base
    // (the C ABI guarantees L*stride < 2^31), advanced by one tile per iteration; rows past L-1 are
    // redirected to row L-1 (they are causally masked for every stored query row, any finite data do).
    const unsigned k_sl = (unsigned)p.k_sl, v_sl = (unsigned)p.v_sl;
    unsigned k_off[CPT], v_off[CPT];
    int ld_key[CPT];
    int st_k[CPT], st_v[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * NT;
        const int key = c / CPR, ch = c % CPR;
        ld_key[i] = key;
        k_off[i] = (unsigned)key * k_sl + ch * 8;
        v_off[i] = (unsigned)key * v_sl + ch * 8;
        const int sw = (D == 64) ? ((key >> 1) & 7) : (key & 15);
        st_k[i] = key * ROWB + ((ch ^ sw) << 4);
        st_v[i] = (ch >> 2) * VBLK_B + (key >> 3) * 512 + (key & 7) * 64 + (ch & 3) * 16;
    }
    const unsigned k_last = (unsigned)(L - 1) * k_sl, v_last = (unsigned)(L - 1) * v_sl;

    // ---- LDS read addresses (tile-invariant, per lane)
    // K: row (32kb + r), chunk (2ks + h) ^ swz(row)
    int k_rd[KS];
    {
        const int sw = (D == 64) ? ((r >> 1) & 7) : (r & 15);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * ROWB + (((2 * ks + h) ^ sw) << 4);
    }
    // V (transposed read): group gq = lane>>4 -> h = gq>>1, 16-col half c16 = gq&1; in-group lane
    // 4*q4 + pp supplies row (4h + q4) of the 8-key block, cols 4pp..4pp+3 of the 16-col half.
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    const bool causal = p.causal != 0;             // 0: no mask, every key (only keys past L are masked)
    const int kv_end = (causal && (qi + 1) * BQ < L) ? (qi + 1) * BQ : L;
    const int n_tiles = (kv_end + BK - 1) / BK;
    const int wave_last = causal ? q_row0 : L - 1, wave_kv_last = causal ? q_row0 + 31 : L - 1, klim = causal ? qrow : L - 1;

    u32x4 kreg[CPT], vreg[CPT];
#define FA_ISSUE_LOADS(T_)                                                                        \
    {                                                                                             \
        const int kv0_ = (T_) * BK;                                                               \
        const unsigned kt_ = (unsigned)kv0_ * k_sl, vt_ = (unsigned)kv0_ * v_sl;                  \
        if (kv0_ + BK <= L) {                                                                     \
            _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                     \
                kreg[i] = *(const u32x4*)(kg + (kt_ + k_off[i]));                                 \
                vreg[i] = *(const u32x4*)(vg + (vt_ + v_off[i]));                                 \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                     \
                const bool in_ = kv0_ + ld_key[i] < L;                                            \
                const unsigned ko_ = in_ ? kt_ + k_off[i] : k_last + (k_off[i] - ld_key[i] * k_sl); \
                const unsigned vo_ = in_ ? vt_ + v_off[i] : v_last + (v_off[i] - ld_key[i] * v_sl); \
                kreg[i] = *(const u32x4*)(kg + ko_);                                              \
                vreg[i] = *(const u32x4*)(vg + vo_);                                              \
            }                                                                                     \
        }                                                                                         \
    }
#define FA_COMMIT_LOADS(BUF_)                                                                     \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                             \
        *(u32x4*)(smem + (BUF_) * KTILE_B + st_k[i]) = kreg[i];                                   \
        *(u32x4*)(smem_v + (BUF_) * VTILE_B + st_v[i]) = vreg[i];                                 \
    }

    // ---- running state
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    float m = -INFINITY;   // running reference max (raw logits); lags the true max by <= RESCALE_THR/c
    float l = 0.f;         // running row sum over THIS half's keys (halves are added in the epilogue)
    const float c = p.scale_log2;

    FA_ISSUE_LOADS(0)
    FA_COMMIT_LOADS(0)
    __syncthreads();
    // Q must have LANDED before the loop: otherwise hipcc carries "Q loads pending" into the loop
    // and drains vmcnt to 0 in front of the QK^T MFMAs of EVERY tile, i.e. waits for the prefetch of
    // tile t+1 it has just issued.  Using the registers here retires the wait once.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        const int kv0 = t * BK;
        if (t + 1 < n_tiles) FA_ISSUE_LOADS(t + 1)

        if (wave_active && kv0 <= wave_kv_last) {
            const unsigned char* kb_ = smem + buf * KTILE_B;
            const unsigned char* vb_ = smem_v + buf * VTILE_B;

            // ---- S^T = K Q^T : two 32-key blocks
            f32x16 s0, s1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 k0 = *(const vec8*)(kb_ + k_rd[ks]);
                const vec8 k1 = *(const vec8*)(kb_ + 32 * ROWB + k_rd[ks]);
                s0 = E::mfma(k0, qf[ks], s0);
                s1 = E::mfma(k1, qf[ks], s1);
            }

            // ---- causal predicate, diagonal tiles only (wave-uniform branch)
            if (kv0 + BK - 1 > wave_last) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key > klim) s0[e] = -INFINITY;
                    if (key + 32 > klim) s1[e] = -INFINITY;
                }
            }

            // ---- online softmax (row = lane; keys = registers x 2 halves)
#if FA_CHAIN_SPLIT
            // (two independent chains: a single 16-deep dependent chain stalls the wave)
            float mxa = max3(s0[0], s0[1], s0[2]), mxb = max3(s1[0], s1[1], s1[2]);
#pragma unroll
            for (int e = 3; e < 15; e += 2) {
                mxa = max3(mxa, s0[e], s0[e + 1]);
                mxb = max3(mxb, s1[e], s1[e + 1]);
            }
            float mx = half_swap_max(max3(mxa, mxb, max2(s0[15], s1[15])));
#else
            float mx = max2(s0[0], s1[0]);
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = max3(mx, s0[e], s1[e]);
            mx = half_swap_max(mx);
#endif
            // lazy rescale: only when some row of the wave outgrew its reference max by > THR
            // (first tile: m = -inf, always taken).  Decision BEFORE this tile's P is formed, so
            // everything already in O and l is scaled exactly once.
            if (__builtin_amdgcn_ballot_w64((mx - m) * c > RESCALE_THR) != 0) {
                const float m_new = max2(m, mx);
                const float alpha = fast_exp2((m - m_new) * c);
                m = m_new;
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            }
            const float mc = m * c;
#if FA_CHAIN_SPLIT
            float ls0 = 0.f, ls1 = 0.f, ls2 = 0.f, ls3 = 0.f;   // four short chains, not one of 32
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                s0[e] = fast_exp2(__builtin_fmaf(s0[e], c, -mc));
                s1[e] = fast_exp2(__builtin_fmaf(s1[e], c, -mc));
                s0[e + 1] = fast_exp2(__builtin_fmaf(s0[e + 1], c, -mc));
                s1[e + 1] = fast_exp2(__builtin_fmaf(s1[e + 1], c, -mc));
                ls0 += s0[e];
                ls1 += s1[e];
                ls2 += s0[e + 1];
                ls3 += s1[e + 1];
            }
            l += (ls0 + ls1) + (ls2 + ls3);
#else
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s0[e] = fast_exp2(__builtin_fmaf(s0[e], c, -mc));
                s1[e] = fast_exp2(__builtin_fmaf(s1[e], c, -mc));
                ls += s0[e];
                ls += s1[e];
            }
            l += ls;
#endif

            // ---- P^T fragments (B operand of O^T += V^T P^T): registers 8s..8s+7 of each block
            vec8 pf[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pf[0][j] = (T)s0[j];
                pf[1][j] = (T)s0[8 + j];
                pf[2][j] = (T)s1[j];
                pf[3][j] = (T)s1[8 + j];
            }

            // ---- O^T += V^T P^T
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const unsigned char* vp = vb_ + dt * VBLK_B + kk * 1024 + v_rd;
                    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (i16x4 __attribute__((address_space(3)))*)(vp));
                    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (i16x4 __attribute__((address_space(3)))*)(vp + 512));
                    const i16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = E::mfma(__builtin_bit_cast(vec8, v8), pf[kk], o[dt]);
                }
            }
        }

        if (t + 1 < n_tiles) { FA_COMMIT_LOADS(buf ^ 1) }
        __syncthreads();
    }
#undef FA_ISSUE_LOADS
#undef FA_COMMIT_LOADS

    // ---- epilogue: O /= l, optional LSE, store
    if (!wave_active || qrow >= L) return;
    const float lt = half_swap_sum(l);
    const float inv = __builtin_amdgcn_rcpf(lt);
    T* __restrict__ og = (T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)qrow * p.o_sl;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            vec4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (T)(o[dt][4 * rg + j] * inv);
            *(vec4*)(og + 32 * dt + 8 * rg + 4 * h) = w;
        }
    }
    if (p.lse != nullptr && h == 0) {
        // LSE of the scaled logits in natural log units: scale*m + ln(l)   (FA.cpp:316)
        p.lse[(int64_t)g * L + qrow] = m * p.scale + __logf(lt);
    }
}

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
#if FA_EXPERIMENTAL   /* the register-staged A/B baseline is not product code: make EXPERIMENTAL=1 (fa_capi.cpp) */
template <typename T, int D, int NWAVES>
static hipError_t launch_tile(const FwdParams& p, hipStream_t stream) {
    const int64_t grid = (int64_t)p.G8 * p.nq_tiles;
    return launch1((const void*)(fa_fwd_tile_kernel<T, D, NWAVES>), dim3((unsigned)grid), dim3(NWAVES * 64), 0, stream, p);
}

template <typename T, int D>
static hipError_t launch_tile_w(const FwdParams& p, int waves, hipStream_t stream) {
    switch (waves) {
        case 8: return launch_tile<T, D, 8>(p, stream);
        case 4: return launch_tile<T, D, 4>(p, stream);
        case 2: return launch_tile<T, D, 2>(p, stream);
        default: return launch_tile<T, D, 1>(p, stream);
    }
}

hipError_t launch_fwd_tile(const FwdParams& p, int dtype, int waves, hipStream_t stream) {
    if (dtype == 1) {
        return p.D == 64 ? launch_tile_w<__bf16, 64>(p, waves, stream) : launch_tile_w<__bf16, 128>(p, waves, stream);
    }
    return p.D == 64 ? launch_tile_w<_Float16, 64>(p, waves, stream) : launch_tile_w<_Float16, 128>(p, waves, stream);
}
#endif

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
