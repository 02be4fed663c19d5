// Software-pipelined FlashAttention forward for MI355X (gfx950): the default kernel for head_dim 64.
//
// Same data layout, LDS images, MFMA operand maps and arithmetic as fa_fwd_tile_kernel
// (fa_fwd_kernels.hip - read its header first).  What changes is the SCHEDULE inside a wave: at
// head_dim 64 the softmax (1 fma + 1 exp + max + convert per score) needs more VALU issue time than the
// two GEMMs need matrix-pipe time, so the matrix pipe must run UNDER the VALU work of the same wave
// instead of before/after it.  Each iteration therefore carries two score tiles:
//
//     S_next = K(t+1) Q^T        (8 MFMAs)   ||   P(t) = 2^(S_cur*c - m*c), bf16 pack      (VALU)
//     O^T   += V(t)^T P(t)^T     (8 MFMAs)   ||   row max of S_next                         (VALU)
//     l     += 1^T  P(t)^T       (4 MFMAs: the row sum rides the matrix pipe, not 32 v_add)
//
// K is staged ONE TILE AHEAD of V (K(t+2) and V(t+1) are fetched during iteration t), the loop is
// unrolled by two so the two score tiles swap roles without register moves, and each wave has exactly
// one diagonal (masked) tile - its last - so the mask lives in a separate straight-line body.
// Row sums come from the bf16-rounded P actually fed to the PV product (normaliser == weights used).

#include "fa_device.h"

namespace fa {

template <typename T, int D, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, ((D == 64 && NWAVES >= 4) ? 2 : 1)) void fa_fwd_pipe_kernel(const FwdParams p) {
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int NT = NWAVES * 64;
    constexpr int BQ = NWAVES * 32;
    constexpr int BK = 64;
    constexpr int ROWB = D * 2;
    constexpr int KTILE_B = BK * ROWB;
    constexpr int VBLK_B = BK * 64 + 64;
    constexpr int VTILE_B = (D / 32) * VBLK_B;
    constexpr int CHUNKS = KTILE_B / 16;
    constexpr int CPT = CHUNKS / NT;
    constexpr int CPR = D / 8;
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
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
    const T* __restrict__ kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)hh * p.k_sh;
    const T* __restrict__ vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)hh * p.v_sh;

    const int q_row0 = qi * BQ + wave * 32;
    const int qrow = q_row0 + r;
    const bool wave_active = q_row0 < L;
    // the wave's last kv tile, which is also its only diagonal (masked) one: q_row0 is a multiple of 32,
    // so rows [q_row0, q_row0+32) sit inside one 64-key tile
    const int t_last = wave_active ? (q_row0 >> 6) : -1;

    vec8 qf[KS];
    {
        const int qr = qrow < L ? qrow : L - 1;
        const T* qp = qg + (int64_t)qr * p.q_sl + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const vec8*)(qp + 16 * ks);
    }

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

    int k_rd[KS];
    {
        const int sw = (D == 64) ? ((r >> 1) & 7) : (r & 15);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * ROWB + (((2 * ks + h) ^ sw) << 4);
    }
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    const int kv_end = (qi + 1) * BQ < L ? (qi + 1) * BQ : L;
    const int n_tiles = (kv_end + BK - 1) / BK;

    u32x4 kreg[CPT], vreg[CPT];
#define FA_ISSUE_ONE(T_, REG_, BASE_, OFF_, SL_, LAST_)                                           \
    {                                                                                             \
        const int kv0_ = (T_) * BK;                                                               \
        const unsigned t0_ = (unsigned)kv0_ * SL_;                                                \
        if (kv0_ + BK <= L) {                                                                     \
            _Pragma("unroll") for (int i = 0; i < CPT; ++i)                                       \
                REG_[i] = *(const u32x4*)(BASE_ + (t0_ + OFF_[i]));                               \
        } else {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                     \
                const unsigned o_ = (kv0_ + ld_key[i] < L) ? t0_ + OFF_[i]                        \
                                                           : LAST_ + (OFF_[i] - ld_key[i] * SL_);  \
                REG_[i] = *(const u32x4*)(BASE_ + o_);                                            \
            }                                                                                     \
        }                                                                                         \
    }
#define FA_ISSUE_K(T_) FA_ISSUE_ONE(T_, kreg, kg, k_off, k_sl, k_last)
#define FA_ISSUE_V(T_) FA_ISSUE_ONE(T_, vreg, vg, v_off, v_sl, v_last)
#define FA_COMMIT_K(BUF_) \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) *(u32x4*)(smem + (BUF_) * KTILE_B + st_k[i]) = kreg[i];
#define FA_COMMIT_V(BUF_) \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) *(u32x4*)(smem_v + (BUF_) * VTILE_B + st_v[i]) = vreg[i];

    // S^T = K Q^T for the K tile in buffer BUF_
#define FA_QK(S0_, S1_, BUF_)                                                                     \
    {                                                                                             \
        const unsigned char* kb_ = smem + (BUF_) * KTILE_B;                                       \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) { S0_[e] = 0.f; S1_[e] = 0.f; }            \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                       \
            const vec8 k0_ = *(const vec8*)(kb_ + k_rd[ks]);                                      \
            const vec8 k1_ = *(const vec8*)(kb_ + 32 * ROWB + k_rd[ks]);                          \
            S0_ = E::mfma(k0_, qf[ks], S0_);                                                      \
            S1_ = E::mfma(k1_, qf[ks], S1_);                                                      \
        }                                                                                         \
    }
#define FA_MASK(S0_, S1_, KV0_)                                                                   \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                              \
        const int key_ = (KV0_) + (e & 3) + 8 * (e >> 2) + 4 * h;                                 \
        if (key_ > qrow) S0_[e] = -INFINITY;                                                      \
        if (key_ + 32 > qrow) S1_[e] = -INFINITY;                                                 \
    }
    // two independent max chains, then the two halves of the wave
#define FA_ROWMAX(S0_, S1_)                                                                       \
    {                                                                                             \
        float ma_ = max3(S0_[0], S0_[1], S0_[2]), mb_ = max3(S1_[0], S1_[1], S1_[2]);             \
        ma_ = max3(ma_, S0_[3], S0_[4]);  mb_ = max3(mb_, S1_[3], S1_[4]);                        \
        ma_ = max3(ma_, S0_[5], S0_[6]);  mb_ = max3(mb_, S1_[5], S1_[6]);                        \
        ma_ = max3(ma_, S0_[7], S0_[8]);  mb_ = max3(mb_, S1_[7], S1_[8]);                        \
        ma_ = max3(ma_, S0_[9], S0_[10]); mb_ = max3(mb_, S1_[9], S1_[10]);                       \
        ma_ = max3(ma_, S0_[11], S0_[12]); mb_ = max3(mb_, S1_[11], S1_[12]);                     \
        ma_ = max3(ma_, S0_[13], S0_[14]); mb_ = max3(mb_, S1_[13], S1_[14]);                     \
        mx = half_swap_max(max3(ma_, mb_, max2(S0_[15], S1_[15])));                               \
    }
    // lazy rescale decision for the tile whose row max is `mx` (taken on the first tile: m = -inf)
#define FA_RESCALE()                                                                              \
    if (__builtin_amdgcn_ballot_w64((mx - m) * c > RESCALE_THR) != 0) {                           \
        const float m_new_ = max2(m, mx);                                                         \
        const float alpha_ = fast_exp2((m - m_new_) * c);                                         \
        m = m_new_;                                                                               \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) lacc[e] *= alpha_;                         \
        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                         \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) o[dt][e] *= alpha_;                    \
    }
    // P^T = 2^(S*c - m*c) as bf16/f16 B-operand fragments
#define FA_PCALC(S0_, S1_)                                                                        \
    {                                                                                             \
        const float mc_ = m * c;                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                           \
            pf[0][j] = (T)fast_exp2(__builtin_fmaf(S0_[j], c, -mc_));                             \
            pf[1][j] = (T)fast_exp2(__builtin_fmaf(S0_[8 + j], c, -mc_));                         \
        }                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                           \
            pf[2][j] = (T)fast_exp2(__builtin_fmaf(S1_[j], c, -mc_));                             \
            pf[3][j] = (T)fast_exp2(__builtin_fmaf(S1_[8 + j], c, -mc_));                         \
        }                                                                                         \
    }
    // O^T += V^T P^T and l += 1^T P^T for the V tile in buffer BUF_
#define FA_PV(BUF_)                                                                               \
    {                                                                                             \
        const unsigned char* vb_ = smem_v + (BUF_) * VTILE_B;                                     \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                        \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                   \
                const unsigned char* vp_ = vb_ + dt * VBLK_B + kk * 1024 + v_rd;                  \
                const i16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                        \
                    (i16x4 __attribute__((address_space(3)))*)(vp_));                             \
                const i16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                        \
                    (i16x4 __attribute__((address_space(3)))*)(vp_ + 512));                       \
                const i16x8 v8_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);      \
                o[dt] = E::mfma(__builtin_bit_cast(vec8, v8_), pf[kk], o[dt]);                    \
            }                                                                                     \
            lacc = E::mfma(ones, pf[kk], lacc);                                                   \
        }                                                                                         \
    }

    // one kv tile: C = scores of tile T_ (ready, masked, row max in mx), N = scores of tile T_+1
#define FA_ITER(T_, C0_, C1_, N0_, N1_)                                                           \
    {                                                                                             \
        const int t_ = (T_);                                                                      \
        const int buf_ = t_ & 1;                                                                  \
        if (t_ + 2 < n_tiles) FA_ISSUE_K(t_ + 2)                                                  \
        if (t_ + 1 < n_tiles) FA_ISSUE_V(t_ + 1)                                                  \
        if (t_ <= t_last) {                                                                       \
            FA_RESCALE()                                                                          \
            if (t_ + 1 < t_last) {                                                                \
                FA_QK(N0_, N1_, buf_ ^ 1)                                                         \
                FA_PCALC(C0_, C1_)                                                                \
                FA_PV(buf_)                                                                       \
                FA_ROWMAX(N0_, N1_)                                                               \
            } else if (t_ < t_last) {                                                             \
                FA_QK(N0_, N1_, buf_ ^ 1)                                                         \
                FA_PCALC(C0_, C1_)                                                                \
                FA_PV(buf_)                                                                       \
                FA_MASK(N0_, N1_, (t_ + 1) * BK)                                                  \
                FA_ROWMAX(N0_, N1_)                                                               \
            } else {                                                                              \
                FA_PCALC(C0_, C1_)                                                                \
                FA_PV(buf_)                                                                       \
            }                                                                                     \
        }                                                                                         \
        if (t_ + 2 < n_tiles) { FA_COMMIT_K(buf_) }                                               \
        if (t_ + 1 < n_tiles) { FA_COMMIT_V(buf_ ^ 1) }                                           \
        __syncthreads();                                                                          \
    }

    // ---- running state
    f32x16 o[DT], lacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) lacc[e] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    vec8 ones, pf[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
    float m = -INFINITY, mx = -INFINITY;
    const float c = p.scale_log2;
    f32x16 sa0, sa1, sb0, sb1;

    // ---- prologue: K(0), V(0), K(1) staged; S(0) computed
    FA_ISSUE_K(0)
    FA_ISSUE_V(0)
    FA_COMMIT_K(0)
    FA_COMMIT_V(0)
    if (n_tiles > 1) {
        FA_ISSUE_K(1)
        FA_COMMIT_K(1)
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // Q landed (see fa_fwd_tile_kernel)
    if (wave_active) {
        FA_QK(sa0, sa1, 0)
        if (t_last == 0) FA_MASK(sa0, sa1, 0)
        FA_ROWMAX(sa0, sa1)
    }
    __syncthreads();   // iteration 0 restages K buffer 0 (with K(2)): every wave must be done reading K(0)

    for (int t = 0; t < n_tiles; t += 2) {
        FA_ITER(t, sa0, sa1, sb0, sb1)
        if (t + 1 < n_tiles) FA_ITER(t + 1, sb0, sb1, sa0, sa1)
    }

    // ---- epilogue
    if (!wave_active || qrow >= L) return;
    const float lt = lacc[0];   // every row of 1^T P^T holds the full row sum (both halves' keys)
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
    if (p.lse != nullptr && h == 0) p.lse[(int64_t)g * L + qrow] = m * p.scale + __logf(lt);
}

template <typename T, int D>
static hipError_t launch_pipe_w(const FwdParams& p, int waves, hipStream_t stream) {
    const int64_t grid = (int64_t)p.G8 * p.nq_tiles;
    switch (waves) {
        case 8: hipLaunchKernelGGL((fa_fwd_pipe_kernel<T, D, 8>), dim3((unsigned)grid), dim3(512), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((fa_fwd_pipe_kernel<T, D, 4>), dim3((unsigned)grid), dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((fa_fwd_pipe_kernel<T, D, 2>), dim3((unsigned)grid), dim3(128), 0, stream, p); break;
        default: hipLaunchKernelGGL((fa_fwd_pipe_kernel<T, D, 1>), dim3((unsigned)grid), dim3(64), 0, stream, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_fwd_pipe(const FwdParams& p, int dtype, int waves, hipStream_t stream) {
    if (dtype == 1) {
        return p.D == 64 ? launch_pipe_w<__bf16, 64>(p, waves, stream) : launch_pipe_w<__bf16, 128>(p, waves, stream);
    }
    return p.D == 64 ? launch_pipe_w<_Float16, 64>(p, waves, stream) : launch_pipe_w<_Float16, 128>(p, waves, stream);
}

}  // namespace fa
