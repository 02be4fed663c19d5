// FlashAttention forward for fp32 inputs on MI355X (gfx950): v_mfma_f32_32x32x2_f32.
//
// The reference op accepts FLOAT16 and FLOAT (flash_attention_qkv_packed.cpp:609); its fp32 tests hold the tightest
// bound (max|delta| <= 1e-4, t_cpp:249-262).  fp32 products with fp32 accumulation meet it without any splitting
// tricks; the fp32 matrix instruction runs at 1/16 of the bf16 rate (256 FLOP/cycle/CU = 157 TFLOP/s dense), which
// is also this kernel's roofline - far below the 16-bit kernels, ~100x above the one-workgroup-per-row generic kernel
// that used to serve fp32.  With a 64-cycle MFMA every other cost (LDS operands, global loads, softmax) hides behind
// the matrix pipe, so the structure is the plain one: register-staged double-buffered K/V tiles of 32 keys, one
// barrier per tile, 4 waves x 32 query rows per workgroup, exact running max per tile.
//
// Same transposed formulation as the 16-bit kernels (fa_fwd_kernels.hip): S^T = K Q^T, O^T += V^T P^T, query row on
// the lane.  Operand maps of v_mfma_f32_32x32x2_f32 (lane l: r = l & 31, h = l >> 5):
//   A[32 x 2]: lane holds A[r][h]        B[2 x 32]: lane holds B[h][r]        C/D as every 32x32 MFMA:
//   column = r, row = (reg & 3) + 8 (reg >> 2) + 4 h.
//   S^T: A = K[key r][d = 2s + h] (LDS), B = Q[row r][d = 2s + h] (registers), s = 0 .. D/2-1.
//   O^T: the k index of step j is the key (j & 3) + 8 (j >> 2) + 4 h - exactly the key that accumulator register j
//   of lane (r, h) holds - so B = P register j as it stands and A = V[that key][32 dt + r] (LDS).

#include "fa_device.h"

namespace fa {

template <int D, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void fa_fwd_f32_kernel(const FwdParams p) {
    constexpr int BQ = NWAVES * 32;
    constexpr int BK = 32;
    constexpr int KP = D + 1;                  // padded K row (floats): the A-operand read walks a column
    constexpr int DT = D / 32;
    constexpr int NT = NWAVES * 64;
    constexpr int CH = BK * D / 4;             // float4 chunks per tile
    constexpr int CPT = CH / NT;               // chunks per thread
    static_assert(CH % NT == 0, "tile chunks must split evenly over the threads");

    __shared__ float ks[2][BK * KP];
    __shared__ __attribute__((aligned(16))) float vs[2][BK * D];

    const int b = blockIdx.x;
    const int nq = p.nq_tiles;
    int g, qi;
    {   // XCD-aware bunched LPT map (fa_fwd_kernels.hip)
        const int xcd = b & 7, slot = b >> 3;
        const int ng = p.G8 >> 3;
        const int per_bunch = p.sched * nq;
        const int bunch = slot / per_bunch;
        const int rem = slot - bunch * per_bunch;
        const int g0 = bunch * p.sched;
        const int bc = (ng - g0) < p.sched ? (ng - g0) : p.sched;
        const int qd = rem / bc;
        qi = nq - 1 - qd;
        g = (g0 + (rem - qd * bc)) * 8 + xcd;
    }
    if (g >= p.G) return;
    if (p.xs > 1) {   // q-tile subsets (fa_params.h FwdParams::xs): g counts (group, subset) pairs; depth d of subset s is the group's depth d xs + s
        const int s_ = g % p.xs, qd_ = (nq - 1 - qi) * p.xs + s_;
        g /= p.xs;
        qi = p.nq_real - 1 - qd_;
        if (qi < 0) return;
    }

    const int L = p.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.H, hh = g - bb * p.H;
    const float* __restrict__ qg = (const float*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh;
    const float* __restrict__ kg = (const float*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;
    const float* __restrict__ vg = (const float*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;

    const int q_row0 = qi * BQ + wave * 32;
    const int qrow = q_row0 + r;
    const bool wave_active = q_row0 < L;

    // B operands of S^T: Q[qrow][2 s + h]
    float qf[D / 2];
    {
        const int qr = qrow < L ? qrow : L - 1;
        const float* qp = qg + (int64_t)qr * p.q_sl + h;
#pragma unroll
        for (int s = 0; s < D / 2; ++s) qf[s] = qp[2 * s];
    }

    const bool causal = p.causal != 0;             // 0: no mask, every key (only keys past L are masked)
    const int kv_end = (causal && (qi + 1) * BQ < L) ? (qi + 1) * BQ : L;
    const int n_tiles = (kv_end + BK - 1) / BK;
    const int wave_last = causal ? q_row0 : L - 1, wave_kv_last = causal ? q_row0 + 31 : L - 1, klim = causal ? qrow : L - 1;

    // global -> registers -> LDS staging: chunk c = tid + j NT covers floats 4c .. 4c+3 of the tile
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 kreg[CPT], vreg[CPT];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = tid + j * NT;
            const int row = (4 * c) / D, col = (4 * c) % D;
            const int key = t * BK + row;
            if (key < L) {
                kreg[j] = *(const f4*)(kg + (int64_t)key * p.k_sl + col);
                vreg[j] = *(const f4*)(vg + (int64_t)key * p.v_sl + col);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { kreg[j][e] = 0.f; vreg[j][e] = 0.f; }
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = tid + j * NT;
            const int row = (4 * c) / D, col = (4 * c) % D;
#pragma unroll
            for (int e = 0; e < 4; ++e) ks[buf][row * KP + col + e] = kreg[j][e];
            *(f4*)(&vs[buf][row * D + col]) = vreg[j];
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c = p.scale_log2;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) load_tile(t + 1);          // in flight under this tile's MFMAs
        const int kv0 = t * BK;
        if (wave_active && kv0 <= wave_kv_last) {
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const float* kb = &ks[buf][r * KP + h];
#pragma unroll
            for (int d2 = 0; d2 < D / 2; ++d2) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[2 * d2], qf[d2], s, 0, 0, 0);
            if (kv0 + BK - 1 > wave_last) {             // the wave's diagonal tile / the ragged last tile
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key > klim) s[e] = -INFINITY;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = max2(mx, s[e]);
            mx = half_swap_max(mx);
            const float m_new = max2(m, mx);
            const float alpha = fast_exp2((m - m_new) * c);
            m = m_new;
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            const float mc = m * c;
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = fast_exp2(__builtin_fmaf(s[e], c, -mc));
                ls += s[e];
            }
            l += ls;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float* vb = &vs[buf][((j & 3) + 8 * (j >> 2) + 4 * h) * D + r];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[32 * dt], s[j], o[dt], 0, 0, 0);
            }
        }
        if (t + 1 < n_tiles) store_tile(buf ^ 1);       // the other buffer was released by the previous barrier
        __syncthreads();
    }

    if (!wave_active || qrow >= L) return;
    const float lt = half_swap_sum(l);
    const float inv = 1.0f / lt;
    float* __restrict__ op = (float*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)qrow * p.o_sl;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = o[dt][4 * rg + j] * inv;
            *(f4*)(op + 32 * dt + 8 * rg + 4 * h) = w;
        }
    }
    if (p.lse != nullptr && h == 0) p.lse[(int64_t)g * L + qrow] = m * p.scale + __logf(lt);
}

// fp32, D in {64, 128}, rows 16-byte aligned; p.nq_tiles = ceil(L / 128)
hipError_t launch_fwd_f32(const FwdParams& p, hipStream_t stream) {
    const int64_t grid = (int64_t)p.G8 * p.nq_tiles;
    hipError_t e;
    if (p.D == 64) e = launch1((const void*)(fa_fwd_f32_kernel<64, 4>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else e = launch1((const void*)(fa_fwd_f32_kernel<128, 4>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    return e;
}

}  // namespace fa
