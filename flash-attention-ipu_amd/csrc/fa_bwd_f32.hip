// FlashAttention backward for fp32 inputs on MI355X (gfx950): v_mfma_f32_32x32x2_f32.
//
// The reference's grad op accepts FLOAT as well as FLOAT16 (flash_attention_qkv_packed.cpp:348-512, :609) and its
// C++ test holds fp32 gradients to max|delta| <= 1e-4 (tests/cpp/test_flash_attention.cpp:273-287).  fp32 products with
// fp32 accumulation meet that directly; the fp32 matrix instruction runs at 1/16 of the bf16 rate (157 TFLOP/s dense),
// which is this file's roofline - two orders of magnitude above the one-workgroup-per-row generic kernels that
// served fp32 gradients before.  Same split as the 16-bit backward (fa_bwd_kernels.hip): no gradient is summed across
// workgroups (deterministic, no atomics), scores are recomputed in both kernels from the forward's logsumexp.
//
//   fa_bwd_f32_dq_kernel    workgroup = 4 waves x 32 query rows; sweeps 32-key tiles up to the diagonal.
//       query row on the lane:  S^T = K Q^T,  dP^T = V dO^T  (A = K / V rows from LDS, B = Q / dO in registers),
//       P^T = 2^(c S^T - lse2), dS^T = P^T o (dP^T - Dsum)  - lane scalars lse2, Dsum -, dQ^T += K^T dS^T with the
//       accumulator registers of dS^T as the B operand as they stand (k index of step j = the key register j holds).
//       Its prologue forms Dsum = rowsum(dO o O) (FA.cpp:375-378) for its own rows and publishes it for the next kernel.
//   fa_bwd_f32_dkdv_kernel  workgroup = 4 waves x 32 keys; sweeps 32-row query tiles from the diagonal down.
//       key on the lane:  S = Q K^T,  dP = dO V^T  (A = Q / dO rows from LDS, B = K / V in registers); the row
//       constants lse2[q], Dsum[q] vary along the accumulator registers and come from a small LDS array;
//       dV^T += dO^T P and dK^T += Q^T dS take P / dS registers as B operands as they stand.
//
// Operand maps of v_mfma_f32_32x32x2_f32 (lane l: r = l & 31, h = l >> 5): A[32 x 2]: A[r][h]; B[2 x 32]: B[h][r];
// C/D: column r, row (reg & 3) + 8 (reg >> 2) + 4 h  (see fa_fwd_f32.hip).
// One wave per SIMD (the D = 128 kernels hold ~300 registers): a 64-cycle MFMA leaves the single wave all the issue
// slots it needs for the LDS operand reads in between.

#include "fa_device.h"

namespace fa {

namespace {

constexpr float LOG2E_F = 1.4426950408889634f;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool f32_block_map(int G8, int G, int sched, int nblk, bool heavy_is_last, int xs, int nblk_real, int& g, int& blk) {
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int ng = G8 >> 3;
    const int per_bunch = sched * nblk;
    const int bunch = slot / per_bunch;
    const int rem = slot - bunch * per_bunch;
    const int g0 = bunch * sched;
    const int bc = (ng - g0) < sched ? (ng - g0) : sched;
    int qd = rem / bc;
    g = (g0 + (rem - qd * bc)) * 8 + xcd;
    if (g >= G) return false;
    if (xs > 1) {   // block subsets (BwdParams::xs_q / xs_kv): g counts (group, subset) pairs; depth d of subset s is the group's depth d xs + s
        const int s_ = g % xs;
        g /= xs;
        qd = qd * xs + s_;
        if (qd >= nblk_real) return false;
        blk = heavy_is_last ? nblk_real - 1 - qd : qd;
        return true;
    }
    blk = heavy_is_last ? nblk - 1 - qd : qd;
    return true;
}

// two [32][D] fp32 tiles (rows row0 .. row0+31 of a and b, zero past L) global -> registers -> LDS [32][D + 1]
template <int D, int NT>
struct TilePair {
    static constexpr int CH = 32 * D / 4, CPT = CH / NT, KP = D + 1;
    static_assert(CH % NT == 0, "tile chunks must split evenly over the threads");
    f4 ra[CPT], rb[CPT];
    __device__ __forceinline__ void load(const float* a, int64_t a_sl, const float* b, int64_t b_sl, int row0, int L, int tid) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = tid + j * NT;
            const int row = (4 * c) / D, col = (4 * c) % D;
            if (row0 + row < L) {
                ra[j] = *(const f4*)(a + (int64_t)(row0 + row) * a_sl + col);
                rb[j] = *(const f4*)(b + (int64_t)(row0 + row) * b_sl + col);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ra[j][e] = 0.f; rb[j][e] = 0.f; }
            }
        }
    }
    __device__ __forceinline__ void store(float* sa, float* sb, int tid) const {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = tid + j * NT;
            const int row = (4 * c) / D, col = (4 * c) % D;
#pragma unroll
            for (int e = 0; e < 4; ++e) { sa[row * KP + col + e] = ra[j][e]; sb[row * KP + col + e] = rb[j][e]; }
        }
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// dQ (and Dsum)
// ---------------------------------------------------------------------------------------------
template <int D, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 1) void fa_bwd_f32_dq_kernel(const BwdParams p) {
    constexpr int BQ = NWAVES * 32, BK = 32, KP = D + 1, DT = D / 32, NT = NWAVES * 64;
    __shared__ float ks[2][BK * KP];
    __shared__ float vs[2][BK * KP];

    int g, qi;
    if (!f32_block_map(p.G8, p.G, p.sched, p.nq_tiles, true, p.xs_q, p.nq_real, g, qi)) return;
    const int L = p.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.H, hh = g - bb * p.H;
    const float* __restrict__ kg = (const float*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;   // (grouped-query K/V head)
    const float* __restrict__ vg = (const float*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;

    const int q_row0 = qi * BQ + wave * 32;
    const int qrow = q_row0 + r;
    const bool wave_active = q_row0 < L;
    const int qr = qrow < L ? qrow : L - 1;

    // B operands: Q[qrow][2 s + h], dO[qrow][2 s + h]; Dsum = rowsum(dO o O) from the same elements
    float qf[D / 2], dof[D / 2];
    float dsum;
    {
        const float* qp = (const float*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh + (int64_t)qr * p.q_sl + h;
        const float* gp = (const float*)p.dout + (int64_t)bb * p.do_sb + (int64_t)hh * p.do_sh + (int64_t)qr * p.do_sl + h;
        const float* op = (const float*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)qr * p.o_sl + h;
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < D / 2; ++s) {
            qf[s] = qp[2 * s];
            dof[s] = gp[2 * s];
            part = __builtin_fmaf(dof[s], op[2 * s], part);
        }
        dsum = half_swap_sum(part);
    }
    if (wave_active && h == 0 && qrow < L) p.dsum[(int64_t)g * L + qrow] = dsum;
    const float lse2 = qrow < L ? p.lse[(int64_t)g * L + qrow] * LOG2E_F : INFINITY;   // rows past L: P = 0

    const bool causal = p.causal != 0;
    const int kv_end = (causal && (qi + 1) * BQ < L) ? (qi + 1) * BQ : L;
    const int n_tiles = (kv_end + BK - 1) / BK;
    const int wave_last = causal ? q_row0 : L - 1, wave_kv_last = causal ? q_row0 + 31 : L - 1, klim = causal ? qrow : L - 1;
    const float c = p.scale_log2;

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;

    TilePair<D, NT> stage;
    stage.load(kg, p.k_sl, vg, p.v_sl, 0, L, tid);
    stage.store(ks[0], vs[0], tid);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) stage.load(kg, p.k_sl, vg, p.v_sl, (t + 1) * BK, L, tid);   // in flight under this tile's MFMAs
        const int kv0 = t * BK;
        if (wave_active && kv0 <= wave_kv_last) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
            const float* kb = &ks[buf][r * KP + h];
            const float* vb = &vs[buf][r * KP + h];
#pragma unroll
            for (int d2 = 0; d2 < D / 2; ++d2) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[2 * d2], qf[d2], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[2 * d2], dof[d2], dp, 0, 0, 0);
            }
            if (kv0 + BK - 1 > wave_last) {             // the wave's diagonal tile / the ragged last tile
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key > klim) s[e] = -INFINITY;
                }
            }
            // dS^T = P^T o (dP^T - Dsum)   (the softmax scale is applied once, to dQ, at the end)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = fast_exp2(__builtin_fmaf(s[e], c, -lse2)) * (dp[e] - dsum);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float* kr = &ks[buf][((j & 3) + 8 * (j >> 2) + 4 * h) * KP + r];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[32 * dt], s[j], dq[dt], 0, 0, 0);
            }
        }
        if (t + 1 < n_tiles) stage.store(ks[buf ^ 1], vs[buf ^ 1], tid);   // released by the previous barrier
        __syncthreads();
    }

    if (!wave_active || qrow >= L) return;
    float* __restrict__ dqp = (float*)p.dq + (int64_t)bb * p.dq_sb + (int64_t)hh * p.dq_sh + (int64_t)qrow * p.dq_sl;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = dq[dt][4 * rg + j] * p.scale;
            *(f4*)(dqp + 32 * dt + 8 * rg + 4 * h) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dK, dV
// ---------------------------------------------------------------------------------------------
template <int D, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 1) void fa_bwd_f32_dkdv_kernel(const BwdParams p) {
    constexpr int BKW = NWAVES * 32, BQ = 32, KP = D + 1, DT = D / 32, NT = NWAVES * 64;
    __shared__ float qs[2][BQ * KP];
    __shared__ float gs[2][BQ * KP];
    __shared__ float rowc[2][2 * BQ];          // per q tile: lse * log2(e) for 32 rows, then Dsum for 32 rows

    int g, kb;
    // (grouped-query attention: one K/V head per workgroup, its kv_div query heads swept one after the other as one stream
    // of (head, query tile) items - see fa_bwd_dkdv_kernel)
    if (!f32_block_map(p.G8kv, p.Gkv, p.sched_kv, p.nk_tiles, false, p.xs_kv, p.nk_real, g, kb)) return;   // key block 0 is the heaviest
    const int L = p.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.Hkv, hh = g - bb * p.Hkv;
    const int hq0 = hh * p.kv_div;
    const float* __restrict__ qg0 = (const float*)p.q + (int64_t)bb * p.q_sb + (int64_t)hq0 * p.q_sh;
    const float* __restrict__ gg0 = (const float*)p.dout + (int64_t)bb * p.do_sb + (int64_t)hq0 * p.do_sh;
    const float* __restrict__ lse_g0 = p.lse + ((int64_t)bb * p.H + hq0) * L;
    const float* __restrict__ dsum_g0 = p.dsum + ((int64_t)bb * p.H + hq0) * L;

    const int key0 = kb * BKW + wave * 32;
    const int key = key0 + r;
    const bool wave_active = key0 < L;
    const int kr = key < L ? key : L - 1;

    float kf[D / 2], vf[D / 2];
    {
        const float* kp = (const float*)p.k + (int64_t)bb * p.k_sb + (int64_t)hh * p.k_sh + (int64_t)kr * p.k_sl + h;
        const float* vp = (const float*)p.v + (int64_t)bb * p.v_sb + (int64_t)hh * p.v_sh + (int64_t)kr * p.v_sl + h;
#pragma unroll
        for (int s = 0; s < D / 2; ++s) { kf[s] = kp[2 * s]; vf[s] = vp[2 * s]; }
    }

    const bool causal = p.causal != 0;
    const int n_qt = (L + BQ - 1) / BQ;
    const int i0 = causal ? (kb * BKW) / BQ : 0;          // query tiles that can see this workgroup's keys: i0 .. n_qt-1
    const float c = p.scale_log2;

    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }

    TilePair<D, NT> stage;
    float rc_stage = 0.f;
    const int n_sweep = n_qt - i0;                 // query tiles per head
    const int n_items = n_sweep > 0 ? n_sweep * p.kv_div : 0;
    auto item = [&](int j, int& hq, int& i) { hq = p.kv_div == 1 ? 0 : j / n_sweep; i = i0 + j - hq * n_sweep; };
    auto load_rowc = [&](int hq, int i) {   // threads 0..31: lse2, 32..63: Dsum; rows past L: lse2 = +inf (P = 0), Dsum = 0
        const int rr = i * BQ + (tid & 31);
        rc_stage = tid < 32 ? INFINITY : 0.f;
        if (tid < 32 && rr < L) rc_stage = lse_g0[(int64_t)hq * L + rr] * LOG2E_F;
        if (tid >= 32 && tid < 64 && rr < L) rc_stage = dsum_g0[(int64_t)hq * L + rr];
    };
    if (n_items > 0) {
        stage.load(qg0, p.q_sl, gg0, p.do_sl, i0 * BQ, L, tid);
        load_rowc(0, i0);
        stage.store(qs[0], gs[0], tid);
        if (tid < 64) rowc[0][tid] = rc_stage;
    }
    __syncthreads();

    for (int j = 0; j < n_items; ++j) {
        const int buf = j & 1;
        int hq, i;
        item(j, hq, i);
        if (j + 1 < n_items) {
            int hn, in;
            item(j + 1, hn, in);
            stage.load(qg0 + (int64_t)hn * p.q_sh, p.q_sl, gg0 + (int64_t)hn * p.do_sh, p.do_sl, in * BQ, L, tid);
            load_rowc(hn, in);
        }
        const int q0 = i * BQ;
        if (wave_active && (!causal || q0 + BQ - 1 >= key0)) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
            const float* qb = &qs[buf][r * KP + h];
            const float* gb = &gs[buf][r * KP + h];
#pragma unroll
            for (int d2 = 0; d2 < D / 2; ++d2) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[2 * d2], kf[d2], s, 0, 0, 0);     // S[query][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gb[2 * d2], vf[d2], dp, 0, 0, 0);   // dP[query][key]
            }
            if (causal && q0 < key0 + 31) {             // diagonal tile: query < key is masked
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int qq = q0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (qq < key) s[e] = -INFINITY;
                }
            }
            // P = 2^(c S - lse2[q]), dS = P o (dP - Dsum[q]): the row constants vary along the registers
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qi_ = (e & 3) + 8 * (e >> 2) + 4 * h;
                s[e] = fast_exp2(__builtin_fmaf(s[e], c, -rowc[buf][qi_]));
                dp[e] = s[e] * (dp[e] - rowc[buf][BQ + qi_]);
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int qq = (jj & 3) + 8 * (jj >> 2) + 4 * h;
                const float* gr = &gs[buf][qq * KP + r];
                const float* qr = &qs[buf][qq * KP + r];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[32 * dt], s[jj], dv[dt], 0, 0, 0);    // dV^T += dO^T P
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[32 * dt], dp[jj], dk[dt], 0, 0, 0);   // dK^T += Q^T dS
                }
            }
        }
        if (j + 1 < n_items) {
            stage.store(qs[buf ^ 1], gs[buf ^ 1], tid);
            if (tid < 64) rowc[buf ^ 1][tid] = rc_stage;
        }
        __syncthreads();
    }

    if (!wave_active || key >= L) return;
    float* __restrict__ dkp = (float*)p.dk + (int64_t)bb * p.dk_sb + (int64_t)hh * p.dk_sh + (int64_t)key * p.dk_sl;
    float* __restrict__ dvp = (float*)p.dv + (int64_t)bb * p.dv_sb + (int64_t)hh * p.dv_sh + (int64_t)key * p.dv_sl;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f4 wk, wv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { wk[j] = dk[dt][4 * rg + j] * p.scale; wv[j] = dv[dt][4 * rg + j]; }
            *(f4*)(dkp + 32 * dt + 8 * rg + 4 * h) = wk;
            *(f4*)(dvp + 32 * dt + 8 * rg + 4 * h) = wv;
        }
    }
}

// fp32, D in {64, 128}, 16-byte aligned rows; p.nq_tiles = p.nk_tiles = ceil(L / 128)
hipError_t launch_bwd_f32(const BwdParams& p, hipStream_t stream) {
    const dim3 gq((unsigned)((int64_t)p.G8 * p.nq_tiles)), gk((unsigned)((int64_t)p.G8kv * p.nk_tiles));
    hipError_t e;
    if (p.D == 64) {
        e = launch1((const void*)(fa_bwd_f32_dq_kernel<64, 4>), gq, dim3(256), 0, stream, p);
        if (e == hipSuccess) e = launch1((const void*)(fa_bwd_f32_dkdv_kernel<64, 4>), gk, dim3(256), 0, stream, p);
    } else {
        e = launch1((const void*)(fa_bwd_f32_dq_kernel<128, 4>), gq, dim3(256), 0, stream, p);
        if (e == hipSuccess) e = launch1((const void*)(fa_bwd_f32_dkdv_kernel<128, 4>), gk, dim3(256), 0, stream, p);
    }
    return e;
}

}  // namespace fa
