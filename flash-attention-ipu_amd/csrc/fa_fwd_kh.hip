// FlashAttention forward for MI355X (gfx950): 64-row workgroups whose waves split every 64-key tile into KEY HALVES - for launches
// that are bound by the serial chain of their heaviest workgroup, not by the matrix pipe.
//
// Same LDS images, MFMA operand maps, LDS-DMA staging, double-buffered 64-key tiles and skip-max arithmetic (optimistic pass + exact
// fallback pass) as fa_fwd_dma_kernel - read fa_fwd_kernels.hip and fa_fwd_dma.hip first.  What changes is the shape of a wave's work.
//
// A causal launch whose workgroups are all resident at once (BASELINE config 4's attention call: 96 groups x L = 1024 = 768 128-row
// workgroups on 1024 slots) lasts as long as its heaviest workgroup's kv loop: 16 tiles x ~0.95 us, each tile one dependency chain per
// wave - QK^T (8 MFMA) -> exponentials of 32 scores per lane -> PV (8 MFMA) - while the CU's pipes are half idle.
// Splitting that loop over more waves of a bigger workgroup costs slots (8-wave workgroups start in two rounds: measured, 3.0e).
// Here the workgroup stays at 4 waves and 33 KB of LDS but owns 64 query rows: wave w = (row block w & 1, key half w >> 1) computes,
// of every streamed tile, the scores of ITS 32 rows against ITS 32 keys - 4 + 4 MFMAs and 16 scores per lane: half the chain per tile -
// with softmax statistics of its own; the two key halves of a row block are merged through LDS once, before the epilogue
// (m, l, O of the upper half folded into the lower half's, as the kv split groups of fa_fwd_dma_kernel are).  Twice as many
// workgroups of half the work each: the grid fills the chip's slots better, and the heaviest chain is half as long.
//
// SPLIT > 1 adds the kv split of fa_fwd_dma_kernel on top (grids of a few hundred workgroups at long L): SPLIT groups of 4 waves own the
// SAME 64 query rows, group s streams the tiles s, s + SPLIT, ... through LDS buffers of its own in lock step with the others, and all
// 2 SPLIT (group, key half) states of a row block are folded into one before the epilogue.
// (Measured and dropped, DESIGN.md 3.0h: a ring of three tile buffers - slower everywhere, the request issued one tile ahead has landed -
// and a 128-row shape in which every wave takes both key halves.)

#include "fa_device.h"

namespace fa {

template <typename T, int D, int SPLIT>
__device__ __forceinline__ void fa_fwd_kh_body(const FwdParams& p) {
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int NW = 4;                       // waves per split group: 2 row blocks x 2 key halves
    constexpr int BQ = 64, BK = 64;
    constexpr int ROWB = D * 2;
    constexpr int KTILE_B = BK * ROWB;
    constexpr int VBLK_B = BK * 64 + 64;
    constexpr int VTILE_B = (D / 32) * VBLK_B;
    constexpr int PIECES = KTILE_B / 1024;      // 1 KiB DMA pieces per K (or V) tile
    constexpr int PPW = PIECES / NW;
    constexpr int KS = D / 16, DT = D / 32;
    constexpr int GROUP_B = 2 * KTILE_B + 2 * VTILE_B;     // K0 K1 V0 V1 of one split group
    static_assert(PIECES % NW == 0 && PPW >= 1, "pieces must split evenly over the waves");

    __shared__ __attribute__((aligned(16))) unsigned char smem_all[SPLIT * GROUP_B];
    __shared__ int flags[NW * SPLIT];

    const int nq = p.nq_tiles;
    int g, qi;
    {
        int b = blockIdx.x;
        if (p.perm_n > 0) {                            // placement-balanced order of a grid below one round of slots (fa_capi.cpp)
            const int s_ = b >> 3;
            b = (s_ < p.perm_n ? (int)p.perm[s_] : s_) * 8 + (b & 7);
        }
        const int xcd = b & 7, slot = b >> 3;
        const int ng = p.G8 >> 3;
        const int per_bunch = p.sched * nq;
        const int bunch = slot / per_bunch;
        const int rem = slot - bunch * per_bunch;
        const int g0 = bunch * p.sched;
        const int bc = (ng - g0) < p.sched ? (ng - g0) : p.sched;
        const int qd = rem / bc;
        qi = nq - 1 - qd;                          // heaviest first
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = SPLIT == 1 ? 0 : wave_all >> 2;           // kv split group
    const int wave = wave_all & 3;
    const int rb = wave & 1, kh = wave >> 1;                 // row block, key half
    const int vg = 2 * ws + kh;                              // which of the 2 SPLIT partial states of its row block this wave holds
    unsigned char* const smem = smem_all + ws * GROUP_B;
    unsigned char* const smem_v = smem + 2 * KTILE_B;
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* __restrict__ qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;
    const T* __restrict__ vg_ = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;

    const int q_row0 = qi * BQ + rb * 32;
    const int qrow = q_row0 + r;
    const bool wave_active = q_row0 < L;

    const int dchunks = p.Dr >> 3;     // the operands' true head dim in 16-byte chunks (= D / 8 unless the call runs zero-extended: FwdParams::Dr)
    vec8 qf[KS];
    {
        const T* qp = qg + (int64_t)(qrow < L ? qrow : L - 1) * p.q_sl;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {       // (columns past the true head dim: zeros)
            const int ch_ = 2 * ks + h;
            const vec8 ld_ = *(const vec8*)(qp + 8 * (ch_ < dchunks ? ch_ : 0));
            vec8 z_;
#pragma unroll
            for (int j = 0; j < 8; ++j) z_[j] = (T)0.f;
            qf[ks] = ch_ < dchunks ? ld_ : z_;
        }
    }

    // ---- LDS-DMA geometry (fa_fwd_dma.hip): piece pc = wave + j NW of a tile, lane -> source byte offset
    const unsigned k_slb = (unsigned)p.k_sl * 2, v_slb = (unsigned)p.v_sl * 2;
    unsigned k_vo[PPW], v_vo[PPW];
    int k_dst[PPW], v_dst[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pc = wave + j * NW;
        const int row = pc * (1024 / ROWB) + lane / (ROWB / 16);
        const int cp = lane % (ROWB / 16);
        const int sw = (D == 64) ? ((row >> 1) & 7) : (row & 15);
        // (a chunk past the true head dim: an offset beyond the descriptor's extent - the DMA writes zeros, as for rows past L)
        k_vo[j] = (cp ^ sw) < dchunks ? (unsigned)row * k_slb + ((cp ^ sw) << 4) : 0x80000000u;
        k_dst[j] = pc * 1024;
        const int dblk = pc >> 2, q2 = pc & 3;
        const int key = (2 * q2 + (lane >> 5)) * 8 + ((lane >> 2) & 7);
        v_vo[j] = (dblk * 4 + (lane & 3)) < dchunks ? (unsigned)key * v_slb + ((dblk * 4 + (lane & 3)) << 4) : 0x80000000u;
        v_dst[j] = dblk * VBLK_B + q2 * 1024;
    }
    const unsigned k_bytes = ((unsigned)(L - 1) * (unsigned)p.k_sl + (unsigned)p.Dr) * 2;
    const unsigned v_bytes = ((unsigned)(L - 1) * (unsigned)p.v_sl + (unsigned)p.Dr) * 2;
    const unsigned smem_lds = lds_offset(smem);
    auto dma_tile = [&](int t, int buf) {          // descriptors start at the tile's first row; rows past L read as zeros
        const unsigned ko = (unsigned)t * BK * k_slb, vo = (unsigned)t * BK * v_slb;
        const i32x4 kr = make_raw_rsrc((const char*)kg + ko, k_bytes - ko);
        const i32x4 vr = make_raw_rsrc((const char*)vg_ + vo, v_bytes - vo);
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            lds_dma_b128(kr, smem_lds + buf * KTILE_B + k_dst[j], k_vo[j]);
            lds_dma_b128(vr, smem_lds + 2 * KTILE_B + buf * VTILE_B + v_dst[j], v_vo[j]);
        }
    };
    auto tile_barrier = [&]() { lds_dma_wait_all(); __syncthreads(); };

    // LDS read offsets of this wave's key half: K rows 32 kh + r of a tile, V^T fragments kk = 2 kh, 2 kh + 1 (16 keys each)
    int k_rd[KS];
    {
        const int sw = (D == 64) ? ((r >> 1) & 7) : (r & 15);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] = (32 * kh + r) * ROWB + (((2 * ks + h) ^ sw) << 4);
    }
    const int v_rd = 2 * kh * 1024 + (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    const bool causal = p.causal != 0;
    const int kv_end = (causal && (qi + 1) * BQ < L) ? (qi + 1) * BQ : L;
    const int n_tiles = (kv_end + BK - 1) / BK;
    const int n_rounds = (n_tiles + SPLIT - 1) / SPLIT;        // rounds of the workgroup's lock-step loop (uniform over all its waves)
    const int wave_last = causal ? q_row0 : L - 1;             // a half tile reaching past this key carries a mask for the wave
    const int wave_kv_last = causal ? q_row0 + 31 : L - 1;     // half tiles starting after it hold nothing for the wave
    const int klim = causal ? qrow : L - 1;

    f32x16 o[DT];
    float m;
    f32x4 lsum;
    const vec8 asel = rowsum_selector<T>(lane);
    const float c = p.scale_log2;
    auto reset_state = [&]() {
        m = -INFINITY;
        lsum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    };

    // this wave's half of tile t; EXACT: per-tile row max with lazy rescale, else the reference max is set on the wave's first half
    // tile and left alone (fa_fwd_dma.hip: skip-max)
    auto compute = [&](int t, int buf, bool exact, bool& first) {
        const int k0 = t * BK + 32 * kh;            // first key of the half tile
        if (!(wave_active && t < n_tiles && k0 <= wave_kv_last)) return;
        const unsigned char* kb = smem + buf * KTILE_B;
        const unsigned char* vb = smem_v + buf * VTILE_B;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = E::mfma(*(const vec8*)(kb + k_rd[ks]), qf[ks], s);
        if (k0 + 31 > wave_last) {                  // the wave's diagonal half tile / the ragged last one
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (key > klim) s[e] = -INFINITY;
            }
        }
        if (exact || first) {
            float mx = max2(s[0], s[1]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) mx = max3(mx, s[e], s[e + 1]);
            mx = half_swap_max(mx);
            const float thr = first ? 0.0f : RESCALE_THR;
            if (__builtin_amdgcn_ballot_w64((mx - m) * c > thr) != 0) {
                const float m_new = max2(m, mx);
                const float alpha = fast_exp2((m - m_new) * c);
                m = m_new;
#pragma unroll
                for (int j = 0; j < 4; ++j) lsum[j] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            }
            first = false;
        }
        const float mc = m * c;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = fast_exp2(__builtin_fmaf(s[e], c, -mc));
        vec8 pf[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) { pf[0][j] = (T)s[j]; pf[1][j] = (T)s[8 + j]; }
        lsum = E::mfma16(asel, pf[0], lsum);
        lsum = E::mfma16(asel, pf[1], lsum);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned char* vp = vb + dt * VBLK_B + kk * 1024 + v_rd;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp + 512));
                const i16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = E::mfma(__builtin_bit_cast(vec8, v8), pf[kk], o[dt]);
            }
        }
    };
    // one pass over the workgroup's kv tiles: group ws streams the tiles ws, ws + SPLIT, ... (the loop is unrolled by two: LDS addresses
    // are base + immediate; every bound is workgroup-uniform)
    auto pass = [&](bool exact) {
        reset_state();
        bool first = true;
        if (ws < n_tiles) dma_tile(ws, 0);
        tile_barrier();
        for (int rr = 0; rr < n_rounds; rr += 2) {
            const int t = ws + rr * SPLIT;               // this group's tile of the round
            if (t + SPLIT < n_tiles) dma_tile(t + SPLIT, 1);
            compute(t, 0, exact, first);
            tile_barrier();
            if (rr + 1 < n_rounds) {
                if (t + 2 * SPLIT < n_tiles) dma_tile(t + 2 * SPLIT, 0);
                compute(t + SPLIT, 1, exact, first);
                tile_barrier();
            }
        }
    };

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // Q loads issued before the first tile is waited for

    const bool optimistic = !E::is_f16 || p.skipmax != 0;
    pass(!optimistic);
    if (optimistic) {
        // Did any row of the workgroup leave the safe range?  (All waves must agree: the exact pass re-streams K/V through the same barriers.)
        const bool bad = wave_active && sum_left_safe_range(lsum[0]);
        flags[wave_all] = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
        __syncthreads();
        int any_bad = 0;
#pragma unroll
        for (int w = 0; w < NW * SPLIT; ++w) any_bad |= flags[w];
        if (any_bad) {
            __syncthreads();
            pass(true);
        }
    }

    // ---- fold the other (split group, key half) states of a row block into state 0, one at a time, through LDS (the K/V buffers are free
    // after the pass's last barrier; the exchange area lies behind the epilogue's row slabs; a lane's values sit at [value][lane]:
    // conflict-free).  A state that saw no key of a row hands over m = -inf, l = 0, O = 0: its weight 2^((m - m_new) c) is 0.
    {
        constexpr int NVAL = DT * 16 + 2;
        float* xb = (float*)(smem_all + 2 * 32 * ROWB) + rb * (NVAL * 64) + lane;
        static_assert(2 * 32 * ROWB + 2 * NVAL * 64 * 4 <= SPLIT * GROUP_B, "exchange area must fit the K/V buffers");
        for (int v = 1; v < 2 * SPLIT; ++v) {
            if (v > 1) __syncthreads();
            if (vg == v) {
                xb[0] = m;
                xb[64] = lsum[0];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xb[(2 + dt * 16 + e) * 64] = o[dt][e];
            }
            __syncthreads();
            if (vg == 0 && wave_active) {
                const float m1 = xb[0], l1 = xb[64];
                const float mn = max2(m, m1);
                const float a0 = fast_exp2((m - mn) * c), a1 = fast_exp2((m1 - mn) * c);
                m = mn;
                const float ln = lsum[0] * a0 + l1 * a1;
#pragma unroll
                for (int j = 0; j < 4; ++j) lsum[j] = ln;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] = o[dt][e] * a0 + xb[(2 + dt * 16 + e) * 64] * a1;
            }
        }
        if (vg != 0) return;
    }
    // ---- epilogue (fa_fwd_dma.hip): O /= l staged through a private 32-row LDS slab so that every store writes whole rows
    if (!wave_active) return;
    unsigned char* slab = smem_all + rb * (32 * ROWB);
    constexpr int CPR = D / 8;      // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;   // rows per store instruction
    T* __restrict__ og = (T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh;
    const float lt = lsum[0];
    const float inv = __builtin_amdgcn_rcpf(lt);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            vec4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (T)(o[dt][4 * rg + j] * inv);
            const int ch = 4 * dt + rg;
            *(vec4*)(slab + r * ROWB + ((ch ^ (r & (CPR - 1))) << 4) + 8 * h) = w;
        }
    }
    if (p.lse != nullptr && h == 0 && qrow < L) p.lse[(int64_t)g * L + qrow] = m * p.scale + __logf(lt);
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int rr = i * RPI + lane / CPR, ch = lane % CPR;
        const u32x4 val = *(const u32x4*)(slab + rr * ROWB + ((ch ^ (rr & (CPR - 1))) << 4));
        const int orow = q_row0 + rr;
        if (orow < L && ch < dchunks) *(u32x4*)(og + (int64_t)orow * p.o_sl + ch * 8) = val;
    }
}

template <typename T, int D, int SPLIT>
__global__ __launch_bounds__(256 * SPLIT) void fa_fwd_kh_kernel(const FwdParams p) {
    fa_fwd_kh_body<T, D, SPLIT>(p);
}

template <typename T, int D>
static hipError_t launch_kh_t(const FwdParams& p, int split, hipStream_t stream) {
    const dim3 grid((unsigned)((int64_t)p.G8 * p.nq_tiles));
    if (split == 2) return launch1((const void*)(fa_fwd_kh_kernel<T, D, 2>), grid, dim3(512), 0, stream, p);
    if (split == 4) {   // 16 waves, 132 KB of LDS at head_dim 64 (264 KB at 128: does not exist)
        if constexpr (D == 64) {
            return launch1((const void*)(fa_fwd_kh_kernel<T, D, 4>), grid, dim3(1024), 0, stream, p);
        } else {
            return hipErrorInvalidValue;
        }
    }
    return launch1((const void*)(fa_fwd_kh_kernel<T, D, 1>), grid, dim3(256), 0, stream, p);
}

// callers guarantee: D in {64, 128}; split = kv split groups per workgroup (1 | 2; 4 at head_dim 64); p.nq_tiles = 64-row q tiles per
// group; grid = G8 x q tiles
hipError_t launch_fwd_kh(const FwdParams& p, int dtype, int split, hipStream_t stream) {
    if (dtype == 1) return p.D == 64 ? launch_kh_t<__bf16, 64>(p, split, stream) : launch_kh_t<__bf16, 128>(p, split, stream);
    return p.D == 64 ? launch_kh_t<_Float16, 64>(p, split, stream) : launch_kh_t<_Float16, 128>(p, split, stream);
}

}  // namespace fa
