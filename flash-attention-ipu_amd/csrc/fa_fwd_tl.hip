// FlashAttention forward for MI355X (gfx950): TEAM LISTS - a static plan for causal launches that last as long as their heaviest
// workgroup's kv chain (BASELINE config 4's attention call, [8,12,1024,64]: 96 groups x 8 q tiles = 768 workgroups of 2 .. 16 kv tiles on
// 1 024 slots; the launch takes the 16-tile chain's 22 us while its 6 912 tile units are 7 us of matrix time - profiles/r05_cfg4).
//
// Causal block skipping (the reference's doBlock, FA.cpp:244) is what makes the q tiles of a group unequal.  Here a workgroup is 8 waves =
// TWO TEAMS of 4 waves x 32 query rows, each with K0 K1 V0 V1 buffers of its own (2 x 33 KB = 66 KB: two workgroups per CU, four waves per
// SIMD), and each team runs a short host-built LIST of units (fa_capi.cpp build_team_lists):
//   * a HEAVY q tile is cut in two kv ranges that live in the two teams of ONE workgroup - their (m, l, O) meet in LDS, exactly as the
//     kv split groups of fa_fwd_dma.hip (SPLIT = 2) fold: no atomics, no workspace, no partial results in HBM;
//   * LIGHT q tiles are whole units chained in front of a team's half (or paired up in a workgroup of their own), so slots are not wasted.
// [96,1024,64]: {q7a | q7b} {q0 q6a | q6b} {q5a | q5b} {q1 q4a | q4b} {q3 | q2} = 5 workgroups per group = 480 <= 512 slots, the longest
// chain 9 tiles instead of 16.  The plan is static and the hardware dispatcher still places the workgroups (what the round-5 work-list
// kernel lost: returning atomics, folds through memory, list scheduling - DESIGN.md 5).
//
// The two teams advance in LOCK STEP - one workgroup barrier per round, a round = one 64-key tile per team (gfx950 has no named barriers) -
// so the host packs both teams of a workgroup to about the same number of rounds.  Tile arithmetic, LDS images and MFMA operand maps are
// those of fa_fwd_dma_kernel (read its header first) at head_dim 64 with one 32-row block per wave; what is new is the unit machinery:
//   * round r of a team computes tile r of its FLATTENED tile sequence from buffer r & 1 and requests tile r + 1 - which may be the first
//     tile of the next unit: K/V addressing does not depend on the q tile - into the other buffer;
//   * after a unit's last tile the next unit's Q rows are requested into the registers the loop has just released; the finished unit's
//     epilogue runs at the TOP of the next round, staged through the LDS pieces THIS WAVE itself is about to overwrite by DMA (K pieces
//     w, w + 4 and V pieces w, w + 4 of the buffer consumed one round ago: a wave-private 32-row slab, so no barrier is needed);
//   * a split q tile is the LAST unit of both teams; a team that is done early parks its state in registers and keeps the barriers.

#include "fa_device.h"

namespace fa {

template <typename T, bool SKIP>
__device__ __forceinline__ void fa_fwd_tl_body(const TlParams& tp) {
    const FwdParams& p = tp.p;
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int D = 64, NW = 4, WQ = 32, BQ = NW * WQ, BK = 64;
    constexpr int ROWB = D * 2;
    constexpr int KTILE_B = BK * ROWB;          // 8 KiB
    constexpr int VBLK_B = BK * 64 + 64;
    constexpr int VTILE_B = (D / 32) * VBLK_B;
    constexpr int PPW = (KTILE_B / 1024) / NW;  // 1 KiB DMA pieces per wave and tile (K, and as many of V): 2
    constexpr int KS = D / 16, DT = D / 32;
    constexpr int TEAM_B = 2 * KTILE_B + 2 * VTILE_B;   // K0 K1 V0 V1 of one team
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[2 * TEAM_B];
    __shared__ int flags[2 * NW];

    // ---- block -> (group, workgroup type): groups go to XCDs by index mod 8 (a group's K/V live in ONE L2), types heaviest first
    int s_ = blockIdx.x >> 3;
    const int xcd = blockIdx.x & 7;
    if (s_ < p.perm_n) s_ = (int)p.perm[s_];          // placement-balanced order (fa_capi.cpp balanced_order)
    const int ng = p.G8 >> 3;
    const int wt = s_ / ng;
    const int g = (s_ - wt * ng) * 8 + xcd;
    if (g >= p.G) return;

#ifdef FA_STAMPS   /* measurement build only (scripts/tl_stamps.py): s_memtime stamps of this wave, written to p.stamps at the end */
    unsigned long long st_[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st_[i] = 0;
    int st_round = 0;
    (void)st_round;
    st_[0] = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define TL_STAMP(I_) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_[I_] = __builtin_amdgcn_s_memtime(); }
#else
#define TL_STAMP(I_) {}
#endif
    const int L = p.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave_all >> 2, wave = wave_all & 3;
    unsigned char* const smem = smem_all + team * TEAM_B;
    unsigned char* const smem_v = smem + 2 * KTILE_B;
    const unsigned smem_lds = lds_offset(smem);
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* __restrict__ qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;
    const T* __restrict__ vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;
    T* __restrict__ og = (T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh;
    const int dchunks = p.Dr >> 3;

    // this team's unit list: qi | t0 << 8 | t1 << 16 | kind << 24 (kind 0 none, 1 whole q tile, 2 | 3 first | second kv range of a split q tile)
    const unsigned uw0 = tp.prog[wt][team][0], uw1 = tp.prog[wt][team][1], uw2 = tp.prog[wt][team][2];
    const int rounds = (int)tp.rounds[wt];
    const bool wg_split = ((tp.prog[wt][0][0] >> 24) == 2u) || ((tp.prog[wt][0][1] >> 24) == 2u) || ((tp.prog[wt][0][2] >> 24) == 2u);

    vec8 qf[KS];
    // (through an opaque copy of the lane index: the compiler otherwise forms the per-lane row addresses of EVERY unit's load up front and
    //  keeps them - eight registers each - across the kv loop; at 128 registers that spills)
#define TL_LOAD_Q(QI_)                                                                             \
    {                                                                                              \
        int lq_ = lane;                                                                            \
        asm volatile("" : "+v"(lq_));                                                              \
        const int r = lq_ & 31, h = lq_ >> 5;                                                      \
        const int qr0_ = (QI_) * BQ + wave * WQ + r;                                               \
        const T* qp = qg + (int64_t)(qr0_ < L ? qr0_ : L - 1) * p.q_sl;                            \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                        \
            const int ch_ = 2 * ks + h;                                                            \
            const vec8 ld_ = *(const vec8*)(qp + 8 * (ch_ < dchunks ? ch_ : 0));                   \
            vec8 z_;                                                                               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) z_[j] = (T)0.f;                          \
            qf[ks] = ch_ < dchunks ? ld_ : z_;                                                     \
        }                                                                                          \
    }

    // ---- LDS-DMA geometry (fa_fwd_dma.hip): piece pc = wave + j * NW of a tile, lane -> source byte offset
    const unsigned k_slb = (unsigned)p.k_sl * 2, v_slb = (unsigned)p.v_sl * 2;
    unsigned k_vo[PPW], v_vo[PPW];
    int k_dst[PPW], v_dst[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pc = wave + j * NW;
        const int row = (pc * 64 + lane) / (ROWB / 16);
        const int cp = (pc * 64 + lane) % (ROWB / 16);
        const int sw = (row >> 1) & 7;
        k_vo[j] = (cp ^ sw) < dchunks ? (unsigned)row * k_slb + ((cp ^ sw) << 4) : 0x80000000u;
        k_dst[j] = pc * 1024;
        const int dblk = pc >> 2, q2 = pc & 3;
        const int key = (2 * q2 + (lane >> 5)) * 8 + ((lane >> 2) & 7);
        v_vo[j] = (dblk * 4 + (lane & 3)) < dchunks ? (unsigned)key * v_slb + ((dblk * 4 + (lane & 3)) << 4) : 0x80000000u;
        v_dst[j] = dblk * VBLK_B + q2 * 1024;
    }
    const unsigned k_bytes = ((unsigned)(L - 1) * (unsigned)p.k_sl + (unsigned)p.Dr) * 2;
    const unsigned v_bytes = ((unsigned)(L - 1) * (unsigned)p.v_sl + (unsigned)p.Dr) * 2;
#define TL_DMA_TILE(T_, BUF_)                                                                      \
    {                                                                                              \
        const unsigned ko_ = (unsigned)(T_) * BK * k_slb, vo_ = (unsigned)(T_) * BK * v_slb;       \
        const i32x4 kr_ = make_raw_rsrc((const char*)kg + ko_, k_bytes - ko_);                     \
        const i32x4 vr_ = make_raw_rsrc((const char*)vg + vo_, v_bytes - vo_);                     \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) {                                          \
            lds_dma_b128(kr_, smem_lds + (BUF_) * KTILE_B + k_dst[j], k_vo[j]);                    \
            lds_dma_b128(vr_, smem_lds + 2 * KTILE_B + (BUF_) * VTILE_B + v_dst[j], v_vo[j]);      \
        }                                                                                          \
    }
#define TL_BARRIER() { lds_dma_wait_all(); __syncthreads(); }

    int k_rd[KS];
    {
        const int sw = (r >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * ROWB + (((2 * ks + h) ^ sw) << 4);
    }
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    f32x16 o[DT];
    float m;
    f32x4 lsum;
    const vec8 asel = rowsum_selector<T>(lane);
#define TL_RESET_STATE()                                                                           \
    {                                                                                              \
        m = -INFINITY;                                                                             \
        lsum = f32x4{0.f, 0.f, 0.f, 0.f};                                                          \
        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                          \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;                         \
    }
    const float c = p.scale_log2;

    // ---- one kv tile of the current unit (fa_fwd_dma.hip's tile at QB = 1; the launch is causal)
#define TL_QK(BUF_)                                                                                \
    {                                                                                              \
        const unsigned char* kb_ = smem + (BUF_) * KTILE_B;                                        \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }               \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                        \
            const vec8 k0 = *(const vec8*)(kb_ + k_rd[ks]);                                        \
            const vec8 k1 = *(const vec8*)(kb_ + 32 * ROWB + k_rd[ks]);                            \
            s0 = E::mfma(k0, qf[ks], s0);                                                          \
            s1 = E::mfma(k1, qf[ks], s1);                                                          \
        }                                                                                          \
        if (kv0 + BK - 1 > q_row0) {      /* a diagonal tile of the wave */                        \
            const int klim = q_row0 + r;                                                           \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                       \
                const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * h;                              \
                if (key > klim) s0[e] = -INFINITY;                                                 \
                if (key + 32 > klim) s1[e] = -INFINITY;                                            \
            }                                                                                      \
        }                                                                                          \
    }
#define TL_MAX_RESCALE(THR_)                                                                       \
    {                                                                                              \
        float mx = max2(s0[0], s1[0]);                                                             \
        _Pragma("unroll") for (int e = 1; e < 16; ++e) mx = max3(mx, s0[e], s1[e]);                \
        mx = half_swap_max(mx);                                                                    \
        if (__builtin_amdgcn_ballot_w64((mx - m) * c > (THR_)) != 0) {                             \
            const float m_new = max2(m, mx);                                                       \
            const float alpha = fast_exp2((m - m_new) * c);                                        \
            m = m_new;                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) lsum[j] *= alpha;                        \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                      \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;                  \
        }                                                                                          \
    }
#define TL_SOFTMAX()                                                                               \
    {                                                                                              \
        const float mc = m * c;                                                                    \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                           \
            s0[e] = fast_exp2(__builtin_fmaf(s0[e], c, -mc));                                      \
            s1[e] = fast_exp2(__builtin_fmaf(s1[e], c, -mc));                                      \
        }                                                                                          \
    }
#define TL_PV(BUF_)                                                                                \
    {                                                                                              \
        const unsigned char* vb_ = smem_v + (BUF_) * VTILE_B;                                      \
        vec8 pf[4];                                                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
            pf[0][j] = (T)s0[j];                                                                   \
            pf[1][j] = (T)s0[8 + j];                                                               \
            pf[2][j] = (T)s1[j];                                                                   \
            pf[3][j] = (T)s1[8 + j];                                                               \
        }                                                                                          \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) lsum = E::mfma16(asel, pf[kk], lsum);     \
        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                        \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                     \
                const unsigned char* vp = vb_ + dt * VBLK_B + kk * 1024 + v_rd;                    \
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                          \
                    (i16x4 __attribute__((address_space(3)))*)(vp));                               \
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                          \
                    (i16x4 __attribute__((address_space(3)))*)(vp + 512));                         \
                const i16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);          \
                o[dt] = E::mfma(__builtin_bit_cast(vec8, v8), pf[kk], o[dt]);                      \
            }                                                                                      \
        }                                                                                          \
    }
    // tile t of the unit that starts at tile t_first (skip-max: the reference max is set exactly on the unit's first tile - every wave that
    // computes any tile of a unit computes its first one, tiles only leave a wave's causal range towards the end)
#define TL_COMPUTE(BUF_, MODE_)                                                                    \
    {                                                                                              \
        const int kv0 = t * BK;                                                                    \
        if (q_row0 < L && kv0 <= q_row0 + WQ - 1) {                                                \
            f32x16 s0, s1;                                                                         \
            TL_QK(BUF_)                                                                            \
            if ((MODE_) == 0) {                                                                    \
                if (t == t_first) TL_MAX_RESCALE(0.0f)                                             \
            } else {                                                                               \
                TL_MAX_RESCALE(RESCALE_THR)                                                        \
            }                                                                                      \
            TL_SOFTMAX()                                                                           \
            TL_PV(BUF_)                                                                            \
        }                                                                                          \
    }

    // ---- epilogue of a finished unit whose rows start at row0_ (this wave: row0_ + 0..31): O /= l through a wave-private 32-row slab so that
    // every store instruction writes whole rows (fa_fwd_dma.hip).  The slab is FOUR 1 KiB regions (8 rows each) given by their LDS pointers:
    // in the round loop the pieces of the idle buffer this wave itself fills by DMA, after the loop 4 KiB of the team's free buffers.
    const auto epilogue = [&](int row0_, unsigned char* rg0, unsigned char* rg1, unsigned char* rg2, unsigned char* rg3) {
        if (row0_ >= L) return;
        int le_ = lane;                                          // (opaque: the slab and output addresses are formed here, not carried through the kv loop)
        asm volatile("" : "+v"(le_));
        const int lane = le_, r = le_ & 31, h = le_ >> 5;
        const float lt = lsum[0];
        const float inv = __builtin_amdgcn_rcpf(lt);
        unsigned char* const wr = ((r >> 3) == 0 ? rg0 : (r >> 3) == 1 ? rg1 : (r >> 3) == 2 ? rg2 : rg3) + (r & 7) * ROWB;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (T)(o[dt][4 * q4 + j] * inv);
                const int ch = 4 * dt + q4;
                *(vec4*)(wr + ((ch ^ (r & 7)) << 4) + 8 * h) = w;
            }
        }
        const int row_b = row0_ + r;
        if (p.lse != nullptr && h == 0 && row_b < L) p.lse[(int64_t)g * L + row_b] = m * p.scale + __logf(lt);
        asm volatile("" ::: "memory");                       // (the slab rows are read back by other lanes of this wave)
        const int rl = lane >> 3, ch = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                            // 64 chunks per store instruction = 8 rows
            unsigned char* const rg = i == 0 ? rg0 : i == 1 ? rg1 : i == 2 ? rg2 : rg3;
            const u32x4 val = *(const u32x4*)(rg + rl * ROWB + ((ch ^ rl) << 4));
            const int orow = row0_ + 8 * i + rl;
            if (orow < L && ch < dchunks) *(u32x4*)(og + (int64_t)orow * p.o_sl + ch * 8) = val;
        }
    };

    // ---- the program: prologue, `rounds` lock-step rounds (unrolled by two: every LDS address is base + immediate), fold, last epilogues
#define TL_ROUND(BUF_, MODE_)                                                                      \
    {                                                                                              \
        if (active) {                                                                              \
            if (pend) {      /* the previous unit's epilogue, through the pieces of the idle buffer this wave fills next */ \
                unsigned char* const kb_ = smem + ((BUF_) ^ 1) * KTILE_B;                          \
                unsigned char* const vb_ = smem_v + ((BUF_) ^ 1) * VTILE_B;                        \
                epilogue(ep_row0, kb_ + k_dst[0], kb_ + k_dst[1], vb_ + v_dst[0], vb_ + v_dst[1]); \
                TL_RESET_STATE()                                                                   \
                pend = false;                                                                      \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* slab reads done before the DMA below rewrites the pieces */ \
            }                                                                                      \
            const bool last_ = t + 1 == t_end;                                                     \
            int nt_ = t + 1;                                                                       \
            unsigned nuw_ = 0u;                                                                    \
            bool have_next_ = true;                                                                \
            if (last_) {                                                                           \
                nuw_ = ui == 0 ? uw1 : ui == 1 ? uw2 : 0u;                                         \
                have_next_ = (nuw_ >> 24) != 0u;                                                   \
                nt_ = (int)((nuw_ >> 8) & 255u);                                                   \
            }                                                                                      \
            if (have_next_) TL_DMA_TILE(nt_, (BUF_) ^ 1)                                           \
            TL_COMPUTE(BUF_, MODE_)                                                                \
            if (last_) {                                                                           \
                if ((MODE_) == 0) bad |= q_row0 < L && sum_left_safe_range(lsum[0]);               \
                if (kind == 1) { pend = true; ep_row0 = q_row0; }                                  \
                if (have_next_) {                                                                  \
                    ++ui;                                                                          \
                    t_first = nt_;                                                                 \
                    t_end = (int)((nuw_ >> 16) & 255u);                                            \
                    kind = (int)(nuw_ >> 24);                                                      \
                    q_row0 = (int)(nuw_ & 255u) * BQ + wave * WQ;                                  \
                    TL_LOAD_Q((int)(nuw_ & 255u))                                                  \
                    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks])); \
                } else {                                                                           \
                    active = false;                                                                \
                }                                                                                  \
            }                                                                                      \
            t = nt_;                                                                               \
        }                                                                                          \
        TL_BARRIER()                                                                               \
    }
#ifdef FA_STAMPS
#define TL_STAMP_ROUND(R_) { if ((R_) < 10) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); _Pragma("unroll") for (int i = 0; i < 10; ++i) if (i == (R_)) st_[6 + i] = t_; } }
#else
#define TL_STAMP_ROUND(R_) {}
#endif
#define TL_PROGRAM(MODE_)                                                                          \
    {                                                                                              \
        int ui = 0;                                                                                \
        int kind = (int)(uw0 >> 24);                                                               \
        bool active = kind != 0;                                                                   \
        int t = (int)((uw0 >> 8) & 255u), t_first = t, t_end = (int)((uw0 >> 16) & 255u);          \
        int q_row0 = (int)(uw0 & 255u) * BQ + wave * WQ;                                           \
        bool pend = false;                                                                         \
        int ep_row0 = 0;                                                                           \
        TL_RESET_STATE()                                                                           \
        if (active) {                                                                              \
            TL_LOAD_Q((int)(uw0 & 255u))                                                           \
            TL_DMA_TILE(t, 0)                                                                      \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));     \
        }                                                                                          \
        TL_BARRIER()                                                                               \
        TL_STAMP(1)                                                                                \
        for (int rr = 0; rr < rounds; rr += 2) {       /* (workgroup-uniform bounds) */            \
            TL_ROUND(0, MODE_)                                                                     \
            TL_STAMP_ROUND(rr)                                                                     \
            if (rr + 1 < rounds) { TL_ROUND(1, MODE_) TL_STAMP_ROUND(rr + 1) }                     \
        }                                                                                          \
        TL_STAMP(2)                                                                                \
        /* every buffer is free now.  A split q tile: team 1 hands its (m, l, O) to team 0 through LDS (a lane's values at [value][lane]: */ \
        /* conflict-free; the area lies behind team 0's epilogue slabs).  A wave that saw no key of a row hands over m = -inf, l = 0, O = 0. */ \
        if (wg_split) {                                                                            \
            constexpr int NVAL = DT * 16 + 2;                                                      \
            float* xb = (float*)(smem_all + NW * WQ * ROWB) + wave * (NVAL * 64) + lane;           \
            static_assert(NW * WQ * ROWB + NW * NVAL * 64 * 4 <= 2 * TEAM_B, "exchange area must fit the K/V buffers"); \
            if (team == 1 && kind == 3) {                                                          \
                xb[0] = m;                                                                         \
                xb[64] = lsum[0];                                                                  \
                _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                  \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) xb[(2 + dt * 16 + e) * 64] = o[dt][e]; \
            }                                                                                      \
            __syncthreads();                                                                       \
            if (team == 0 && kind == 2 && q_row0 < L) {                                            \
                const float m1 = xb[0], l1 = xb[64];                                               \
                const float mn = max2(m, m1);                                                      \
                const float a0 = fast_exp2((m - mn) * c), a1 = fast_exp2((m1 - mn) * c);           \
                m = mn;                                                                            \
                const float ln = lsum[0] * a0 + l1 * a1;                                           \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) lsum[j] = ln;                        \
                _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                  \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e)                                 \
                        o[dt][e] = o[dt][e] * a0 + xb[(2 + dt * 16 + e) * 64] * a1;                \
                pend = true;                                                                       \
                ep_row0 = q_row0;                                                                  \
            }                                                                                      \
        }                                                                                          \
        if (pend) {                                                                                \
            unsigned char* const sl_ = smem + wave * (WQ * ROWB);                                  \
            epilogue(ep_row0, sl_, sl_ + 1024, sl_ + 2048, sl_ + 3072);                            \
        }                                                                                          \
    }

    bool bad = false;
    if constexpr (SKIP) {
        TL_PROGRAM(0)
        // Did any row of the workgroup leave the safe range (fa_fwd_dma.hip)?  Practically never: then the whole program runs again
        // with the exact per-tile max and its stores replace the first pass's.
        flags[wave_all] = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
        __syncthreads();
        int any_bad = 0;
#pragma unroll
        for (int w = 0; w < 2 * NW; ++w) any_bad |= flags[w];
        if (any_bad) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            TL_PROGRAM(1)
        }
    } else {
        TL_PROGRAM(1)
    }
    (void)bad;
#ifdef FA_STAMPS
    TL_STAMP(3)
    if (p.stamps != nullptr && lane == 0) {   // 16 u64 per wave: entry, prologue done, rounds done, end | hw id | rounds, team, type | ten round stamps
        unsigned long long* sp = p.stamps + ((int64_t)blockIdx.x * (2 * NW) + wave_all) * 16;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st_[4] = ((unsigned long long)xcc << 32) | hwid;
        st_[5] = (unsigned long long)rounds | ((unsigned long long)team << 8) | ((unsigned long long)wt << 16) | ((unsigned long long)(__builtin_amdgcn_s_memrealtime() - rt0_) << 32);
#pragma unroll
        for (int i = 0; i < 16; ++i) sp[i] = st_[i];
    }
#endif
#undef TL_LOAD_Q
#undef TL_DMA_TILE
#undef TL_BARRIER
#undef TL_RESET_STATE
#undef TL_QK
#undef TL_MAX_RESCALE
#undef TL_SOFTMAX
#undef TL_PV
#undef TL_COMPUTE
#undef TL_ROUND
#undef TL_PROGRAM
}

// 8 waves, two workgroups per CU (66 KB of LDS each) = four waves per SIMD: 128 registers
template <typename T, bool SKIP>
__global__ __launch_bounds__(512, 4) void fa_fwd_tl_kernel(const TlParams tp) {
    fa_fwd_tl_body<T, SKIP>(tp);
}

hipError_t launch_fwd_tl(const TlParams& tp, int dtype, hipStream_t stream) {
    const dim3 grid((unsigned)((int64_t)tp.p.G8 * tp.nw));
    if (tp.p.D != 64 || !tp.p.causal || tp.nw < 1 || tp.nw > TL_MAXW) return hipErrorInvalidValue;
    if (dtype == 1) return launch1((const void*)(fa_fwd_tl_kernel<__bf16, true>), grid, dim3(512), 0, stream, tp);
    if (tp.p.skipmax) return launch1((const void*)(fa_fwd_tl_kernel<_Float16, true>), grid, dim3(512), 0, stream, tp);
    return launch1((const void*)(fa_fwd_tl_kernel<_Float16, false>), grid, dim3(512), 0, stream, tp);
}

}  // namespace fa
