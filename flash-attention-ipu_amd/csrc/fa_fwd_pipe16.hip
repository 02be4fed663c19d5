// FlashAttention forward for MI355X (gfx950): the software-pipelined kernel of fa_fwd_pipe.hip rebuilt on
// v_mfma_f32_16x16x32 (read fa_fwd_pipe.hip first: same rings, LDS-DMA staging, step barrier, optimistic skip-max
// arithmetic with an exact fallback pass, PV-first step order with the next step's V^T fragments requested mid-step).
//
// Why a second MFMA shape: the forward is power-bound - the in-kernel clock under the pipelined 32x32x16 kernels is
// 1.65-1.75 GHz of 2.4 (scripts/phase_times.py, profiles/r02_cfg3) - and on this chip the clock held under a bf16 MFMA
// stream depends on the MFMA shape: 16x16x32 delivers 1.12-1.14x the FLOP/s of 32x32x16 at equal cycles with operands
// read from LDS (MI355X_MICROARCH.md, DVFS give-back item 7; cdna_hip_programming.md rule 28).  The shape costs issue
// slots (twice the MFMAs, each holding the issue port 8 cycles):
//   head_dim 128, 32 rows per wave: 34 MFMAs x 8 + ~224 VALU cycles = ~500 issue cycles per 32-key step against 544
//     matrix cycles - it fits, and pays: +4.4 % over the 32x32x16 pipelined kernel;
//   head_dim 64, 64 rows per wave (NQH = 4): 36 x 8 + ~456 = ~750 against 576 - issue-bound, and the clock gain only
//     just covers it: +0.8 % at L = 4096, +1.9 % at L = 32768, -5 % at L = 1024: a tuner candidate.
//
// Operand maps (lane l: n = l & 15, g = l >> 4):
//   A[16 x 32]: A[row n][k = 8 g + j]      B[32 x 16]: B[k = 8 g + j][col n]      C/D[16 x 16]: [row 4 g + j][col n], j = 0..3
//   S^T tile (kb, qh) = K(16 keys of block kb) Q^T(16 queries of half qh), summed over four 32-wide d slices:
//     A = K[key 16 kb + pi(n)][32 ds + 8 g ..+7]  (one ds_read_b128),  B = Q[q_row0 + 16 qh + n][32 ds + 8 g ..+7]  (registers)
//     -> lane holds S^T[key 16 kb + pi(4 g + j)][query 16 qh + n]:  a query on the lane, as in the 32x32 kernels.
//     pi(r) = 8 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3) renumbers the keys of a block so that the two lane groups of a
//     half-wave own keys 8 apart (g = 0: 0-3, g = 1: 8-11, g = 2: 4-7, g = 3: 12-15): that makes the transposed V reads
//     below bank-conflict free; any fixed renumbering is legal because the same keys meet again in the PV contraction.
//   O^T tile (db, qh) += V^T(16 d of block db x 32 keys) P^T(32 keys x 16 queries of half qh):
//     B = the lane's 8 probabilities of the step: j' = 0..3 from S^T tile (0, qh), 4..7 from tile (1, qh), i.e. contraction
//     index k' = 8 g + j' <-> key 16 (j' >> 2) + pi(4 g + (j' & 3));  A = V[that key][16 db + n] = two ds_read_b64_tr_b16
//     (4 keys x 16 columns each, transposed by the LDS hardware) from a plain row-major V image.
//   row sums: one more MFMA per query half with an all-ones A: every lane receives the full 32-key sum of its query.
// LDS images (32-key slots of 2 D-byte rows, both filled by LDS-DMA with per-lane source addresses); head_dim 128 | 64:
//   K: 16-byte chunk c of row r at chunk slot c ^ (r & 15) | c ^ ((r >> 1) & 7)          (conflict-free ds_read_b128 rows)
//   V: chunk c of row r at slot c ^ (((r & 3) << 2) | ((r >> 2) & 3)) | c ^ (2 ((r >> 1) & 1) + 4 ((r >> 3) & 1))
//      (conflict-free transposed reads: the 8 rows a half-wave touches, 32 bytes each, cover the 64 banks once)

#include "fa_device.h"

#ifndef FA_P16_PF
#define FA_P16_PF 8          // hand-scheduled step: LDS operands are requested this many operands ahead
#endif
// timing-only ablations (wrong results): -DFA_P16_ABL=1 no per-step barrier, 2 no per-step DMA
#ifndef FA_P16_ABL
#define FA_P16_ABL 0
#endif
#ifndef FA_P16_NPRE
#define FA_P16_NPRE 8        // V^T fragments of the next step requested mid-step and held across the step barrier
#endif

namespace fa {

// D = head_dim (64 | 128), NQH = 16-row query blocks per wave (2 | 4: every K / V^T fragment read feeds NQH MFMAs)
template <typename T, int D, int NWAVES, int NQH, bool CAUSAL>
__device__ __forceinline__ void fa_fwd_pipe16_body(const FwdParams& p) {
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int RW = 16 * NQH;                // query rows per wave
    constexpr int BQ = NWAVES * RW, HK = 32, ROWB = D * 2;
    constexpr int SLOT_B = HK * ROWB;           // K slot = V slot (8 KiB at head_dim 128, 4 KiB at 64)
    constexpr int PIECES = SLOT_B / 1024;
    constexpr int PPW = PIECES / NWAVES;
    constexpr int DPS = 2 * PPW;                // LDS-DMA instructions a wave issues per step
    constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
    constexpr int RPP = 1024 / ROWB;            // rows per 1-KiB DMA piece
    constexpr int NDIAG = RW / HK;              // a wave's last NDIAG steps carry the causal mask
    static_assert(RW >= HK && PPW >= 1 && PIECES % NWAVES == 0, "32 | 64 rows per wave; pieces must split evenly over the waves");
    constexpr int NDS = D / 32;                 // 32-wide d slices of the QK^T contraction
    constexpr int NDB = D / 16;                 // 16-row d blocks of O^T
    constexpr int NS = 4, NSM = NS - 1;         // ring slots
    constexpr int NPRE = FA_P16_NPRE < NDB ? FA_P16_NPRE : NDB;   // V^T fragments of the next step held in registers across the step barrier
    // image swizzles: the 16-byte chunk c of row r sits at chunk slot c ^ kswz(r) (K) / c ^ vswz(r) (V)
    auto kswz = [](int r) { return D == 128 ? (r & 15) : ((r >> 1) & 7); };
    auto vswz = [](int r) { return D == 128 ? (((r & 3) << 2) | ((r >> 2) & 3)) : ((((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2)); };

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NS * SLOT_B];   // K0..K3 V0..V3
    unsigned char* const smem_v = smem + NS * SLOT_B;
    __shared__ int flags[NWAVES];

    const int b = blockIdx.x;
    const int nq = p.nq_tiles;
    int g, qi;
    {
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, gq = lane >> 4;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* __restrict__ qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;
    const T* __restrict__ vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;

    const int q_row0 = qi * BQ + wave * RW;
    const bool wave_active = q_row0 < L;
    const int dchunks = p.Dr >> 3;               // the operands' true head dim in 16-byte chunks (= D / 8 unless the call runs zero-extended)
    int qrow[NQH];                               // this lane's query row in each 16-row block
    vec8 qf[NQH][NDS];
#pragma unroll
    for (int qh = 0; qh < NQH; ++qh) {
        qrow[qh] = q_row0 + 16 * qh + n;
        const int qr = qrow[qh] < L ? qrow[qh] : L - 1;
        const T* qp = qg + (int64_t)qr * p.q_sl;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {      // (columns past the operands' true head dim, FwdParams::Dr: zeros)
            const int ch_ = 4 * ds + gq;
            const vec8 ld_ = *(const vec8*)(qp + 8 * (ch_ < dchunks ? ch_ : 0));
            vec8 z_;
#pragma unroll
            for (int j = 0; j < 8; ++j) z_[j] = (T)0.f;
            qf[qh][ds] = ch_ < dchunks ? ld_ : z_;
        }
    }

    // ---- LDS-DMA geometry: piece pc = wave + j NWAVES = rows 4 pc .. 4 pc + 3 of a slot; the lane's LDS destination is
    // linear (row 4 pc + (lane >> 4), chunk slot lane & 15), its SOURCE chunk is the one the image stores in that slot
    const unsigned k_slb = (unsigned)p.k_sl * 2, v_slb = (unsigned)p.v_sl * 2;
    unsigned k_vo[PPW], v_vo[PPW];
    int dst[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pc = wave + j * NWAVES;
        const int row = RPP * pc + lane / CPR, cp = lane % CPR;
        // (a chunk past the true head dim: an offset beyond the descriptor's extent - the DMA writes zeros, as for rows past L)
        k_vo[j] = (cp ^ kswz(row)) < dchunks ? (unsigned)row * k_slb + ((cp ^ kswz(row)) << 4) : 0x80000000u;
        v_vo[j] = (cp ^ vswz(row)) < dchunks ? (unsigned)row * v_slb + ((cp ^ vswz(row)) << 4) : 0x80000000u;
        dst[j] = pc * 1024;
    }
    const unsigned k_bytes = ((unsigned)(L - 1) * (unsigned)p.k_sl + (unsigned)p.Dr) * 2;
    const unsigned v_bytes = ((unsigned)(L - 1) * (unsigned)p.v_sl + (unsigned)p.Dr) * 2;
    const unsigned smem_lds = lds_offset(smem);

#define P16_DMA_K(J_, SLOT_)                                                                       \
    {                                                                                              \
        const unsigned ko_ = (unsigned)(J_) * HK * k_slb;                                          \
        const i32x4 kr_ = make_raw_rsrc((const char*)kg + ko_, k_bytes - ko_);                     \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j)                                            \
            lds_dma_b128(kr_, smem_lds + (SLOT_) * SLOT_B + dst[j], k_vo[j]);                      \
    }
#define P16_DMA_V(J_, SLOT_)                                                                       \
    {                                                                                              \
        const unsigned vo_ = (unsigned)(J_) * HK * v_slb;                                          \
        const i32x4 vr_ = make_raw_rsrc((const char*)vg + vo_, v_bytes - vo_);                     \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j)                                            \
            lds_dma_b128(vr_, smem_lds + (NS + (SLOT_)) * SLOT_B + dst[j], v_vo[j]);               \
    }

    // K fragment (kb, ds): row 16 kb + pi(n), chunk 4 ds + g
    const int pin = 8 * ((n >> 2) & 1) + 4 * (n >> 3) + (n & 3);
    int k_rd[2][NDS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) k_rd[kb][ds] = (16 * kb + pin) * ROWB + (((4 * ds + gq) ^ kswz(pin)) << 4);
    // V^T fragment db: lane 4 q + pp of group g supplies row pi(4 g + q) (+16 for the second read), columns 16 db + 4 pp ..+3
    const int vq = (lane & 15) >> 2, vpp = lane & 3;
    const int vrow = 8 * (gq & 1) + 4 * (gq >> 1) + vq;
    const int vsw = vswz(vrow);                  // (= vswz(vrow + 16): the second read of a fragment)
    int v_rd[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) v_rd[db] = vrow * ROWB + (((2 * db + (vpp >> 1)) ^ vsw) << 4) + 8 * (vpp & 1);
    const int keyg = 8 * (gq & 1) + 4 * (gq >> 1);         // key of accumulator register j of S^T tile (kb, .): 16 kb + keyg + j

    constexpr bool causal = CAUSAL;                        // false: no mask - every wave computes every step, keys past L are masked
    const int kv_end = (causal && (qi + 1) * BQ < L) ? (qi + 1) * BQ : L;
    const int nh_wg = (kv_end + HK - 1) / HK;              // steps the workgroup streams
    const int nh_w = wave_active ? (causal ? q_row0 / HK + NDIAG : nh_wg) : 0;   // steps this wave computes; the last NDIAG carry the mask
#define P16_KLIM(QH_) (CAUSAL ? qrow[QH_] : L - 1)        /* the last key a row may see */

    f32x4 o[NDB][NQH];
    f32x4 lsum[NQH];
    float m[NQH];
    const vec8 ones = __builtin_bit_cast(vec8, u32x4{E::one_bits * 0x10001u, E::one_bits * 0x10001u, E::one_bits * 0x10001u, E::one_bits * 0x10001u});
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define P16_RESET()                                                                                \
    _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) {                                             \
        m[qh] = -INFINITY;                                                                         \
        lsum[qh] = zero4;                                                                          \
        _Pragma("unroll") for (int db = 0; db < NDB; ++db) o[db][qh] = zero4;                      \
    }
    P16_RESET()
    const float c = p.scale_log2;

#define P16_KFRAG(KB_, DS_, SLOT_) (*(const vec8*)(smem + (SLOT_) * SLOT_B + k_rd[KB_][DS_]))
#define P16_VFRAG(OUT_, DB_, VB_)                                                                  \
    {                                                                                              \
        const unsigned char* vp = (VB_) + v_rd[DB_];                                               \
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp)); \
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp + 16 * ROWB)); \
        OUT_ = __builtin_bit_cast(vec8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));  \
    }
    // ---- compiler-scheduled pieces (first step, drain, idle steps, the exact fallback pass)
#define P16_QK(S_, SLOT_)                                                                          \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                             \
        _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) S_[kb][qh] = zero4;                       \
        _Pragma("unroll") for (int ds = 0; ds < NDS; ++ds) {                                       \
            const vec8 kf_ = P16_KFRAG(kb, ds, SLOT_);                                             \
            _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) S_[kb][qh] = E::mfma16(kf_, qf[qh][ds], S_[kb][qh]); \
        }                                                                                          \
    }
#define P16_PV(P_, SLOT_)                                                                          \
    _Pragma("unroll") for (int db = 0; db < NDB; ++db) {                                           \
        vec8 vf_;                                                                                  \
        P16_VFRAG(vf_, db, smem_v + (SLOT_) * SLOT_B)                                              \
        _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) o[db][qh] = E::mfma16(vf_, P_[qh], o[db][qh]); \
    }
#define P16_MASK(S_, I_)                                                                           \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                               \
        _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh)                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
                if ((I_) * HK + 16 * kb + keyg + j > P16_KLIM(qh)) S_[kb][qh][j] = -INFINITY;
    // exact row max of the step (a query's 32 scores sit in the 4 lanes n, n+16, n+32, n+48); raise m (rescaling O and the
    // row sums) wherever it is exceeded
#define P16_MAXR(S_)                                                                               \
    _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) {                                             \
        float mx = max2(max2(S_[0][qh][0], S_[0][qh][1]), max2(S_[0][qh][2], S_[0][qh][3]));       \
        mx = max3(mx, max2(S_[1][qh][0], S_[1][qh][1]), max2(S_[1][qh][2], S_[1][qh][3]));         \
        mx = max2(mx, __shfl_xor(mx, 16, 64));                                                     \
        mx = max2(mx, __shfl_xor(mx, 32, 64));                                                     \
        if (__builtin_amdgcn_ballot_w64((mx - m[qh]) * c > 0.0f) != 0) {                           \
            const float m_new = max2(m[qh], mx);                                                   \
            const float alpha = fast_exp2((m[qh] - m_new) * c);                                    \
            m[qh] = m_new;                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) lsum[qh][j] *= alpha;                    \
            _Pragma("unroll") for (int db = 0; db < NDB; ++db)                                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) o[db][qh][j] *= alpha;               \
        }                                                                                          \
    }
    // P = 2^(c S - c m), packed as the B fragment of the PV product (j' = 0..3 from key block 0, 4..7 from block 1), row sums
#define P16_SM_PACK(PC_, S_)                                                                       \
    _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) {                                             \
        const float mc = m[qh] * c;                                                                \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
                PC_[qh][4 * kb + j] = (T)fast_exp2(__builtin_fmaf(S_[kb][qh][j], c, -mc));         \
        lsum[qh] = E::mfma16(ones, PC_[qh], lsum[qh]);                                             \
    }
    // every step: issue the DMA for K(i+3), V(i+2); compute; publish (everything issued before this step has landed)
#define P16_ISSUE(I_, KSLOT_, VSLOT_)                                                              \
        const bool dk_ = (I_) + NS - 1 < nh_wg, dv_ = (I_) + NS - 2 < nh_wg;                       \
        if (dk_ && FA_P16_ABL != 2) P16_DMA_K((I_) + NS - 1, KSLOT_)                               \
        if (dv_ && FA_P16_ABL != 2) P16_DMA_V((I_) + NS - 2, VSLOT_)
#define P16_PUBLISH()                                                                              \
        if (dk_ && dv_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS * (NS - 3)) : "memory");      \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        if (FA_P16_ABL != 1) __syncthreads();
#define P16_PRE(VB_)                                                                               \
        fa_static_for<0, NPRE>([&](auto ti_) { constexpr int t2_ = decltype(ti_)::value; P16_VFRAG(pre[t2_], t2_, VB_) });
#define P16_GENERIC(I_, SC_, SN_, PP_, PC_, SAFE_)                                                 \
    {                                                                                              \
        const int i_ = (I_);                                                                       \
        P16_ISSUE(i_, (i_ + NS - 1) & NSM, (i_ + NS - 2) & NSM)                                    \
        if (i_ + 1 < nh_w) P16_QK(SN_, (i_ + 1) & NSM)                                             \
        if (i_ >= 1 && i_ <= nh_w) P16_PV(PP_, (i_ + NS - 1) & NSM)                                \
        if (i_ < nh_w) {                                                                           \
            if (i_ >= nh_w - NDIAG) P16_MASK(SC_, i_)                                              \
            if (i_ == 0 || (SAFE_)) P16_MAXR(SC_)                                                  \
            P16_SM_PACK(PC_, SC_)                                                                  \
            if (!(SAFE_)) P16_PRE(smem_v + (i_ & NSM) * SLOT_B)   /* V(i) fragments for a hand-scheduled step i + 1 */ \
        }                                                                                          \
        P16_PUBLISH()                                                                              \
    }
    // ---- hand-scheduled step.  32 (or 16 on a wave's last step) MFMAs in a fixed order: PV first (operands x = 0..7 =
    // V^T fragments of V(i-1), the first NPRE of them held in `pre` since the middle of the previous step), then QK^T
    // (operands x = 8..15 = K(i+1) fragments (kb, ds), requested at the top of the step); every operand feeds the two query
    // halves.  The softmax of S(i) is cut into 8 items (2 scores -> one dword of P) and spread one sub-operation per MFMA
    // gap: [2 fma] [exp] [exp] [cvt (+ row-sum MFMA when a half's fragment is complete)] - at most 8 VALU issue cycles
    // beside a 16-cycle MFMA.  Frozen with sched_barrier(0).
#define P16_ITEM_FMA(SC_, V_, MASK_)                                                               \
        {                                                                                          \
            constexpr int qh_ = (V_) >> 2, kb_ = ((V_) >> 1) & 1, e_ = 2 * ((V_) & 1);             \
            if constexpr (MASK_) {                                                                 \
                const int key_ = key0_ + 16 * kb_ + e_;                                            \
                if (key_ > P16_KLIM(qh_)) SC_[kb_][qh_][e_] = -INFINITY;                           \
                if (key_ + 1 > P16_KLIM(qh_)) SC_[kb_][qh_][e_ + 1] = -INFINITY;                   \
            }                                                                                      \
            SC_[kb_][qh_][e_] = __builtin_fmaf(SC_[kb_][qh_][e_], c, -mc[qh_]);                    \
            SC_[kb_][qh_][e_ + 1] = __builtin_fmaf(SC_[kb_][qh_][e_ + 1], c, -mc[qh_]);            \
        }
#define P16_ITEM_EXP(SC_, V_, EL_)                                                                 \
        {                                                                                          \
            constexpr int qh_ = (V_) >> 2, kb_ = ((V_) >> 1) & 1, e_ = 2 * ((V_) & 1) + (EL_);     \
            SC_[kb_][qh_][e_] = fast_exp2(SC_[kb_][qh_][e_]);                                      \
        }
#define P16_ITEM_CVT(SC_, PC_, V_)                                                                 \
        {                                                                                          \
            constexpr int qh_ = (V_) >> 2, kb_ = ((V_) >> 1) & 1, e_ = 2 * ((V_) & 1);             \
            PC_[qh_][4 * kb_ + e_] = (T)SC_[kb_][qh_][e_];                                         \
            PC_[qh_][4 * kb_ + e_ + 1] = (T)SC_[kb_][qh_][e_ + 1];                                 \
            if constexpr (((V_) & 3) == 3) lsum[qh_] = E::mfma16(ones, PC_[qh_], lsum[qh_]);       \
        }
#define P16_OPND_LOAD(X_)                                                                          \
        if constexpr ((X_) < NDB) {                                                                \
            P16_VFRAG(opnd[(X_)], (X_), vb_)                                                       \
        } else {                                                                                   \
            opnd[(X_)] = *(const vec8*)(kb_ + k_rd[((X_) - NDB) / NDS][((X_) - NDB) % NDS]);       \
        }
#define P16_FAST(I_, KN_, VP_, KI_, VI_, SC_, SN_, PP_, PC_, MASK_, QK_)                            \
    {                                                                                              \
        P16_ISSUE(I_, KI_, VI_)                                                                    \
        constexpr int NM = NDB + ((QK_) ? 2 * NDS : 0), NMF = NQH * NM, NV = 4 * NQH, GAP = NMF / NV;  \
        static_assert(GAP == 1 || GAP == 2 || GAP == 4, "softmax items per MFMA gap");             \
        constexpr int PF_ = FA_P16_PF < NM ? FA_P16_PF : NM;                                       \
        const unsigned char* kb_ = smem + (KN_) * SLOT_B;                                          \
        const unsigned char* vb_ = smem_v + (VP_) * SLOT_B;                                        \
        const unsigned char* vnext_ = smem_v + (((VP_) + 1) & NSM) * SLOT_B;                       \
        const int key0_ = (I_) * HK + keyg;                                                        \
        (void)kb_; (void)key0_; (void)vnext_;                                                      \
        vec8 opnd[NM];                                                                             \
        fa_static_for<0, NPRE>([&](auto xi_) { constexpr int x_ = decltype(xi_)::value; opnd[x_] = pre[x_]; }); \
        fa_static_for<NPRE, (NPRE > PF_ ? NPRE : PF_)>([&](auto xi_) { constexpr int x_ = decltype(xi_)::value; P16_OPND_LOAD(x_) }); \
        float mc[NQH];                                                                             \
        _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) mc[qh] = m[qh] * c;                     \
        P16_ITEM_FMA(SC_, 0, MASK_)                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        fa_static_for<0, NMF>([&](auto gi_) {                                                      \
            constexpr int n_ = decltype(gi_)::value, g_ = n_ / NQH, qh_ = n_ % NQH;                \
            if constexpr (g_ < NDB) {                                                              \
                o[g_][qh_] = E::mfma16(opnd[g_], PP_[qh_], o[g_][qh_]);                            \
            } else {                                                                               \
                constexpr int kbx_ = (g_ - NDB) / NDS, ds_ = (g_ - NDB) % NDS;                     \
                if constexpr (ds_ == 0) SN_[kbx_][qh_] = E::mfma16(opnd[g_], qf[qh_][ds_], zero4); \
                else SN_[kbx_][qh_] = E::mfma16(opnd[g_], qf[qh_][ds_], SN_[kbx_][qh_]);           \
            }                                                                                      \
            if constexpr (qh_ == NQH - 1 && g_ + PF_ < NM && g_ + PF_ >= NPRE) { P16_OPND_LOAD(g_ + PF_) } \
            /* the NEXT step's first V^T fragments, requested once this step's held ones have been consumed */ \
            if constexpr ((QK_) && qh_ == NQH - 1 && g_ >= NPRE && g_ < 2 * NPRE) { P16_VFRAG(pre[g_ - NPRE], g_ - NPRE, vnext_) } \
            /* this gap's share of the softmax of S(i) */                                          \
            constexpr int w_ = n_ / GAP, ph_ = n_ % GAP;                                           \
            if constexpr (GAP == 4) {                                                              \
                if constexpr (ph_ == 0 && w_ + 1 < NV) { P16_ITEM_FMA(SC_, w_ + 1, MASK_) }        \
                if constexpr (ph_ == 1) { P16_ITEM_EXP(SC_, w_, 0) }                               \
                if constexpr (ph_ == 2) { P16_ITEM_EXP(SC_, w_, 1) }                               \
                if constexpr (ph_ == 3) { P16_ITEM_CVT(SC_, PC_, w_) }                             \
            } else if constexpr (GAP == 2) {                                                       \
                if constexpr (ph_ == 0) { if constexpr (w_ + 1 < NV) { P16_ITEM_FMA(SC_, w_ + 1, MASK_) } P16_ITEM_EXP(SC_, w_, 0) } \
                if constexpr (ph_ == 1) { P16_ITEM_EXP(SC_, w_, 1) P16_ITEM_CVT(SC_, PC_, w_) }    \
            } else {                                                                               \
                if constexpr (w_ + 1 < NV) { P16_ITEM_FMA(SC_, w_ + 1, MASK_) }                    \
                P16_ITEM_EXP(SC_, w_, 0) P16_ITEM_EXP(SC_, w_, 1) P16_ITEM_CVT(SC_, PC_, w_)       \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        });                                                                                        \
        /* pin this step's vector work to this step */                                             \
        _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) asm volatile("" : "+v"(PC_[qh]), "+v"(lsum[qh])); \
        P16_PUBLISH()                                                                              \
    }
#define P16_STEADY(I_, R_, SC_, SN_, PP_, PC_)                                                     \
    P16_FAST(I_, ((R_) + 1) & NSM, ((R_) + NS - 1) & NSM, ((R_) + NS - 1) & NSM, ((R_) + NS - 2) & NSM, SC_, SN_, PP_, PC_, false, true)
#define P16_ROTATE()                                                                               \
    _Pragma("unroll") for (int qh = 0; qh < NQH; ++qh) {                                             \
        sa[0][qh] = sb[0][qh];                                                                     \
        sa[1][qh] = sb[1][qh];                                                                     \
        pb[qh] = pa[qh];                                                                           \
    }
#define P16_RAGGED(I_, MASK_, QK_)                                                                 \
    P16_FAST(I_, ((I_) + 1) & NSM, ((I_) + NS - 1) & NSM, ((I_) + NS - 1) & NSM, ((I_) + NS - 2) & NSM, sa, sb, pb, pa, MASK_, QK_)
    // one pass over the kv steps of this workgroup.  S(i) lives in sa for even i and in sb for odd i; P(i) in pa / pb
    // likewise.  Groups of four steady steps start at i = 1.
#define P16_RUN(SAFE_)                                                                             \
    {                                                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < NS - 1; ++j_) {       /* K(0..NS-2), V(0..NS-3) */    \
            if (j_ < nh_wg) P16_DMA_K(j_, j_)                                                      \
            if (j_ < NS - 2 && j_ < nh_wg) P16_DMA_V(j_, j_)                                       \
        }                                                                                          \
        lds_dma_wait_all();                                                                        \
        __syncthreads();                                                                           \
        if (nh_w >= 1) P16_QK(sa, 0)                                                               \
        P16_GENERIC(0, sa, sb, pb, pa, SAFE_)                                                      \
        int i = 1;                                                                                 \
        if (!(SAFE_)) {                                                                            \
            for (; i + NS - 1 + NDIAG < nh_w; i += NS) {   /* steps i .. i+NS-1 carry no mask */     \
                fa_static_for<0, NS>([&](auto ki_) {                                               \
                    constexpr int k_ = decltype(ki_)::value, R_ = (1 + k_) % NS;                   \
                    if constexpr (k_ % 2 == 0) P16_STEADY(i + k_, R_, sb, sa, pa, pb)              \
                    else P16_STEADY(i + k_, R_, sa, sb, pb, pa)                                    \
                });                                                                                \
            }                                                                                      \
        }                                                                                          \
        P16_ROTATE()                                           /* i is odd: S(i) sb -> sa, P(i-1) pa -> pb */ \
        if (!(SAFE_)) {                                                                            \
            for (; i + NDIAG < nh_w; ++i) {                    /* steps the groups left over */    \
                P16_RAGGED(i, false, true)                                                         \
                P16_ROTATE()                                                                       \
            }                                                                                      \
            if (NDIAG == 2 && i + 1 < nh_w) {                  /* the first diagonal step of a 64-row wave */ \
                P16_RAGGED(i, true, true)                                                          \
                P16_ROTATE()                                                                       \
                ++i;                                                                               \
            }                                                                                      \
            if (i < nh_w) {                                    /* the wave's last (diagonal) step */ \
                P16_RAGGED(i, true, false)                                                         \
                P16_ROTATE()                                                                       \
                ++i;                                                                               \
            }                                                                                      \
        }                                                                                          \
        for (; i <= nh_wg; ++i) {                              /* drain, idle (and the exact pass) */ \
            P16_GENERIC(i, sa, sb, pb, pa, SAFE_)                                                  \
            P16_ROTATE()                                                                           \
        }                                                                                          \
    }

    f32x4 sa[2][NQH], sb[2][NQH];
    vec8 pa[NQH], pb[NQH];
    vec8 pre[NPRE];
#pragma unroll
    for (int qh = 0; qh < NQH; ++qh) {
        sa[0][qh] = zero4; sa[1][qh] = zero4; sb[0][qh] = zero4; sb[1][qh] = zero4;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pa[qh][j] = (T)0.f; pb[qh][j] = (T)0.f; }
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) asm volatile("" : "+v"(qf[qh][ds]));   // Q loads issued
    }
#pragma unroll
    for (int t = 0; t < NPRE; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[t][j] = (T)0.f;

    P16_RUN(false)
    // optimistic pass: an overflowed P made a row sum inf / NaN for good - one check per pass, all waves must agree
    bool bad = false;
#pragma unroll
    for (int qh = 0; qh < NQH; ++qh) bad |= wave_active && sum_left_safe_range(lsum[qh][0]);
    flags[wave] = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
    __syncthreads();
    int any_bad = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) any_bad |= flags[w];
    if (any_bad) {
        __syncthreads();
        P16_RESET()
        P16_RUN(true)
    }
#undef P16_KLIM
#undef P16_DMA_K
#undef P16_DMA_V
#undef P16_RESET
#undef P16_KFRAG
#undef P16_VFRAG
#undef P16_QK
#undef P16_PV
#undef P16_MASK
#undef P16_MAXR
#undef P16_SM_PACK
#undef P16_ISSUE
#undef P16_PUBLISH
#undef P16_PRE
#undef P16_GENERIC
#undef P16_ITEM_FMA
#undef P16_ITEM_EXP
#undef P16_ITEM_CVT
#undef P16_OPND_LOAD
#undef P16_FAST
#undef P16_STEADY
#undef P16_ROTATE
#undef P16_RAGGED
#undef P16_RUN

    // ---- epilogue: O /= l through LDS (the K ring is free now), whole-row stores.  A lane holds, per d block, 4 consecutive
    // d of ONE row: 8 bytes at byte 32 db + 8 g of that row; the 16-byte chunk index is XORed with the row so that the 16
    // rows of a half land in different banks.
    if (!wave_active) return;
    unsigned char* slab = smem + wave * (RW * ROWB);
    T* __restrict__ og = (T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh;
#pragma unroll
    for (int qh = 0; qh < NQH; ++qh) {
        const float lt = lsum[qh][0];
        const float inv = __builtin_amdgcn_rcpf(lt);
        const int rr = 16 * qh + n;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            vec4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (T)(o[db][qh][j] * inv);
            const int ch = 2 * db + (gq >> 1);
            *(vec4*)(slab + rr * ROWB + ((ch ^ (rr & (CPR - 1))) << 4) + 8 * (gq & 1)) = w;
        }
        if (p.lse != nullptr && gq == 0 && qrow[qh] < L) p.lse[(int64_t)g * L + qrow[qh]] = m[qh] * p.scale + __logf(lt);
    }
    constexpr int RPI = 64 / CPR;           // rows per store instruction (4 x 16 chunks at head_dim 128, 8 x 8 at 64)
#pragma unroll
    for (int i = 0; i < RW / RPI; ++i) {
        const int rr = RPI * i + lane / CPR, ch = lane % CPR;
        const u32x4 val = *(const u32x4*)(slab + rr * ROWB + ((ch ^ (rr & (CPR - 1))) << 4));
        const int orow = q_row0 + rr;
        if (orow < L && ch < dchunks) *(u32x4*)(og + (int64_t)orow * p.o_sl + ch * 8) = val;
    }
}

template <typename T, int D, int NWAVES, int NQH, bool CAUSAL>
__global__ __launch_bounds__(NWAVES * 64, 2) void fa_fwd_pipe16_kernel(const FwdParams p) {
    fa_fwd_pipe16_body<T, D, NWAVES, NQH, CAUSAL>(p);
}

// callers guarantee: D = 128, skip-max arithmetic wanted (bf16, or f16 with p.skipmax), waves in {4, 8},
// p.nq_tiles = ceil(L / (32 waves))
template <typename T, bool CAUSAL>
static hipError_t launch_pipe16_c(const FwdParams& p, int waves, int qb, hipStream_t stream) {
    const dim3 grid((unsigned)((int64_t)p.G8 * p.nq_tiles));
    // 8 waves = 256-row workgroups: half the K/V LDS-DMA pieces per wave and step, half the L2 -> LDS traffic per FLOP
    if (p.D == 64) {   // 4 waves; qb = 32-row query blocks per wave (the host sized p.nq_tiles for it)
        if (qb == 2) return launch1((const void*)(fa_fwd_pipe16_kernel<T, 64, 4, 4, CAUSAL>), grid, dim3(256), 0, stream, p);
        return launch1((const void*)(fa_fwd_pipe16_kernel<T, 64, 4, 2, CAUSAL>), grid, dim3(256), 0, stream, p);
    }
    if (waves == 8) return launch1((const void*)(fa_fwd_pipe16_kernel<T, 128, 8, 2, CAUSAL>), grid, dim3(512), 0, stream, p);
    return launch1((const void*)(fa_fwd_pipe16_kernel<T, 128, 4, 2, CAUSAL>), grid, dim3(256), 0, stream, p);
}

hipError_t launch_fwd_pipe16(const FwdParams& p, int dtype, int waves, int qb, hipStream_t stream) {
    if (dtype == 1) return p.causal ? launch_pipe16_c<__bf16, true>(p, waves, qb, stream) : launch_pipe16_c<__bf16, false>(p, waves, qb, stream);
    return p.causal ? launch_pipe16_c<_Float16, true>(p, waves, qb, stream) : launch_pipe16_c<_Float16, false>(p, waves, qb, stream);
}

}  // namespace fa
