// FlashAttention backward for MI355X (gfx950): dQ, dK, dV of the causal forward in fa_fwd_dma.hip.
//
// Replaces the Poplar program built by flashAttentionQKVPackedGrad
// (flash_attention_ipu/cpp/flash_attention_qkv_packed.cpp:348-512): recompute the scores from Q, K and
// the forward's logSumExp, then   P = exp(S - lse),  dV += P^T dO,  dP = dO V^T,
// dS = P o (dP - rowsum(dO o O)),  dQ += dS K,  dK += dS^T Q   (FA.cpp:452-472).
// The reference runs one kv-outer / q-inner double loop that updates dQ in place; here the work is
// split so that no gradient is ever summed across workgroups (deterministic, no atomics):
//
//   fa_bwd_dq_kernel     one workgroup per (group, 128 query rows): sweeps kv tiles <= diagonal,
//                        dQ^T += K^T dS^T with the query row on the lane (same shape as the forward);
//                        its prologue also forms Dsum[g,r] = sum_d dO[g,r,d] * O[g,r,d] (FA.cpp:375-378)
//                        for its own rows and publishes it to the workspace for the next kernel
//   fa_bwd_dkdv_kernel   one workgroup per (group, 128 keys): sweeps 64-row query tiles >= diagonal,
//                        dV^T += dO^T P and dK^T += Q^T dS with the KEY on the lane, so the score and
//                        dP accumulators are already the B operands of those two products.
//
// Every streamed tile (K, V in the dq kernel; Q, dO in the dkdv kernel) is consumed twice: by rows
// (A operand of the score products, ds_read_b128) and transposed (A operand of the gradient products,
// ds_read_b64_tr_b16).  One LDS image serves both: [d/32][row/8][8 rows][4 x 16-B chunks], chunk slot
// XORed with (row>>2)&3 - conflict-free for both read kinds - filled by LDS-DMA with per-lane source
// addresses, exactly like the forward's V image.
// MFMA operand maps, the accumulator-as-operand trick and the DMA descriptor handling are the ones
// documented in fa_fwd_kernels.hip / fa_fwd_dma.hip.

#include "fa_device.h"
#include "fa_extras.h"

#ifndef FA_BWD_NBUF
#define FA_BWD_NBUF(D_) 2   /* streamed-tile buffers per operand; 3 at head_dim 64 (-DFA_BWD_NBUF\(D_\)=...) measured +0.2..0.9 %: not worth 16 KB of LDS */
#endif
#ifndef FA_BWD_ABL
#define FA_BWD_ABL 0    /* timing-only ablations (wrong results): 1 no K/V DMA in dq, 2 no Q/dO DMA in dkdv */
#endif

namespace fa {

// ---------------------------------------------------------------------------------------------
// shared pieces of the two gradient kernels
// ---------------------------------------------------------------------------------------------
constexpr float LOG2E = 1.4426950408889634f;

// dual-use tile image (see header): byte offset of 16-B chunk `chunk` of row `row` (row < 64)
__device__ __forceinline__ int img_addr(int row, int chunk) {
    return (chunk >> 2) * 4096 + (row >> 3) * 512 + (row & 7) * 64 + (((chunk & 3) ^ ((row >> 2) & 3)) << 4);
}

template <typename T, int D, int NWAVES>
struct BwdGeom {
    static constexpr int IMG_B = (D / 32) * 4096;    // bytes per 64-row tile image
    static constexpr int PIECES = IMG_B / 1024;      // 1 KiB DMA pieces per image
    static constexpr int PPW = PIECES / NWAVES;      // pieces per wave
    static constexpr int KS = D / 16;
    static constexpr int DT = D / 32;
    // streamed-tile buffers per operand (a ring: with three, a tile's LDS-DMA has two tile times to land)
    static constexpr int NB = FA_BWD_NBUF(D);
    static_assert(PIECES % NWAVES == 0 && PPW >= 1, "pieces must split evenly over the waves");
};

// which (group, block-of-128) does this workgroup own: same XCD-aware bunched order as the forward;
// `heavy_first_desc` picks whether block index 0 or nblk-1 is the heaviest.
__device__ __forceinline__ bool bwd_block_map(int G8, int G, int sched, int nblk, bool heavy_is_last, int xs, int nblk_real, int& g, int& blk) {
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

// ---------------------------------------------------------------------------------------------
// dQ: one workgroup = NWAVES x 32 query rows; streams K and V tiles of 64 keys
// ---------------------------------------------------------------------------------------------
// EXTRA: the forward ran with an attention mask / bias and / or dropout (fa_extras.h): the same additive terms go onto the recomputed
// scores, the dropout decisions are regenerated from (seed, offset, g, row, col):  dP = keep ? dP_raw / (1 - p) : 0  before
// dS = P o (dP - Dsum)  (Dsum = rowsum(dO o O) still equals sum_j P_j dP_j);  dV sees  keep ? P / (1 - p) : 0.
// EXTRA: 0 plain, 1 mask / dropout with the mask loaded in place, 2 the same with the mask's raw data requested one tile ahead (masks
// that differ from row to row: [.., L, L]; a key-padding or key-bias row is cache-resident and gains nothing from 8 - 32 more registers)
// QS > 1 (kv split groups, for grids that cannot fill the chip - the mirror of the forward's SPLIT, fa_fwd_dma.hip): QS groups of NWAVES waves own
// the SAME query rows; group s streams the kv tiles s, s + QS, ... through buffers of its own, in lock step with the others (one workgroup
// barrier per round), and the groups' dQ partial sums meet in LDS after the loop, added in group order (deterministic).  A launch of at most
// one workgroup per CU lasts as long as its heaviest row block's kv chain; the chain shrinks QS-fold.
template <typename T, int D, int NWAVES, int EXTRA = 0, int QS = 1, int BAL = 0>
__device__ __forceinline__ void fa_bwd_dq_body(const BwdParams& p) {
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    using GM = BwdGeom<T, D, NWAVES>;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int BQ = NWAVES * 32, BK = 64, KS = GM::KS, DT = GM::DT, IMG_B = GM::IMG_B, PPW = GM::PPW, NB = GM::NB;
    static_assert(QS == 1 || EXTRA == 0, "kv split groups: the plain kernels only");

    // K0 .. K(NB-1)  V0 .. V(NB-1); EXTRA == 2 | 3: plus per buffer the tile's mask image (128 rows x 64 keys: 8 KB of boolean bytes, 16 KB of bias)
    constexpr int MIMG_B = EXTRA == 3 ? 16384 : 8192, MOFF = 2 * NB * IMG_B;
    // (EXTRA == 1: per buffer the 64 per-key terms of a row-broadcast mask, fp32, and the tile's "every key masked" flag)
    constexpr int MT_F = 80;
    constexpr int GROUP_B = MOFF + (EXTRA >= 2 ? NB * MIMG_B : EXTRA == 1 ? NB * MT_F * 4 : 0);
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[QS * GROUP_B];

    int g, qi;
    // BAL (balanced units, for causal grids of at most one workgroup per CU - where the launch lasts as long as its heaviest row block, twice the
    // mean): the grid runs over UNITS instead of row blocks.  The heavy half of a group's row blocks (the upper half: their kv sweeps are the long
    // ones) are two units each - the first and the second half of the sweep, `half` = 1 | 2 - whose fp32 partial sums meet in a reduce pass
    // (fa_bwd_dq_reduce_kernel); a light row block stays one unit (`half` = 0) and stores dQ itself.  1.5x the workgroups, the longest half as long.
    // The partial sums of the heavy rows need 2 x (L/2 rows x D floats) per group = exactly the bytes of the group's dK and dV, which nobody reads or
    // writes before the dK/dV kernel (next on the stream) fills them: half 1 parks its sums in dK, half 2 in dV (partial row i = rows 2i, 2i + 1 there).
    int half = 0;
    if constexpr (BAL) {
        int qd;
        if (!bwd_block_map(p.G8, p.G, p.sched, p.nq_tiles, false, p.xs_q, p.nq_real, g, qd)) return;
        // units in descending length for the dispatcher (the row-block count is a multiple of 4): level l = nh .. nh/2 + 1 holds the four halves of
        // row blocks 2l - 1 and 2l - 2 (l and l - 1/2 rounds) and the whole light block l - 1 (l rounds); below that the remaining light blocks
        const int nq_true = (p.L + BQ - 1) / BQ, nh = nq_true >> 1, nlev = nh >> 1;
        if (qd < 5 * nlev) {
            const int lv = nh - qd / 5, sl = qd % 5;
            if (sl == 4) qi = lv - 1;
            else { qi = 2 * lv - 1 - (sl >> 1); half = 1 + (sl & 1); }
        } else qi = nlev - 1 - (qd - 5 * nlev);
    } else {
        if (!bwd_block_map(p.G8, p.G, p.sched, p.nq_tiles, true, p.xs_q, p.nq_real, g, qi)) return;
    }

    const int L = p.L;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int ws = QS == 1 ? 0 : wave_all / NWAVES;                    // kv split group of this wave
    const int wave = QS == 1 ? wave_all : wave_all - ws * NWAVES;      // its row block inside the group
    const int tid = (int)threadIdx.x - ws * (NWAVES * 64), lane = tid & 63;
    unsigned char* const smem = smem_all + ws * GROUP_B;
    unsigned char* const smem_v = smem + NB * IMG_B;
    float* const mterm = (float*)(smem + MOFF);
    (void)mterm;
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* __restrict__ kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;   // (grouped-query K/V head)
    const T* __restrict__ vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;

    const int q_row0 = qi * BQ + wave * 32;
    const int qrow = q_row0 + r;
    const bool wave_active = q_row0 < L;
    const int qr = qrow < L ? qrow : L - 1;

    const int dchunks = p.Dr >> 3;
    // B operands: Q^T and dO^T fragments of this lane's query row; per-row constants
    vec8 qf[KS], dof[KS];
    {
        // (columns past the operands' true head dim - BwdParams::Dr, in 16-byte chunks - read as zeros: predicated here, out-of-range
        //  LDS-DMA chunks below; the gradients' columns past it are not stored)
        const T* qp = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh + (int64_t)qr * p.q_sl;
        const T* gp = (const T*)p.dout + (int64_t)bb * p.do_sb + (int64_t)hh * p.do_sh + (int64_t)qr * p.do_sl;
        vec8 z_;
#pragma unroll
        for (int j = 0; j < 8; ++j) z_[j] = (T)0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch_ = 2 * ks + h, cc_ = ch_ < dchunks ? ch_ : 0;
            const vec8 ql_ = *(const vec8*)(qp + 8 * cc_), gl_ = *(const vec8*)(gp + 8 * cc_);
            qf[ks] = ch_ < dchunks ? ql_ : z_;
            dof[ks] = ch_ < dchunks ? gl_ : z_;
        }
    }
    // (a row whose every key was masked has lse = -inf and P = 0: its exponent offset is taken as 0, never inf - inf)
    const float lse_row = p.lse[(int64_t)g * L + qr];
    const float lse2 = (EXTRA && lse_row == -INFINITY) ? 0.f : lse_row * LOG2E;
    const int mask_esz = p.ex.mask_kind == 2 ? (int)sizeof(T) : 1;
    const char* const mrow = EXTRA && p.ex.mask ? (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)hh * p.ex.m_sh + (int64_t)qr * p.ex.m_sl) * mask_esz : nullptr;
    const unsigned drop_kg = EXTRA ? drop_group_key(drop_key0_of(p.ex), (unsigned)g) : 0u;
    // Dsum = rowsum(dO o O) of this lane's row (FA.cpp:375-378), from the dO fragments already in registers and the
    // matching O fragments; the two half-waves hold the two halves of the row.  Published for fa_bwd_dkdv_kernel,
    // which runs after this kernel on the same stream (this replaces a separate pass over dO and O).
    float dsum;
    {
        const T* op = (const T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)qr * p.o_sl;
        float acc = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const vec8 of = *(const vec8*)(op + 8 * ((2 * ks + h) < dchunks ? 2 * ks + h : 0));   // (times a zero dO fragment past the head dim)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)of[j] * (float)dof[ks][j];
        }
        dsum = half_swap_sum(acc);
        if (h == 0 && qrow < L && ws == 0 && half != 2) p.dsum[(int64_t)g * L + qrow] = dsum;
    }
    const float ndsum = -dsum;
    const float c = p.scale_log2;

    // LDS-DMA geometry (pieces pc = wave + j*NWAVES of each image)
    const unsigned k_slb = (unsigned)p.k_sl * 2, v_slb = (unsigned)p.v_sl * 2;
    unsigned k_vo[PPW], v_vo[PPW];
    int dst[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pc = wave + j * NWAVES;
        const int row = (2 * (pc & 3) + (lane >> 5)) * 8 + ((lane >> 2) & 7);
        const int chunk = (pc >> 2) * 4 + ((lane & 3) ^ ((row >> 2) & 3));
        k_vo[j] = chunk < dchunks ? (unsigned)row * k_slb + (chunk << 4) : 0x80000000u;   // (past the true head dim: zeros)
        v_vo[j] = chunk < dchunks ? (unsigned)row * v_slb + (chunk << 4) : 0x80000000u;
        dst[j] = (pc >> 2) * 4096 + (pc & 3) * 1024;
    }
    const unsigned k_bytes = ((unsigned)(L - 1) * (unsigned)p.k_sl + (unsigned)p.Dr) * 2;
    const unsigned v_bytes = ((unsigned)(L - 1) * (unsigned)p.v_sl + (unsigned)p.Dr) * 2;
#define FA_DMA_KV(T_, BUF_)                                                                        \
    {                                                                                              \
        const unsigned ko_ = (unsigned)(T_) * BK * k_slb, vo_ = (unsigned)(T_) * BK * v_slb;       \
        const i32x4 kr_ = make_raw_rsrc((const char*)kg + ko_, k_bytes - ko_);                     \
        const i32x4 vr_ = make_raw_rsrc((const char*)vg + vo_, v_bytes - vo_);                     \
        if (FA_BWD_ABL != 1) {                                                                     \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) {                                          \
            lds_dma_b128(kr_, smem_lds + (BUF_) * IMG_B + dst[j], k_vo[j]);                        \
            lds_dma_b128(vr_, smem_lds + (NB + (BUF_)) * IMG_B + dst[j], v_vo[j]);                 \
        }                                                                                          \
        }                                                                                          \
    }
#define FA_TILE_BARRIER() { lds_dma_wait_all(); __syncthreads(); }
    const unsigned smem_lds = lds_offset(smem);

    // LDS read addresses: row reads (A operand rows = 32*blk + r) and transposed reads
    int a_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_rd[ks] = img_addr(r, 2 * ks + h);
    const int rowin8 = 4 * h + ((lane & 15) >> 2);
    const int c4 = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const int trb0 = rowin8 * 64 + ((c4 ^ h) << 4) + (lane & 1) * 8;
    const int trb1 = rowin8 * 64 + ((c4 ^ (2 + h)) << 4) + (lane & 1) * 8;

    const bool causal = p.causal != 0;             // 0: no mask, every key (keys past L are masked)
    const int kv_end = (causal && (qi + 1) * BQ < L) ? (qi + 1) * BQ : L;
    const int n_tiles_all = (kv_end + BK - 1) / BK;
    const int t_lo = half == 2 ? n_tiles_all >> 1 : 0;
    const int n_tiles = half == 0 ? n_tiles_all : half == 1 ? n_tiles_all >> 1 : n_tiles_all - (n_tiles_all >> 1);
    const int wave_last = causal ? q_row0 : L - 1, wave_kv_last = causal ? q_row0 + 31 : L - 1, klim = causal ? qrow : L - 1;

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
    // EXTRA == 2 | 3 (a boolean mask | a bias that differs from row to row; the kind is a template argument: decided per element at run
    // time it costs a branch and a copy of the score tile per four keys): every lane walks its own mask row, so a 4- or 8-byte load per lane touches 32
    // cache lines per instruction, eight instructions per tile and wave.  The tile's mask image (the workgroup's 128 rows x 64 keys)
    // travels to LDS with the K and V images instead: LDS-DMA, 16 bytes per lane, whole rows (2 / 4 instructions per wave and tile, one
    // tile ahead); a lane then reads its 4 keys of a (half, j) with one ds_read_b32 / _b64.  16-byte chunk c of row i sits at chunk
    // position c ^ (i >> 2 & 3) (bias: c ^ (i >> 1 & 7)): the 32 rows of a wave read the same chunk, which a row-major image keeps on
    // 4 (2) bank groups.  Needs ExtraParams::mask_vec bit 1; any other mask is loaded in place.
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr bool mbias = EXTRA == 3;
    const bool mdma = EXTRA >= 2 && p.ex.mask != nullptr && (p.ex.mask_vec & 2) != 0;
    unsigned m_vo[EXTRA >= 2 ? 4 : 1];
    int m_rd = 0;
    (void)mbias; (void)mdma; (void)m_vo; (void)m_rd;
    if constexpr (EXTRA >= 2) {
        const unsigned m_slb = (unsigned)p.ex.m_sl * (unsigned)mask_esz;
        const int rl_ = L - 1 - qi * BQ;                   // (rows past L: the last row again - they are never stored)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pc = wave + 4 * j;
            const int row = mbias ? 8 * pc + (lane >> 3) : 16 * pc + (lane >> 2);
            const int cpos = mbias ? (lane & 7) : (lane & 3);
            m_vo[j] = (unsigned)(row < rl_ ? row : rl_) * m_slb + ((unsigned)(cpos ^ (mbias ? (row >> 1) & 7 : (row >> 2) & 3)) << 4);
        }
        const int il = wave * 32 + r;
        m_rd = mbias ? il * 128 + (((il >> 1) & 7) << 4) + 8 * h : il * 64 + (((il >> 2) & 3) << 4) + 4 * h;
    }
    // EXTRA == 1 with a mask - a row-broadcast one (row stride 0: fa_bwd_extra.hip sends every other mask to EXTRA == 2 | 3): the term of
    // a score depends on its key alone.  Threads 0..63 fetch and decode the 64 terms of a tile one tile ahead (requested BEFORE that
    // tile's DMA - vmcnt retires in order - and written to LDS at the end of the tile during which they arrived); the tile body adds
    // them with four broadcast ds_read_b128 per half: no decoding and no mask-kind branch per element.  A tile whose 64 keys are
    // all masked (a padded tail) is skipped by the whole workgroup: its P is 0.
    float mt_stage = 0.f;
    const bool mbc = EXTRA == 1 && p.ex.mask != nullptr;
    (void)mt_stage; (void)mbc;
#define FA_MTERM_LOAD(T_)                                                                          \
    if constexpr (EXTRA == 1) {                                                                    \
        if (mbc && tid < 64) {                                                                     \
            const int k_ = (T_) * BK + tid;                                                        \
            mt_stage = mask_load1<T>(p.ex.mask_kind, mrow, 0, k_ < L ? k_ : L - 1, p.ex.inv_scale); \
        }                                                                                          \
    }
#define FA_MTERM_STORE(BUF_)                                                                       \
    if constexpr (EXTRA == 1) {                                                                    \
        if (mbc && tid < 64) {                                                                     \
            mterm[(BUF_) * MT_F + tid] = mt_stage;                                                 \
            const bool dead_ = __builtin_amdgcn_ballot_w64(mt_stage != -INFINITY) == 0;            \
            if (tid == 0) mterm[(BUF_) * MT_F + 64] = dead_ ? 1.f : 0.f;                           \
        }                                                                                          \
    }
    // (the tiles are requested in order: the image's address in the mask and the bytes left to the end of the slice's last row advance
    //  by one tile per request - two scalar pairs carried through the loop instead of the strides and indices they come from)
    unsigned long long m_at = 0, m_left = 0;
    if constexpr (EXTRA >= 2) {
        m_at = (unsigned long long)((const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)hh * p.ex.m_sh + (int64_t)(qi * BQ) * p.ex.m_sl) * mask_esz);
        m_left = (unsigned long long)((((int64_t)(L - 1 - qi * BQ) * p.ex.m_sl + L) * mask_esz + 3) & ~3ll);   // (whole dwords: fa_fwd_dma.hip FA_MASK_EXTENT)
        m_at = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m_at >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m_at);
        m_left = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m_left >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m_left);
    }
    (void)m_at; (void)m_left;
#define FA_DMA_MASK(T_, BUF_)                                                                      \
    if constexpr (EXTRA >= 2) {                                                                    \
        if (mdma) {   /* (tile T_ is the next one in order: asserted by the loop structure below) */ \
            const i32x4 mr_ = make_raw_rsrc((const char*)m_at, m_left < 0xffffffffull ? (unsigned)m_left : 0xffffffffu); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
                if (mbias || j < 2) lds_dma_b128(mr_, smem_lds + MOFF + (BUF_) * MIMG_B + (wave + 4 * j) * 1024, m_vo[j]); \
            m_at += (unsigned)(BK * mask_esz);                                                     \
            m_left -= (unsigned)(BK * mask_esz);                                                   \
        }                                                                                          \
    }

#ifndef FA_BWD_HALVES
#define FA_BWD_HALVES 1   /* 1: a 64-row (dK/dV) / 64-key (dQ) tile is consumed as two 32-wide halves one after the other: one score and
                             one dP tile live at a time - the head_dim 128 dK/dV kernel then fits two waves per SIMD, the others gain a
                             wave, and a half that lies wholly behind the causal diagonal is skipped; 0: both halves at once */
#endif
    static_assert(!EXTRA || FA_BWD_HALVES, "mask / dropout: the two-halves tile bodies only");
#if FA_BWD_HALVES
#define FA_DQ_TILE(T_, BUF_)                                                                       \
    {                                                                                              \
        const int kv0 = (T_) * BK;                                                                 \
        bool live_ = wave_active && kv0 <= wave_kv_last;                                           \
        if constexpr (EXTRA == 1) {                                                                \
            if (mbc && __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mterm[(BUF_) * MT_F + 64])) != 0) live_ = false; \
        }                                                                                          \
        if (live_) {                                                                               \
            const unsigned char* kb_ = smem + (BUF_) * IMG_B;                                      \
            const unsigned char* vb_ = smem_v + (BUF_) * IMG_B;                                    \
            _Pragma("unroll") for (int hb = 0; hb < 2; ++hb) {        /* keys kv0 + 32 hb .. +31 */ \
                if (kv0 + 32 * hb > wave_kv_last) continue;           /* every key of this half lies after the wave's rows */ \
                f32x16 s0, t0;                                                                     \
                /* the accumulators of dP^T start at -Dsum: the MFMA chain delivers dP^T - Dsum */ \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) { s0[e] = 0.f; t0[e] = ndsum; }     \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                \
                    s0 = E::mfma(*(const vec8*)(kb_ + 2048 * hb + a_rd[ks]), qf[ks], s0);          \
                    t0 = E::mfma(*(const vec8*)(vb_ + 2048 * hb + a_rd[ks]), dof[ks], t0);         \
                }                                                                                  \
                if constexpr (EXTRA) {                                                             \
                    if (mrow != nullptr) {                                                         \
                        const bool fast_ = (p.ex.mask_vec & 1) != 0 && kv0 + BK <= L;              \
                        if (EXTRA == 1) {           /* the tile's per-key terms, staged one tile ago */ \
                            typedef float f32x4_ __attribute__((ext_vector_type(4)));              \
                            const float* mt_ = mterm + (BUF_) * MT_F + 32 * hb + 4 * h;            \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                        \
                                const f32x4_ a_ = *(const f32x4_*)(mt_ + 8 * j);                   \
                                _Pragma("unroll") for (int i = 0; i < 4; ++i) s0[4 * j + i] += a_[i]; \
                            }                                                                      \
                        } else if (EXTRA >= 2 && mdma) {   /* this lane's row of the tile image */ \
                            const unsigned char* mi_ = smem + MOFF + (BUF_) * MIMG_B;              \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                        \
                                if (mbias) {                                                       \
                                    const vec4 b_ = __builtin_bit_cast(vec4, *(const u32x2*)(mi_ + (m_rd ^ ((4 * hb + j) << 4)))); \
                                    _Pragma("unroll") for (int i = 0; i < 4; ++i) s0[4 * j + i] += bias_term((float)b_[i], p.ex.inv_scale); \
                                } else {                                                           \
                                    const unsigned w_ = *(const unsigned*)(mi_ + (m_rd ^ ((2 * hb + (j >> 1)) << 4)) + 8 * (j & 1)); \
                                    _Pragma("unroll") for (int i = 0; i < 4; ++i) if (((w_ >> (8 * i)) & 0xffu) == 0) s0[4 * j + i] = -INFINITY; \
                                }                                                                  \
                            }                                                                      \
                        } else {                                                                   \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                        \
                                float a_[4];                                                       \
                                mask_load4<T>(p.ex.mask_kind, fast_, mrow, kv0 + 32 * hb + 8 * j + 4 * h, L, p.ex.inv_scale, a_); \
                                _Pragma("unroll") for (int i = 0; i < 4; ++i) s0[4 * j + i] += a_[i]; \
                            }                                                                      \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
                if (kv0 + 32 * hb + 31 > wave_last) {                                              \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                               \
                        const int key = kv0 + 32 * hb + (e & 3) + 8 * (e >> 2) + 4 * h;            \
                        if (key > klim) s0[e] = -INFINITY;                                         \
                    }                                                                              \
                }                                                                                  \
                if constexpr (EXTRA) {                                                             \
                    if (p.ex.drop_thr != 0u) {   /* t0 = dP_raw - Dsum  ->  keep ? dP_raw / (1 - p) - Dsum : -Dsum */ \
                        const unsigned col0_ = (unsigned)(kv0 + 32 * hb);                          \
                        const unsigned base_ = drop_tile_key(drop_kg, (unsigned)qrow, col0_) ^ (((unsigned)qrow & 0xffffu) << 16) ^ ((col0_ & 0xffffu) | (unsigned)(4 * h)); \
                        const float dd_ = ndsum * (1.f - p.ex.drop_rp);     /* = Dsum (rp - 1) */  \
                        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                           \
                            const unsigned ce_ = (unsigned)((e & 3) + 8 * (e >> 2));               \
                            t0[e] = mix32(base_ ^ ce_) < p.ex.drop_thr ? ndsum : __builtin_fmaf(t0[e], p.ex.drop_rp, dd_); \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
                /* dS^T = P^T o (dP^T - Dsum), P^T = 2^(S^T c - lse2): the row constants are lane scalars */ \
                _Pragma("unroll") for (int e = 0; e < 16; ++e)                                     \
                    s0[e] = fast_exp2(__builtin_fmaf(s0[e], c, -lse2)) * t0[e];                    \
                vec8 dsf[2];                                                                       \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) { dsf[0][j] = (T)s0[j]; dsf[1][j] = (T)s0[8 + j]; } \
                /* dQ^T += K^T dS^T */                                                             \
                _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                \
                    _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) {                             \
                        const unsigned char* ap = kb_ + dt * 4096 + (2 * hb + k2) * 1024;          \
                        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                  \
                            (i16x4 __attribute__((address_space(3)))*)(ap + trb0));                \
                        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                  \
                            (i16x4 __attribute__((address_space(3)))*)(ap + 512 + trb1));          \
                        const i16x8 a8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);  \
                        dq[dt] = E::mfma(__builtin_bit_cast(vec8, a8), dsf[k2], dq[dt]);           \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
#else
#define FA_DQ_TILE(T_, BUF_)                                                                       \
    {                                                                                              \
        const int kv0 = (T_) * BK;                                                                 \
        if (wave_active && kv0 <= wave_kv_last) {                                                  \
            const unsigned char* kb_ = smem + (BUF_) * IMG_B;                                      \
            const unsigned char* vb_ = smem_v + (BUF_) * IMG_B;                                    \
            f32x16 s0, s1, t0, t1;                                                                 \
            /* the accumulators of dP^T start at -Dsum: the MFMA chain delivers dP^T - Dsum (no v_sub per element) */ \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; t0[e] = ndsum; t1[e] = ndsum; } \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                    \
                s0 = E::mfma(*(const vec8*)(kb_ + a_rd[ks]), qf[ks], s0);                          \
                s1 = E::mfma(*(const vec8*)(kb_ + 2048 + a_rd[ks]), qf[ks], s1);                   \
                t0 = E::mfma(*(const vec8*)(vb_ + a_rd[ks]), dof[ks], t0);                         \
                t1 = E::mfma(*(const vec8*)(vb_ + 2048 + a_rd[ks]), dof[ks], t1);                  \
            }                                                                                      \
            if (kv0 + BK - 1 > wave_last) {                                                        \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                   \
                    const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * h;                          \
                    if (key > klim) s0[e] = -INFINITY;                                             \
                    if (key + 32 > klim) s1[e] = -INFINITY;                                        \
                }                                                                                  \
            }                                                                                      \
            /* dS^T = P^T o (dP^T - Dsum), P^T = 2^(S^T c - lse2): the row constants are lane scalars */ \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                       \
                s0[e] = fast_exp2(__builtin_fmaf(s0[e], c, -lse2)) * t0[e];                        \
                s1[e] = fast_exp2(__builtin_fmaf(s1[e], c, -lse2)) * t1[e];                        \
            }                                                                                      \
            vec8 dsf[4];                                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
                dsf[0][j] = (T)s0[j];                                                              \
                dsf[1][j] = (T)s0[8 + j];                                                          \
                dsf[2][j] = (T)s1[j];                                                              \
                dsf[3][j] = (T)s1[8 + j];                                                          \
            }                                                                                      \
            /* dQ^T += K^T dS^T */                                                                 \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                    \
                _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                 \
                    const unsigned char* ap = kb_ + dt * 4096 + kk * 1024;                         \
                    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                      \
                        (i16x4 __attribute__((address_space(3)))*)(ap + trb0));                    \
                    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                      \
                        (i16x4 __attribute__((address_space(3)))*)(ap + 512 + trb1));              \
                    const i16x8 a8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);      \
                    dq[dt] = E::mfma(__builtin_bit_cast(vec8, a8), dsf[kk], dq[dt]);               \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }

#endif

    // ring of NB buffers: tile t lives in buffer t % NB; at the top of tile t the DMA for tile t + NB - 1 goes out (its buffer
    // was released by the barrier that ended tile t - 1); at the end of tile t only that newest DMA may still be in flight
    // (kv split groups: the group's j-th tile is tile ws + QS j; n_local of them; every group keeps the workgroup's n_rounds barriers)
    const int n_rounds = QS == 1 ? n_tiles : (n_tiles + QS - 1) / QS;
    const int n_local = QS == 1 ? n_tiles : (n_tiles - ws + QS - 1) / QS;
    fa_static_for<0, NB - 1>([&](auto bi_) { constexpr int b_ = decltype(bi_)::value; if (b_ < n_local) { FA_MTERM_LOAD(b_) FA_DMA_KV(t_lo + ws + QS * b_, b_) FA_DMA_MASK(b_, b_) FA_MTERM_STORE(b_) } });
    FA_TILE_BARRIER()
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));   // row operands landed

    for (int t0 = 0; t0 < n_rounds; t0 += NB) {
        fa_static_for<0, NB>([&](auto ki_) {
            constexpr int k_ = decltype(ki_)::value;
            const int t = t0 + k_;
            if (t < n_rounds) {                                  // (workgroup-uniform)
                const bool more = t + NB - 1 < n_local;
                if (more) { FA_MTERM_LOAD(t + NB - 1) FA_DMA_KV(t_lo + ws + QS * (t + NB - 1), (k_ + NB - 1) % NB) FA_DMA_MASK(t + NB - 1, (k_ + NB - 1) % NB) }
                if (QS == 1 || t < n_local) FA_DQ_TILE(t_lo + ws + QS * t, k_)
                if (more) FA_MTERM_STORE((k_ + NB - 1) % NB)
                if (more && NB > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW * (NB - 2)) : "memory");
                else lds_dma_wait_all();
                __syncthreads();
            }
        });
    }
#undef FA_DMA_KV
#undef FA_TILE_BARRIER
#undef FA_DQ_TILE
#undef FA_DMA_MASK
#undef FA_MTERM_LOAD
#undef FA_MTERM_STORE

    // kv split groups: fold the other groups' dQ^T partial sums into group 0 through LDS, one group at a time in group order (every buffer is
    // free after the loop's last barrier; a lane's values sit at [value][lane]: conflict-free writes and reads)
    if constexpr (QS > 1) {
        float* xb = (float*)smem_all + wave * (DT * 16 * 64) + lane;
        static_assert(NWAVES * DT * 16 * 64 * 4 <= QS * GROUP_B, "exchange area must fit the K/V buffers");
        for (int sg = 1; sg < QS; ++sg) {
            if (sg > 1) __syncthreads();                // (group 0 has read the previous group's values)
            if (ws == sg) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xb[(dt * 16 + e) * 64] = dq[dt][e];
            }
            __syncthreads();
            if (ws == 0) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dq[dt][e] += xb[(dt * 16 + e) * 64];
            }
        }
        if (ws != 0) return;
    }
    if (!wave_active || qrow >= L) return;
    if constexpr (BAL) {
        if (half != 0) {      // fp32 partial sums for the reduce pass, parked in the group's dK (half 1) / dV (half 2): see above
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
            const int hr = qrow - (((L + BQ - 1) / BQ) >> 1) * BQ;        // row among the heavy rows
            T* const base = half == 1 ? (T*)p.dk + (int64_t)bb * p.dk_sb + (int64_t)hh * p.dk_sh : (T*)p.dv + (int64_t)bb * p.dv_sb + (int64_t)hh * p.dv_sh;
            const int64_t sl = half == 1 ? p.dk_sl : p.dv_sl;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                float* const wp = (float*)(base + (int64_t)(2 * hr + (dt >= DT / 2 ? 1 : 0)) * sl) + 32 * (dt % (DT / 2));
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4_ w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = dq[dt][4 * rg + j] * p.scale;
                    *(f32x4_*)(wp + 8 * rg + 4 * h) = w;
                }
            }
            return;
        }
    }
    T* __restrict__ og = (T*)p.dq + (int64_t)bb * p.dq_sb + (int64_t)hh * p.dq_sh + (int64_t)qrow * p.dq_sl;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            vec4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (T)(dq[dt][4 * rg + j] * p.scale);
            if (4 * dt + rg < dchunks) *(vec4*)(og + 32 * dt + 8 * rg + 4 * h) = w;
        }
    }
}

#ifndef FA_BWD_BODY_ONLY
template <typename T, int D, int NWAVES, int QS = 1, int BAL = 0>
__global__ __launch_bounds__(NWAVES * QS * 64) void fa_bwd_dq_kernel(const BwdParams p) {
    fa_bwd_dq_body<T, D, NWAVES, 0, QS, BAL>(p);
}
// BAL: dQ rows of the heavy row blocks = the sum of the two fp32 partial sums parked in dK and dV (fa_bwd_dq_body), one thread per 4 columns
template <typename T, int D>
__global__ __launch_bounds__(256) void fa_bwd_dq_reduce_kernel(const BwdParams p, int row0, int Gt) {
    typedef T vec4 __attribute__((ext_vector_type(4)));
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = (int)(i % (D / 4));
    const int64_t rr = i / (D / 4);
    const int rows = p.L - row0;
    const int g = (int)(rr / rows), hr = (int)(rr % rows);
    if (g >= Gt) return;
    const int bb = g / p.H, hh = g - bb * p.H;
    const int up = c4 >= D / 8 ? 1 : 0, cc = 4 * (c4 - up * (D / 8));      // which of the two parked rows, float column in it
    const f32x4_ a = *(const f32x4_*)((const float*)((const T*)p.dk + (int64_t)bb * p.dk_sb + (int64_t)hh * p.dk_sh + (int64_t)(2 * hr + up) * p.dk_sl) + cc);
    const f32x4_ b = *(const f32x4_*)((const float*)((const T*)p.dv + (int64_t)bb * p.dv_sb + (int64_t)hh * p.dv_sh + (int64_t)(2 * hr + up) * p.dv_sl) + cc);
    vec4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (T)(a[j] + b[j]);
    *(vec4*)((T*)p.dq + (int64_t)bb * p.dq_sb + (int64_t)hh * p.dq_sh + (int64_t)(row0 + hr) * p.dq_sl + 4 * c4) = w;
}
#endif

// ---------------------------------------------------------------------------------------------
// dK, dV: one workgroup = NWAVES x 32 keys; streams Q and dO tiles of 64 query rows
// ---------------------------------------------------------------------------------------------
// PASS: 0 = dK and dV in one sweep; 1 = dV alone, 2 = dK alone - the same body with one of the two results never stored, so that the compiler
// drops its accumulators, its MFMAs and (dV alone) the V^T fragments and the dP product: the head_dim 256 mask / dropout launches run the two
// passes one after the other (the single sweep fills all 512 registers without the mask terms), at the price of a second S = Q K^T.
// QS > 1 (query-sweep split groups, for grids that cannot fill the chip): QS groups of NWAVES waves own the SAME 128 keys; group s streams the
// items s, s + QS, ... of the (head, query tile) stream through buffers of its own, in lock step with the others, and the groups' dK^T / dV^T
// partial sums meet in LDS after the loop, added in group order (deterministic: no atomics, as everywhere in this backward).
// QH = 2 (row halves, head_dim 128's form of the same idea - its single sweep fills 256 registers, so a second group cannot run a sweep of its own): the
// workgroup has TWO groups of NWAVES waves on the same 128 keys AND the same streamed tiles (one set of LDS buffers: no extra LDS); group h computes
// the 32-row half h of every 64-row query tile, i.e. half of each item's products - twice the waves per SIMD on half the chain per item - and
// the two groups' dK^T / dV^T partial sums meet in LDS after the loop (group 1's added to group 0's: deterministic).  Group 0 issues the Q tile
// DMA, group 1 the dO tile's.
template <typename T, int D, int NWAVES, int EXTRA = 0, int PASS = 0, int QS = 1, int QH = 1>
__device__ __forceinline__ void fa_bwd_dkdv_body(const BwdParams& p) {
    static_assert(QS == 1 || (EXTRA == 0 && PASS == 0), "query-sweep split groups: the plain kernels only");
    static_assert(QH == 1 || (QH == 2 && QS == 1 && EXTRA == 0 && PASS == 0), "row halves: the plain unsplit sweep only");
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    using GM = BwdGeom<T, D, NWAVES>;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int BKW = NWAVES * 32, BQ = 64, KS = GM::KS, DT = GM::DT, IMG_B = GM::IMG_B, PPW = GM::PPW, NB = GM::NB;

    // Q0 .. Q(NB-1), dO0 .. dO(NB-1) images, then per-buffer row constants: lse2[64], dsum[64]
    // (EXTRA == 2 | 3: plus per buffer the item's mask tile, 64 rows x the workgroup's 128 keys: 8 KB of boolean bytes or 16 KB of bias)
    constexpr int MIMG_B = EXTRA == 3 ? 16384 : 8192, MOFF = 2 * NB * IMG_B + NB * 512;
    // (head_dim 256 with a bias: the Q / dO images fill 128 KB - two 16 KB bias images do not fit beside them.  There every wave keeps the bias of ITS 32
    //  keys only, in half-item pieces - 32 rows x 32 keys x 2 B = 2 KB - that travel through a wave-private ring of three: half k of the (item, row half)
    //  sequence lives in slot k % 3, the first half of item j + 1 is requested at the top of round j (the slot the second half of item j - 1 has left), its
    //  second half as soon as the wave has read the first half of item j (whose slot it takes): 24 KB in all, no barrier of its own - a wave is the only
    //  writer and the only reader of its ring - and every piece has 85 % of an item's time or more to land before the round's closing wait.)
    constexpr bool MPRIV = D == 256 && EXTRA == 3;
    constexpr int MHALF_B = 2048, MRING_B = 3 * MHALF_B;
    constexpr int GROUP_B = MOFF + (EXTRA >= 2 ? (MPRIV ? NWAVES * MRING_B : NB * MIMG_B) : 0);
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[QS * GROUP_B];

    // Grouped-query attention: the workgroup owns 128 keys of ONE K/V head and sweeps the query tiles of all kv_div query
    // heads that read it, one head after the other, as one stream of (head, query tile) items: dK and dV of the shared
    // head are summed in registers - no expanded K/V, no atomics, no second pass.
    int g, kb;
    if (!bwd_block_map(p.G8kv, p.Gkv, p.sched_kv, p.nk_tiles, false, p.xs_kv, p.nk_real, g, kb)) return;   // key block 0 is the heaviest

    const int L = p.L;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int ws = QS == 1 ? 0 : wave_all / NWAVES;                    // query-sweep split group of this wave
    const int wh = QH == 1 ? 0 : wave_all / NWAVES;                    // row-half group of this wave (QH = 2)
    const int wave = (QS == 1 && QH == 1) ? wave_all : wave_all - (ws + wh) * NWAVES;      // its 32 keys inside the group
    const int tid = (int)threadIdx.x, lane = tid & 63;                 // (tid: the unsplit kernel's thread index; split groups use wave / lane)
    // (per group: Q / dO images, then rowc = [buf][ lse2 0..63 | dsum 0..63 ] - the pointers are formed inside the group's copy of the loop)
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.Hkv, hh = g - bb * p.Hkv;                 // batch entry, K/V head
    const int hq0 = hh * p.kv_div;                                 // its first query head
    const T* __restrict__ qg0 = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hq0 * p.q_sh;
    const T* __restrict__ gg0 = (const T*)p.dout + (int64_t)bb * p.do_sb + (int64_t)hq0 * p.do_sh;

    const int key0 = kb * BKW + wave * 32;        // first key of this wave
    const int key = key0 + r;                     // this lane's key (both halves hold the same key)
    const bool wave_active = key0 < L;
    const int kr = key < L ? key : L - 1;

    const int dchunks = p.Dr >> 3;
    // B operands: K^T and V^T fragments of this lane's key
    vec8 kf[KS], vf[KS];
    {
        const T* kp = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)hh * p.k_sh + (int64_t)kr * p.k_sl;
        const T* vp = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)hh * p.v_sh + (int64_t)kr * p.v_sl;
        vec8 z_;
#pragma unroll
        for (int j = 0; j < 8; ++j) z_[j] = (T)0.f;
        if constexpr (D >= 192) {             // (the half-wave's chunk offset through an opaque copy: written plainly, the 512-register kernel's
            int h_p = h;                      //  prologue parked a - never reloaded - address pair in a stack slot)
            asm volatile("" : "+v"(h_p));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int ch_ = 2 * ks + h_p, cc_ = ch_ < dchunks ? ch_ : 0;
                const vec8 kl_ = *(const vec8*)(kp + 8 * cc_), vl_ = *(const vec8*)(vp + 8 * cc_);
                kf[ks] = ch_ < dchunks ? kl_ : z_;
                vf[ks] = ch_ < dchunks ? vl_ : z_;
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {     // (columns past the true head dim, BwdParams::Dr: zeros)
            const int ch_ = 2 * ks + h, cc_ = ch_ < dchunks ? ch_ : 0;
            const vec8 kl_ = *(const vec8*)(kp + 8 * cc_), vl_ = *(const vec8*)(vp + 8 * cc_);
            kf[ks] = ch_ < dchunks ? kl_ : z_;
            vf[ks] = ch_ < dchunks ? vl_ : z_;
        }
        }
    }
    const float c = p.scale_log2;
    const int mask_esz = p.ex.mask_kind == 2 ? (int)sizeof(T) : 1;

    const unsigned q_slb = (unsigned)p.q_sl * 2, g_slb = (unsigned)p.do_sl * 2;
    // LDS-DMA geometry: piece pc = wave + 4 j of an image holds rows 16 (pc & 3) .. +15 of its 64-byte column block pc >> 2 - with four waves
    // every piece of a wave holds the SAME rows and piece j starts 64 j bytes further along them: one lane offset per operand, the piece in the
    // instruction's immediate (a lane offset per piece is 2 x PPW registers - 16 at head_dim 256, which that kernel spilled)
    static_assert(NWAVES == 4, "piece = wave + 4 j");
    const bool d_full = dchunks == D / 8;               // (else: chunks past the true head dim get an offset beyond any extent - zeros)
    // (head_dim 128 runs at the 256-register limit of two waves per SIMD: there the two offsets are formed per item from an opaque copy of the
    //  lane index - a dozen scalar-fed vector instructions per 64 MFMAs - instead of being kept, spilled, and reloaded behind a vmcnt(0))
#define FA_DMA_LANE_OFFSETS()                                                                      \
        int le_ = lane;                                                                            \
        if constexpr (D == 128) asm volatile("" : "+v"(le_));                                      \
        const int d_row = (2 * wave + (le_ >> 5)) * 8 + ((le_ >> 2) & 7);                          \
        const int d_c0 = (le_ & 3) ^ ((d_row >> 2) & 3);                                           \
        const unsigned q_vb = (unsigned)d_row * q_slb + (d_c0 << 4), g_vb = (unsigned)d_row * g_slb + (d_c0 << 4);
    const unsigned q_bytes = ((unsigned)(L - 1) * (unsigned)p.q_sl + (unsigned)p.Dr) * 2;
    const unsigned g_bytes = ((unsigned)(L - 1) * (unsigned)p.do_sl + (unsigned)p.Dr) * 2;
    // item J_ of the stream = query tile i0 + J_ % n_sweep of query head hq0 + J_ / n_sweep
#define FA_ITEM(J_, HQ_, I_)                                                                       \
        const int HQ_ = p.kv_div == 1 ? 0 : (J_) / n_sweep;                                        \
        const int I_ = i0 + (J_) - HQ_ * n_sweep;
#define FA_DMA_QG(J_, BUF_)                                                                        \
    {                                                                                              \
        FA_ITEM(J_, hq_, it_)                                                                      \
        FA_DMA_LANE_OFFSETS()                                                                      \
        const unsigned qo_ = (unsigned)it_ * BQ * q_slb, go_ = (unsigned)it_ * BQ * g_slb;         \
        const i32x4 qr_ = make_raw_rsrc((const char*)(qg0 + (int64_t)hq_ * p.q_sh) + qo_, q_bytes - qo_);   \
        const i32x4 gr_ = make_raw_rsrc((const char*)(gg0 + (int64_t)hq_ * p.do_sh) + go_, g_bytes - go_);  \
        if (FA_BWD_ABL != 2) {                                                                     \
        if (d_full) {                                                                              \
            fa_static_for<0, PPW>([&](auto ji_) {                                                  \
                constexpr int j = decltype(ji_)::value;                                            \
                if (QH == 1 || wh == 0) lds_dma_b128_imm<64 * j>(qr_, smem_lds + (BUF_) * IMG_B + j * 4096 + wave * 1024, q_vb); \
                if (QH == 1 || wh == 1) lds_dma_b128_imm<64 * j>(gr_, smem_lds + (NB + (BUF_)) * IMG_B + j * 4096 + wave * 1024, g_vb); \
            });                                                                                    \
        } else {   /* (the lane's chunk through an opaque copy: the per-piece offsets are formed here, not kept in 2 PPW registers across the loop) */ \
            int c0_ = d_c0;                                                                        \
            asm volatile("" : "+v"(c0_));                                                          \
            _Pragma("unroll") for (int j = 0; j < PPW; ++j) {                                      \
                const bool in_ = 4 * j + c0_ < dchunks;                                            \
                if (QH == 1 || wh == 0) lds_dma_b128(qr_, smem_lds + (BUF_) * IMG_B + j * 4096 + wave * 1024, in_ ? q_vb + 64 * j : 0x80000000u); \
                if (QH == 1 || wh == 1) lds_dma_b128(gr_, smem_lds + (NB + (BUF_)) * IMG_B + j * 4096 + wave * 1024, in_ ? g_vb + 64 * j : 0x80000000u); \
            }                                                                                      \
        }                                                                                          \
        }                                                                                          \
    }
    // row constants of q tile I_: threads 0..63 fetch lse*log2e, 64..127 fetch -Dsum; rows past L get
    // lse2 = +inf (P = 0) and Dsum = 0
    const float* __restrict__ lse_g0 = p.lse + ((int64_t)bb * p.H + hq0) * L;       // [G, L] rows of query head hq0
    const float* __restrict__ dsum_g0 = p.dsum + ((int64_t)bb * p.H + hq0) * L;
    // (the group's threads 0..63 / 64..127: with split groups told by the wave index, a scalar - the group-local thread index would be one more
    //  register in a kernel that has none left at head_dim 128)
#define RC_W0 (QS == 1 ? tid < 64 : wave == 0)
#define RC_W1 (QS == 1 ? (tid >= 64 && tid < 128) : wave == 1)
#define RC_IDX (QS == 1 ? tid : wave * 64 + lane)
#define FA_ROWC_LOAD(J_)                                                                           \
    {                                                                                              \
        FA_ITEM(J_, hq_, it_)                                                                      \
        const int rr_ = it_ * BQ + (tid & 63);                                                     \
        rc_stage = RC_W0 ? INFINITY : 0.f;                                                         \
        if (RC_W0 && rr_ < L) {                                                                    \
            const float l_ = lse_g0[(int64_t)hq_ * L + rr_];                                       \
            rc_stage = (EXTRA && l_ == -INFINITY) ? INFINITY : l_ * LOG2E;   /* a fully masked row: P = 0 */ \
        }                                                                                          \
        if (RC_W1 && rr_ < L) rc_stage = -dsum_g0[(int64_t)hq_ * L + rr_];   /* negated: see the dP accumulators */ \
    }
#define FA_ROWC_STORE(BUF_) \
    if (RC_W0 || RC_W1) rowc[(BUF_) * 128 + RC_IDX] = rc_stage;

    int a_rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_rd[ks] = img_addr(r, 2 * ks + h);
    const int rowin8 = 4 * h + ((lane & 15) >> 2);
    const int c4 = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const int trb0 = rowin8 * 64 + ((c4 ^ h) << 4) + (lane & 1) * 8;
    const int trb1 = rowin8 * 64 + ((c4 ^ (2 + h)) << 4) + (lane & 1) * 8;

    // query tiles that can see this workgroup's keys: i0 .. n_qt-1
    const int n_qt = (L + BQ - 1) / BQ;
    const bool causal = p.causal != 0;             // 0: every query tile sees this workgroup's keys, no mask
    const int i0 = causal ? (kb * BKW) / BQ : 0;
    const int n_sweep = n_qt - i0;                 // query tiles per head
    const int n_items = n_sweep > 0 ? n_sweep * p.kv_div : 0;

    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
    float rc_stage = 0.f;
    // EXTRA == 2 | 3 (a boolean mask | a bias that differs from row to row): a key-on-the-lane kernel reads the mask down its COLUMNS - from global memory that
    // is one byte (or one bf16) per lane and row, 32 gather instructions per item and wave on a texture path four waves share (measured:
    // 5.7 ms against 0.8 ms for the unmasked kernel).  So the item's mask tile travels to LDS the way its Q and dO images do: LDS-DMA,
    // 16 bytes per lane, one item ahead, rows whole (2 / 4 instructions per wave and item); the lanes then read their column with
    // ds_read_u8 / _u16.  16-byte chunk c of row i sits at chunk position c ^ 4 [i & 4] (bias: c ^ 8 [i & 4]): the two half-waves of one
    // read hold rows i and i + 4, which a plain row-major image would put on the same banks.  Needs mask_vec (pointer and strides multiples
    // of four elements); any other mask is loaded in place, element by element.
    constexpr bool mbias = EXTRA == 3;
    const bool mdma = EXTRA >= 2 && p.ex.mask != nullptr && (p.ex.mask_vec & 2) != 0;
    const unsigned m_slb = (unsigned)p.ex.m_sl * (unsigned)mask_esz;
    int m_row[EXTRA >= 2 ? 4 : 1];
    unsigned m_col[EXTRA >= 2 ? 4 : 1];
    int m_rd = 0;
    (void)mbias; (void)mdma; (void)m_slb; (void)m_row; (void)m_col; (void)m_rd;
    if constexpr (MPRIV) {
        // a piece = 16 rows x the wave's 32 keys (64 B a row), lane-linear in LDS: 64-byte slot lane >> 2 holds row b0 + 2 b1 + 4 b2 + 8 b3 of the
        // piece at slot b2 + 2 b0 + 4 b1 + 8 b3 - the two half-waves of one read hold rows i and i + 4 (b2), which land on neighbouring slots, i.e. on
        // different banks, and every register's row is the lane's own address plus a constant
        const int sl_ = lane >> 2;
        m_row[0] = ((sl_ >> 1) & 1) + 2 * ((sl_ >> 2) & 1) + 4 * (sl_ & 1) + 8 * (sl_ >> 3);
        m_col[0] = (unsigned)(lane & 3) << 4;
#pragma unroll
        for (int j = 1; j < 4; ++j) { m_row[j] = 0; m_col[j] = 0; }
        m_rd = wave * MRING_B + h * 64 + r * 2;
    } else if constexpr (EXTRA >= 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pc = wave + 4 * j;
            const int row = mbias ? 4 * pc + (lane >> 4) : 8 * pc + (lane >> 3);
            const int cpos = mbias ? (lane & 15) : (lane & 7);
            m_row[j] = row;
            m_col[j] = (unsigned)(cpos ^ (((row >> 2) & 1) * (mbias ? 8 : 4))) << 4;
        }
        m_rd = mbias ? 4 * h * 256 + (((4 * wave + (r >> 3)) ^ (8 * h)) << 4) + (r & 7) * 2
                     : 4 * h * 128 + (((2 * wave + (r >> 4)) ^ (4 * h)) << 4) + (r & 15);
    }
    // MPRIV: half K_ = 2 item + row half of the wave's bias sequence -> ring slot K_ % 3 (two 1 KiB pieces; the wave's own reads of the slot's
    // previous tenant are drained first)
#define FA_DMA_MHALF(K_)                                                                           \
    if constexpr (MPRIV) {                                                                         \
        if (mdma && wave_active) {                                                                 \
            const int km_ = (K_), jm_ = km_ >> 1;                                                  \
            FA_ITEM(jm_, hqm_, itm_)                                                               \
            const int q0m_ = itm_ * BQ, rl_ = L - 1 - q0m_;                                        \
            const char* mb_ = (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)(hq0 + hqm_) * p.ex.m_sh + (int64_t)q0m_ * p.ex.m_sl + kb * BKW + wave * 32) * mask_esz; \
            const int64_t left_ = (((int64_t)rl_ * p.ex.m_sl + (L - kb * BKW - wave * 32)) * mask_esz + 3) & ~3ll;   /* to the end of the slice's last row, whole dwords (> 0: the wave is active) */ \
            i32x4 mr_ = make_raw_rsrc(mb_, left_ < 0xffffffffll ? (unsigned)left_ : 0xffffffffu);   \
            _Pragma("unroll") for (int w_ = 0; w_ < 3; ++w_) mr_[w_] = __builtin_amdgcn_readfirstlane(mr_[w_]);   \
            const unsigned dst_ = smem_lds + MOFF + wave * MRING_B + (km_ % 3) * MHALF_B;          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {   /* (rows past L: the last row again - their P is 0) */ \
                const int row_ = 32 * (km_ & 1) + 16 * i_ + m_row[0];                              \
                lds_dma_b128(mr_, dst_ + i_ * 1024, (unsigned)(row_ < rl_ ? row_ : rl_) * m_slb + m_col[0]); \
            }                                                                                      \
        }                                                                                          \
    }
#define FA_DMA_MASK(J_, BUF_)                                                                      \
    if constexpr (MPRIV) {   /* the first half of the item; its second half follows the wave's reads of the slot's tenant (FA_DKDV_TILE) */ \
        FA_DMA_MHALF(2 * (J_))                                                                     \
    } else if constexpr (EXTRA >= 2) {                                                             \
        if (mdma) {                                                                                \
            FA_ITEM(J_, hqm_, itm_)                                                                \
            const int q0m_ = itm_ * BQ, rl_ = L - 1 - q0m_;                                        \
            const char* mb_ = (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)(hq0 + hqm_) * p.ex.m_sh + (int64_t)q0m_ * p.ex.m_sl + kb * BKW) * mask_esz; \
            const int64_t left_ = (((int64_t)rl_ * p.ex.m_sl + (L - kb * BKW)) * mask_esz + 3) & ~3ll;   /* to the end of the slice's last row, whole dwords (fa_fwd_dma.hip FA_MASK_EXTENT) */ \
            i32x4 mr_ = make_raw_rsrc(mb_, left_ < 0xffffffffll ? (unsigned)left_ : 0xffffffffu);   \
            _Pragma("unroll") for (int w_ = 0; w_ < 3; ++w_) mr_[w_] = __builtin_amdgcn_readfirstlane(mr_[w_]);   /* (64-bit products leave the scalar unit) */ \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
                if (mbias || j < 2) {   /* (rows past L: the last row again - their P is 0) */     \
                    const int rw_ = m_row[j] < rl_ ? m_row[j] : rl_;                               \
                    lds_dma_b128(mr_, smem_lds + MOFF + (BUF_) * MIMG_B + (wave + 4 * j) * 1024, (unsigned)rw_ * m_slb + m_col[j]); \
                }                                                                                  \
        }                                                                                          \
    }
    // EXTRA == 1 with a mask: a row-broadcast one (row stride 0: fa_bwd_extra.hip sends every other mask to EXTRA == 2 | 3), so the term of a
    // score depends on its key alone - one load per lane and query head, not one per element; a wave whose 32 keys are all masked for
    // that head (a padded tail) skips the head's tiles: its P is 0, its dK / dV contribution exactly the zeros already there
    float kterm = 0.f;
    int kterm_hq = -1;
    bool kskip = false;
    (void)kterm; (void)kterm_hq; (void)kskip;

#if FA_BWD_HALVES
#define FA_DKDV_TILE(J_, BUF_)                                                                     \
    {                                                                                              \
        FA_ITEM(J_, hqt_, itt_)                                                                    \
        (void)hqt_; (void)mask_esz;                                                                \
        const int q0 = itt_ * BQ;                                                                  \
        /* (head_dim 128 and up: the half-wave offset through an opaque copy - the row-constant addresses and the diagonal's row numbers \
           are then formed where they are used; hoisted out of the item loop they are a dozen registers the 256-register kernel spills) */ \
        int h_t = h;                                                                               \
        if constexpr (D >= 128) asm volatile("" : "+v"(h_t));                                      \
        if constexpr (EXTRA == 1) {                                                                \
            if (p.ex.mask != nullptr && wave_active && kterm_hq != (p.ex.m_sh == 0 ? 0 : hqt_)) {  \
                kterm_hq = p.ex.m_sh == 0 ? 0 : hqt_;                                              \
                const char* mb_ = (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)(hq0 + hqt_) * p.ex.m_sh) * mask_esz; \
                kterm = mask_load1<T>(p.ex.mask_kind, mb_, 0, kr, p.ex.inv_scale);                 \
                kskip = __builtin_amdgcn_ballot_w64(kterm != -INFINITY) == 0;                      \
            }                                                                                      \
        }                                                                                          \
        bool mh_sent_ = false;   /* (MPRIV) the next item's second bias half is on its way */       \
        (void)mh_sent_;                                                                            \
        if (wave_active && !(EXTRA == 1 && kskip) && (!causal || q0 + BQ - 1 >= key0)) {           \
            const unsigned char* qb_ = smem + (BUF_) * IMG_B;                                      \
            const unsigned char* gb_ = smem_g + (BUF_) * IMG_B;                                    \
            const float* rc_ = rowc + (BUF_) * 128;                                                \
            _Pragma("unroll") for (int hb = 0; hb < 2; ++hb) {       /* query rows q0 + 32 hb .. +31 */ \
                if (QH == 2 && hb != wh) continue;                  /* row halves: the other group's half */ \
                if (causal && q0 + 32 * hb + 31 < key0) continue;   /* every query of this half lies before the wave's keys */ \
                f32x16 s0, t0;                                                                     \
                /* dP accumulators start at -Dsum[q] (rowc holds the negated sums) */              \
                _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                 \
                    const f32x4 da = *(const f32x4*)(rc_ + 64 + 32 * hb + 8 * gq + 4 * h_t);         \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) { s0[4 * gq + j] = 0.f; t0[4 * gq + j] = da[j]; } \
                }                                                                                  \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                \
                    s0 = E::mfma(*(const vec8*)(qb_ + 2048 * hb + a_rd[ks]), kf[ks], s0);          \
                    t0 = E::mfma(*(const vec8*)(gb_ + 2048 * hb + a_rd[ks]), vf[ks], t0);          \
                }                                                                                  \
                if constexpr (EXTRA) {                                                             \
                    if (p.ex.mask != nullptr) {   /* mask[row][this lane's key], one row per register */ \
                        if constexpr (EXTRA >= 2) {                                                \
                            if (MPRIV && mdma) {   /* this lane's column of the wave's half (ring slot (2 item + hb) % 3) */ \
                                const unsigned char* mi_ = smem + MOFF + m_rd + ((2 * (J_) + hb) % 3) * MHALF_B; \
                                _Pragma("unroll") for (int e = 0; e < 16; ++e)                     \
                                    s0[e] += bias_term((float)*(const T*)(mi_ + (e >> 3) * 1024 + ((e >> 2) & 1) * 512 + (e & 3) * 128), p.ex.inv_scale); \
                                if (hb == 0 && more) {   /* the slot is read: the next item's second half takes it */ \
                                    FA_DMA_MHALF(2 * (J_) + 3)                                     \
                                    mh_sent_ = true;                                               \
                                }                                                                  \
                            } else if (mdma) {   /* this lane's column of the tile image */        \
                                const unsigned char* mi_ = smem + MOFF + (BUF_) * MIMG_B + m_rd;   \
                                if (mbias) {                                                       \
                                    _Pragma("unroll") for (int e = 0; e < 16; ++e)                 \
                                        s0[e] += bias_term((float)*(const T*)(mi_ + (32 * hb + (e & 3) + 8 * (e >> 2)) * 256), p.ex.inv_scale); \
                                } else {                                                           \
                                    _Pragma("unroll") for (int e = 0; e < 16; ++e)                 \
                                        s0[e] += mi_[(32 * hb + (e & 3) + 8 * (e >> 2)) * 128] ? 0.f : -INFINITY; \
                                }                                                                  \
                            } else {                                                               \
                                const char* mb_ = (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)(hq0 + hqt_) * p.ex.m_sh) * mask_esz; \
                                _Pragma("unroll") for (int e = 0; e < 16; ++e) {                   \
                                    const int qq = q0 + 32 * hb + (e & 3) + 8 * (e >> 2) + 4 * h_t;  \
                                    s0[e] += mask_load1<T>(p.ex.mask_kind, mb_, (int64_t)(qq < L ? qq : L - 1) * p.ex.m_sl, kr, p.ex.inv_scale); \
                                }                                                                  \
                            }                                                                      \
                        } else {   /* one term per key (loaded where the item's query head changed) */ \
                            _Pragma("unroll") for (int e = 0; e < 16; ++e) s0[e] += kterm;         \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
                if (causal && q0 + 32 * hb < key0 + 31) { /* diagonal: query < key is masked */    \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                               \
                        const int qq = q0 + 32 * hb + (e & 3) + 8 * (e >> 2) + 4 * h_t;              \
                        if (qq < key) s0[e] = -INFINITY;                                           \
                    }                                                                              \
                }                                                                                  \
                _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                 \
                    const f32x4 la = *(const f32x4*)(rc_ + 32 * hb + 8 * gq + 4 * h_t);              \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                \
                        const int e = 4 * gq + j;                                                  \
                        s0[e] = fast_exp2(__builtin_fmaf(s0[e], c, -la[j]));                       \
                    }                                                                              \
                }                                                                                  \
                f32x16 pd_ = s0;   /* what dV sees: P, or with dropout keep ? P / (1 - p) : 0 */    \
                if constexpr (EXTRA) {                                                             \
                    if (p.ex.drop_thr != 0u) {                                                     \
                        const unsigned row0_ = (unsigned)(q0 + 32 * hb);                           \
                        const unsigned kgq_ = drop_group_key(drop_key0_of(p.ex), (unsigned)(bb * p.H + hq0 + hqt_)); \
                        const unsigned base_ = drop_tile_key(kgq_, row0_, (unsigned)key0) ^ (((row0_ & 0xffffu) | (unsigned)(4 * h_t)) << 16) ^ ((unsigned)key & 0xffffu); \
                        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                         \
                            const f32x4 da = *(const f32x4*)(rc_ + 64 + 32 * hb + 8 * gq + 4 * h_t);   /* -Dsum[row] */ \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                        \
                                const int e = 4 * gq + j;                                          \
                                const bool keep_ = mix32(base_ ^ ((unsigned)((e & 3) + 8 * (e >> 2)) << 16)) >= p.ex.drop_thr; \
                                pd_[e] = keep_ ? s0[e] * p.ex.drop_rp : 0.f;                       \
                                t0[e] = keep_ ? __builtin_fmaf(t0[e], p.ex.drop_rp, da[j] * (1.f - p.ex.drop_rp)) : da[j]; \
                            }                                                                      \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) t0[e] = s0[e] * t0[e];              \
                vec8 pf[2], dsf[2];                                                                \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                    \
                    pf[0][j] = (T)pd_[j];  pf[1][j] = (T)pd_[8 + j];                               \
                    dsf[0][j] = (T)t0[j]; dsf[1][j] = (T)t0[8 + j];                                \
                }                                                                                  \
                _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                \
                    _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2) {                             \
                        const int kk = 2 * hb + k2;                                                \
                        const unsigned char* gp_ = gb_ + dt * 4096 + kk * 1024;                    \
                        const unsigned char* qp_ = qb_ + dt * 4096 + kk * 1024;                    \
                        const i16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                 \
                            (i16x4 __attribute__((address_space(3)))*)(gp_ + trb0));               \
                        const i16x4 ghi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                 \
                            (i16x4 __attribute__((address_space(3)))*)(gp_ + 512 + trb1));         \
                        const i16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                 \
                            (i16x4 __attribute__((address_space(3)))*)(qp_ + trb0));               \
                        const i16x4 qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                 \
                            (i16x4 __attribute__((address_space(3)))*)(qp_ + 512 + trb1));         \
                        const i16x8 g8 = __builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7); \
                        const i16x8 q8 = __builtin_shufflevector(qlo, qhi, 0, 1, 2, 3, 4, 5, 6, 7); \
                        dv[dt] = E::mfma(__builtin_bit_cast(vec8, g8), pf[k2], dv[dt]);            \
                        dk[dt] = E::mfma(__builtin_bit_cast(vec8, q8), dsf[k2], dk[dt]);           \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        if constexpr (MPRIV) {   /* (a wave that skipped the first half of this item - or all of it - sends the next item's second half here) */ \
            if (more && !mh_sent_) { FA_DMA_MHALF(2 * (J_) + 3) }                                  \
        }                                                                                          \
    }
#else
#define FA_DKDV_TILE(J_, BUF_)                                                                     \
    {                                                                                              \
        FA_ITEM(J_, hqt_, itt_)                                                                    \
        (void)hqt_;                                                                                \
        const int q0 = itt_ * BQ;                                                                  \
        if (wave_active && (!causal || q0 + BQ - 1 >= key0)) {                                     \
            const unsigned char* qb_ = smem + (BUF_) * IMG_B;                                      \
            const unsigned char* gb_ = smem_g + (BUF_) * IMG_B;                                    \
            const float* rc_ = rowc + (BUF_) * 128;                                                \
            f32x16 s0, s1, t0, t1;                                                                 \
            /* the accumulators of dP start at -Dsum[q] (rowc holds the negated sums; the row constants vary along the \
               registers): the MFMA chain delivers dP - Dsum, no v_sub per element */              \
            _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                     \
                const f32x4 da = *(const f32x4*)(rc_ + 64 + 8 * gq + 4 * h);                       \
                const f32x4 db = *(const f32x4*)(rc_ + 96 + 8 * gq + 4 * h);                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                    \
                    s0[4 * gq + j] = 0.f; s1[4 * gq + j] = 0.f;                                    \
                    t0[4 * gq + j] = da[j]; t1[4 * gq + j] = db[j];                                \
                }                                                                                  \
            }                                                                                      \
            /* S = Q K^T and dP = dO V^T, key on the lane, query rows in the registers */          \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                    \
                s0 = E::mfma(*(const vec8*)(qb_ + a_rd[ks]), kf[ks], s0);                          \
                s1 = E::mfma(*(const vec8*)(qb_ + 2048 + a_rd[ks]), kf[ks], s1);                   \
                t0 = E::mfma(*(const vec8*)(gb_ + a_rd[ks]), vf[ks], t0);                          \
                t1 = E::mfma(*(const vec8*)(gb_ + 2048 + a_rd[ks]), vf[ks], t1);                   \
            }                                                                                      \
            if (causal && q0 < key0 + 31) { /* diagonal tile: query < key is masked */             \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                   \
                    const int qq = q0 + (e & 3) + 8 * (e >> 2) + 4 * h;                            \
                    if (qq < key) s0[e] = -INFINITY;                                               \
                    if (qq + 32 < key) s1[e] = -INFINITY;                                          \
                }                                                                                  \
            }                                                                                      \
            /* P = 2^(S c - lse2[q]) ; dS = P o (dP - Dsum[q]) : row constants vary along registers */ \
            _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                     \
                const f32x4 la = *(const f32x4*)(rc_ + 8 * gq + 4 * h);                            \
                const f32x4 lb = *(const f32x4*)(rc_ + 32 + 8 * gq + 4 * h);                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                    \
                    const int e = 4 * gq + j;                                                      \
                    s0[e] = fast_exp2(__builtin_fmaf(s0[e], c, -la[j]));                           \
                    s1[e] = fast_exp2(__builtin_fmaf(s1[e], c, -lb[j]));                           \
                    t0[e] = s0[e] * t0[e];                                                         \
                    t1[e] = s1[e] * t1[e];                                                         \
                }                                                                                  \
            }                                                                                      \
            vec8 pf[4], dsf[4];                                                                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
                pf[0][j] = (T)s0[j];  pf[1][j] = (T)s0[8 + j];  pf[2][j] = (T)s1[j];  pf[3][j] = (T)s1[8 + j]; \
                dsf[0][j] = (T)t0[j]; dsf[1][j] = (T)t0[8 + j]; dsf[2][j] = (T)t1[j]; dsf[3][j] = (T)t1[8 + j]; \
            }                                                                                      \
            /* dV^T += dO^T P   and   dK^T += Q^T dS */                                            \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                    \
                _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                 \
                    const unsigned char* gp_ = gb_ + dt * 4096 + kk * 1024;                        \
                    const unsigned char* qp_ = qb_ + dt * 4096 + kk * 1024;                        \
                    const i16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                     \
                        (i16x4 __attribute__((address_space(3)))*)(gp_ + trb0));                   \
                    const i16x4 ghi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                     \
                        (i16x4 __attribute__((address_space(3)))*)(gp_ + 512 + trb1));             \
                    const i16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                     \
                        (i16x4 __attribute__((address_space(3)))*)(qp_ + trb0));                   \
                    const i16x4 qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                     \
                        (i16x4 __attribute__((address_space(3)))*)(qp_ + 512 + trb1));             \
                    const i16x8 g8 = __builtin_shufflevector(glo, ghi, 0, 1, 2, 3, 4, 5, 6, 7);    \
                    const i16x8 q8 = __builtin_shufflevector(qlo, qhi, 0, 1, 2, 3, 4, 5, 6, 7);    \
                    dv[dt] = E::mfma(__builtin_bit_cast(vec8, g8), pf[kk], dv[dt]);                \
                    dk[dt] = E::mfma(__builtin_bit_cast(vec8, q8), dsf[kk], dk[dt]);               \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }

#endif

    // ring of NB buffers (see fa_bwd_dq_kernel).  The row constants of an item travel through one staging register: loaded
    // BEFORE that item's DMA is issued - vmcnt retires in order, so waiting for them at the end of a tile does not wait for
    // the newer DMA - and stored to the item's buffer at the end of the tile during which they were loaded.
    // (query-sweep split groups: the group's j-th item is item ws + QS j; n_local of them; every group keeps the workgroup's n_rounds barriers.
    //  Each group runs a copy of the loop of its own in which its index - and with it every LDS address - is a compile-time constant, as in the
    //  unsplit kernel: with the group's LDS base in a register the head_dim 128 kernel, which has none to spare, spilled seven.)
    const int n_rounds = QS == 1 ? n_items : (n_items + QS - 1) / QS;
    const int ws_dyn = ws;
    fa_static_for<0, QS>([&](auto wi_) {
    constexpr int ws = decltype(wi_)::value;
    if (ws_dyn != ws) return;
    unsigned char* const smem = smem_all + ws * GROUP_B;
    unsigned char* const smem_g = smem + NB * IMG_B;
    float* const rowc = (float*)(smem + 2 * NB * IMG_B);
    const unsigned smem_lds = lds_offset(smem);
    const int n_local = QS == 1 ? n_items : (n_items - ws + QS - 1) / QS;
    fa_static_for<0, NB - 1>([&](auto bi_) {
        constexpr int b_ = decltype(bi_)::value;
        if (b_ < n_local) {
            FA_ROWC_LOAD(ws + QS * b_)
            FA_DMA_QG(ws + QS * b_, b_)
            FA_DMA_MASK(ws + QS * b_, b_)
            FA_DMA_MHALF(2 * (ws + QS * b_) + 1)
            FA_ROWC_STORE(b_)
        }
    });
    lds_dma_wait_all();
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {   // row operands landed
        if constexpr (PASS == 1) asm volatile("" : "+v"(kf[ks]));
        else asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
    }

    for (int j0 = 0; j0 < n_rounds; j0 += NB) {
        fa_static_for<0, NB>([&](auto ki_) {
            constexpr int k_ = decltype(ki_)::value;
            const int j = j0 + k_;
            if (j < n_rounds) {                                  // (workgroup-uniform)
                const bool more = j + NB - 1 < n_local;
                if (more) {
                    FA_ROWC_LOAD(ws + QS * (j + NB - 1))
                    FA_DMA_QG(ws + QS * (j + NB - 1), (k_ + NB - 1) % NB)
                    FA_DMA_MASK(ws + QS * (j + NB - 1), (k_ + NB - 1) % NB)
                }
                if (QS == 1 || j < n_local) FA_DKDV_TILE(ws + QS * j, k_)
                if (more) FA_ROWC_STORE((k_ + NB - 1) % NB)
                if (more && NB > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW * (NB - 2)) : "memory");
                else lds_dma_wait_all();
                __syncthreads();
            }
        });
    }
    });   // (split group ws)
#undef FA_ITEM
#undef FA_DMA_QG
#undef FA_DMA_LANE_OFFSETS
#undef FA_ROWC_LOAD
#undef FA_ROWC_STORE
#undef RC_W0
#undef RC_W1
#undef RC_IDX
#undef FA_DKDV_TILE
#undef FA_DMA_MASK
#undef FA_DMA_MHALF

    // query-sweep split groups: fold the other groups' dK^T and dV^T partial sums into group 0 through LDS - one group and one of the two
    // gradients at a time, in that order (every buffer is free after the loop's last barrier; a lane's values at [value][lane])
    if constexpr (QS > 1) {
        float* xb = (float*)smem_all + wave * (DT * 16 * 64) + lane;
        static_assert(NWAVES * DT * 16 * 64 * 4 <= QS * GROUP_B, "exchange area must fit the Q / dO buffers");
        for (int sg = 1; sg < QS; ++sg) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                if (sg > 1 || which > 0) __syncthreads();           // (group 0 has read the previous values)
                if (ws == sg) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) xb[(dt * 16 + e) * 64] = which == 0 ? dk[dt][e] : dv[dt][e];
                }
                __syncthreads();
                if (ws == 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            if (which == 0) dk[dt][e] += xb[(dt * 16 + e) * 64];
                            else dv[dt][e] += xb[(dt * 16 + e) * 64];
                        }
                }
            }
        }
        if (ws != 0) return;
    }
    if constexpr (QH == 2) {      // row halves: group 1's dK^T, then its dV^T, into group 0's through LDS (every buffer is free after the loop's last barrier)
        float* xb = (float*)smem_all + wave * (DT * 16 * 64) + lane;
        static_assert(QH == 1 || NWAVES * DT * 16 * 64 * 4 <= GROUP_B, "exchange area must fit the Q / dO buffers");
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            if (which > 0) __syncthreads();           // (group 0 has read dK's values)
            if (wh == 1) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xb[(dt * 16 + e) * 64] = which == 0 ? dk[dt][e] : dv[dt][e];
            }
            __syncthreads();
            if (wh == 0) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (which == 0) dk[dt][e] += xb[(dt * 16 + e) * 64];
                        else dv[dt][e] += xb[(dt * 16 + e) * 64];
                    }
            }
        }
        if (wh != 0) return;
    }
    if (!wave_active || key >= L) return;
    int key_e = key;                                // (the output rows' addresses are formed here, not carried through the loop)
    asm volatile("" : "+v"(key_e));
    T* __restrict__ dkg = (T*)p.dk + (int64_t)bb * p.dk_sb + (int64_t)hh * p.dk_sh + (int64_t)key_e * p.dk_sl;
    T* __restrict__ dvg = (T*)p.dv + (int64_t)bb * p.dv_sb + (int64_t)hh * p.dv_sh + (int64_t)key_e * p.dv_sl;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            vec4 wk, wv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wk[j] = (T)(dk[dt][4 * rg + j] * p.scale);
                wv[j] = (T)dv[dt][4 * rg + j];
            }
            if (4 * dt + rg < dchunks) {
                if constexpr (PASS != 1) *(vec4*)(dkg + 32 * dt + 8 * rg + 4 * h) = wk;
                if constexpr (PASS != 2) *(vec4*)(dvg + 32 * dt + 8 * rg + 4 * h) = wv;
            }
        }
    }
}

#ifndef FA_BWD_BODY_ONLY
// (head_dim 128: 254 + 4 registers as allocated freely - held to 256 = two waves per SIMD)
template <typename T, int D, int NWAVES, int QS = 1, int QH = 1>
__global__ __launch_bounds__(NWAVES * QS * QH * 64, D == 128 ? 2 : QS == 3 ? 3 : 1) void fa_bwd_dkdv_kernel(const BwdParams p) {
    fa_bwd_dkdv_body<T, D, NWAVES, 0, 0, QS, QH>(p);
}

// ---------------------------------------------------------------------------------------------
// host-side launcher: dQ (which also publishes Dsum), then dK/dV on the same stream (4-wave workgroups)
// ---------------------------------------------------------------------------------------------
// (p.qs_q / p.qs_kv: split groups of the two grids - 1, or for grids of at most one workgroup per CU 3 at head_dim 64 and 2 at head_dim 128:
//  what the kernels' registers leave room for - fa_capi.cpp bwd_impl)
template <typename T, int D>
static hipError_t launch_bwd_t(const BwdParams& p, hipStream_t stream) {
    const dim3 qgrid((unsigned)((int64_t)p.G8 * p.nq_tiles)), kgrid((unsigned)((int64_t)p.G8kv * p.nk_tiles));
 hipError_t e;
    if constexpr (D == 64 || D == 128) {
        if (p.bal) {   // balanced dQ units (fa_bwd_dq_body BAL), the reduce pass, then dK/dV as always
            if constexpr (D == 64)
                e = p.qs_q == 3   ? launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 3, 1>), qgrid, dim3(768), 0, stream, p)
                    : p.qs_q == 2 ? launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 2, 1>), qgrid, dim3(512), 0, stream, p)
                                  : launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 1, 1>), qgrid, dim3(256), 0, stream, p);
            else
                e = p.qs_q == 2 ? launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 2, 1>), qgrid, dim3(512), 0, stream, p)
                                : launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 1, 1>), qgrid, dim3(256), 0, stream, p);
            if (e != hipSuccess) return e;
            const int nq_true = (p.L + 127) / 128, row0 = (nq_true / 2) * 128, Gt = p.G / p.xs_q;
            const int64_t n = (int64_t)Gt * (p.L - row0) * (D / 4);
            if (n > 0) {
                void* args[] = {(void*)&p, (void*)&row0, (void*)&Gt};
                e = hipLaunchKernel((const void*)(fa_bwd_dq_reduce_kernel<T, D>), dim3((unsigned)((n + 255) / 256)), dim3(256), args, 0, stream);
                if (e != hipSuccess) return e;
            }
            if constexpr (D == 64)
                return p.qs_kv == 3   ? launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4, 3>), kgrid, dim3(768), 0, stream, p)
                       : p.qs_kv == 2 ? launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4, 2>), kgrid, dim3(512), 0, stream, p)
                                      : launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4>), kgrid, dim3(256), 0, stream, p);
            else
                return p.qs_kv == 2 ? launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4, 1, 2>), kgrid, dim3(512), 0, stream, p)
                                    : launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4>), kgrid, dim3(256), 0, stream, p);
        }
    }
    if constexpr (D == 64) {
        e = p.qs_q == 3   ? launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 3>), qgrid, dim3(768), 0, stream, p)
            : p.qs_q == 2 ? launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 2>), qgrid, dim3(512), 0, stream, p)
                          : launch1((const void*)(fa_bwd_dq_kernel<T, D, 4>), qgrid, dim3(256), 0, stream, p);
        if (e != hipSuccess) return e;
        return p.qs_kv == 3   ? launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4, 3>), kgrid, dim3(768), 0, stream, p)
               : p.qs_kv == 2 ? launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4, 2>), kgrid, dim3(512), 0, stream, p)
                              : launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4>), kgrid, dim3(256), 0, stream, p);
    } else if constexpr (D == 128) {
        e = p.qs_q == 2 ? launch1((const void*)(fa_bwd_dq_kernel<T, D, 4, 2>), qgrid, dim3(512), 0, stream, p)
                        : launch1((const void*)(fa_bwd_dq_kernel<T, D, 4>), qgrid, dim3(256), 0, stream, p);
        if (e != hipSuccess) return e;
        // (head_dim 128: the single sweep fills its 256 registers, so a second group cannot run a sweep of its own - it takes the other ROW HALF of
        //  every item instead: QH = 2, the same LDS buffers, two waves per SIMD)
        return p.qs_kv == 2 ? launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4, 1, 2>), kgrid, dim3(512), 0, stream, p)
                            : launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4>), kgrid, dim3(256), 0, stream, p);
    } else {
        e = launch1((const void*)(fa_bwd_dq_kernel<T, D, 4>), qgrid, dim3(256), 0, stream, p);
        if (e == hipSuccess) e = launch1((const void*)(fa_bwd_dkdv_kernel<T, D, 4>), kgrid, dim3(256), 0, stream, p);
        return e;
    }
}

hipError_t launch_bwd(const BwdParams& p, int dtype, hipStream_t stream) {
    if (p.D == 192) return dtype == 1 ? launch_bwd_t<__bf16, 192>(p, stream) : launch_bwd_t<_Float16, 192>(p, stream);   // (and 136 .. 184 zero-extended)
    if (p.D == 256) return dtype == 1 ? launch_bwd_t<__bf16, 256>(p, stream) : launch_bwd_t<_Float16, 256>(p, stream);   // (and 200 .. 248 zero-extended)
    if (dtype == 1) return p.D == 64 ? launch_bwd_t<__bf16, 64>(p, stream) : launch_bwd_t<__bf16, 128>(p, stream);
    return p.D == 64 ? launch_bwd_t<_Float16, 64>(p, stream) : launch_bwd_t<_Float16, 128>(p, stream);
}

#endif  // FA_BWD_BODY_ONLY

}  // namespace fa
