// FlashAttention forward for MI355X (gfx950) with LDS-DMA staging - 4 waves per SIMD at head_dim 64.
//
// Same tiling, LDS images, MFMA operand maps and arithmetic as fa_fwd_tile_kernel
// (fa_fwd_kernels.hip - read its header first).  What changes is how K/V tiles reach LDS:
//
//   * `buffer_load_dwordx4 ... lds` (LDS-DMA): global -> LDS with no VGPR destination, no ds_write
//     and no per-tile address arithmetic.  One wave-instruction moves a 1 KiB piece (64 lanes x 16 B,
//     lane-linear in LDS); the K swizzle and the V [d/32][key/8][8][32] image are produced by
//     choosing each lane's SOURCE address (the LDS destination cannot scatter).
//   * the per-tile advance lives in the buffer descriptor (scalar adds), whose byte limit also
//     zero-fills key rows past L - no clamping code;
//   * the kv loop is unrolled by two, so every LDS address is base + immediate.
//
// That frees ~22 VGPRs (no staging registers, no staging addresses): the D=64 kernel fits 128 VGPRs
// = 4 waves per SIMD (4 workgroups of 4 waves per CU), and removes ~25 VALU instructions per tile
// from the VALU issue port, which is the binding resource of this kernel at head_dim 64.

#include "fa_device.h"
#include "fa_extras.h"

// 1: no per-tile row max (bf16).  The reference max m is set exactly on a wave's first tile and then left
// alone; a tile whose row-sum part exceeds P_LIMIT (some P outgrew the safe range) is redone with an
// exact max.  Any reference value cancels in O/l; it only has to keep P finite.  Worth +7 % at D=64:
// the 16-deep v_max3 chain, the half-wave swap and the rescale decision leave every tile.
#ifndef FA_DMA_SKIPMAX
#define FA_DMA_SKIPMAX 1
#endif

namespace fa {

// K tile rows in LDS: chunk position cp of a row holds the row's 16-byte chunk cp ^ k_swizzle(row) - an XOR inside aligned groups of
// 8 (head_dim 64: rows are 8 chunks; head_dim 192: 24 = three groups of 8) or 16 chunks (128, 256)
template <int D>
__device__ __forceinline__ int k_swizzle(int row) {
    return D == 64 ? ((row >> 1) & 7) : (D % 128 == 0) ? (row & 15) : (row & 7);
}

// (the body lives in a __device__ function: the buffer/LDS-DMA builtins do not exist for the host pass,
// and a __global__ template that names them loses its host-side launch stub)
//
// SPLIT > 1 (split-KV inside the workgroup, for grids that cannot fill the chip): SPLIT groups of NWAVES waves own the SAME
// query rows; group s streams the kv tiles s, s + SPLIT, ... through LDS buffers of its own, in lock step with the others
// (one workgroup barrier per tile round), and the groups' (m, l, O) are merged through LDS before the epilogue - the kv
// loop of a q tile, the critical path of a launch with fewer workgroups than slots, shrinks SPLIT-fold with no second
// kernel and no partial results in HBM.
//
// PAIR (causal, SPLIT == 1, QB == 1): the workgroup owns TWO q tiles, (nq - 1 - i) then i, one after the other.  Every workgroup
// of a group then streams the same number of kv tiles (nq + 1 of 128-row q tiles: no light and heavy workgroups, no tail), the
// dispatcher refills half as many slots, and the second tile's prologue disappears: its Q rows and its first K/V tile are requested
// right after the first tile's loop, into the registers / LDS buffers that loop has just released, and land under the first tile's
// epilogue (whose row slabs move to the K1 / V1 buffers for that).  Short sequences, where these fixed phases are 25-40 % of a
// workgroup's life, are what it is for.
//
// EXTRA: attention mask / additive bias and dropout (fa_extras.h), exact-max arithmetic only (a mask can hide every key of a
// tile from a row, so the fixed reference max of the skip-max mode has nothing to stand on); instantiated in fa_fwd_extra.hip.
// MM (EXTRA only) - how a mask reaches the scores: 0 = none (dropout alone) or a row-broadcast mask (row stride 0: its 64 per-key terms of
// a tile are decoded by 64 threads one tile ahead and staged through LDS); 2 | 3 = a boolean mask | a bias that differs from row to row:
// the tile's mask image (128 rows x 64 keys) travels to LDS by DMA with the K / V tile.
template <typename T, int D, int NWAVES, bool SKIP, int QB, int SPLIT, bool PAIR, bool EXTRA = false, int MM = 0>
__device__ __forceinline__ void fa_fwd_dma_body(const FwdParams& p) {
    static_assert(!EXTRA || !SKIP, "mask / dropout launches use the exact-max arithmetic");
    using E = Elt<T>;
    using vec8 = typename E::vec8;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int WQ = 32 * QB;                 // query rows per wave: QB blocks of 32 sharing every K/V operand read
    constexpr int BQ = NWAVES * WQ;
    constexpr int BK = 64;
    constexpr int ROWB = D * 2;
    constexpr int KTILE_B = BK * ROWB;
    constexpr int VBLK_B = BK * 64 + 64;
    constexpr int VTILE_B = (D / 32) * VBLK_B;
    constexpr int PIECES = KTILE_B / 1024;      // 1 KiB DMA pieces per K (or V) tile
    constexpr int PPW = PIECES / NWAVES;        // pieces per wave
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
    static_assert(PIECES % NWAVES == 0 && PPW >= 1, "pieces must split evenly over the waves");

    constexpr int GROUP_B = 2 * KTILE_B + 2 * VTILE_B;                                      // K0 K1 V0 V1 of one group
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[SPLIT * GROUP_B];
    __shared__ int flags[NWAVES * SPLIT];

    static_assert(!PAIR || (SPLIT == 1 && QB == 1), "q-tile pairs: unsplit workgroups with one query block per wave");
    const int nq = p.nq_tiles;
    const int nslot = PAIR ? (nq + 1) >> 1 : nq;      // workgroups per group
    int g, qd;
    {
        int b = blockIdx.x;
        if (p.perm_n > 0) {                            // placement-balanced order of a grid below one round of slots (fa_capi.cpp)
            const int s_ = b >> 3;
            b = (s_ < p.perm_n ? (int)p.perm[s_] : s_) * 8 + (b & 7);
        }
        const int xcd = b & 7, slot = b >> 3;
        const int ng = p.G8 >> 3;
        const int per_bunch = p.sched * nslot;
        const int bunch = slot / per_bunch;
        const int rem = slot - bunch * per_bunch;
        const int g0 = bunch * p.sched;
        const int bc = (ng - g0) < p.sched ? (ng - g0) : p.sched;
        qd = rem / bc;
        g = (g0 + (rem - qd * bc)) * 8 + xcd;
    }
    if (g >= p.G) return;
    if (p.xs > 1) {   // q-tile subsets (fa_params.h FwdParams::xs): g counts (group, subset) pairs; depth d of subset s is the group's depth d xs + s
        const int s_ = g % p.xs;
        g /= p.xs;
        qd = qd * p.xs + s_;
        if (qd >= p.nq_real) return;
    }
    const int nq_r = p.xs > 1 ? p.nq_real : nq;      // q tiles per (real) group
    const int n_pass = (PAIR && qd != nq_r - 1 - qd) ? 2 : 1;
#ifdef FA_STAMPS   /* measurement build only (scripts/phase_times.py): phase time stamps, written to p.stamps at the end */
    const unsigned long long st0_ = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
    unsigned long long st1_ = 0, st2_ = 0;
#define FA_STAMP(V_) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); V_ = __builtin_amdgcn_s_memtime(); }
#else
#define FA_STAMP(V_) {}
#endif

    const int L = p.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base -> M0)
    const int ws = SPLIT == 1 ? 0 : wave_all / NWAVES;               // kv split group of this wave
    const int wave = SPLIT == 1 ? wave_all : wave_all - ws * NWAVES; // its row block inside the group
    constexpr int tstride = SPLIT;                                   // a wave group streams the kv tiles ws, ws + SPLIT, ...
    const int tfirst = ws;
    unsigned char* const smem = smem_all + ws * GROUP_B;
    unsigned char* const smem_v = smem + 2 * KTILE_B;
    const int r = lane & 31, h = lane >> 5;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* __restrict__ qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh;
    const T* __restrict__ kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;
    const T* __restrict__ vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;
    // EXTRA: this group's mask rows (byte addressed) and dropout key
    const int mask_esz = p.ex.mask_kind == 2 ? (int)sizeof(T) : 1;
    const char* const mask_g = EXTRA && p.ex.mask ? (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)hh * p.ex.m_sh) * mask_esz : nullptr;
    const unsigned drop_kg = EXTRA ? drop_group_key(drop_key0_of(p.ex), (unsigned)g) : 0u;
    // EXTRA, MM 2 | 3 (a mask that differs from row to row): every lane walks its own mask row, so a 4- or 8-byte load per lane touches 32
    // cache lines per instruction, eight instructions per tile and wave.  The tile's mask image travels to LDS with the K and V tiles
    // instead: LDS-DMA, 16 bytes per lane, whole rows (2 / 4 instructions per wave and tile, one tile ahead); a lane then reads the 4
    // keys of a (half, j) with one ds_read_b32 / _b64.  16-byte chunk c of row i sits at chunk position c ^ (i >> 2 & 3) (bias:
    // c ^ (i >> 1 & 7)): the 32 rows of a wave read the same chunk, which a row-major image keeps on 4 (2) bank groups.  Needs
    // ExtraParams::mask_vec bit 1; any other mask is loaded in place.  The mask kind is a template argument: decided per element
    // at run time it costs a branch and a copy of the score tile per four keys.
    // MM 0 with a mask - a row-broadcast one (fa_fwd_extra.hip sends every other mask to MM 2 | 3): threads 0..63 fetch and decode the 64
    // per-key terms of a tile one tile ahead (requested BEFORE that tile's DMA - vmcnt retires in order - and written to LDS before the
    // barrier that publishes the tile); the tile body adds them with broadcast ds_read_b128.  A tile whose 64 keys are all masked (a
    // padded tail) is skipped by the whole workgroup.
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    constexpr int MIMG_B = MM == 3 ? 16384 : 8192, MT_F = 80;
    // (head_dim 256: K / V fill 132 KB - two 16 KB bias images do not fit beside them.  There every wave keeps the bias of ITS 32 rows only, in half-tile
    //  pieces - 32 rows x 32 keys x 2 B = 2 KB - that travel through a wave-private ring of three: half k of the (tile, key half) sequence lives in slot
    //  k % 3; the first half of tile t + 1 is requested with that tile's K / V, its second half as soon as the wave has read tile t's bias (it takes the
    //  slot of tile t's first half): 24 KB in all, and no barrier of its own - a wave is the only writer and the only reader of its ring.)
    constexpr bool MPRIV = EXTRA && D == 256 && MM == 3;
    constexpr int MHALF_B = 2048, MRING_B = 3 * MHALF_B;
    __shared__ __attribute__((aligned(16))) unsigned char smem_mk[EXTRA ? (MM >= 2 ? (MPRIV ? NWAVES * MRING_B : 2 * MIMG_B) : 2 * MT_F * 4) : 16];
    float* const mterm = (float*)smem_mk;
    constexpr bool mbias = MM == 3;
    const bool mdma = EXTRA && MM >= 2 && mask_g != nullptr && (p.ex.mask_vec & 2) != 0;
    const bool mbc = EXTRA && MM == 0 && mask_g != nullptr;
    static_assert(!EXTRA || (QB == 1 && SPLIT == 1 && NWAVES == 4), "mask staging: one 32-row block per wave, unsplit 4-wave workgroups");
    unsigned m_vo[(EXTRA && MM >= 2) ? 4 : 1];
    int m_rd = 0;
    float mt_stage = 0.f;
    (void)mterm; (void)mdma; (void)mbc; (void)m_vo; (void)m_rd; (void)mt_stage;
#define FA_MASK_GEOM(QI_)                                                                          \
    if constexpr (MPRIV) {   /* piece i of a half = rows 16 i .. + 15 of the wave (64 B a row, lane-linear in LDS): 16-byte chunk c of row-slot s sits at \
                                chunk position c ^ (s >> 2) - the 16 rows one read instruction covers per half-wave then spread over all banks */ \
        const unsigned m_slb_ = (unsigned)p.ex.m_sl * (unsigned)mask_esz;                          \
        const int rl_ = L - 1 - (QI_) * BQ;          /* (rows past L: the last row again - they are never stored) */ \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                         \
            const int row = wave * 32 + 16 * i_ + (lane >> 2);                                     \
            m_vo[i_] = (unsigned)(row < rl_ ? row : rl_) * m_slb_ + ((unsigned)((lane & 3) ^ (lane >> 4)) << 4); \
        }                                                                                          \
        m_vo[2] = m_vo[3] = 0u;                                                                    \
        m_rd = wave * MRING_B + r * 64 + (((r >> 2) & 3) << 4) + 8 * h;                            \
    } else if constexpr (EXTRA && MM >= 2) {                                                       \
        const unsigned m_slb_ = (unsigned)p.ex.m_sl * (unsigned)mask_esz;                          \
        const int rl_ = L - 1 - (QI_) * BQ;          /* (rows past L: the last row again - they are never stored) */ \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
            const int pc = wave + 4 * j;                                                           \
            const int row = mbias ? 8 * pc + (lane >> 3) : 16 * pc + (lane >> 2);                  \
            const int cpos = mbias ? (lane & 7) : (lane & 3);                                      \
            m_vo[j] = (unsigned)(row < rl_ ? row : rl_) * m_slb_ + ((unsigned)(cpos ^ (mbias ? (row >> 1) & 7 : (row >> 2) & 3)) << 4); \
        }                                                                                          \
        const int il = wave * 32 + r;                                                              \
        m_rd = mbias ? il * 128 + (((il >> 1) & 7) << 4) + 8 * h : il * 64 + (((il >> 2) & 3) << 4) + 4 * h; \
    }
    // FA_MASK_EXTENT: a mask tile's descriptor ends with the slice's last row - rounded UP to whole dwords: the range check of a buffer load is per
    // dword, and with L odd (a bias; L % 4 != 0 for a boolean mask) the last key of the last row shares its dword with the element after it - cut
    // exactly at L that dword read as zero (bias[L-1][L-1] lost, the last row's last keys masked out: found by tests/tools/fuzz_extra.py on a strided
    // view).  Pointer and strides are multiples of four elements here (mask_vec), so the dword is aligned and belongs to the allocation; keys >= L are
    // masked by the ragged-tile path.
    // MPRIV: half HALF_ (keys 32 HALF_ .. + 31) of tile T_ -> ring slot (2 T_ + HALF_) % 3; the wave's own reads of the slot's tenant are drained first
#define FA_DMA_MHALF(QI_, T_, HALF_)                                                               \
    if constexpr (MPRIV) {                                                                         \
        if (mdma && wave_active) {                                                                 \
            const int c0_ = (T_) * BK + 32 * (HALF_);                                              \
            const char* mb_ = mask_g + ((int64_t)((QI_) * BQ) * p.ex.m_sl + c0_) * mask_esz;       \
            const int64_t left_ = (((int64_t)(L - 1 - (QI_) * BQ) * p.ex.m_sl + (L - c0_)) * mask_esz + 3) & ~3ll;   /* to the end of the slice's last row in whole dwords, FA_MASK_EXTENT (<= 0: a half past L - zeros) */ \
            i32x4 mr_ = make_raw_rsrc(mb_, left_ <= 0 ? 0u : left_ < 0xffffffffll ? (unsigned)left_ : 0xffffffffu); \
            _Pragma("unroll") for (int w_ = 0; w_ < 3; ++w_) mr_[w_] = __builtin_amdgcn_readfirstlane(mr_[w_]);   \
            const unsigned dst_ = lds_offset(smem_mk) + wave * MRING_B + ((2 * (T_) + (HALF_)) % 3) * MHALF_B; \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
            lds_dma_b128(mr_, dst_, m_vo[0]);                                                      \
            lds_dma_b128(mr_, dst_ + 1024, m_vo[1]);                                               \
        }                                                                                          \
    }
#define FA_DMA_MASKTILE(QI_, T_, BUF_)                                                             \
    if constexpr (MPRIV) {   /* the tile's first half; its second half follows the wave's reads of tile T_ - 1 (FA_COMPUTE) */ \
        FA_DMA_MHALF(QI_, T_, 0)                                                                   \
    } else if constexpr (EXTRA && MM >= 2) {                                                       \
        if (mdma) {                                                                                \
            const char* mb_ = mask_g + ((int64_t)((QI_) * BQ) * p.ex.m_sl + (T_) * BK) * mask_esz; \
            const int64_t left_ = (((int64_t)(L - 1 - (QI_) * BQ) * p.ex.m_sl + (L - (T_) * BK)) * mask_esz + 3) & ~3ll;   /* to the end of the slice's last row (whole dwords) */ \
            i32x4 mr_ = make_raw_rsrc(mb_, left_ < 0xffffffffll ? (unsigned)left_ : 0xffffffffu);   \
            _Pragma("unroll") for (int w_ = 0; w_ < 3; ++w_) mr_[w_] = __builtin_amdgcn_readfirstlane(mr_[w_]);   /* (64-bit products leave the scalar unit) */ \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                          \
                if (mbias || j < 2) lds_dma_b128(mr_, lds_offset(smem_mk) + (BUF_) * MIMG_B + (wave + 4 * j) * 1024, m_vo[j]); \
        }                                                                                          \
    }
#define FA_MTERM_LOAD(T_)                                                                          \
    if constexpr (EXTRA && MM == 0) {                                                              \
        if (mbc && tid < 64) {                                                                     \
            const int k_ = (T_) * BK + tid;                                                        \
            mt_stage = mask_load1<T>(p.ex.mask_kind, mask_g, 0, k_ < L ? k_ : L - 1, p.ex.inv_scale); \
        }                                                                                          \
    }
#define FA_MTERM_STORE(BUF_)                                                                       \
    if constexpr (EXTRA && MM == 0) {                                                              \
        if (mbc && tid < 64) {                                                                     \
            mterm[(BUF_) * MT_F + tid] = mt_stage;                                                 \
            const bool dead_ = __builtin_amdgcn_ballot_w64(mt_stage != -INFINITY) == 0;            \
            if (tid == 0) mterm[(BUF_) * MT_F + 64] = dead_ ? 1.f : 0.f;                           \
        }                                                                                          \
    }

    // the operands' true head dim in 16-byte chunks (= D / 8 unless the call runs zero-extended: FwdParams::Dr)
    const int dchunks = p.Dr >> 3;
    // Q^T fragments of q tile QI_ (B operands of S^T = K Q^T): block b of this wave owns row QI_ * BQ + wave * WQ + 32 b + r
    vec8 qf[QB][KS];
#define FA_LOAD_Q(QI_)                                                                             \
    _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                               \
        const int qr0_ = (QI_) * BQ + wave * WQ + r + 32 * b;                                      \
        const T* qp = qg + (int64_t)(qr0_ < L ? qr0_ : L - 1) * p.q_sl;                            \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                        \
            const int ch_ = 2 * ks + h;            /* 16-byte chunk of the row: columns 8 ch_ ..+7 */ \
            const vec8 ld_ = *(const vec8*)(qp + 8 * (ch_ < dchunks ? ch_ : 0));                   \
            vec8 z_;                                                                               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) z_[j] = (T)0.f;                          \
            qf[b][ks] = ch_ < dchunks ? ld_ : z_;  /* (columns past the true head dim: zeros) */   \
        }                                                                                          \
    }
    FA_LOAD_Q(nq_r - 1 - qd)

    // ---- LDS-DMA geometry: piece pc = wave + j*NWAVES of a tile, lane -> source byte offset
    const unsigned k_slb = (unsigned)p.k_sl * 2, v_slb = (unsigned)p.v_sl * 2;   // row strides in bytes
    unsigned k_vo[PPW], v_vo[PPW];
    int k_dst[PPW], v_dst[PPW];    // wave-uniform LDS byte offsets of the pieces inside a tile buffer
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pc = wave + j * NWAVES;
        // K: lane-linear piece = 1024/ROWB rows; LDS chunk position cp of row holds global chunk cp ^ swz(row)
        // (head_dim 192: a row is 24 chunks, a piece 2 2/3 rows - the piece's 64 chunks are numbered through the tile)
        const int row = (pc * 64 + lane) / (ROWB / 16);
        const int cp = (pc * 64 + lane) % (ROWB / 16);
        const int sw = k_swizzle<D>(row);
        // (a chunk past the true head dim: an offset beyond any descriptor's extent - the DMA writes zeros for it, as for rows past L)
        k_vo[j] = (cp ^ sw) < dchunks ? (unsigned)row * k_slb + ((cp ^ sw) << 4) : 0x80000000u;
        k_dst[j] = pc * 1024;
        // V: piece = two 8-key blocks of one 32-column d-block
        const int dblk = pc >> 2, q2 = pc & 3;
        const int key = (2 * q2 + (lane >> 5)) * 8 + ((lane >> 2) & 7);
        v_vo[j] = (dblk * 4 + (lane & 3)) < dchunks ? (unsigned)key * v_slb + ((dblk * 4 + (lane & 3)) << 4) : 0x80000000u;
        v_dst[j] = dblk * VBLK_B + q2 * 1024;
    }
    const unsigned k_bytes = ((unsigned)(L - 1) * (unsigned)p.k_sl + (unsigned)p.Dr) * 2;   // extent of this group's K
    const unsigned v_bytes = ((unsigned)(L - 1) * (unsigned)p.v_sl + (unsigned)p.Dr) * 2;

    // tile T_ -> buffer BUF_: descriptors start at the tile's first row; rows past L read as zeros
#define FA_DMA_TILE(T_, BUF_)                                                                      \
    {                                                                                              \
        const unsigned ko_ = (unsigned)(T_) * BK * k_slb, vo_ = (unsigned)(T_) * BK * v_slb;       \
        const i32x4 kr_ = make_raw_rsrc((const char*)kg + ko_, k_bytes - ko_);                     \
        const i32x4 vr_ = make_raw_rsrc((const char*)vg + vo_, v_bytes - vo_);                     \
        if (FA_DMA_ABL != 2) {                                                                     \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) {                                          \
            lds_dma_b128(kr_, smem_lds + (BUF_) * KTILE_B + k_dst[j], k_vo[j]);                    \
            lds_dma_b128(vr_, smem_lds + 2 * KTILE_B + (BUF_) * VTILE_B + v_dst[j], v_vo[j]);      \
        }                                                                                          \
        }                                                                                          \
    }
    // publish a DMA'd tile: drain this wave's DMA, then the workgroup barrier
#ifndef FA_DMA_ABL
#define FA_DMA_ABL 0    /* timing-only ablations (wrong results): 2 no K/V DMA, 3 no softmax, 4 no PV, 5 no QK, 6 no O stores */
#endif
#if defined(FA_DMA_ABL_NOBAR)   /* timing-only ablation: wrong results */
#define FA_TILE_BARRIER() { lds_dma_wait_all(); }
#else
#define FA_TILE_BARRIER() { lds_dma_wait_all(); __syncthreads(); }
#endif
    const unsigned smem_lds = lds_offset(smem);

    int k_rd[KS];
    {
        const int sw = k_swizzle<D>(r);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * ROWB + (((2 * ks + h) ^ sw) << 4);
    }
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    const bool causal = p.causal != 0;
    f32x16 o[QB][DT];
    float m[QB];
    // Softmax denominators are formed on the matrix pipe (fa_device.h, rowsum_selector): lsum[b] holds the FULL row
    // sum of query block b's rows (both half-waves), identical in its four registers.
    f32x4 lsum[QB];
    const vec8 asel = rowsum_selector<T>(lane);
#define FA_RESET_STATE()                                                                           \
    _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                               \
        m[b] = -INFINITY;                                                                          \
        lsum[b] = f32x4{0.f, 0.f, 0.f, 0.f};                                                       \
        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                          \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) o[b][dt][e] = 0.f;                      \
    }
    const float c = p.scale_log2;
    // Two arithmetic modes per kv tile:
    //   fast (MODE_ 0, "skip-max"): the reference max m is set exactly on the wave's first tile and then left alone -
    //     any reference value cancels in O/l, it only has to keep P finite.  bf16 P keeps the fp32 exponent range, so
    //     the max may lag by ~2^100 before anything overflows; f16 P must stay below 65504, which 1/sqrt(D)-scaled
    //     logits practically always satisfy (the host picks this mode for f16 from the softmax scale, fa_capi.cpp).
    //     Nothing is checked per tile: an overflowed P turns the row sum into inf/NaN for good, the pass is
    //     OPTIMISTIC, and if any wave of the workgroup finds such a sum at its end, the workgroup re-streams K/V
    //     through an exact pass.  Worth +7 % at D=64 when it replaced the per-tile max (the 16-deep v_max3 chain, the
    //     half-wave swap and the rescale decision leave every tile) and +2..4 % more when the row sums moved to the
    //     matrix pipe (32 v_add_f32 per tile and block replaced by 4 small MFMAs; no per-tile overflow test).
    //   exact (MODE_ 1): per-tile row max with lazy rescale (O, l rescaled only when a row max outgrows m by more
    //     than 2^RESCALE_THR) - unscaled f16, and the fallback pass of the fast mode.
    constexpr bool SKIPMAX = FA_DMA_SKIPMAX && SKIP;

    // ---- the pieces of one kv tile (buffer index BUF_ is compile-time); same arithmetic as fa_fwd_tile_kernel
#define FA_QK(BUF_)                                                                                \
    {                                                                                              \
        const unsigned char* kb_ = smem + (BUF_) * KTILE_B;                                        \
        _Pragma("unroll") for (int b = 0; b < QB; ++b)                                             \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) { s0[b][e] = 0.f; s1[b][e] = 0.f; }     \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                        \
            const vec8 k0 = *(const vec8*)(kb_ + k_rd[ks]);                                        \
            const vec8 k1 = *(const vec8*)(kb_ + 32 * ROWB + k_rd[ks]);                            \
            _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                       \
                s0[b] = E::mfma(k0, qf[b][ks], s0[b]);                                             \
                s1[b] = E::mfma(k1, qf[b][ks], s1[b]);                                             \
            }                                                                                      \
        }                                                                                          \
        if constexpr (EXTRA) {                                                                     \
            if (MM == 0 && mbc) {          /* the tile's per-key terms, staged one tile ago */     \
                typedef float f32x4_ __attribute__((ext_vector_type(4)));                          \
                const float* mt_ = mterm + (BUF_) * MT_F + 4 * h;                                  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                    \
                    const f32x4_ a0_ = *(const f32x4_*)(mt_ + 8 * j), a1_ = *(const f32x4_*)(mt_ + 32 + 8 * j); \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) { s0[0][4 * j + i] += a0_[i]; s1[0][4 * j + i] += a1_[i]; } \
                }                                                                                  \
            } else if (MPRIV && mdma) {   /* this lane's row of the wave's two halves (ring slots (2 tile), (2 tile + 1) % 3) */ \
                const unsigned char* mi0_ = smem_mk + ((2 * (kv0 / BK)) % 3) * MHALF_B;            \
                const unsigned char* mi1_ = smem_mk + ((2 * (kv0 / BK) + 1) % 3) * MHALF_B;        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                    \
                    const vec4 b0_ = __builtin_bit_cast(vec4, *(const u32x2*)(mi0_ + (m_rd ^ (j << 4)))); \
                    const vec4 b1_ = __builtin_bit_cast(vec4, *(const u32x2*)(mi1_ + (m_rd ^ (j << 4)))); \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
                        s0[0][4 * j + i] += bias_term((float)b0_[i], p.ex.inv_scale);              \
                        s1[0][4 * j + i] += bias_term((float)b1_[i], p.ex.inv_scale);              \
                    }                                                                              \
                }                                                                                  \
                if ((kv0 / BK) + tstride < n_tiles) {   /* the first half's slot is read: the next tile's second half takes it */ \
                    FA_DMA_MHALF(qi, kv0 / BK + tstride, 1)                                        \
                    mh_sent_ = true;                                                               \
                }                                                                                  \
            } else if (MM >= 2 && mdma) {   /* this lane's row of the tile image */                \
                const unsigned char* mi_ = smem_mk + (BUF_) * MIMG_B;                              \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                    \
                    if (mbias) {                                                                   \
                        const vec4 b0_ = __builtin_bit_cast(vec4, *(const u32x2*)(mi_ + (m_rd ^ (j << 4)))); \
                        const vec4 b1_ = __builtin_bit_cast(vec4, *(const u32x2*)(mi_ + (m_rd ^ ((4 + j) << 4)))); \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                            \
                            s0[0][4 * j + i] += bias_term((float)b0_[i], p.ex.inv_scale);          \
                            s1[0][4 * j + i] += bias_term((float)b1_[i], p.ex.inv_scale);          \
                        }                                                                          \
                    } else {                                                                       \
                        const unsigned w0_ = *(const unsigned*)(mi_ + (m_rd ^ ((j >> 1) << 4)) + 8 * (j & 1)); \
                        const unsigned w1_ = *(const unsigned*)(mi_ + (m_rd ^ ((2 + (j >> 1)) << 4)) + 8 * (j & 1)); \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                            \
                            if (((w0_ >> (8 * i)) & 0xffu) == 0) s0[0][4 * j + i] = -INFINITY;     \
                            if (((w1_ >> (8 * i)) & 0xffu) == 0) s1[0][4 * j + i] = -INFINITY;     \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
            } else if (mask_g != nullptr) {   /* additive terms on the raw scores: bias / scale, or 0 | -inf */ \
                const bool fast_ = (p.ex.mask_vec & 1) != 0 && kv0 + BK <= L;                      \
                _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                   \
                    const int mr_ = qrow + 32 * b < L ? qrow + 32 * b : L - 1;                     \
                    const char* mrow_ = mask_g + (int64_t)mr_ * p.ex.m_sl * mask_esz;              \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                \
                        float a0_[4], a1_[4];                                                      \
                        mask_load4<T>(p.ex.mask_kind, fast_, mrow_, kv0 + 8 * j + 4 * h, L, p.ex.inv_scale, a0_);      \
                        mask_load4<T>(p.ex.mask_kind, fast_, mrow_, kv0 + 32 + 8 * j + 4 * h, L, p.ex.inv_scale, a1_); \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i) { s0[b][4 * j + i] += a0_[i]; s1[b][4 * j + i] += a1_[i]; } \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        if (kv0 + BK - 1 > wave_last) { /* a diagonal tile of the wave / the ragged last tile */   \
            _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                       \
                const int klim = causal ? qrow + 32 * b : L - 1;                                   \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                   \
                    const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * h;                          \
                    if (key > klim) s0[b][e] = -INFINITY;                                          \
                    if (key + 32 > klim) s1[b][e] = -INFINITY;                                     \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
    // row max of the tile, then raise the reference max m (rescaling O and l) where it is outgrown by
    // more than THR_ (log2 units); decided BEFORE the tile's P is formed
#define FA_MAX_RESCALE(THR_)                                                                       \
    _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                               \
        float mx = max2(s0[b][0], s1[b][0]);                                                       \
        _Pragma("unroll") for (int e = 1; e < 16; ++e) mx = max3(mx, s0[b][e], s1[b][e]);          \
        mx = half_swap_max(mx);                                                                    \
        if (__builtin_amdgcn_ballot_w64((mx - m[b]) * c > (THR_)) != 0) {                          \
            const float m_new = max2(m[b], mx);                                                    \
            /* (EXTRA: a row that has met no admitted key yet keeps m = -inf; its O and l are 0 and stay 0) */ \
            const float alpha = (EXTRA && m_new == -INFINITY) ? 1.f : fast_exp2((m[b] - m_new) * c); \
            m[b] = m_new;                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) lsum[b][j] *= alpha;                     \
            _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                      \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) o[b][dt][e] *= alpha;               \
        }                                                                                          \
    }
    // P = 2^(s*c - m*c) in place
#define FA_SOFTMAX()                                                                               \
    _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                               \
        const float mc = (EXTRA && m[b] == -INFINITY) ? 0.f : m[b] * c;   /* (all scores of such a row are -inf: P = 0) */ \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                           \
            s0[b][e] = fast_exp2(__builtin_fmaf(s0[b][e], c, -mc));                                \
            s1[b][e] = fast_exp2(__builtin_fmaf(s1[b][e], c, -mc));                                \
        }                                                                                          \
    }
    // pack P to 16 bits, row sums += selector x P (matrix pipe), O^T += V^T P^T
#define FA_PV(BUF_)                                                                                \
    {                                                                                              \
        const unsigned char* vb_ = smem_v + (BUF_) * VTILE_B;                                      \
        vec8 pf[QB][4];                                                                            \
        _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
                pf[b][0][j] = (T)s0[b][j];                                                         \
                pf[b][1][j] = (T)s0[b][8 + j];                                                     \
                pf[b][2][j] = (T)s1[b][j];                                                         \
                pf[b][3][j] = (T)s1[b][8 + j];                                                     \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int b = 0; b < QB; ++b)                                             \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) lsum[b] = E::mfma16(asel, pf[b][kk], lsum[b]); \
        if constexpr (EXTRA) {                                                                     \
            if (p.ex.drop_thr != 0u) {   /* the denominator keeps every P; the PV product sees the kept ones */ \
                _Pragma("unroll") for (int b = 0; b < QB; ++b) {                                   \
                    const unsigned row_ = (unsigned)(qrow + 32 * b);                               \
                    const unsigned base_ = drop_tile_key(drop_kg, row_, (unsigned)kv0) ^ ((row_ & 0xffffu) << 16) ^ (unsigned)((kv0 & 0xffff) | (4 * h)); \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                               \
                        const unsigned ce_ = (unsigned)((e & 3) + 8 * (e >> 2));                   \
                        if (mix32(base_ ^ ce_) < p.ex.drop_thr) s0[b][e] = 0.f;                    \
                        if (mix32(base_ ^ (ce_ + 32u)) < p.ex.drop_thr) s1[b][e] = 0.f;            \
                    }                                                                              \
                    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                \
                        pf[b][0][j] = (T)s0[b][j];                                                 \
                        pf[b][1][j] = (T)s0[b][8 + j];                                             \
                        pf[b][2][j] = (T)s1[b][j];                                                 \
                        pf[b][3][j] = (T)s1[b][8 + j];                                             \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                        \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                     \
                const unsigned char* vp = vb_ + dt * VBLK_B + kk * 1024 + v_rd;                    \
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                          \
                    (i16x4 __attribute__((address_space(3)))*)(vp));                               \
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                          \
                    (i16x4 __attribute__((address_space(3)))*)(vp + 512));                         \
                const i16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);          \
                _Pragma("unroll") for (int b = 0; b < QB; ++b)                                     \
                    o[b][dt] = E::mfma(__builtin_bit_cast(vec8, v8), pf[b][kk], o[b][dt]);         \
            }                                                                                      \
        }                                                                                          \
    }
#define FA_TILE_DEAD(BUF_) (EXTRA && MM == 0 && mbc && __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mterm[(BUF_) * MT_F + 64])) != 0)
#define FA_COMPUTE(T_, BUF_, MODE_)                                                                \
    {                                                                                              \
        const int kv0 = (T_) * BK;                                                                 \
        bool mh_sent_ = false;   /* (MPRIV) the next tile's second bias half is on its way */       \
        (void)mh_sent_;                                                                            \
        if (wave_active && (SPLIT == 1 || (T_) < n_tiles) && kv0 <= wave_kv_last && !FA_TILE_DEAD(BUF_)) { \
            f32x16 s0[QB], s1[QB];                                                                 \
            if (FA_DMA_ABL != 5) { FA_QK(BUF_) } else {                                            \
                _Pragma("unroll") for (int b = 0; b < QB; ++b)                                     \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) { s0[b][e] = o[b][0][e]; s1[b][e] = o[b][1][e]; } \
            }                                                                                      \
            if ((MODE_) == 0) {                                                                    \
                if ((T_) == tfirst) FA_MAX_RESCALE(0.0f)   /* the group's first tile: m = -inf -> exact max */ \
            } else {                                                                               \
                FA_MAX_RESCALE(RESCALE_THR)                                                        \
            }                                                                                      \
            if (FA_DMA_ABL != 3) FA_SOFTMAX()                                                      \
            if (FA_DMA_ABL != 4) { FA_PV(BUF_) } else {                                            \
                _Pragma("unroll") for (int b = 0; b < QB; ++b)                                     \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) { o[b][0][e] += s0[b][e]; o[b][1][e] += s1[b][e]; } \
            }                                                                                      \
        }                                                                                          \
        if constexpr (MPRIV) {   /* (a wave that skipped this tile sends the next tile's second half here) */ \
            if (!mh_sent_ && (T_) + tstride < n_tiles) { FA_DMA_MHALF(qi, (T_) + tstride, 1) }     \
        }                                                                                          \
    }
    // one pass over the workgroup's kv tiles (the kv loop is unrolled by two: LDS addresses are base + immediate)
#define FA_PASS(MODE_)                                                                             \
    {                                                                                              \
        if (!tile0_requested && (SPLIT == 1 || tfirst < n_tiles)) { FA_MTERM_LOAD(tfirst) FA_DMA_TILE(tfirst, 0) FA_DMA_MASKTILE(qi, tfirst, 0) FA_DMA_MHALF(qi, tfirst, 1) FA_MTERM_STORE(0) } \
        tile0_requested = false;                                                                   \
        FA_TILE_BARRIER()                                                                          \
        if (pass == 0) FA_STAMP(st1_)                                                              \
        for (int rr = 0; rr < n_rounds; rr += 2) {              /* (workgroup-uniform bounds) */   \
            const int t = tfirst + rr * tstride;                /* this group's tile of the round */ \
            if (t + tstride < n_tiles) { FA_MTERM_LOAD(t + tstride) FA_DMA_TILE(t + tstride, 1) FA_DMA_MASKTILE(qi, t + tstride, 1) } \
            FA_COMPUTE(t, 0, MODE_)                                                                \
            if (t + tstride < n_tiles) FA_MTERM_STORE(1)                                           \
            FA_TILE_BARRIER()                                                                      \
            if (rr + 1 < n_rounds) {                                                               \
                if (t + 2 * tstride < n_tiles) { FA_MTERM_LOAD(t + 2 * tstride) FA_DMA_TILE(t + 2 * tstride, 0) FA_DMA_MASKTILE(qi, t + 2 * tstride, 0) } \
                FA_COMPUTE(t + tstride, 1, MODE_)                                                  \
                if (t + 2 * tstride < n_tiles) FA_MTERM_STORE(0)                                   \
                FA_TILE_BARRIER()                                                                  \
            }                                                                                      \
        }                                                                                          \
    }

    bool tile0_requested = false;     // pairs: the next q tile's first K/V tile is already on its way to buffer 0
    T* __restrict__ og = (T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh;
#pragma nounroll
    for (int pass = 0; pass < n_pass; ++pass) {
    const int qi = pass == 0 ? nq_r - 1 - qd : qd;     // heaviest first; pairs: then the light partner
    const int q_row0 = qi * BQ + wave * WQ;
    const int qrow = q_row0 + r;                 // block b owns row qrow + 32 b
    const bool wave_active = q_row0 < L;
    // causal: keys up to the workgroup's last row; otherwise every key.  `klim`: the last key a row may see is
    // min(row, L - 1) under the causal mask and L - 1 without it (rows past L are clamped duplicates, never stored)
    const int kv_end = (causal && (qi + 1) * BQ < L) ? (qi + 1) * BQ : L;
    const int n_tiles = (kv_end + BK - 1) / BK;
    const int n_rounds = (n_tiles + SPLIT - 1) / SPLIT;   // rounds of the workgroup's lock-step loop (uniform over all its waves)
    const int wave_last = causal ? q_row0 : L - 1;      // a tile reaching past this key carries a mask for the wave
    const int wave_kv_last = causal ? q_row0 + WQ - 1 : L - 1;   // tiles starting after it hold nothing for the wave
    FA_RESET_STATE()
    FA_MASK_GEOM(qi)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int b = 0; b < QB; ++b) asm volatile("" : "+v"(qf[b][ks]));   // Q loads issued (see fa_fwd_tile_kernel)

    if constexpr (SKIPMAX) {
        FA_PASS(0)
        // Did any row of the workgroup leave the safe range?  (All waves must agree: the exact pass re-streams K/V
        // through the same barriers.)  Practically never taken: logits must outgrow the first tile's row max by
        // ~2^100 (bf16) / 2^16 (f16, 1/sqrt(D)-scaled logits only).
        bool bad = false;
#pragma unroll
        for (int b = 0; b < QB; ++b) bad |= wave_active && sum_left_safe_range(lsum[b][0]);
        flags[wave_all] = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
        __syncthreads();
        int any_bad = 0;
#pragma unroll
        for (int w = 0; w < NWAVES * SPLIT; ++w) any_bad |= flags[w];
        if (any_bad) {
            __syncthreads();   // (flags are rewritten only after later barriers; this orders the LDS buffer reuse)
            FA_RESET_STATE()
            FA_PASS(1)
        }
    } else {
        FA_PASS(1)
    }
    if (pass + 1 == n_pass) FA_STAMP(st2_)
    if constexpr (PAIR) {
        if (pass + 1 < n_pass) {
            // the pair's second q tile: its Q rows go into the registers the loop has just released, its first K/V tile into
            // buffer 0 (every wave is past the loop's last barrier); both land under the epilogue below
            {   // (through opaque copies of the lane's row and half: the compiler otherwise keeps this load's per-lane addresses - eight
                //  registers - alive across the whole kv loop of the first q tile, and at 128 registers spills them)
                const int r0_ = r, h0_ = h;
                int r = r0_, h = h0_;
                asm volatile("" : "+v"(r), "+v"(h));
                FA_LOAD_Q(qd)
            }
            FA_DMA_TILE(0, 0)
            tile0_requested = true;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int b = 0; b < QB; ++b) asm volatile("" : "+v"(qf[b][ks]));
        }
    }
    // ---- split-KV: fold the other groups' (m, l, O) into group 0, one group at a time, through LDS (the K/V buffers are free
    // after the pass's last barrier).  The exchange area lies behind the epilogue's row slabs; a lane's values sit at
    // [value][lane], so writes and reads are conflict-free.  A group that saw no key of a row hands over m = -inf, l = 0,
    // O = 0: its weight 2^((m - m_new) c) is 0.
    if constexpr (SPLIT > 1) {
        constexpr int NVAL = QB * (DT * 16 + 2);
        float* xb = (float*)(smem_all + NWAVES * WQ * ROWB) + wave * (NVAL * 64) + lane;
        static_assert(NWAVES * WQ * ROWB + NWAVES * NVAL * 64 * 4 <= SPLIT * GROUP_B, "exchange area must fit the K/V buffers");
        for (int sgrp = 1; sgrp < SPLIT; ++sgrp) {
            __syncthreads();
            if (ws == sgrp) {
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    xb[(b * (DT * 16 + 2) + 0) * 64] = m[b];
                    xb[(b * (DT * 16 + 2) + 1) * 64] = lsum[b][0];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) xb[(b * (DT * 16 + 2) + 2 + dt * 16 + e) * 64] = o[b][dt][e];
                }
            }
            __syncthreads();
            if (ws == 0 && wave_active) {
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    const float m1 = xb[(b * (DT * 16 + 2) + 0) * 64], l1 = xb[(b * (DT * 16 + 2) + 1) * 64];
                    const float mn = max2(m[b], m1);
                    const float a0 = fast_exp2((m[b] - mn) * c), a1 = fast_exp2((m1 - mn) * c);
                    m[b] = mn;
                    const float ln = lsum[b][0] * a0 + l1 * a1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) lsum[b][j] = ln;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            o[b][dt][e] = o[b][dt][e] * a0 + xb[(b * (DT * 16 + 2) + 2 + dt * 16 + e) * 64] * a1;
                }
            }
        }
        if (ws != 0) return;
    }
    // ---- epilogue: O /= l, staged through LDS so that every store instruction writes whole rows.
    // (A lane owns 8-byte pieces of ONE row at a 16-byte pitch; stored directly, one instruction touches
    // 32 rows x 2 pieces.)  The K buffers are free after the last barrier; each wave uses a private
    // 32-row slab [32][D] with the 16-byte chunk index XORed by the row, so writes and reads are
    // conflict-free and no workgroup barrier is needed.
    // Pairs: the slabs live in the K1 / V1 buffers (two waves each), K0 / V0 are receiving the next q tile's first K/V tile.
    unsigned char* slab = PAIR ? smem_all + (wave < 2 ? KTILE_B + wave * (WQ * ROWB) : 2 * KTILE_B + VTILE_B + (wave - 2) * (WQ * ROWB))
                               : smem_all + wave * (WQ * ROWB);
    static_assert(!PAIR || (NWAVES <= 4 && 2 * WQ * ROWB <= KTILE_B && 2 * WQ * ROWB <= VTILE_B), "pair epilogue slabs must fit K1 / V1");
    constexpr int CPR = D / 8;   // 16-byte chunks per row
    constexpr int ESW = CPR & -CPR;   // the slab's XOR swizzle stays inside aligned groups of this many chunks (the largest power of two in CPR: 8 at head_dim 192)
    if (wave_active) {
    int lane_e = lane;                                  // (pairs: the store loop's per-lane rows and chunks are recomputed per pass, not kept
    if constexpr (PAIR) asm volatile("" : "+v"(lane_e));   //  in registers - and spilled - across the kv loops)
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const float lt = lsum[b][0];
        const float inv = EXTRA ? __builtin_amdgcn_rcpf(lt) * p.ex.drop_rp : __builtin_amdgcn_rcpf(lt);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (T)(o[b][dt][4 * rg + j] * inv);
                const int ch = 4 * dt + rg;                                   // chunk holding d = 32dt + 8rg + 4h ..+3
                *(vec4*)(slab + r * ROWB + ((ch ^ (r & (ESW - 1))) << 4) + 8 * h) = w;
            }
        }
        const int row_b = qrow + 32 * b;
        if (p.lse != nullptr && h == 0 && row_b < L) p.lse[(int64_t)g * L + row_b] = m[b] * p.scale + __logf(lt);
#pragma unroll
        for (int i = 0; i < CPR / 2; ++i) {                      // 64 chunks per store instruction: 8 rows at D=64, 4 at 128, 2 2/3 at 192, 2 at 256
            const int rr = (i * 64 + lane_e) / CPR, ch = (i * 64 + lane_e) % CPR;
            const u32x4 val = *(const u32x4*)(slab + rr * ROWB + ((ch ^ (rr & (ESW - 1))) << 4));
            const int orow = q_row0 + 32 * b + rr;
            if (orow < L && ch < dchunks && (FA_DMA_ABL != 6 || val.x == 0x12345u)) *(u32x4*)(og + (int64_t)orow * p.o_sl + ch * 8) = val;
        }
    }
    }   // wave_active
    }   // pass
#undef FA_DMA_TILE
#undef FA_MASK_GEOM
#undef FA_TILE_DEAD
#undef FA_DMA_MASKTILE
#undef FA_DMA_MHALF
#undef FA_MTERM_LOAD
#undef FA_MTERM_STORE
#undef FA_TILE_BARRIER
#undef FA_COMPUTE
#undef FA_PASS
#undef FA_RESET_STATE
#undef FA_QK
#undef FA_MAX_RESCALE
#undef FA_SOFTMAX
#undef FA_PV

#undef FA_LOAD_Q
#ifdef FA_STAMPS
    {   // stamps go to a buffer of their own (fa_debug_set_stamp_buffer), never to an output: 16 u64 per wave
        unsigned long long st3_;
        FA_STAMP(st3_)
        if (p.stamps != nullptr && lane == 0) {
            // (q-tile subsets: g is the REAL group and qd the group-wide depth - the record index uses the group's real tile count, which the
            //  buffer covers: G8v * nq_tiles >= G * nq_real)
            unsigned long long* sp = p.stamps + (((int64_t)g * (p.xs > 1 ? nq_r : nslot) + qd) * (NWAVES * SPLIT) + wave_all) * 16;
            int tot_tiles = 0;   // kv tiles this workgroup streamed (both q tiles of a pair)
            for (int ps = 0; ps < n_pass; ++ps) {
                const int qi_ = ps == 0 ? nq_r - 1 - qd : qd;
                const int ke_ = (causal && (qi_ + 1) * BQ < L) ? (qi_ + 1) * BQ : L;
                tot_tiles += (ke_ + BK - 1) / BK;
            }
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            sp[0] = st0_; sp[1] = st1_; sp[2] = st2_; sp[3] = st3_;
            sp[4] = ((unsigned long long)xcc << 32) | hwid;
            sp[5] = (unsigned long long)tot_tiles;
            sp[9] = (unsigned long long)blockIdx.x;
            sp[10] = rt0_; sp[11] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

#ifndef FA_DMA_BODY_ONLY   /* fa_fwd_extra.hip includes this file for the body template alone */
// (pairs at head_dim 64: held to 128 registers = 4 waves per SIMD like the unpaired kernel - allocated freely it takes 138)
template <typename T, int D, int NWAVES, bool SKIP, int QB, int SPLIT, bool PAIR = false>
__global__ __launch_bounds__(NWAVES * SPLIT * 64, (PAIR && D == 64) ? 4 : 1) void fa_fwd_dma_kernel(const FwdParams p) {
    fa_fwd_dma_body<T, D, NWAVES, SKIP, QB, SPLIT, PAIR>(p);
}

template <typename T, int D, bool SKIP>
static hipError_t launch_dma_w(const FwdParams& p, int waves, int qb, int split, int pair, hipStream_t stream) {
    if (pair) {   // q-tile pairs (4 waves, one query block per wave, unsplit, causal): (nq + 1) / 2 workgroups per group
        if (!(waves == 4 && qb == 1 && split == 1 && p.causal)) return hipErrorInvalidValue;
        const int64_t pgrid = (int64_t)p.G8 * ((p.nq_tiles + 1) / 2);
        return launch1((const void*)(fa_fwd_dma_kernel<T, D, 4, SKIP, 1, 1, true>), dim3((unsigned)pgrid), dim3(256), 0, stream, p);
    }
    const int64_t grid = (int64_t)p.G8 * p.nq_tiles;
    if (split == 2) return launch1((const void*)(fa_fwd_dma_kernel<T, D, 4, SKIP, 1, 2>), dim3((unsigned)grid), dim3(512), 0, stream, p);
    if (split == 4) {   // 16 waves = 4 per SIMD: needs the 128-register kernel (head_dim 64)
        if constexpr (D == 64) {
            return launch1((const void*)(fa_fwd_dma_kernel<T, D, 4, SKIP, 1, 4>), dim3((unsigned)grid), dim3(1024), 0, stream, p);
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (qb == 2) {   // 64 query rows per wave (head_dim 64, 4 waves): half the LDS operand reads and half the K/V DMA per FLOP
        if constexpr (D == 64) {
            return launch1((const void*)(fa_fwd_dma_kernel<T, D, 4, SKIP, 2, 1>), dim3((unsigned)grid), dim3(256), 0, stream, p);
        } else {
            return hipErrorInvalidValue;
        }
    }
    hipError_t e;
    switch (waves) {
        case 8: e = launch1((const void*)(fa_fwd_dma_kernel<T, D, 8, SKIP, 1, 1>), dim3((unsigned)grid), dim3(512), 0, stream, p); break;
        case 4: e = launch1((const void*)(fa_fwd_dma_kernel<T, D, 4, SKIP, 1, 1>), dim3((unsigned)grid), dim3(256), 0, stream, p); break;
        case 2: e = launch1((const void*)(fa_fwd_dma_kernel<T, D, 2, SKIP, 1, 1>), dim3((unsigned)grid), dim3(128), 0, stream, p); break;
        default: e = launch1((const void*)(fa_fwd_dma_kernel<T, D, 1, SKIP, 1, 1>), dim3((unsigned)grid), dim3(64), 0, stream, p); break;
    }
    return e;
}

// head_dim 192 / 256 (and 136 .. 184 / 200 .. 248 zero-extended): one form - 4 waves x 32 rows, 64-key tiles: K0 K1 V0 V1 = 99 / 132 KB of LDS,
// one workgroup per CU; 234 VGPRs at 192, 256 VGPRs + the accumulators' overflow in AGPRs at 256 (no scratch)
template <typename T, int D, bool SKIP>
static hipError_t launch_dma_wide(const FwdParams& p, int waves, int qb, int split, int pair, hipStream_t stream) {
    if (!((waves == 4 || (waves == 8 && D == 192)) && qb == 1 && split == 1 && !pair)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((int64_t)p.G8 * p.nq_tiles));
    if constexpr (D == 192) {   // (8 waves = two per SIMD = 256 registers: head_dim 256 would spill 33 of them - not instantiated)
        if (waves == 8) return launch1((const void*)(fa_fwd_dma_kernel<T, D, 8, SKIP, 1, 1>), grid, dim3(512), 0, stream, p);
    }
    return launch1((const void*)(fa_fwd_dma_kernel<T, D, 4, SKIP, 1, 1>), grid, dim3(256), 0, stream, p);
}
template <typename T, bool SKIP>
static hipError_t launch_dma_wide_d(const FwdParams& p, int waves, int qb, int split, int pair, hipStream_t stream) {
    return p.D == 192 ? launch_dma_wide<T, 192, SKIP>(p, waves, qb, split, pair, stream) : launch_dma_wide<T, 256, SKIP>(p, waves, qb, split, pair, stream);
}

// waves = waves per kv split group (the workgroup has waves x split); qb = 32-row query blocks per wave (1, or 2 for head_dim 64
// with 4 waves); split = kv split groups (1; 2 | 4 with 4 waves and qb 1).  The host sized p.nq_tiles for 32 waves qb rows.
hipError_t launch_fwd_dma(const FwdParams& p, int dtype, int waves, int qb, int split, int pair, hipStream_t stream) {
    if (p.D > 128) {
        if (dtype == 1) return launch_dma_wide_d<__bf16, true>(p, waves, qb, split, pair, stream);
        return p.skipmax ? launch_dma_wide_d<_Float16, true>(p, waves, qb, split, pair, stream) : launch_dma_wide_d<_Float16, false>(p, waves, qb, split, pair, stream);
    }
    if (dtype == 1) {   // bf16: always skip-max
        return p.D == 64 ? launch_dma_w<__bf16, 64, true>(p, waves, qb, split, pair, stream) : launch_dma_w<__bf16, 128, true>(p, waves, qb, split, pair, stream);
    }
    if (p.skipmax) {
        return p.D == 64 ? launch_dma_w<_Float16, 64, true>(p, waves, qb, split, pair, stream) : launch_dma_w<_Float16, 128, true>(p, waves, qb, split, pair, stream);
    }
    return p.D == 64 ? launch_dma_w<_Float16, 64, false>(p, waves, qb, split, pair, stream) : launch_dma_w<_Float16, 128, false>(p, waves, qb, split, pair, stream);
}

#endif  // FA_DMA_BODY_ONLY

}  // namespace fa
