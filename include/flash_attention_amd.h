/*
 * flash_attention_amd.h - C ABI of libflash_attention.so (MI355X / gfx950 FlashAttention).
 *
 * This is the drop-in boundary for the ONE hot path of graphcore-research/flash-attention-ipu:
 * the forward of `flash_attention_qkv_packed` and the causal SDPA override (plus, as the first
 * widening step, its backward).  Every entry point is
 * `extern "C"`, takes plain pointers and sizes (device pointers unless stated), owns no memory,
 * never throws and never synchronises the host: kernels are enqueued on the caller's HIP stream.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *
 *   fa_fwd_qkv_packed   <- poptorch.custom_op("FlashAttentionQKVPacked", "ai.graphcore", v1,
 *                          inputs=[qkv], attributes={num_chunks_q, num_chunks_kv})
 *                          flash_attention_ipu/_impl/flash_attention_qkv_packed.py:39-46, which PopART
 *                          resolves to FlashAttentionQKVPackedOpx::grow -> flashAttentionQKVPacked
 *                          flash_attention_ipu/cpp/flash_attention_qkv_packed.cpp:618-641, :336-346
 *   fa_fwd              <- flashAttentionQKVPackedWithStash (output + logSumExp stash)
 *                          flash_attention_ipu/cpp/flash_attention_qkv_packed.cpp:175-334 and
 *                          flash_attention_qkv_packed.hpp:10-21; takes q/k/v as three strided views so
 *                          the SDPA override (scaled_dot_product_attention.py:55-59) needs no
 *                          torch.stack copy, and takes the softmax scale as a scalar instead of a
 *                          pre-multiplied q.
 *   fa_infer_output_shape <- FlashAttentionQKVPackedOp::setup() shape inference [3,G,L,D]->[G,L,D]
 *                          flash_attention_qkv_packed.cpp:571-577
 *   fa_plan             <- the reference's (num_chunks_q, num_chunks_kv) -> chunk-size arithmetic
 *                          flash_attention_qkv_packed.cpp:193-195 (chunkedQueryLen / chunkedKVLen);
 *                          reports the hardware tile this library maps the request to.
 *   fa_bwd              <- flashAttentionQKVPackedGrad + FlashAttentionQKVPackedGradOp/Opx
 *                          flash_attention_qkv_packed.cpp:348-512, :522-558, :643-673 (next-row scope)
 *   fa_abi_version / fa_last_error / fa_error_string
 *                       <- (no reference counterpart: the reference's .so ABI is static-init
 *                          registration only, flash_attention_ipu/__init__.py:4-23; its errors are C++
 *                          asserts, flash_attention_qkv_packed.cpp:183,190-191)
 *
 * Semantics (identical to the reference):
 *   O[g] = softmax_rows(scale * Q[g] K[g]^T + M) V[g],  M[r,c] = -inf-equivalent for c > r (causal);
 *   LSE[g,r] = log sum_c exp(scale * S[r,c]).  `scale` is 1.0 for the packed op (no 1/sqrt(D) inside
 *   the op, flash_attention_qkv_packed.cpp:254); the SDPA override passes D^-0.5.
 *   The result does not depend on num_chunks_q / num_chunks_kv.
 */
#ifndef FLASH_ATTENTION_AMD_H
#define FLASH_ATTENTION_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_ABI_VERSION 7   /* 3: fa_bwd takes heads_kv (grouped-query backward without expanded K/V); 4: fa_plan_t gains pair / balanced,
                              fa_fwd_ex / fa_bwd_ex (attention mask or bias, dropout); 5: fa_fwd_ex / fa_bwd_ex take `dropout_state`
                              (seed and offset in DEVICE memory: graph-safe dropout); 6: fa_plan_t gains xs, the A/B kernels (plan kernel 5,
                              debug kernels 't' / 'P', options "fold" / "mirror") are gone; 7: what only existed so that older callers link is
                              gone too - fa_experimental_kernels() (always 0 since ABI 6) and the never-returned FA_ERR_UNSUPPORTED (-13 stays
                              unassigned) - and the debug option "bsplit" names the backward's split groups */

/* element types (reference allows FLOAT16 and FLOAT, FA.cpp:609; bf16 is new here) */
enum { FA_DTYPE_F16 = 0, FA_DTYPE_BF16 = 1, FA_DTYPE_F32 = 2 };

/* return codes */
enum {
    FA_OK = 0,
    FA_ERR_NULL_POINTER = -1,
    FA_ERR_BAD_SHAPE = -2,          /* G, L or D < 1 */
    FA_ERR_BAD_DTYPE = -3,
    FA_ERR_CHUNKS_Q = -4,           /* L % num_chunks_q != 0   (FA.cpp:190, fa.py:28-31) */
    FA_ERR_CHUNKS_KV = -5,          /* L % num_chunks_kv != 0  (FA.cpp:191, fa.py:32-35) */
    FA_ERR_MISALIGNED = -6,         /* MFMA path needs 16-byte aligned rows */
    FA_ERR_NOT_CAUSAL = -7,         /* causal must be 0 or 1 (the reference path is causal only, sdpa.py:29-32;
                                       causal = 0 is a widening) */
    FA_ERR_LAUNCH = -8,             /* the launch itself failed (hipLaunchKernel's own return value) */
    FA_ERR_BAD_STRIDE = -9,
    FA_ERR_TOO_LARGE = -10,         /* grid or offsets exceed what the kernel indexes */
    FA_ERR_BAD_SCALE = -11,         /* softmax_scale must be finite and > 0 */
    FA_ERR_BAD_OPTION = -12,        /* fa_debug_set_option: unknown name or value out of range */
    /* (-13 is unassigned: ABI 4 drafts returned FA_ERR_UNSUPPORTED for a mask / dropout on operands the MFMA kernels do not take; the generic
        kernels have served those since, and the name left the header with ABI 7) */
    FA_ERR_BAD_MASK = -14,          /* mask_kind out of range, or mask pointer and mask_kind disagree */
    FA_ERR_BAD_DROPOUT = -15        /* dropout_p outside [0, 1) */
};

/* hardware tile the request was mapped to (fa_plan) */
typedef struct fa_plan_t {
    int32_t kernel;      /* 0 = MFMA kernel, 64-key tiles (bf16/f16, D in {64,128,192,256}; other multiples of 8 up to 256 run zero-extended
                                at the next of the four, in this and - up to 128 - in kernels 2, 4, 6), 1 = generic kernel,
                            2 = MFMA kernel, software-pipelined 32-key steps,
                            3 = fp32 MFMA kernel (fp32 inputs, D in {64,128}, 16-byte rows),
                            4 = software-pipelined kernel on v_mfma_f32_16x16x32 (bf16/f16, D in {64,128}),
                            (5 is unassigned: the complementary-phase kernel of rounds 3-4 left the library in ABI 6),
                            6 = 64-key tiles, 64-row workgroups whose 4 waves are (row block, key half) pairs (bf16/f16, D in {64,128}) */
    int32_t block_q;     /* q rows per workgroup */
    int32_t block_kv;    /* keys per LDS tile / step */
    int32_t waves;       /* wavefronts per workgroup; waves * qb * 32 / block_q = kv split groups of the workgroup (1, or 2 | 4
                            when a grid too small to fill the chip is served by splitting each q tile's kv loop) */
    int32_t chunk_q;     /* L / num_chunks_q  : the reference's chunkedQueryLen */
    int32_t chunk_kv;    /* L / num_chunks_kv : the reference's chunkedKVLen */
    int64_t grid;        /* workgroups launched */
    int64_t lds_bytes;   /* LDS per workgroup */
    int32_t qb;          /* 32-row query blocks per wave (block_q = 32 * waves * qb / kv split groups) */
    int32_t sched;       /* groups per XCD bunch of the block -> (group, q tile) map */
    int32_t skipmax;     /* 1 = reference max fixed on a wave's first tile, optimistic pass + exact fallback pass
                            (bf16 always; f16 when softmax_scale <= 0.25), 0 = per-tile row max with lazy rescale */
    int32_t source;      /* 0 = built-in rules, 1 = a row of the measured tuning table (fa_tuning_table.h),
                            2 = a debug override was in force (fa_debug_set_option / FA_AMD_* at load) */
    int32_t pair;        /* 1 = a workgroup owns two q tiles, (n - 1 - i) then i, one after the other (causal, 64-key tiles) [ABI 4] */
    int32_t balanced;    /* 1 = placement-balanced block order (a causal grid below one round of the chip's workgroup slots) [ABI 4] */
    int32_t xs;          /* q-tile subsets per group [ABI 6]: groups go to the 8 XCDs by index mod 8; where G wastes an eighth of the XCD slots or
                            more (G < 8, G = 12, ...) the launch is laid out over G * xs (group, subset) pairs, subset s of a group = its q
                            tiles s, s + xs, ... counted from the last.  xs = the SMALLEST of 2, 4, 8 whose G * xs pairs waste less than 9 % of
                            the slots (G = 25 -> 4, not 8); 1 = groups map whole - also when every subset would not get a tile
                            (ceil(L / 256) < xs) or when the pairs one XCD then holds no longer share its L2 (pairs per XCD x K/V bytes of
                            a group > 8 MB with K/V < 4 MB per group: G = 7 at large L -> 1).  fa_capi.cpp subsets_for();
                            measured in profiles/r05_subsets/ab_final_rule.txt */
} fa_plan_t;

int fa_abi_version(void);

/* thread-local text of the last failure on this thread ("" if none) */
const char* fa_last_error(void);
const char* fa_error_string(int code);

/* Bench utility (NOT a data-path call: allocates a scratch buffer and synchronises the device): measured
 * dense MFMA throughput of the current device in TFLOP/s for FA_DTYPE_BF16 / FA_DTYPE_F16, from a register-only
 * loop of v_mfma_f32_32x32x16 at 4 waves per SIMD on every CU; <= 0 on failure.  random_data = 0 feeds
 * near-constant operands (the device holds its full clock), 1 feeds N(0,1)-like operands that differ in every
 * lane (realistic switching activity: the device drops its clock to stay inside its power limit).  bench.py
 * reports the attention kernel against both as well as against the nominal 2.5 PFLOP/s. */
double fa_probe_mfma_peak(int32_t dtype, int32_t iters, int32_t random_data);

/* Bench utility: enqueue on `stream` (use a side stream) a one-wave kernel that samples the shader clock counter
 * against the constant 100 MHz counter for `spins` sleeps of ~2 us and writes two uint64 to the device buffer
 * `out2`: {shader cycles, 10 ns ticks}.  cycles / ticks * 100 = the core clock in MHz the device held meanwhile -
 * bench.py runs it next to the timed steps (the attention kernels are power-bound: DESIGN.md 3.0).
 * (No reference counterpart.) */
int fa_probe_clock(void* out2, int32_t spins, void* stream);

/* [3,G,L,D] -> [G,L,D]; returns FA_OK or FA_ERR_BAD_SHAPE (rank != 4 or dim0 != 3). */
int fa_infer_output_shape(const int64_t* qkv_shape, int32_t rank, int64_t out_shape[3]);

/* Host-only: which kernel/tile would serve this request (contiguous, 16-byte aligned operands assumed).
 * No GPU needed.  fa_fwd() goes through the same selection function with the same inputs, so for aligned
 * operands the reported plan IS the launched plan.
 * num_chunks_q  : L / num_chunks_q  is the upper bound of the q tile  (>= 128 -> 4-wave workgroups, 64 -> 2, 32 -> 1)
 * num_chunks_kv : L / num_chunks_kv is the upper bound of the kv step: >= 64 -> 64-key tiles (or 32-key pipelined
 *                 steps where those are faster), 32..63 -> 32-key steps; the hardware minimum is 32 keys, smaller
 *                 chunks are served by 32-key steps (the result does not depend on the chunking). */
int fa_plan(int64_t G, int64_t L, int64_t D, int32_t dtype, int32_t num_chunks_q, int32_t num_chunks_kv,
            float softmax_scale, int32_t causal, fa_plan_t* plan);

/* TEST / TUNING ONLY - not part of the data path contract.  Overrides one knob of the tile plan for the whole
 * process until set back to its "auto" value.  The FA_AMD_* environment variables of round 1 are read ONCE, at the
 * first call into the library, into the same knobs; nothing reads the environment per launch.
 *   "kernel"  0 auto | 'd' (64-key tiles, LDS-DMA) | 'p' (pipelined 32-key steps) | 'x' (the same on 16x16x32 MFMAs)
 *             | 'h' (64-row workgroups, waves = (row block, key half) of the 64-key tiles)
 *   "qb"      0 auto | 1 | 2        "waves"  0 auto | 1 | 2 | 4 | 8        "sched"  0 auto | n >= 1
 *   "skipmax" -1 auto | 0 | 1       "f32_generic" 0 | 1                    "table"  1 use the tuning table | 0 rules only
 *   "split"   0 auto | 1 | 2 | 4    kv split groups per workgroup ('d' kernel: 4 waves x 32 rows per group; 'h' kernel when forced with it)
 *   "pair"    -1 auto | 0 | 1       q-tile pairs (n - 1 - i, i) per workgroup ('d' kernel, causal, 4 waves)
 *   "balance" -1 auto | 0 | 1       placement-balanced block order of a causal grid below one round of workgroup slots
 *   "subsets" -1 auto | 0 off       q-tile subsets for group counts that leave XCDs idle (fa_plan_t.xs)
 *   "bsplit"  -1 auto | 0 | 1 off | 2 | 3   split groups inside the backward's workgroups (fa_bwd: kv split groups of a dQ workgroup, query-sweep
 *                                   split groups of a dK/dV workgroup; their partial sums meet in LDS in group order).  Auto, per grid: at most 512
 *                                   workgroups at L >= 2048 or 384 at L >= 1024 - 3 groups at head_dim 64 up to 256 workgroups (L >= 2048), else 2;
 *                                   at head_dim 128 "2" = two kv split groups in dQ, two groups that take the two 32-row halves of every streamed
 *                                   query tile in dK/dV
 *   "bbal"    -1 auto | 0 off | 1   balanced dQ units (fa_bwd, causal, heads == heads_kv, D 64 | 128, row-block count a multiple of 4): the heavy
 *                                   half of a group's 128-row blocks run as two workgroups each (halves of the kv sweep) whose fp32 partial sums
 *                                   are parked in dk / dv until a reduce pass adds them.  Auto: at most 128 row blocks in the grid with >= 16 per
 *                                   group, or at most 256 with >= 64 per group
 * Returns FA_OK or FA_ERR_BAD_OPTION.  (No reference counterpart.) */
int fa_debug_set_option(const char* name, int32_t value);
int fa_debug_get_option(const char* name, int32_t* value);

/* MEASUREMENT BUILDS ONLY (-DFA_STAMPS, scripts/phase_times.py): a device buffer of its own for the in-kernel phase
 * stamps - 16 uint64 per wave of a forward launch, record ((g * q_tiles + q_tile) * waves + wave); never an output
 * tensor.  The shipped library accepts the call and writes nothing.  (NULL, 0) clears it.  (No reference counterpart.) */
int fa_debug_set_stamp_buffer(void* device_buffer, int64_t bytes);

/*
 * Forward on three strided views.  All pointers are device pointers.
 *   q,k,v : element (g, r, d) at  base + (g / heads) * strides[0] + (g % heads) * strides[1]
 *           + r * strides[2] + d   (strides in ELEMENTS; innermost dimension contiguous)
 *   out   : same addressing with o_strides; dtype = input dtype
 *   lse   : optional float32 [G, L] contiguous (NULL to skip)  - the reference's logSumExp stash
 *   heads : inner group count for the 2-level group stride (1 => strides[1] unused, strides[0] is
 *           the group stride)
 *   heads_kv : key/value heads (grouped-query attention; the reference rejects it, sdpa.py:33-37):
 *           must divide `heads`; query head h reads key/value head h / (heads / heads_kv), addressed with
 *           k_strides / v_strides.  heads_kv == heads is ordinary multi-head attention.
 *   softmax_scale : multiplies the logits (1.0f reproduces the packed op); must be finite and > 0
 *   causal : 1 = causal mask (the reference path); 0 = no mask (widening, SURVEY 8f-4: the reference raises
 *           NotImplementedError, sdpa.py:29-32)
 *   stream : hipStream_t (NULL = default stream)
 */
int fa_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
           int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv,
           const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
           const int64_t o_strides[3],
           int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype,
           float softmax_scale, int32_t causal, void* stream);

/*
 * Backward of fa_fwd (replaces flashAttentionQKVPackedGrad, flash_attention_qkv_packed.cpp:348-512, and
 * FlashAttentionQKVPackedGradOpx::grow, :643-658).  Unlike the reference it does not recompute the
 * forward: it takes the forward's `out` and `lse`.
 *   q,k,v,out,dout : inputs; dq,dk,dv : outputs; all addressed like fa_fwd's tensors with
 *                    strides[i] = {outer group stride, inner group stride, row stride} (elements),
 *                    i = 0..7 for q, k, v, out, dout, dq, dk, dv
 *   lse            : float32 [G, L] contiguous, as written by fa_fwd (log-sum-exp of the scaled logits)
 *   dsum_workspace : float32 [G, L] contiguous scratch (rowsum(dout * out)), caller-owned
 *   heads, heads_kv : as in fa_fwd.  With heads_kv < heads (grouped-query attention) k, v, dk and dv have
 *                    G / (heads / heads_kv) groups; dk / dv of a key/value head are the sums over the query heads that
 *                    read it, formed inside ONE workgroup per 128 keys (no expanded K/V, no atomics)   [ABI 3]
 * f16/bf16/fp32 with D in {64, 128}, f16/bf16 with D any other multiple of 8 up to 256 (zero-extended at 64 / 128 / 192 / 256 inside
 * the kernels) and 16-byte aligned rows run the MFMA kernels; every other case (other D <= 1024, unaligned rows) runs generic
 * fp32-arithmetic kernels.
 * Deterministic: no gradient is accumulated across workgroups with atomics; results are bit-reproducible from call to call.
 * dq, dk and dv are pure outputs that the call may ALSO use as scratch before it writes their final values (an under-filled causal dQ grid
 * parks fp32 partial sums of dQ in dk / dv until a reduce pass has added them - "bbal" below): they must not overlap each other or any input.
 */
int fa_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
           void* dq, void* dk, void* dv, float* dsum_workspace,
           int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv, const int64_t strides[8][3],
           int32_t dtype, float softmax_scale, int32_t causal, void* stream);

/* mask_kind of fa_fwd_ex / fa_bwd_ex */
#define FA_MASK_NONE 0
#define FA_MASK_BOOL 1   /* bytes, non-zero = the key takes part (torch's boolean attn_mask) */
#define FA_MASK_BIAS 2   /* additive term on the scaled logits, in the operand dtype (torch's float attn_mask) */

/*
 * fa_fwd / fa_bwd with an attention mask or additive bias and / or dropout - the two modes of scaled_dot_product_attention
 * that the reference rejects (flash_attention_ipu/_impl/scaled_dot_product_attention.py:21-28: `attn_mask`, `dropout_p > 0`;
 * SURVEY 8f-4).  [ABI 4]  Everything before `mask` is as in fa_fwd / fa_bwd.
 *   mask          : device pointer or NULL (mask_kind FA_MASK_NONE).  Element (g, row, key) at
 *                   mask[(g / heads) * mask_strides[0] + (g % heads) * mask_strides[1] + row * mask_strides[2] + key]
 *                   (strides in ELEMENTS of the mask's type; 0 broadcasts that dimension - e.g. a [L, L] mask has strides
 *                   {0, 0, L}, a key-padding mask [B, 1, 1, L] has {L, 0, 0}; the key dimension is contiguous).
 *                   FA_MASK_BOOL: bytes, non-zero = the key takes part;  FA_MASK_BIAS: values of the operand dtype, added to
 *                   softmax_scale * (q . k) - use -inf to exclude a key.  A row that excludes every key returns NaN (as torch
 *                   does) and lse = -inf.  May be combined with causal = 1.  No gradient is computed for a bias.
 *                   A mask whose pointer and non-zero strides are multiples of four ELEMENTS travels to LDS in 16-byte pieces; its
 *                   loads stay inside the aligned 4-byte word that holds the last element of the last row (up to 3 bytes past
 *                   mask[..][L-1][L-1], never another word); any other mask is read element by element.
 *   dropout_p     : probability in [0, 1) of dropping an attention weight (0 = off); kept weights are scaled by 1 / (1 - p).
 *                   The decision for (g, row, key) is a pure function of (dropout_seed, dropout_offset, g, row, key)
 *                   (csrc/fa_extras.h), so fa_bwd_ex given the same three values regenerates the forward's mask.
 *   dropout_state : NULL, or a DEVICE pointer to two uint64 {seed, offset} that the kernels read INSTEAD of the two immediate values
 *                   [ABI 5].  A launch captured in a HIP graph bakes its immediates into every replay - one dropout mask forever;
 *                   with the two words in device memory the caller captures its own update of them (copy the generator's state to
 *                   a per-call snapshot, advance the generator) and every replay sees fresh values.  fa_bwd_ex must be given the
 *                   snapshot its forward read.  8-byte aligned; read on `stream`, never written.
 * f16 / bf16 operands with 16-byte aligned rows run MFMA kernels (csrc/fa_fwd_extra.hip, fa_bwd_extra.hip) at head dims up to 256
 * (multiples of 8: zero-extended at 64 / 128 / 192 / 256; at 256 fa_bwd_ex forms dV and dK in two passes);
 * everything else fa_fwd serves (fp32, other head dims <= 1024, unaligned rows) runs the generic fp32-arithmetic kernels with the same
 * mask and the same dropout decisions (a bias then has the operand dtype, fp32 included).  With mask = NULL and dropout_p = 0 they
 * ARE fa_fwd / fa_bwd.
 */
int fa_fwd_ex(const void* q, const void* k, const void* v, void* out, float* lse,
              int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv,
              const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
              const int64_t o_strides[3],
              int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype, float softmax_scale, int32_t causal,
              const void* mask, int32_t mask_kind, const int64_t mask_strides[3],
              float dropout_p, uint64_t dropout_seed, uint64_t dropout_offset, const uint64_t* dropout_state, void* stream);
int fa_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
              void* dq, void* dk, void* dv, float* dsum_workspace,
              int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv, const int64_t strides[8][3],
              int32_t dtype, float softmax_scale, int32_t causal,
              const void* mask, int32_t mask_kind, const int64_t mask_strides[3],
              float dropout_p, uint64_t dropout_seed, uint64_t dropout_offset, const uint64_t* dropout_state, void* stream);

/* Forward on a contiguous packed qkv [3,G,L,D] -> out [G,L,D] (scale 1, causal). */
int fa_fwd_qkv_packed(const void* qkv, void* out, float* lse, int64_t G, int64_t L, int64_t D,
                      int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLASH_ATTENTION_AMD_H */
