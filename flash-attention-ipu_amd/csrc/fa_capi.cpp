// C ABI of libflash_attention.so - see include/flash_attention_amd.h for the contract and for the
// reference interface (file:line) each entry point replaces.  No torch types, no exceptions, no host
// synchronisation, no allocation: validate, pick a tile plan, enqueue one kernel on the caller's stream.

#include "flash_attention_amd.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "fa_params.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int elt_size(int dtype) { return dtype == FA_DTYPE_F32 ? 4 : 2; }

// num_chunks -> hardware tile.  The reference's chunk sizes (L/num_chunks, FA.cpp:193-195) are
// upper bounds on the scheduling tile: the result is independent of them (semantic
// contract), so a chunk larger than one workgroup can hold is subdivided into hardware tiles, and a
// chunk that IS a hardware tile size is honoured exactly.  Tiles: 32 q rows per wave, 1..8 waves.
// The base tile is 4 waves = 128 rows (4 workgroups per CU at D=64, 2 at D=128): fastest on short sequences and
// grids that under-fill the chip; long sequences on full grids get 256-row workgroups (make_plan / pick_kernel:
// two query blocks per wave at head_dim 64, 8 waves at head_dim 128).
int pick_waves(int64_t L, int64_t chunk_q) {
    int64_t want = chunk_q < L ? chunk_q : L;
    const char* env = getenv("FA_AMD_WAVES");
    if (env && *env) {
        int w = atoi(env);
        if (w == 1 || w == 2 || w == 4 || w == 8) return w;
    }
    if (want >= 128) return 4;
    if (want >= 64) return 2;
    return 1;
}

bool tile_kernel_ok(int64_t D, int dtype, const void* const ptrs[4], const int64_t* const strides[4]) {
    if (!(dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16)) return false;
    if (!(D == 64 || D == 128)) return false;
    for (int i = 0; i < 4; ++i) {
        if (ptrs[i] && (reinterpret_cast<uintptr_t>(ptrs[i]) & 15)) return false;
        if (strides[i]) {
            for (int j = 0; j < 3; ++j)
                if (strides[i][j] % 8) return false;   // 16-byte rows for 16-bit elements
        }
    }
    return true;
}

// fp32 inputs: the v_mfma_f32_32x32x2_f32 kernel (fa_fwd_f32.hip) needs head_dim 64 / 128 and 16-byte rows
bool f32_kernel_ok(int64_t D, int dtype, const void* const ptrs[4], const int64_t* const strides[4]) {
    if (dtype != FA_DTYPE_F32 || !(D == 64 || D == 128)) return false;
    if (const char* e = getenv("FA_AMD_F32")) { if (e[0] == 'g') return false; }   // FA_AMD_F32=generic: A/B knob
    for (int i = 0; i < 4; ++i) {
        if (ptrs[i] && (reinterpret_cast<uintptr_t>(ptrs[i]) & 15)) return false;
        if (strides[i]) {
            for (int j = 0; j < 3; ++j)
                if (strides[i][j] % 4) return false;
        }
    }
    return true;
}

// Which MFMA kernel serves a request:
//   'd' fa_fwd_dma_kernel   64-key tiles, LDS-DMA staged (fa_fwd_dma.hip)
//   'p' fa_fwd_pipe_kernel  software-pipelined 32-key steps (fa_fwd_pipe.hip); skip-max arithmetic only
//   't' fa_fwd_tile_kernel  register-staged A/B baseline (FA_AMD_KERNEL=tile)
// and qb = 32-row query blocks per wave (2 = 64 rows per wave, head_dim 64 only).
// Default: head_dim 64 with 4 waves, skip-max arithmetic, L >= 4096 and a grid of >= 1024 workgroups -> 'p' with
// qb 2 (256-row workgroups: +1..6 % measured at L 4096 .. 32768 under sustained load; slower on short sequences
// and under-filled grids); everything else -> 'd' with qb 1.
// FA_AMD_KERNEL=dma|pipe|tile and FA_AMD_QB=1|2 override (developer knobs).
struct KernelChoice { char which; int qb; };
KernelChoice pick_kernel(int64_t G, int64_t L, int64_t D, int waves, int64_t chunk_q, bool skip_ok) {
    const char* kv = getenv("FA_AMD_KERNEL");
    const char forced = (kv && (kv[0] == 't' || kv[0] == 'd' || kv[0] == 'p')) ? kv[0] : 0;
    const char* qe = getenv("FA_AMD_QB");
    const int qb_env = (qe && *qe) ? atoi(qe) : 0;
    KernelChoice kc{'d', 1};
    const bool pipe_ok = skip_ok && waves <= 4;
    if (forced == 't') return KernelChoice{'t', 1};
    if (forced == 'p' && pipe_ok) kc.which = 'p';
    // (256-row workgroups at 2 per CU: below ~2 workgroups per slot of a 256-CU device the coarser grid loses)
    const int64_t wgs_qb2 = (G + 7) / 8 * 8 * ((L + 255) / 256);
    if (forced == 0 && pipe_ok && D == 64 && waves == 4 && chunk_q >= 256 && L >= 4096 && wgs_qb2 >= 1024) kc = KernelChoice{'p', 2};
    const bool qb2_ok = D == 64 && L >= 256 && waves == 4 && chunk_q >= 256;   // (the chunk bounds the q tile)
    if (qb_env == 1) kc.qb = 1;
    if (qb_env == 2 && qb2_ok) kc.qb = 2;
    if (forced == 'p' && kc.which == 'p' && qb_env == 0) kc.qb = (qb2_ok && waves == 4) ? 2 : 1;
    return kc;
}

int make_plan(int64_t G, int64_t L, int64_t D, int32_t dtype, int32_t nq, int32_t nk, bool tile_ok, bool f32_ok, bool skip_ok,
              fa_plan_t* plan, KernelChoice* choice) {
    if (G < 1 || L < 1 || D < 1) return fail(FA_ERR_BAD_SHAPE, "G, L, D must be >= 1 (got %lld, %lld, %lld)", (long long)G, (long long)L, (long long)D);
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    if (nq < 1 || L % nq != 0) return fail(FA_ERR_CHUNKS_Q, "sequence length %lld is not divisible by num_chunks_q=%d", (long long)L, nq);
    if (nk < 1 || L % nk != 0) return fail(FA_ERR_CHUNKS_KV, "sequence length %lld is not divisible by num_chunks_kv=%d", (long long)L, nk);
    if (L > (1ll << 30) || G > (1ll << 30)) return fail(FA_ERR_TOO_LARGE, "G and L must be < 2^30");
    memset(plan, 0, sizeof(*plan));
    plan->chunk_q = (int32_t)(L / nq);
    plan->chunk_kv = (int32_t)(L / nk);
    if (tile_ok) {
        plan->waves = pick_waves(L, plan->chunk_q);
        // head_dim 128, skip-max arithmetic: 8-wave / 256-row workgroups halve the K/V DMA traffic per FLOP
        // (+2..3 % measured at L >= 4096 on full grids; slower on short sequences, small grids and with the
        // per-tile row max of unscaled f16)
        {
            const char* we = getenv("FA_AMD_WAVES");
            const int64_t wgs256 = (G + 7) / 8 * 8 * ((L + 255) / 256);
            if (!(we && *we) && plan->waves == 4 && D == 128 && skip_ok && plan->chunk_q >= 256 && L >= 4096 && wgs256 >= 512)
                plan->waves = 8;
        }
        const KernelChoice kc = pick_kernel(G, L, D, plan->waves, plan->chunk_q < L ? plan->chunk_q : L, skip_ok);
        if (choice) *choice = kc;
        plan->kernel = kc.which == 'p' ? 2 : 0;
        plan->block_q = 32 * plan->waves * kc.qb;
        plan->block_kv = kc.which == 'p' ? 32 : 64;
        const int64_t nqt = (L + plan->block_q - 1) / plan->block_q;
        const int64_t g8 = (G + 7) / 8 * 8;
        plan->grid = g8 * nqt;
        plan->lds_bytes = kc.which == 'p' ? 4 * 32 * D * 2 + 4 * (D / 32) * (32 * 64 + 64) + 4 * plan->waves
                                          : 2 * 64 * D * 2 + 2 * (D / 32) * (64 * 64 + 64);
    } else if (f32_ok) {
        plan->kernel = 3;
        plan->waves = 4;
        plan->block_q = 128;
        plan->block_kv = 32;
        plan->grid = (G + 7) / 8 * 8 * ((L + 127) / 128);
        plan->lds_bytes = 2 * 32 * (D + 1) * 4 + 2 * 32 * D * 4;
    } else {
        if (D > 1024) return fail(FA_ERR_BAD_SHAPE, "generic kernel supports D <= 1024 (got %lld)", (long long)D);
        plan->kernel = 1;
        plan->waves = 4;
        plan->block_q = 1;
        plan->block_kv = 1024;
        plan->grid = G * L;
        plan->lds_bytes = 1024 * 4 + 16 + D * 4;
    }
    if (plan->grid > 0x7fffffffll) return fail(FA_ERR_TOO_LARGE, "grid of %lld workgroups exceeds 2^31-1", (long long)plan->grid);
    return FA_OK;
}

}  // namespace

extern "C" {

int fa_abi_version(void) { return FA_ABI_VERSION; }

const char* fa_last_error(void) { return g_err; }

const char* fa_error_string(int code) {
    switch (code) {
        case FA_OK: return "ok";
        case FA_ERR_NULL_POINTER: return "null pointer";
        case FA_ERR_BAD_SHAPE: return "bad shape";
        case FA_ERR_BAD_DTYPE: return "bad dtype";
        case FA_ERR_CHUNKS_Q: return "sequence length not divisible by num_chunks_q";
        case FA_ERR_CHUNKS_KV: return "sequence length not divisible by num_chunks_kv";
        case FA_ERR_MISALIGNED: return "misaligned pointer or stride";
        case FA_ERR_NOT_CAUSAL: return "only causal attention is supported";
        case FA_ERR_LAUNCH: return "kernel launch failed";
        case FA_ERR_BAD_STRIDE: return "bad stride";
        case FA_ERR_TOO_LARGE: return "problem too large";
        default: return "unknown error";
    }
}

double fa_probe_mfma_peak(int32_t dtype, int32_t iters, int32_t random_data) {
    if (!(dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16) || iters < 1) return -1.0;
    return fa::probe_mfma_peak(dtype, iters, random_data);
}

int fa_probe_clock(void* out2, int32_t spins, void* stream) {
    if (!out2) return fail(FA_ERR_NULL_POINTER, "fa_probe_clock: null output pointer");
    if (spins < 1) return fail(FA_ERR_BAD_SHAPE, "fa_probe_clock: spins must be >= 1");
    hipError_t e = fa::launch_clock_sampler(static_cast<unsigned long long*>(out2), spins, (hipStream_t)stream);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

int fa_infer_output_shape(const int64_t* qkv_shape, int32_t rank, int64_t out_shape[3]) {
    if (!qkv_shape || !out_shape) return fail(FA_ERR_NULL_POINTER, "null shape pointer");
    if (rank != 4) return fail(FA_ERR_BAD_SHAPE, "qkv must have 4 dimensions (got %d)", rank);
    if (qkv_shape[0] != 3) return fail(FA_ERR_BAD_SHAPE, "qkv must have size 3 at dimension 0 (got %lld)", (long long)qkv_shape[0]);
    out_shape[0] = qkv_shape[1];
    out_shape[1] = qkv_shape[2];
    out_shape[2] = qkv_shape[3];
    return FA_OK;
}

int fa_plan(int64_t G, int64_t L, int64_t D, int32_t dtype, int32_t num_chunks_q, int32_t num_chunks_kv, fa_plan_t* plan) {
    if (!plan) return fail(FA_ERR_NULL_POINTER, "null plan pointer");
    const void* const ptrs[4] = {nullptr, nullptr, nullptr, nullptr};
    const int64_t* const strides[4] = {nullptr, nullptr, nullptr, nullptr};
    // (the packed op has softmax scale 1: f16 keeps the per-tile row max, bf16 runs the skip-max arithmetic)
    return make_plan(G, L, D, dtype, num_chunks_q, num_chunks_kv, tile_kernel_ok(D, dtype, ptrs, strides),
                     f32_kernel_ok(D, dtype, ptrs, strides), dtype == FA_DTYPE_BF16, plan, nullptr);
}

int fa_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
           int64_t G, int64_t L, int64_t D, int64_t heads,
           const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],
           int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype,
           float softmax_scale, int32_t causal, void* stream) {
    if (!q || !k || !v || !out) return fail(FA_ERR_NULL_POINTER, "q, k, v and out must be non-null");
    if (!q_strides || !k_strides || !v_strides || !o_strides) return fail(FA_ERR_NULL_POINTER, "stride arrays must be non-null");
    if (!causal) return fail(FA_ERR_NOT_CAUSAL, "only causal attention is supported (reference: sdpa.py:29-32)");
    if (heads < 1 || (G >= 1 && G % heads != 0)) return fail(FA_ERR_BAD_SHAPE, "G=%lld must be a positive multiple of heads=%lld", (long long)G, (long long)heads);
    const void* const ptrs[4] = {q, k, v, out};
    const int64_t* const strides[4] = {q_strides, k_strides, v_strides, o_strides};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
            if (strides[i][j] < 0) return fail(FA_ERR_BAD_STRIDE, "negative strides are not supported");
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    for (int i = 0; i < 4; ++i)
        if (reinterpret_cast<uintptr_t>(ptrs[i]) % elt_size(dtype)) return fail(FA_ERR_MISALIGNED, "pointer %d is not element aligned", i);

    fa_plan_t plan;
    bool tile_ok = tile_kernel_ok(D, dtype, ptrs, strides);
    // the MFMA kernel addresses K/V rows with 32-bit element offsets from the group base
    if (tile_ok && (L * k_strides[2] >= (1ll << 31) || L * v_strides[2] >= (1ll << 31))) tile_ok = false;
    // f16 skip-max pays off when logits are small against the 2^15 headroom of f16 P: a softmax scale
    // <= 0.25 means 1/sqrt(D)-style scaled attention (D >= 16); the packed op (scale 1) keeps the row max
    int skipmax = softmax_scale <= 0.25f ? 1 : 0;
    if (const char* sm = getenv("FA_AMD_SKIPMAX")) skipmax = atoi(sm);
    KernelChoice kc{'d', 1};
    bool f32_ok = f32_kernel_ok(D, dtype, ptrs, strides);
    if (f32_ok && (L * k_strides[2] >= (1ll << 31) || L * v_strides[2] >= (1ll << 31))) f32_ok = false;
    int rc = make_plan(G, L, D, dtype, num_chunks_q, num_chunks_kv, tile_ok, f32_ok, dtype == FA_DTYPE_BF16 || skipmax != 0, &plan, &kc);
    if (rc != FA_OK) return rc;

    fa::FwdParams p;
    p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse;
    p.q_sb = q_strides[0]; p.q_sh = q_strides[1]; p.q_sl = q_strides[2];
    p.k_sb = k_strides[0]; p.k_sh = k_strides[1]; p.k_sl = k_strides[2];
    p.v_sb = v_strides[0]; p.v_sh = v_strides[1]; p.v_sl = v_strides[2];
    p.o_sb = o_strides[0]; p.o_sh = o_strides[1]; p.o_sl = o_strides[2];
    p.G = (int32_t)G;
    p.G8 = (int32_t)((G + 7) / 8 * 8);
    p.H = (int32_t)heads;
    p.L = (int32_t)L;
    p.D = (int32_t)D;
    p.nq_tiles = plan.kernel != 1 ? (int32_t)((L + plan.block_q - 1) / plan.block_q) : 0;
    // groups per L2 bunch (see the kernels' block mapping): a bunch's K+V ~ 8 MiB measured best at
    // L=4096, D=64 (bunch 1/2/4/8/16 -> 737/769/800/828/811 TFLOP/s)
    {
        const int64_t kv_bytes = 2 * L * D * elt_size(dtype);
        int64_t bunch = kv_bytes > 0 ? (8ll << 20) / kv_bytes : 1;
        const char* sched = getenv("FA_AMD_SCHED");
        if (sched && *sched) bunch = atoi(sched);
        const int64_t ng = p.G8 / 8;
        if (bunch < 1) bunch = 1;
        if (bunch > ng) bunch = ng;
        p.sched = (int32_t)bunch;
    }
    p.scale = softmax_scale;
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    p.skipmax = skipmax;
    hipError_t e = plan.kernel == 1    ? fa::launch_fwd_generic(p, dtype, (hipStream_t)stream)
                   : plan.kernel == 3 ? fa::launch_fwd_f32(p, (hipStream_t)stream)
                   : kc.which == 'p' ? fa::launch_fwd_pipe(p, dtype, plan.waves, kc.qb, (hipStream_t)stream)
                   : kc.which == 'd' ? fa::launch_fwd_dma(p, dtype, plan.waves, kc.qb, (hipStream_t)stream)
                                     : fa::launch_fwd_tile(p, dtype, plan.waves, (hipStream_t)stream);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

int fa_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
           void* dq, void* dk, void* dv, float* dsum_workspace,
           int64_t G, int64_t L, int64_t D, int64_t heads, const int64_t strides[8][3],
           int32_t dtype, float softmax_scale, int32_t causal, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !dsum_workspace || !strides)
        return fail(FA_ERR_NULL_POINTER, "fa_bwd: every pointer argument must be non-null");
    if (!causal) return fail(FA_ERR_NOT_CAUSAL, "only causal attention is supported (reference: sdpa.py:29-32)");
    if (G < 1 || L < 1 || D < 1) return fail(FA_ERR_BAD_SHAPE, "G, L, D must be >= 1");
    if (heads < 1 || G % heads != 0) return fail(FA_ERR_BAD_SHAPE, "G=%lld must be a positive multiple of heads=%lld", (long long)G, (long long)heads);
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    if (L > (1ll << 30) || G > (1ll << 30)) return fail(FA_ERR_TOO_LARGE, "G and L must be < 2^30");
    const void* const ptrs[8] = {q, k, v, out, dout, dq, dk, dv};
    // MFMA kernels: 16-bit dtypes, head_dim 64/128, 16-byte aligned rows; everything else -> generic kernels
    bool mfma = (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16) && (D == 64 || D == 128);
    for (int i = 0; i < 8; ++i) {
        if (reinterpret_cast<uintptr_t>(ptrs[i]) % elt_size(dtype)) return fail(FA_ERR_MISALIGNED, "fa_bwd: tensor %d is not element aligned", i);
        if (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) mfma = false;
        for (int j = 0; j < 3; ++j) {
            if (strides[i][j] < 0) return fail(FA_ERR_BAD_STRIDE, "negative strides are not supported");
            if (strides[i][j] % 8) mfma = false;
        }
        if (L * strides[i][2] >= (1ll << 31)) mfma = false;
    }
    if (!mfma && D > 1024) return fail(FA_ERR_BAD_SHAPE, "generic backward supports D <= 1024 (got %lld)", (long long)D);
    if (!mfma && G * L > 0x7fffffffll) return fail(FA_ERR_TOO_LARGE, "grid exceeds 2^31-1 workgroups");
    fa::BwdParams p;
    p.q = q; p.k = k; p.v = v; p.o = out; p.dout = dout; p.lse = lse; p.dsum = dsum_workspace;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.q_sb = strides[0][0]; p.q_sh = strides[0][1]; p.q_sl = strides[0][2];
    p.k_sb = strides[1][0]; p.k_sh = strides[1][1]; p.k_sl = strides[1][2];
    p.v_sb = strides[2][0]; p.v_sh = strides[2][1]; p.v_sl = strides[2][2];
    p.o_sb = strides[3][0]; p.o_sh = strides[3][1]; p.o_sl = strides[3][2];
    p.do_sb = strides[4][0]; p.do_sh = strides[4][1]; p.do_sl = strides[4][2];
    p.dq_sb = strides[5][0]; p.dq_sh = strides[5][1]; p.dq_sl = strides[5][2];
    p.dk_sb = strides[6][0]; p.dk_sh = strides[6][1]; p.dk_sl = strides[6][2];
    p.dv_sb = strides[7][0]; p.dv_sh = strides[7][1]; p.dv_sl = strides[7][2];
    p.G = (int32_t)G; p.G8 = (int32_t)((G + 7) / 8 * 8); p.H = (int32_t)heads; p.L = (int32_t)L; p.D = (int32_t)D;
    p.nq_tiles = (int32_t)((L + 127) / 128);
    p.nk_tiles = (int32_t)((L + 127) / 128);
    {
        int64_t bunch = (8ll << 20) / (2 * L * D * 2);
        if (const char* sched = getenv("FA_AMD_SCHED")) { if (*sched) bunch = atoi(sched); }
        const int64_t ng = p.G8 / 8;
        if (bunch < 1) bunch = 1;
        if (bunch > ng) bunch = ng;
        p.sched = (int32_t)bunch;
    }
    if ((int64_t)p.G8 * p.nq_tiles > 0x7fffffffll) return fail(FA_ERR_TOO_LARGE, "grid exceeds 2^31-1 workgroups");
    p.scale = softmax_scale;
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    hipError_t e = mfma ? fa::launch_bwd(p, dtype, (hipStream_t)stream) : fa::launch_bwd_generic(p, dtype, (hipStream_t)stream);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

int fa_fwd_qkv_packed(const void* qkv, void* out, float* lse, int64_t G, int64_t L, int64_t D,
                      int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype, void* stream) {
    if (!qkv || !out) return fail(FA_ERR_NULL_POINTER, "qkv and out must be non-null");
    if (G < 1 || L < 1 || D < 1) return fail(FA_ERR_BAD_SHAPE, "G, L, D must be >= 1");
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    const int64_t es = elt_size(dtype);
    const int64_t strides[3] = {L * D, 0, D};
    const char* base = static_cast<const char*>(qkv);
    return fa_fwd(base, base + G * L * D * es, base + 2 * G * L * D * es, out, lse, G, L, D, 1,
                  strides, strides, strides, strides, num_chunks_q, num_chunks_kv, dtype, 1.0f, 1, stream);
}

}  // extern "C"
