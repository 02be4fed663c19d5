// Host-only test of the C ABI's argument validation and plan selection, built with
// -fsanitize=address,undefined against tests/cpp/hip_stubs.cpp (no GPU, no HIP runtime).  The reference's
// counterparts are C++ asserts (flash_attention_qkv_packed.cpp:183, :190-191, :573-574); here every bad argument
// must come back as an error code with a message, never as a crash, an out-of-bounds read or signed overflow.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

#include "flash_attention_amd.h"

extern int g_stub_launches;
extern char g_stub_last;

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s   (last error: %s)\n", __FILE__, __LINE__, #cond, fa_last_error()); ++fails; } } while (0)

int main() {
    EXPECT(fa_abi_version() == FA_ABI_VERSION);
    // ---- shape inference
    int64_t shp[4] = {3, 8, 128, 64}, out[3] = {0, 0, 0};
    EXPECT(fa_infer_output_shape(shp, 4, out) == FA_OK && out[0] == 8 && out[1] == 128 && out[2] == 64);
    EXPECT(fa_infer_output_shape(shp, 3, out) == FA_ERR_BAD_SHAPE);
    EXPECT(fa_infer_output_shape(nullptr, 4, out) == FA_ERR_NULL_POINTER);
    // ---- plan: every error code, extreme sizes (no overflow: UBSan watches)
    fa_plan_t plan;
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 4, 1, 1.0f, 1, &plan) == FA_ERR_CHUNKS_Q);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 1, 4, 1.0f, 1, &plan) == FA_ERR_CHUNKS_KV);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 0, 1, 1.0f, 1, &plan) == FA_ERR_CHUNKS_Q);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, -3, 1, 1.0f, 1, &plan) == FA_ERR_CHUNKS_Q);
    EXPECT(fa_plan(0, 6, 2, FA_DTYPE_F32, 1, 1, 1.0f, 1, &plan) == FA_ERR_BAD_SHAPE);
    EXPECT(fa_plan(1, 6, 2, 7, 1, 1, 1.0f, 1, &plan) == FA_ERR_BAD_DTYPE);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 1, 1, 0.0f, 1, &plan) == FA_ERR_BAD_SCALE);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 1, 1, -1.0f, 1, &plan) == FA_ERR_BAD_SCALE);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 1, 1, std::numeric_limits<float>::infinity(), 1, &plan) == FA_ERR_BAD_SCALE);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 1, 1, std::nanf(""), 1, &plan) == FA_ERR_BAD_SCALE);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 1, 1, 1.0f, 5, &plan) == FA_ERR_NOT_CAUSAL);
    EXPECT(fa_plan(1, 6, 2, FA_DTYPE_F32, 1, 1, 1.0f, 1, nullptr) == FA_ERR_NULL_POINTER);
    EXPECT(fa_plan((1ll << 30) + 1, 64, 64, FA_DTYPE_BF16, 1, 1, 1.0f, 1, &plan) == FA_ERR_TOO_LARGE);
    EXPECT(fa_plan(8, (1ll << 30) + 8, 64, FA_DTYPE_BF16, 1, 1, 1.0f, 1, &plan) == FA_ERR_TOO_LARGE);
    EXPECT(fa_plan(1ll << 30, 1ll << 30, 64, FA_DTYPE_BF16, 1, 1, 1.0f, 1, &plan) == FA_ERR_TOO_LARGE);   // grid > 2^31-1
    EXPECT(fa_plan(1ll << 62, 1ll << 62, 1ll << 62, FA_DTYPE_F32, 1, 1, 1.0f, 1, &plan) == FA_ERR_TOO_LARGE);
    EXPECT(fa_plan(4, 256, 2048, FA_DTYPE_BF16, 1, 1, 1.0f, 1, &plan) == FA_ERR_BAD_SHAPE);              // generic kernel: D <= 1024
    EXPECT(fa_plan(128, 4096, 64, FA_DTYPE_BF16, 8, 8, 1.0f, 1, &plan) == FA_OK && plan.block_q * plan.grid >= 128 * 4096);
    EXPECT(fa_plan(128, 4096, 64, FA_DTYPE_BF16, 8, 8, 1.0f, 0, &plan) == FA_OK && (plan.kernel == 0 || plan.kernel == 2));   // non-causal: an MFMA kernel
    for (int32_t nq : {1, 2, 4, 8, 16, 32, 64, 128, 4096})       // every chunking of L = 4096 yields a legal tile
        for (int32_t nk : {1, 8, 64, 128, 4096})
            for (int dt : {FA_DTYPE_F16, FA_DTYPE_BF16, FA_DTYPE_F32})
                for (int64_t D : {64, 128, 80}) {
                    EXPECT(fa_plan(24, 4096, D, dt, nq, nk, 0.125f, 1, &plan) == FA_OK);
                    EXPECT(plan.block_q >= 1 && plan.grid >= 1 && plan.waves >= 1 && plan.sched >= 1);
                    if (plan.kernel == 0 || plan.kernel == 2) EXPECT(plan.block_q == 32 * plan.waves * plan.qb && (plan.block_q <= 4096 / nq || plan.block_q == 32));
                }
    // ---- the plan grid: every request x every debug override must come out as something the kernels can launch (round-3 verdict,
    // item 7: select_plan = rules -> table -> overrides -> ONE constraints pass; these are the invariants of that pass)
    {
        struct Ov { const char* name; int32_t value; };
        const std::vector<std::vector<Ov>> overrides = {
            {}, {{"kernel", 'd'}}, {{"kernel", 'p'}}, {{"kernel", 'x'}}, {{"kernel", 'h'}}, {{"waves", 1}}, {{"waves", 2}}, {{"waves", 4}}, {{"waves", 8}},
            {{"qb", 1}}, {{"qb", 2}}, {{"split", 2}}, {{"split", 4}}, {{"table", 0}}, {{"pair", 1}}, {{"balance", 1}}, {{"balance", 0}}, {{"sched", 3}},
            {{"skipmax", 0}}, {{"skipmax", 1}}, {{"kernel", 'p'}, {"qb", 2}}, {{"kernel", 'x'}, {"qb", 2}}, {{"kernel", 'x'}, {"waves", 8}},
            {{"kernel", 'h'}, {"split", 2}}, {{"kernel", 'h'}, {"split", 4}}, {{"kernel", 'd'}, {"split", 4}}, {{"kernel", 'p'}, {"waves", 8}},
            {{"kernel", 'd'}, {"qb", 2}}, {{"kernel", 'h'}, {"balance", 1}}, {{"kernel", 'd'}, {"pair", 1}}, {{"f32_generic", 1}}};
        const char* names[] = {"kernel", "qb", "waves", "sched", "skipmax", "f32_generic", "table", "split", "pair", "balance", "subsets"};
        const int32_t autos[] = {0, 0, 0, 0, -1, 0, 1, 0, -1, -1, -1};
        long checked = 0;
        for (const auto& ov : overrides) {
            for (int i = 0; i < 11; ++i) EXPECT(fa_debug_set_option(names[i], autos[i]) == FA_OK);
            for (const Ov& o : ov) EXPECT(fa_debug_set_option(o.name, o.value) == FA_OK);
            for (int64_t G : {1, 7, 8, 12, 96, 128, 2048})
                for (int64_t L : {1, 6, 32, 100, 256, 1000, 1024, 2048, 4096, 32768})
                    for (int64_t D : {2, 64, 96, 128, 160, 200, 256})
                        for (int dt : {FA_DTYPE_F16, FA_DTYPE_BF16, FA_DTYPE_F32})
                            for (int32_t nc : {1, 2, 8, 64})
                                for (float scale : {1.0f, 0.125f})
                                    for (int causal : {0, 1}) {
                                        if (L % nc) continue;
                                        fa_plan_t pl;
                                        if (fa_plan(G, L, D, dt, nc, nc == 8 ? 4 : nc, scale, causal, &pl) != FA_OK) { EXPECT(L % (nc == 8 ? 4 : nc) != 0); continue; }
                                        ++checked;
                                        const bool mfma16 = (D == 64 || D == 128 || D == 96 || D == 160 || D == 200 || D == 256) && dt != FA_DTYPE_F32;    // (96 / 160 / 200: zero-extended at 128 / 192 / 256)
                                        const int64_t Dk = D == 96 ? 128 : D == 160 ? 192 : D == 200 ? 256 : D;
                                        const int64_t chunk = L / nc;
                                        EXPECT(pl.grid >= 1 && pl.grid <= 0x7fffffffll && pl.waves >= 1 && pl.sched >= 1 && pl.block_q >= 1 && pl.block_kv >= 1);
                                        EXPECT(pl.lds_bytes > 0 && pl.lds_bytes <= 160 * 1024);                  // one workgroup must fit a CU's LDS
                                        EXPECT(pl.chunk_q == chunk);
                                        if (!mfma16) { EXPECT(pl.kernel == 1 || (pl.kernel == 3 && (D == 64 || D == 128))); continue; }
                                        if (Dk > 128) EXPECT(((pl.kernel == 0 && pl.lds_bytes == (Dk == 192 ? 99072 : 132096)) || (pl.kernel == 2 && Dk == 256 && pl.waves == 4 && pl.skipmax == 1 && pl.lds_bytes == 133136))
                                                            && (pl.waves == 4 || pl.waves == 8) && pl.qb == 1 && pl.block_q == 32 * pl.waves && !pl.pair);   // the 64-key-tile kernel unsplit, or (256, skip-max) the pipelined kernel's 4-wave form - whatever was forced
                                        EXPECT(pl.kernel == 0 || pl.kernel == 2 || pl.kernel == 4 || pl.kernel == 6);   // (plan kernel 5 was retired with ABI 6)
                                        EXPECT(pl.qb == 1 || pl.qb == 2);
                                        EXPECT(pl.waves == 1 || pl.waves == 2 || pl.waves == 4 || pl.waves == 8 || pl.waves == 16);
                                        EXPECT(pl.waves <= 16 && 64 * pl.waves <= 1024);                            // a workgroup is at most 1024 threads
                                        EXPECT(pl.skipmax == (dt == FA_DTYPE_BF16 || pl.skipmax));                  // bf16 always runs the skip-max arithmetic
                                        if (pl.kernel == 2 || pl.kernel == 4) EXPECT(pl.skipmax == 1 && pl.block_kv == 32);
                                        if (pl.kernel == 6) { EXPECT(pl.block_q == 64 && pl.qb == 1 && (pl.waves == 4 || pl.waves == 8 || (pl.waves == 16 && Dk == 64))); }
                                        else {
                                            const int64_t rows = 32ll * pl.waves * pl.qb;
                                            EXPECT(rows % pl.block_q == 0);
                                            const int64_t split = rows / pl.block_q;
                                            EXPECT(split == 1 || ((split == 2 || (split == 4 && Dk == 64)) && pl.kernel == 0 && pl.qb == 1));
                                        }
                                        if (pl.qb == 2) EXPECT(Dk == 64 && (pl.kernel == 2 || pl.kernel == 4 || pl.kernel == 0) && pl.block_q == 256);
                                        // an unforced plan honours the q chunk as an upper bound on the workgroup's rows
                                        if (ov.empty()) EXPECT(pl.block_q <= (chunk > 32 ? chunk : 32) || pl.kernel == 6 || Dk > 128);   // (head_dim 192 / 256 have the one 128-row form)
                                        // the grid covers every (group, q tile): 8-aligned groups x tiles (pairs: two tiles per workgroup)
                                        const int64_t g8 = (G + 7) / 8 * 8, tiles = (L + pl.block_q - 1) / pl.block_q;
                                        // (q-tile subsets where G wastes an eighth of the XCDs or more: 8-aligned (group, subset) pairs x tiles per pair)
                                        EXPECT(pl.xs == 1 || pl.xs == 2 || pl.xs == 4 || pl.xs == 8);
                                        if ((g8 - G) * 8 < g8) EXPECT(pl.xs == 1);
                                        if (pl.xs > 1) { const int64_t gv = G * pl.xs, gv8 = (gv + 7) / 8 * 8; EXPECT((gv8 - gv) * 100 < 9 * gv8 && (L + 255) / 256 >= pl.xs); }
                                        if (pl.xs > 1) { EXPECT(!pl.pair && !pl.balanced); EXPECT(pl.grid == ((G * pl.xs + 7) / 8 * 8) * ((tiles + pl.xs - 1) / pl.xs)); }
                                        else EXPECT(pl.grid == (pl.pair ? g8 * ((tiles + 1) / 2) : g8 * tiles));
                                        if (pl.pair) EXPECT(pl.kernel == 0 && causal == 1 && pl.qb == 1);
                                        if (pl.balanced) EXPECT(causal == 1 && (pl.kernel == 0 || pl.kernel == 6));
                                    }
        }
        for (int i = 0; i < 11; ++i) EXPECT(fa_debug_set_option(names[i], autos[i]) == FA_OK);
        EXPECT(checked > 100000);
        EXPECT(fa_debug_set_option("kernel", 't') == FA_ERR_BAD_OPTION && fa_debug_set_option("kernel", 'P') == FA_ERR_BAD_OPTION);   // (retired with ABI 6; round 6's team-list kernel borrowed 't' for one commit: profiles/r06_tl)
    }
    // ---- debug options
    int32_t v = -7;
    EXPECT(fa_debug_set_option(nullptr, 1) == FA_ERR_NULL_POINTER);
    EXPECT(fa_debug_set_option("bogus", 1) == FA_ERR_BAD_OPTION);
    EXPECT(fa_debug_set_option("waves", 3) == FA_ERR_BAD_OPTION);
    EXPECT(fa_debug_set_option("kernel", 'p') == FA_OK && fa_debug_get_option("kernel", &v) == FA_OK && v == 'p');
    EXPECT(fa_plan(128, 2048, 64, FA_DTYPE_BF16, 8, 8, 1.0f, 1, &plan) == FA_OK && plan.kernel == 2 && plan.source == 2);
    EXPECT(fa_debug_set_option("kernel", 0) == FA_OK);
    EXPECT(fa_debug_get_option("kernel", nullptr) == FA_ERR_NULL_POINTER);
    // ---- fa_fwd / fa_bwd argument checks (pointers are never dereferenced on the host; the launchers are stubs)
    alignas(16) static char buf[64];
    const int64_t st[3] = {128 * 64, 0, 64}, neg[3] = {-1, 0, 64}, odd[3] = {128 * 64, 0, 65};
    void* p = buf;
    EXPECT(fa_fwd(nullptr, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_NULL_POINTER);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, nullptr, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_NULL_POINTER);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 3, 128, 64, 2, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_SHAPE);    // G % heads
    EXPECT(fa_fwd(p, p, p, p, nullptr, 6, 128, 64, 6, 4, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_SHAPE);    // heads % heads_kv
    EXPECT(fa_fwd(p, p, p, p, nullptr, 6, 128, 64, 6, 0, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_SHAPE);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, neg, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_STRIDE);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, 9, 1.0f, 1, nullptr) == FA_ERR_BAD_DTYPE);
    EXPECT(fa_fwd(buf + 1, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_MISALIGNED);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 0.0f, 1, nullptr) == FA_ERR_BAD_SCALE);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 2, nullptr) == FA_ERR_NOT_CAUSAL);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 3, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_CHUNKS_Q);
    const int64_t huge[3] = {1ll << 62, 0, 1ll << 62};
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1ll << 62, 1ll << 62, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_TOO_LARGE);
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 1ll << 30, 64, 1, 1, st, huge, huge, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_STRIDE);  // row offsets would overflow
    const int launches0 = g_stub_launches;
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && (g_stub_last == 'd' || g_stub_last == 'p'));   // an MFMA kernel (table or rules)
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, odd, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && g_stub_last == 'g');   // unaligned rows -> generic
    EXPECT(fa_fwd(buf + 2, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && g_stub_last == 'g');
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_F32, 1.0f, 0, nullptr) == FA_OK && g_stub_last == 'f');
    EXPECT(fa_fwd(p, p, p, p, nullptr, 1024, 4096, 64, 1, 1, st, st, st, st, 8, 8, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && (g_stub_last == 'p' || g_stub_last == 'd'));
    EXPECT(fa_fwd_qkv_packed(nullptr, p, nullptr, 1, 8, 64, 1, 1, FA_DTYPE_BF16, nullptr) == FA_ERR_NULL_POINTER);
    EXPECT(fa_fwd_qkv_packed(p, p, nullptr, 0, 8, 64, 1, 1, FA_DTYPE_BF16, nullptr) == FA_ERR_BAD_SHAPE);
    EXPECT(fa_fwd_qkv_packed(p, p, nullptr, 1, 8, 64, 1, 1, 5, nullptr) == FA_ERR_BAD_DTYPE);
    // sizes are bounded before the packed pointer arithmetic (G * L * D * es must not overflow: UBSan would trap here)
    EXPECT(fa_fwd_qkv_packed(p, p, nullptr, 1ll << 62, 8, 64, 1, 1, FA_DTYPE_BF16, nullptr) == FA_ERR_TOO_LARGE);
    EXPECT(fa_fwd_qkv_packed(p, p, nullptr, 1, 1ll << 62, 64, 1, 1, FA_DTYPE_BF16, nullptr) == FA_ERR_TOO_LARGE);
    EXPECT(fa_fwd_qkv_packed(p, p, nullptr, 1, 8, 1ll << 62, 1, 1, FA_DTYPE_F32, nullptr) == FA_ERR_TOO_LARGE);
    EXPECT(fa_fwd_qkv_packed(p, p, nullptr, 1ll << 30, 1ll << 30, 1ll << 20, 1, 1, FA_DTYPE_F32, nullptr) == FA_ERR_TOO_LARGE);   // every size in range, the product not
    EXPECT(fa_debug_set_stamp_buffer(nullptr, 8) == FA_ERR_BAD_OPTION && fa_debug_set_stamp_buffer(buf + 1, 64) == FA_ERR_MISALIGNED);
    EXPECT(fa_debug_set_stamp_buffer(buf, 64) == FA_OK && fa_debug_set_stamp_buffer(nullptr, 0) == FA_OK);
    EXPECT(g_stub_launches == launches0 + 5);
    {   // a zero-extended head dim (96 runs at 128) marks the chunks past it with the LDS-DMA offset 0x80000000: the MFMA kernels take the call
        // only while a group's K / V extent is below 2^31 BYTES; a head-inner layout with a longer extent goes to the generic kernel (round-4 advisor)
        const int64_t inner96[3] = {262144ll * 6144, 96, 6144}, short96[3] = {4096ll * 6144, 96, 6144};
        EXPECT(fa_fwd(p, p, p, p, nullptr, 64, 4096, 96, 64, 64, short96, short96, short96, short96, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && g_stub_last != 'g');
        EXPECT(fa_fwd(p, p, p, p, nullptr, 64, 262144, 96, 64, 64, inner96, inner96, inner96, inner96, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && g_stub_last == 'g');
        const int64_t inner128[3] = {131072ll * 8192, 128, 8192};     // (a native head dim keeps the 2^31-ELEMENT bound)
        EXPECT(fa_fwd(p, p, p, p, nullptr, 64, 131072, 128, 64, 64, inner128, inner128, inner128, inner128, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && g_stub_last != 'g');
    }
    int64_t st8[8][3];
    for (auto& r : st8) { r[0] = 128 * 64; r[1] = 0; r[2] = 64; }
    float fbuf[4];
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && g_stub_last == 'B');
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_F32, 1.0f, 0, nullptr) == FA_OK && g_stub_last == 'F');
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 80, 1, 1, st8, FA_DTYPE_F32, 1.0f, 0, nullptr) == FA_OK && g_stub_last == 'b');
    EXPECT(fa_bwd(p, p, p, p, p, nullptr, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_NULL_POINTER);
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, -2.0f, 1, nullptr) == FA_ERR_BAD_SCALE);
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, 1.0f, 3, nullptr) == FA_ERR_NOT_CAUSAL);
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 3, 128, 64, 2, 2, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_SHAPE);
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 6, 128, 64, 6, 4, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_SHAPE);   // heads % heads_kv
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 8, 128, 64, 4, 2, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_OK && g_stub_last == 'B');   // grouped-query
    // ---- mask / bias / dropout entry points (ABI 4): argument checks, routing, what the plain entry points still do
    {
        const int64_t ms[3] = {0, 0, 128}, ms_neg[3] = {0, -1, 128};
        const int n0 = g_stub_launches;
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_BOOL, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last == 'E');
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_F16, 0.125f, 0, p, FA_MASK_BIAS, ms, 0.25f, 7, 9, nullptr, nullptr) == FA_OK && g_stub_last == 'E');
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 0.1f, 1ull << 63, ~0ull, nullptr, nullptr) == FA_OK && g_stub_last == 'E');
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 0.0f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last != 'E');   // nothing extra = fa_fwd
        EXPECT(g_stub_launches == n0 + 4);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_NONE, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_MASK);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr, FA_MASK_BOOL, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_MASK);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, 3, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_MASK);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_BOOL, nullptr, 0.0f, 0, 0, nullptr, nullptr) == FA_ERR_NULL_POINTER);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_BOOL, ms_neg, 0.0f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_STRIDE);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, buf + 1, FA_MASK_BIAS, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_ERR_MISALIGNED);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 1.0f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_DROPOUT);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, -0.5f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_DROPOUT);
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 0.0f, 1, nullptr, FA_MASK_NONE, nullptr, 0.5f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_SCALE);
        // operands the MFMA mask kernels do not take go to the generic kernels (never to the fp32 MFMA kernel, which has no mask path)
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 64, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_F32, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 0.5f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last == 'g');
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 80, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_BOOL, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last == 'E');   // head_dim 80: zero-extended at 128 by the MFMA mask kernel
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 84, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_BOOL, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last == 'g');   // not a multiple of 8: generic
        EXPECT(fa_fwd_ex(p, p, p, p, nullptr, 1, 128, 2048, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_BOOL, ms, 0.0f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_SHAPE);   // (head_dim > 1024: as fa_fwd)
        EXPECT(fa_bwd_ex(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, 1.0f, 1, p, FA_MASK_BIAS, ms, 0.1f, 3, 4, nullptr, nullptr) == FA_OK && g_stub_last == 'e');
        EXPECT(fa_bwd_ex(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 0.0f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last == 'B');
        EXPECT(fa_bwd_ex(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_F32, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 0.3f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last == 'b');
        EXPECT(fa_bwd_ex(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_F32, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 0.0f, 0, 0, nullptr, nullptr) == FA_OK && g_stub_last == 'F');   // nothing extra: the fp32 MFMA kernels
        EXPECT(fa_bwd_ex(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr, FA_MASK_NONE, nullptr, 2.0f, 0, 0, nullptr, nullptr) == FA_ERR_BAD_DROPOUT);
    }
    st8[3][1] = -5;
    EXPECT(fa_bwd(p, p, p, p, p, fbuf, p, p, p, fbuf, 1, 128, 64, 1, 1, st8, FA_DTYPE_BF16, 1.0f, 1, nullptr) == FA_ERR_BAD_STRIDE);
    // ---- thread-local error text; concurrent plan + option calls (ThreadSanitizer-style smoke under ASan/UBSan)
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
        th.emplace_back([t] {
            fa_plan_t pl;
            for (int i = 0; i < 2000; ++i) {
                if (t == 0) fa_debug_set_option("sched", i % 5);
                const int rc = fa_plan(8 + t, 1024, 64, FA_DTYPE_BF16, (i & 1) ? 3 : 1, 1, 1.0f, 1, &pl);
                if (rc == FA_ERR_CHUNKS_Q && !std::strstr(fa_last_error(), "num_chunks_q=3")) std::abort();
            }
        });
    for (auto& x : th) x.join();
    fa_debug_set_option("sched", 0);
    for (int code = 1; code >= -17; --code) EXPECT(fa_error_string(code) != nullptr && std::strlen(fa_error_string(code)) > 0);
    std::printf(fails ? "FAILED (%d)\n" : "PASSED\n", fails);
    return fails ? 1 : 0;
}
