// C ABI of libflash_attention.so - see include/flash_attention_amd.h for the contract and for the
// reference interface (file:line) each entry point replaces.  No torch types, no exceptions, no host
// synchronisation, no allocation: validate, pick a tile plan, enqueue one kernel on the caller's stream.

#include "flash_attention_amd.h"

#include <math.h>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "fa_extras.h"
#include "fa_params.h"
#include "fa_tuning_table.h"

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

// Developer / test overrides of the tile plan.  They are NOT read per launch: the FA_AMD_* environment variables
// are parsed ONCE (first use), afterwards only fa_debug_set_option() changes them (tests force every kernel
// variant through it).  Every field is a relaxed atomic, 0 / -1 = "no override", so concurrent fa_fwd calls see
// either the old or the new value of one knob and never a torn one; fa_plan() and fa_fwd() go through the same
// make_plan(), so a reported plan is the launched plan.
struct Options {
    std::atomic<int> kernel{0};     // 0 auto, 'd' | 'p' | 'x' | 'h'
    std::atomic<int> qb{0};         // 0 auto, 1 | 2 query blocks per wave
    std::atomic<int> waves{0};      // 0 auto, 1 | 2 | 4 | 8
    std::atomic<int> sched{0};      // 0 auto, else groups per L2 bunch
    std::atomic<int> skipmax{-1};   // -1 auto, 0 | 1
    std::atomic<int> f32_generic{0};
    std::atomic<int> table{1};      // 1 = consult the measured tuning table (fa_tuning_table.h), 0 = built-in rules only
    std::atomic<int> split{0};      // 0 auto, 1 | 2 | 4 kv split groups per workgroup (fa_fwd_dma.hip)
    std::atomic<int> pair{-1};      // -1 auto, 0 | 1 q-tile pairs per workgroup (fa_fwd_dma.hip, causal)
    std::atomic<int> balance{-1};   // -1 auto, 0 | 1 placement-balanced block order for grids below one round of slots
    std::atomic<int> subsets{-1};   // -1 auto, 0 off: q-tile subsets for group counts that leave XCDs idle (FwdParams::xs)
    std::atomic<int> bbal{-1};      // -1 auto, 0 off, 1 forced where the form exists: balanced dQ units (BwdParams::bal)
    std::atomic<int> bsplit{-1};    // -1 auto, 0 off, 2 | 3: split groups inside the backward's workgroups (BwdParams::qs_q / qs_kv), forced where a kernel has that form
};
Options g_opt;
std::once_flag g_opt_once;
// measurement builds (-DFA_STAMPS): device buffer that receives the phase stamps, and its size in bytes
std::atomic<void*> g_stamp_buf{nullptr};
std::atomic<int64_t> g_stamp_bytes{0};

int parse_kernel(const char* v) {   // dma | pipe | pipe16 | kh
    if (!v) return 0;
    if (strcmp(v, "kh") == 0 || v[0] == 'h') return 'h';
    if (strcmp(v, "pipe16") == 0 || v[0] == 'x') return 'x';
    return (v[0] == 'd' || v[0] == 'p') ? v[0] : 0;
}

const Options& options() {
    std::call_once(g_opt_once, [] {
        if (const char* e = getenv("FA_AMD_KERNEL")) g_opt.kernel = parse_kernel(e);
        if (const char* e = getenv("FA_AMD_QB")) g_opt.qb = atoi(e);
        if (const char* e = getenv("FA_AMD_WAVES")) g_opt.waves = atoi(e);
        if (const char* e = getenv("FA_AMD_SCHED")) g_opt.sched = atoi(e);
        if (const char* e = getenv("FA_AMD_SPLIT")) g_opt.split = atoi(e);
        if (const char* e = getenv("FA_AMD_SKIPMAX")) { if (*e) g_opt.skipmax = atoi(e); }
        if (const char* e = getenv("FA_AMD_F32")) g_opt.f32_generic = e[0] == 'g';
        if (const char* e = getenv("FA_AMD_TABLE")) { if (*e) g_opt.table = atoi(e); }
        if (const char* e = getenv("FA_AMD_PAIR")) { if (*e) g_opt.pair = atoi(e); }
        if (const char* e = getenv("FA_AMD_BALANCE")) { if (*e) g_opt.balance = atoi(e); }
        if (const char* e = getenv("FA_AMD_SUBSETS")) { if (*e) g_opt.subsets = atoi(e); }
        if (const char* e = getenv("FA_AMD_BSPLIT")) { if (*e) g_opt.bsplit = atoi(e); }
        if (const char* e = getenv("FA_AMD_BBAL")) { if (*e) g_opt.bbal = atoi(e); }
    });
    return g_opt;
}

// num_chunks -> hardware tile.  The reference's chunk sizes (L/num_chunks, FA.cpp:193-195) are
// upper bounds on the scheduling tile: the result is independent of them (semantic
// contract), so a chunk larger than one workgroup can hold is subdivided into hardware tiles, and a
// chunk that IS a hardware tile size is honoured exactly.  q tiles: 32 rows per wave and query block, 1..8 waves;
// kv steps: 64-key tiles or 32-key pipelined steps.
// The base tile is 4 waves = 128 rows (4 workgroups per CU at D=64, 2 at D=128): fastest on short sequences and
// grids that under-fill the chip; long sequences on full grids get 256-row workgroups (two query blocks per
// wave at head_dim 64, 8 waves at head_dim 128).
int waves_for_chunk(int64_t want) { return want >= 128 ? 4 : want >= 64 ? 2 : 1; }

// Head dims the MFMA kernels serve: 64 and 128 natively (16-bit and fp32), 192 and 256 for 16-bit operands (one workgroup per CU: the 64-key-tile
// kernels, fa_fwd_dma.hip / fa_bwd_kernels.hip, and at 256 the pipelined forward's 4-wave form); for 16-bit operands every other multiple of 8 up to 256 runs ZERO-EXTENDED
// at the next of the four (FwdParams::Dr: out-of-range LDS-DMA chunks and predicated register loads read the missing columns as zeros,
// stores stop at the true head dim) - the reference's op has no head-dim restriction (FA.cpp:175-195).  Returns 0 when there is no such kernel.
int64_t kernel_head_dim(int64_t D, int32_t dtype) {
    if (D == 64 || D == 128) return D;
    if ((dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16) && D >= 8 && D <= 256 && D % 8 == 0) return D <= 64 ? 64 : D <= 128 ? 128 : D <= 192 ? 192 : 256;
    return 0;
}

bool aligned16(int64_t D, int elt, const void* const ptrs[4], const int64_t* const strides[4]) {
    if (!(D == 64 || D == 128 || D == 192 || D == 256)) return false;
    for (int i = 0; i < 4; ++i) {
        if (ptrs[i] && (reinterpret_cast<uintptr_t>(ptrs[i]) & 15)) return false;
        if (strides[i]) {
            for (int j = 0; j < 3; ++j)
                if (strides[i][j] % (16 / elt)) return false;   // 16-byte rows
        }
    }
    return true;
}

// The ONE selection function behind fa_plan() and fa_fwd().
//   'd' fa_fwd_dma_kernel   64-key tiles, LDS-DMA staged (fa_fwd_dma.hip)
//   'p' fa_fwd_pipe_kernel  software-pipelined 32-key steps (fa_fwd_pipe.hip); skip-max arithmetic only
//   'x' fa_fwd_pipe16_kernel the pipelined kernel on v_mfma_f32_16x16x32 (fa_fwd_pipe16.hip): head_dim 128 with 4 | 8 waves and qb 1,
//       head_dim 64 with 4 waves and qb 1 | 2
//   'h' fa_fwd_kh_kernel    64-row workgroups, every wave = (row block, key half) of the streamed 64-key tiles (fa_fwd_kh.hip): half the
//       dependency chain per tile - for causal launches bound by their heaviest workgroup's kv loop (grids of about one round of slots)
// qb = 32-row query blocks per wave (2 = 64 rows per wave, head_dim 64 only).
// split = kv split groups of a 'd' workgroup (4 waves x 32 rows each, all on the same 128 query rows; fa_fwd_dma.hip): for
// grids with fewer workgroups than the chip has slots, where the kv loop of the longest q tile is the launch's critical path.
// Order: debug overrides > measured table (fa_tuning_table.h) > built-in rules; the num_chunks_* bounds and the
// kernels' own constraints are enforced on whatever was picked.
struct Selection {
    fa_plan_t plan;
    char which;      // 'd' | 'p' | 'x' | 'h' | 'g' generic | 'f' fp32 MFMA
    int split;       // kv split groups per workgroup ('d' only; plan.waves counts all of the workgroup's waves)
    int pair;        // 'd' only: 1 = a workgroup owns q tiles (nq - 1 - i) and i, one after the other
    int perm_n;      // 'd' | 'h': > 0 = placement-balanced order of the XCD's first perm_n slots
    uint8_t perm[128];
    int xs;          // q-tile subsets per group (FwdParams::xs); 1 = groups map to XCDs whole
};

// Groups go to XCDs by index mod 8 (a group's K/V then live in ONE XCD's L2).  A group count that is not a multiple of 8 leaves XCDs idle
// (G = 2: six of eight; measured [2,32768,64]: 0.11 of peak) or unevenly loaded (G = 12: 2 + 1): where that wastes an eighth of the chip or more
// the launch is laid out over G xs (group, q-tile subset) pairs instead, xs = 8 / gcd(G, 8) - a multiple of 8 pairs, every XCD the same
// number; a group's K/V are then fetched into xs L2s, which is what the idle XCDs' bandwidth is there for.
// xs = the smallest of 2, 4, 8 whose G xs pairs waste less than 9 % - unless the pairs an XCD then holds no longer share its L2: a group's K/V
// (kv_bytes) are fetched by every XCD that holds one of its subsets, and an XCD with more than ~8 MB of K/V in flight re-fetches them per q tile
// (measured, profiles/r05_subsets: [25,2048,64] with 8 subsets 26.6 -> 31.4 us, [7,4096,128] 60.3 -> 64.0 us); a group whose K/V exceed the
// 4 MB L2 by themselves streams either way.
int subsets_for(int64_t G, int64_t L, int64_t kv_bytes) {
    const int64_t g8 = (G + 7) / 8 * 8;
    if ((g8 - G) * 8 < g8) return 1;          // (less than an eighth of the XCD slots idle: whole groups - [36,1024,64], 10 %, lost 4 % to subsets)
    for (int xs = 2; xs <= 8; xs *= 2) {
        const int64_t gv = G * xs, gv8 = (gv + 7) / 8 * 8;
        if ((gv8 - gv) * 100 >= 9 * gv8) continue;
        if ((L + 255) / 256 < xs) return 1;                    // (every subset gets a tile of the widest workgroup)
        const int64_t per_xcd = gv8 / 8;
        return (kv_bytes >= (4ll << 20) || per_xcd * kv_bytes <= (8ll << 20)) ? xs : 1;
    }
    return 1;
}

// Placement-balanced block order for a causal 'd' launch whose workgroups are all resident at once (a grid of at most one
// round of slots: e.g. GPT-2 small's attention call, 96 groups x 8 q tiles = 768 workgroups on 1024 slots).  Nothing is
// refilled in such a launch, so the dispatcher's list scheduling cannot even out the 2..16-tile workgroups: the launch lasts as
// long as the CU with the largest sum of kv tiles.  Observed placement on MI355X (scripts/phase_times.py, profiles/r03_cfg4:
// speed only, never correctness - any order computes the same result): the blocks of one XCD, b = 8 s + x, go to its 32 CUs
// round-robin, slot s -> CU s mod 32; with heaviest-first order CU 0 gets {16, 12, 6} tiles = 34 and CU 31 {12, 6, 2} = 20.
// Greedy longest-processing-time over the CUs (capacity = that CU's slot count) gives 26..28 on every CU.
//   weight(slot)  kv tiles of the natural order's work item `slot`;  perm[hardware slot] = natural slot
constexpr int kCusPerXcd = 32;
int balanced_order(int n_slots, const int* weight, uint8_t* perm) {
    if (n_slots <= kCusPerXcd || n_slots > 128) return 0;
    int order[128], load[kCusPerXcd] = {0}, cap[kCusPerXcd], cnt[kCusPerXcd] = {0};
    int items[kCusPerXcd][4];
    for (int c = 0; c < kCusPerXcd; ++c) cap[c] = (n_slots - c + kCusPerXcd - 1) / kCusPerXcd;
    for (int i = 0; i < n_slots; ++i) order[i] = i;
    for (int i = 1; i < n_slots; ++i) {          // insertion sort by weight, descending, stable
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && weight[order[j]] < weight[v]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
    for (int i = 0; i < n_slots; ++i) {
        int best = -1;
        for (int c = 0; c < kCusPerXcd; ++c)
            if (cnt[c] < cap[c] && (best < 0 || load[c] < load[best])) best = c;
        items[best][cnt[best]++] = order[i];
        load[best] += weight[order[i]];
    }
    for (int c = 0; c < kCusPerXcd; ++c)
        for (int k = 0; k < cnt[c]; ++k) perm[c + kCusPerXcd * k] = (uint8_t)items[c][k];   // a CU's heaviest item in its first round
    return n_slots;
}

// ---- the selection, in four steps that each do one thing: built-in rules -> measured table -> debug overrides -> ONE constraints pass
// (what the kernels can actually launch, enforced on whatever the first three picked), then the plan's geometry.
// tests/cpp/test_capi_host.cpp walks a grid of requests x every override through fa_plan and asserts the invariants.
struct Request {
    int64_t G, L, D, g8, cq, ck;   // cq / ck: the q / kv chunk lengths, capped at L
    int32_t dtype, causal;
    float scale;
    int es, skip, wmax;            // element size; 1 = skip-max arithmetic; waves the q chunk allows
    bool pipe_ok, want32;          // the pipelined kernels implement skip-max only; a kv chunk below 64 keys wants 32-key steps
    bool d_only;                   // the operands' true head dim is below the kernel's (zero-extended call: FwdParams::Dr)
};
struct Choice {
    char which;                    // 'd' | 'p' | 'x' | 'h'
    int waves, qb, sched, source, split, table_pair;
};

// built-in rules (round-1 .. round-3 measurements)
Choice rules_choice(const Request& r) {
    Choice c{'d', r.wmax, 1, 0, 0, 1, 0};
    const int64_t wgs256 = r.g8 * ((r.L + 255) / 256), wgs128 = r.g8 * ((r.L + 127) / 128);
    if (r.D == 64 && r.pipe_ok && c.waves == 4 && r.cq >= 256 && r.L >= 4096 && wgs256 >= 1024) { c.which = 'p'; c.qb = 2; }
    if (r.D == 128 && r.skip && c.waves == 4 && r.cq >= 256 && r.L >= 4096 && wgs256 >= 512) c.waves = 8;
    // a grid of at most 512 128-row workgroups leaves CUs idle or at one workgroup each: split every q tile's kv loop over
    // 2 | 4 wave groups of the same workgroup (measured: G=8 L=4096 394 -> 514, G=8 L=8192 754 -> 883, G=4 L=8192 D=128 458 -> 512)
    if (c.waves == 4 && r.cq >= 128 && r.L >= 1024 && wgs128 <= 512) c.split = (r.D == 64 && r.L >= 2048) ? 4 : 2;
    // ... and at head_dim 64 up to L = 2048 the 64-row workgroups with (row block, key half) waves do better still: same slots, twice the
    // workgroups, half the chain per tile (measured, profiles/r03_kh: G=12..64 L=1024 +10..17 %, G=8..32 L=2048 +10..22 % over the split plans;
    // L = 4096: level with 4 split groups; 768 workgroups and more: slower than the 128-row workgroups)
    if (r.D == 64 && c.waves == 4 && r.cq >= 128 && !r.want32 && r.L >= 1024 && r.L <= 2048 && wgs128 <= 512) { c.which = 'h'; c.split = 1; }
    // ... with two kv split groups on top where even the 64-row workgroups leave CUs idle (at most 256 128-row workgroups, L = 2048 .. 4096:
    // G = 8 L = 2048 18.0 -> 15.5 us, G = 8 L = 4096 35.0 (4 split groups of 128-row workgroups) -> 30.5 us; L = 8192: the 128-row split plan stays)
    if (r.D == 64 && c.waves == 4 && r.cq >= 128 && !r.want32 && r.L >= 2048 && r.L <= 4096 && wgs128 <= 256) { c.which = 'h'; c.split = 2; }
    return c;
}

// measured table (fa_tuning_table.h): the first row whose bucket holds the request and whose tile the q chunk allows
void table_choice(const Request& r, Choice* c) {
    const int64_t wgs128 = r.g8 * ((r.L + 127) / 128);
    for (const fa::TuneRow& t : fa::kTuneTable) {
        if (t.D != r.D || t.skip != r.skip || t.causal != r.causal || r.L < t.L_lo || r.L > t.L_hi || wgs128 < t.wg_lo || wgs128 > t.wg_hi) continue;
        if (32 * t.waves * t.qb > (r.cq > 32 ? r.cq : 32)) continue;   // the q chunk forbids the row's tile
        if ((t.kernel == 'p' || t.kernel == 'x') && !r.pipe_ok) continue;
        if (t.kernel == 'x' && r.D == 64 && t.waves != 4) continue;
        c->which = t.kernel; c->waves = t.waves; c->qb = t.qb; c->sched = t.sched; c->source = 1;
        c->split = t.split > 1 ? t.split : 1;
        c->table_pair = t.pair;
        return;
    }
}

// debug overrides (fa_debug_set_option / FA_AMD_* at load): a forced variant ignores the chunk bound, as the forced kernel does
void apply_overrides(const Request& r, const Options& opt, Choice* c) {
    const int fk = opt.kernel, fw = opt.waves, fq = opt.qb, fs = opt.split;
    const int64_t D = r.D, L = r.L, cq = r.cq;
    const bool pipe_ok = r.pipe_ok;
    char& which = c->which;
    int &waves = c->waves, &qb = c->qb, &split = c->split;
    if (fk || fw || fq || fs || opt.sched > 0 || (r.dtype == FA_DTYPE_F16 && opt.skipmax >= 0)) c->source = 2;
    if (fk || fw || fq) { split = 1; c->table_pair = 0; } // (a forced tile drops the table's split / pairs unless forced too)
    if ((fw || fq || fs > 1) && !fk && which == 'h') which = 'd';   // (the key-halves kernel has one tile shape: a forced tile leaves it)
    if (fw == 1 || fw == 2 || fw == 4 || fw == 8) { waves = fw; if (!fq) qb = 1; if (!fk && which == 'p' && waves > 4 && D != 128) which = 'd'; }
    if (fk == 'd') { which = 'd'; if (!fq) qb = 1; }
    else if (fk == 'x' && pipe_ok) { which = 'x'; waves = (fw == 8 && D == 128) ? 8 : 4; if (!fq) qb = 1; }
    else if (fk == 'h') { which = 'h'; waves = 4; qb = 1; }
    else if (fk == 'p' && pipe_ok && (waves <= 4 || D == 128)) { which = 'p'; if (!fq) qb = (D == 64 && L >= 256 && waves == 4 && cq >= 256) ? 2 : 1; }
    if (fq == 1) qb = 1;
    if (fq == 2 && D == 64 && L >= 256 && waves == 4 && cq >= 256) qb = 2;
    if (fq == 2 && fk == 'x' && D == 64) qb = 2;
    if (fs == 1) split = 1;
    if (fs > 1 && (fk == 0 || fk == 'd')) { which = 'd'; waves = 4; qb = 1; split = fs; }
    if (fs > 1 && fk == 'h') split = fs;
}

// what the kernels can launch - enforced on whatever rules, table and overrides picked
void apply_constraints(const Request& r, const Options& opt, Choice* c) {
    const int fk = opt.kernel, fs = opt.split;
    const int64_t D = r.D;
    char& which = c->which;
    int &waves = c->waves, &qb = c->qb, &split = c->split;
    if (D > 128) {
        // head_dim 192 / 256: the 64-key-tile kernel alone, one workgroup per CU (99 / 132 KB of LDS).  192 fits 8 waves (225 registers, two
        // waves per SIMD: [3,128,2048,192] 323 -> 269 us, [3,32,8192,192] 970 -> 753 us = 1 096 TFLOP/s); 256 spilled 32 registers in that form and
        // lost 20-35 % (426 -> 531 us): it keeps 4 waves with the whole register file, the 8-wave form is not built
        int w = (D == 192 && r.cq >= 256 && r.L >= 1024 && r.g8 * ((r.L + 255) / 256) >= 256) ? 8 : 4;
        if (opt.waves == 4 || (opt.waves == 8 && D == 192)) w = opt.waves;   // (8 waves at 256 would spill: not built)
        // 256 with the skip-max arithmetic from L = 1024 on: the software-pipelined kernel's 4-wave form (133 KB of rings; one wave per SIMD needs the
        // hand-ordered step to keep the matrix pipe fed: [3,128,2048,256] 428 -> 362 us, [3,32,8192,256] 1 478 -> 1 191 us = 923 TFLOP/s)
        bool pipe256 = D == 256 && r.pipe_ok && r.L >= 1024 && r.cq >= 128;
        if (opt.kernel == 'd') pipe256 = false;
        if (opt.kernel == 'p' && D == 256 && r.pipe_ok) pipe256 = true;
        which = pipe256 ? 'p' : 'd'; waves = pipe256 ? 4 : w; qb = 1; split = 1;
        return;
    }
    if (r.want32 && r.pipe_ok && which == 'd' && fk == 0 && fs <= 1) {   /* (a forced variant ignores the chunk bound) */ which = 'p'; if (waves > 4 && D != 128) waves = 4; }
    if (which == 'p' && ((waves > 4 && D != 128) || !r.pipe_ok)) which = 'd';   // (8-wave pipelined workgroups exist at head_dim 128 only)
    if (which == 'x' && !r.pipe_ok) which = 'd';
    if (which == 'h') { waves = 4; qb = 1; if (!(split == 1 || split == 2 || (split == 4 && D == 64))) split = 1; }
    if (which == 'x') { waves = (waves == 8 && D == 128) ? 8 : 4; if (D == 128) qb = 1; }
    if (qb == 2 && !(D == 64 && waves == 4)) qb = 1;
    if (split > 1 && !((which == 'd' || which == 'h') && waves == 4 && qb == 1 && (split == 2 || (split == 4 && D == 64)))) split = 1;
}

// q-tile pairs and the placement-balanced order ('d', causal), then the plan's geometry
void finish_mfma_plan(const Request& r, const Options& opt, Choice& c, Selection* sel) {
    fa_plan_t* plan = &sel->plan;
    const char which = c.which;
    const int waves = c.waves, qb = c.qb, split = c.split, sched = c.sched;
    const int64_t L = r.L, D = r.D, g8 = r.g8;
    const int es = r.es;
    int pair = 0;
    {
        const int64_t nq128 = (L + 127) / 128;
        const bool pair_ok = which == 'd' && r.causal && waves == 4 && qb == 1 && split == 1 && nq128 >= 2 && D <= 128;
        // measured (profiles/r03_short/sweep_*.txt): pairs LOSE 2..10 % at L = 1024 .. 4096 (4 resident workgroups per CU already hide each
        // other's prologues; the pair's second tile waits for the first tile's stores at its first barrier) - built-in rule: off
        if (c.table_pair && pair_ok && c.source == 1) pair = 1;          // (a measured row may switch them on)
        if (opt.pair >= 0) { pair = (opt.pair != 0 && pair_ok) ? 1 : 0; c.source = 2; }
        if (sel->xs > 1) pair = 0;                                       // (q-tile subsets: plain block map only)
    }
    sel->pair = pair;
    sel->perm_n = 0;
    // (the key-halves kernel takes the order too, but measured no gain from it - G = 8 L = 4096 31.7 vs 31.5 us, profiles/r03_kh/balance_*.txt:
    //  its launches are bound per SIMD, not per CU sum - so there it is applied on request only)
    if (((which == 'd' && !pair && split == 1 && opt.balance != 0) || (which == 'h' && opt.balance == 1)) && r.causal && qb == 1 && waves == 4 && sel->xs == 1) {
        const int rows = which == 'h' ? 64 : 32 * waves;                                    // q rows per workgroup
        const int64_t nqt = (L + rows - 1) / rows, ng = g8 / 8;
        const int64_t nitem = nqt;                                                          // workgroups per group
        // resident workgroups per XCD: by LDS (33 KB per 4-wave group at head_dim 64, 66 KB at 128; 160 KB per CU)
        const int per_cu = which == 'h' ? (D == 64 ? (split == 1 ? 4 : split == 2 ? 2 : 1) : (split == 1 ? 2 : 1)) : (D == 64 ? 4 : D == 128 ? 2 : 1);
        const int64_t slots_per_xcd = (int64_t)per_cu * kCusPerXcd;
        if (ng * nitem <= slots_per_xcd && ng * nitem <= 128) {
            int bunch_sz = (int)((8ll << 20) / (2 * L * D * es));
            if (sched > 0) bunch_sz = sched;
            if (opt.sched > 0) bunch_sz = opt.sched;
            if (bunch_sz < 1) bunch_sz = 1;
            if (bunch_sz > ng) bunch_sz = (int)ng;
            const auto tiles_of = [&](int64_t qi) { const int64_t kv_end = (qi + 1) * rows < L ? (qi + 1) * rows : L; return (int)((kv_end + 63) / 64); };
            int weight[128];
            for (int s_ = 0; s_ < ng * nitem; ++s_) {    // the kernels' natural slot -> (group, work item) map (fa_fwd_dma.hip, fa_fwd_kh.hip)
                const int per_bunch = bunch_sz * (int)nitem, b_ = s_ / per_bunch, rem = s_ - b_ * per_bunch;
                const int g0 = b_ * bunch_sz, bc = (ng - g0) < bunch_sz ? (int)(ng - g0) : bunch_sz;
                const int64_t qd = rem / bc;
                weight[s_] = tiles_of(nqt - 1 - qd);
            }
            sel->perm_n = balanced_order((int)(ng * nitem), weight, sel->perm);
        }
    }
    plan->kernel = which == 'p' ? 2 : which == 'x' ? 4 : which == 'h' ? 6 : 0;
    plan->pair = pair;
    plan->balanced = sel->perm_n > 0;
    plan->waves = waves * split;
    plan->qb = qb;
    plan->block_q = 32 * waves * qb;
    plan->block_kv = (which == 'p' || which == 'x') ? 32 : 64;
    plan->skipmax = r.skip;
    plan->source = c.source;
    // workgroups launched for q tiles of `rows` rows: 8-aligned groups x tiles, or - q-tile subsets - 8-aligned (group, subset) pairs x tiles per pair
    const int64_t xs = sel->xs, g8v = (r.G * xs + 7) / 8 * 8;
    const auto grid_of = [&](int64_t rows) { const int64_t t = (L + rows - 1) / rows; return xs > 1 ? g8v * ((t + xs - 1) / xs) : g8 * t; };
    plan->grid = grid_of(plan->block_q);
    if (pair) plan->grid = g8 * (((L + plan->block_q - 1) / plan->block_q + 1) / 2);
    if (which == 'h') {   // 4 waves = 2 row blocks x 2 key halves per kv split group: 64-row workgroups
        plan->block_q = 64;
        plan->grid = grid_of(64);
    }
    plan->lds_bytes = which == 'x' ? 8 * 32 * D * 2 + 16
                      : which == 'p' ? 4 * 32 * D * 2 + 4 * (D / 32) * (32 * 64 + 64) + 4 * waves
                                   : which == 'h' ? split * (2 * 64 * D * 2 + 2 * (D / 32) * (64 * 64 + 64)) + 16 * split
                                     : split * (2 * 64 * D * 2 + 2 * (D / 32) * (64 * 64 + 64));
    // groups per L2 bunch (see the kernels' block mapping): a bunch's K+V ~ 8 MiB measured best at
    // L=4096, D=64 (bunch 1/2/4/8/16 -> 737/769/800/828/811 TFLOP/s)
    {
        const int64_t kv_bytes = 2 * L * D * es;
        int64_t bunch = sched > 0 ? sched : (8ll << 20) / kv_bytes;
        if (opt.sched > 0) bunch = opt.sched;
        int64_t ng = (xs > 1 ? g8v : g8) / 8;
        if (bunch < 1) bunch = 1;
        if (bunch > ng) bunch = ng;
        plan->sched = (int32_t)bunch;
    }
    plan->xs = sel->xs;
    sel->which = which;
    sel->split = split;
}

int select_plan(int64_t G, int64_t L, int64_t D, int32_t dtype, int32_t nq, int32_t nk, float scale, int32_t causal,
                bool mfma16_ok, bool f32_ok, bool d_only, Selection* sel) {
    if (G < 1 || L < 1 || D < 1) return fail(FA_ERR_BAD_SHAPE, "G, L, D must be >= 1 (got %lld, %lld, %lld)", (long long)G, (long long)L, (long long)D);
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    if (nq < 1 || L % nq != 0) return fail(FA_ERR_CHUNKS_Q, "sequence length %lld is not divisible by num_chunks_q=%d", (long long)L, nq);
    if (nk < 1 || L % nk != 0) return fail(FA_ERR_CHUNKS_KV, "sequence length %lld is not divisible by num_chunks_kv=%d", (long long)L, nk);
    if (L > (1ll << 30) || G > (1ll << 30)) return fail(FA_ERR_TOO_LARGE, "G and L must be < 2^30");
    if (!(scale > 0.0f) || !isfinite(scale)) return fail(FA_ERR_BAD_SCALE, "softmax_scale must be finite and > 0 (got %g)", (double)scale);
    if (!(causal == 0 || causal == 1)) return fail(FA_ERR_NOT_CAUSAL, "causal must be 0 or 1 (got %d)", causal);
    const Options& opt = options();
    fa_plan_t* plan = &sel->plan;
    memset(plan, 0, sizeof(*plan));
    sel->split = 1;
    sel->pair = 0;
    sel->perm_n = 0;
    sel->xs = 1;
    plan->chunk_q = (int32_t)(L / nq);
    plan->chunk_kv = (int32_t)(L / nk);
    int64_t g8 = (G + 7) / 8 * 8;
    const int es = elt_size(dtype);
    sel->xs = (mfma16_ok || f32_ok) ? subsets_for(G, L, 2 * L * D * es) : 1;
    if (options().subsets == 0 || options().pair == 1 || options().balance == 1) sel->xs = 1;   // (a forced pair / balanced order keeps the whole-group map)
    plan->xs = sel->xs;
    if (sel->xs > 1) g8 = ((G * sel->xs + 7) / 8 * 8 + sel->xs - 1) / sel->xs;   // the groups the launch really spreads over the XCDs (~G), for the fill rules
    if (mfma16_ok) {
        Request r;
        r.G = G; r.L = L; r.D = D; r.g8 = g8; r.dtype = dtype; r.causal = causal; r.scale = scale; r.es = es;
        // f16 skip-max pays off when logits are small against the 2^15 headroom of f16 P: a softmax scale
        // <= 0.25 means 1/sqrt(D)-style scaled attention (D >= 16); the packed op (scale 1) keeps the row max
        r.skip = dtype == FA_DTYPE_BF16 ? 1 : (scale <= 0.25f ? 1 : 0);
        if (dtype == FA_DTYPE_F16 && opt.skipmax >= 0) r.skip = opt.skipmax != 0;
        r.cq = plan->chunk_q < L ? plan->chunk_q : L;
        r.ck = plan->chunk_kv < L ? plan->chunk_kv : L;
        r.wmax = waves_for_chunk(r.cq);            // the q chunk bounds the workgroup's rows
        r.pipe_ok = r.skip != 0;                   // (the pipelined kernels implement the skip-max arithmetic only)
        r.want32 = r.ck < 64;                      // a kv chunk below 64 keys is served by 32-key steps
        r.d_only = d_only;
        Choice c = rules_choice(r);
        if (opt.table) table_choice(r, &c);
        apply_overrides(r, opt, &c);
        apply_constraints(r, opt, &c);
        finish_mfma_plan(r, opt, c, sel);
    } else if (f32_ok) {
        plan->kernel = 3;
        plan->waves = 4;
        plan->qb = 1;
        plan->block_q = 128;
        plan->block_kv = 32;
        const int64_t g8v = (G * sel->xs + 7) / 8 * 8;
        plan->grid = sel->xs > 1 ? g8v * (((L + 127) / 128 + sel->xs - 1) / sel->xs) : g8 * ((L + 127) / 128);
        plan->lds_bytes = 2 * 32 * (D + 1) * 4 + 2 * 32 * D * 4;
        {
            int64_t bunch = (8ll << 20) / (2 * L * D * es);
            if (opt.sched > 0) bunch = opt.sched;
            const int64_t ng = (sel->xs > 1 ? g8v : g8) / 8;
            if (bunch < 1) bunch = 1;
            if (bunch > ng) bunch = ng;
            plan->sched = (int32_t)bunch;
        }
        sel->which = 'f';
    } else {
        if (D > 1024) return fail(FA_ERR_BAD_SHAPE, "generic kernel supports D <= 1024 (got %lld)", (long long)D);
        plan->kernel = 1;
        plan->waves = 4;
        plan->qb = 1;
        plan->block_q = 1;
        plan->block_kv = 1024;
        plan->grid = G * L;
        plan->lds_bytes = 1024 * 4 + 16 + D * 4;
        plan->sched = 1;
        sel->which = 'g';
    }
    if (plan->grid > 0x7fffffffll) return fail(FA_ERR_TOO_LARGE, "grid of %lld workgroups exceeds 2^31-1", (long long)plan->grid);
    return FA_OK;
}

struct OptionName { const char* name; std::atomic<int> Options::*field; int lo, hi; };
const OptionName kOptionNames[] = {
    {"kernel", &Options::kernel, 0, 127}, {"qb", &Options::qb, 0, 2}, {"waves", &Options::waves, 0, 8},
    {"sched", &Options::sched, 0, 1 << 20}, {"skipmax", &Options::skipmax, -1, 1},
    {"f32_generic", &Options::f32_generic, 0, 1}, {"table", &Options::table, 0, 1}, {"split", &Options::split, 0, 4},
    {"pair", &Options::pair, -1, 1}, {"balance", &Options::balance, -1, 1}, {"subsets", &Options::subsets, -1, 0},
    {"bsplit", &Options::bsplit, -1, 3},
    {"bbal", &Options::bbal, -1, 1},
};

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
        case FA_ERR_NOT_CAUSAL: return "causal must be 0 or 1";
        case FA_ERR_LAUNCH: return "kernel launch failed";
        case FA_ERR_BAD_STRIDE: return "bad stride";
        case FA_ERR_TOO_LARGE: return "problem too large";
        case FA_ERR_BAD_SCALE: return "softmax_scale must be finite and > 0";
        case FA_ERR_BAD_OPTION: return "unknown debug option or value out of range";
        case FA_ERR_BAD_MASK: return "bad mask kind or mask pointer";
        case FA_ERR_BAD_DROPOUT: return "dropout_p must be in [0, 1)";
        default: return "unknown error";
    }
}

int fa_debug_set_option(const char* name, int32_t value) {
    if (!name) return fail(FA_ERR_NULL_POINTER, "fa_debug_set_option: null name");
    (void)options();   // the environment is parsed before the first override, never after it
    for (const OptionName& o : kOptionNames) {
        if (strcmp(o.name, name) != 0) continue;
        if (value < o.lo || value > o.hi) return fail(FA_ERR_BAD_OPTION, "fa_debug_set_option: %s=%d out of range [%d, %d]", name, value, o.lo, o.hi);
        if (o.field == &Options::kernel && !(value == 0 || value == 'd' || value == 'p' || value == 'x' || value == 'h'))
            return fail(FA_ERR_BAD_OPTION, "fa_debug_set_option: kernel must be 0, 'd', 'p', 'x' or 'h'");
        if (o.field == &Options::waves && !(value == 0 || value == 1 || value == 2 || value == 4 || value == 8))
            return fail(FA_ERR_BAD_OPTION, "fa_debug_set_option: waves must be 0, 1, 2, 4 or 8");
        if (o.field == &Options::split && value == 3) return fail(FA_ERR_BAD_OPTION, "fa_debug_set_option: split must be 0, 1, 2 or 4");
        (g_opt.*(o.field)).store(value);
        return FA_OK;
    }
    return fail(FA_ERR_BAD_OPTION, "fa_debug_set_option: unknown option '%s'", name);
}

int fa_debug_set_stamp_buffer(void* device_buffer, int64_t bytes) {
    if (bytes < 0 || (device_buffer == nullptr) != (bytes == 0)) return fail(FA_ERR_BAD_OPTION, "fa_debug_set_stamp_buffer: need (buffer, bytes > 0) or (NULL, 0)");
    if (reinterpret_cast<uintptr_t>(device_buffer) & 7) return fail(FA_ERR_MISALIGNED, "fa_debug_set_stamp_buffer: buffer must be 8-byte aligned");
    g_stamp_bytes.store(0);
    g_stamp_buf.store(device_buffer);
    g_stamp_bytes.store(bytes);
    return FA_OK;
}

int fa_debug_get_option(const char* name, int32_t* value) {
    if (!name || !value) return fail(FA_ERR_NULL_POINTER, "fa_debug_get_option: null argument");
    const Options& opt = options();
    for (const OptionName& o : kOptionNames) {
        if (strcmp(o.name, name) == 0) { *value = (opt.*(o.field)).load(); return FA_OK; }
    }
    return fail(FA_ERR_BAD_OPTION, "fa_debug_get_option: unknown option '%s'", name);
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

int fa_plan(int64_t G, int64_t L, int64_t D, int32_t dtype, int32_t num_chunks_q, int32_t num_chunks_kv,
            float softmax_scale, int32_t causal, fa_plan_t* plan) {
    if (!plan) return fail(FA_ERR_NULL_POINTER, "null plan pointer");
    const int64_t Dk = (dtype >= 0 && dtype <= 2 && D >= 1) ? kernel_head_dim(D, dtype) : 0;   // the head dim the MFMA kernels would run this call at
    const bool d_ok = Dk != 0;
    Selection sel;
    const int rc = select_plan(G, L, d_ok ? Dk : D, dtype, num_chunks_q, num_chunks_kv, softmax_scale, causal,
                               d_ok && (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16),
                               d_ok && dtype == FA_DTYPE_F32 && !options().f32_generic, d_ok && Dk != D, &sel);
    if (rc == FA_OK) *plan = sel.plan;
    return rc;
}

}  // extern "C"

namespace {

// mask / bias / dropout arguments of fa_fwd_ex / fa_bwd_ex, validated into the kernels' ExtraParams (fa_extras.h)
int make_extra(const void* mask, int32_t mask_kind, const int64_t* mask_strides, int32_t dtype, float softmax_scale,
               float dropout_p, uint64_t seed, uint64_t offset, const uint64_t* state_dev, fa::ExtraParams* ex) {
    memset(ex, 0, sizeof(*ex));
    ex->drop_rp = 1.0f;
    ex->inv_scale = 1.0f / softmax_scale;
    if (mask_kind < FA_MASK_NONE || mask_kind > FA_MASK_BIAS) return fail(FA_ERR_BAD_MASK, "mask_kind must be FA_MASK_NONE/BOOL/BIAS (got %d)", mask_kind);
    if ((mask_kind == FA_MASK_NONE) != (mask == nullptr)) return fail(FA_ERR_BAD_MASK, "mask pointer and mask_kind disagree");
    if (mask_kind != FA_MASK_NONE) {
        if (!mask_strides) return fail(FA_ERR_NULL_POINTER, "mask_strides must be non-null when a mask is given");
        const int esz = mask_kind == FA_MASK_BIAS ? elt_size(dtype) : 1;
        if (reinterpret_cast<uintptr_t>(mask) % esz) return fail(FA_ERR_MISALIGNED, "mask pointer is not element aligned");
        bool vec = reinterpret_cast<uintptr_t>(mask) % (4 * esz) == 0;
        for (int j = 0; j < 3; ++j) {
            if (mask_strides[j] < 0 || mask_strides[j] > (1ll << 40)) return fail(FA_ERR_BAD_STRIDE, "mask strides must be in [0, 2^40] elements");
            if (mask_strides[j] % 4) vec = false;
        }
        ex->mask = mask;
        ex->mask_kind = mask_kind;
        ex->m_sb = mask_strides[0]; ex->m_sh = mask_strides[1]; ex->m_sl = mask_strides[2];
        // bit 1: whole mask tiles may travel to LDS by DMA (the per-row masks of the backward: fa_bwd_kernels.hip) - 128 rows of byte offsets
        // must fit the 32-bit lane offset of a buffer load
        ex->mask_vec = vec ? (mask_strides[2] * esz < (1ll << 24) ? 3 : 1) : 0;
    }
    if (!(dropout_p >= 0.0f) || !(dropout_p < 1.0f)) return fail(FA_ERR_BAD_DROPOUT, "dropout_p must be in [0, 1) (got %g)", (double)dropout_p);
    if (dropout_p > 0.0f) {
        const double t = (double)dropout_p * 4294967296.0;
        ex->drop_thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
        ex->drop_rp = 1.0f / (1.0f - dropout_p);
        ex->drop_key0 = fa::drop_fold_key0(seed, offset);
        if (state_dev) {
            if (reinterpret_cast<uintptr_t>(state_dev) & 7) return fail(FA_ERR_MISALIGNED, "dropout_state must be 8-byte aligned");
            ex->drop_state = reinterpret_cast<const unsigned long long*>(state_dev);
        }
    }
    return FA_OK;
}

int fwd_impl(const void* q, const void* k, const void* v, void* out, float* lse,
           int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv,
           const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],
           int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype,
           float softmax_scale, int32_t causal, void* stream, const fa::ExtraParams* ex) {
    if (!q || !k || !v || !out) return fail(FA_ERR_NULL_POINTER, "q, k, v and out must be non-null");
    if (!q_strides || !k_strides || !v_strides || !o_strides) return fail(FA_ERR_NULL_POINTER, "stride arrays must be non-null");
    if (heads < 1 || (G >= 1 && G % heads != 0)) return fail(FA_ERR_BAD_SHAPE, "G=%lld must be a positive multiple of heads=%lld", (long long)G, (long long)heads);
    if (heads_kv < 1 || heads % heads_kv != 0) return fail(FA_ERR_BAD_SHAPE, "heads=%lld must be a positive multiple of heads_kv=%lld", (long long)heads, (long long)heads_kv);
    const void* const ptrs[4] = {q, k, v, out};
    const int64_t* const strides[4] = {q_strides, k_strides, v_strides, o_strides};
    bool wide_rows = false;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j)
            if (strides[i][j] < 0 || strides[i][j] > (1ll << 40)) return fail(FA_ERR_BAD_STRIDE, "strides must be in [0, 2^40] elements");
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    for (int i = 0; i < 4; ++i)
        if (reinterpret_cast<uintptr_t>(ptrs[i]) % elt_size(dtype)) return fail(FA_ERR_MISALIGNED, "pointer %d is not element aligned", i);
    // (sizes are bounded before anything is multiplied: the products below must not overflow)
    if (G < 1 || L < 1 || D < 1) return fail(FA_ERR_BAD_SHAPE, "G, L, D must be >= 1 (got %lld, %lld, %lld)", (long long)G, (long long)L, (long long)D);
    if (L > (1ll << 30) || G > (1ll << 30) || D > (1ll << 20)) return fail(FA_ERR_TOO_LARGE, "G and L must be < 2^30, D < 2^20");
    if (k_strides[2] > (1ll << 32) || v_strides[2] > (1ll << 32)) wide_rows = true;

    // the MFMA kernels need 16-byte rows and address K/V rows with 32-bit element offsets from the group base
    // mask / dropout: the MFMA mask kernels (fa_fwd_extra.hip; fa_bwd_extra.hip for fa_bwd_ex) for 16-bit operands at every head dim the plain
    // kernels take (up to 256; there the backward's dK/dV runs as a dV pass and a dK pass), the generic kernel for
    // everything else fa_fwd serves (fp32, other head dims, unaligned rows) - never the fp32 MFMA kernel, which has no mask path
    const bool extra = ex && (ex->mask != nullptr || ex->drop_thr != 0u);
    const int64_t Dk0 = kernel_head_dim(D, dtype);
    // (a zero-extended call marks the chunks past the true head dim with the LDS-DMA offset 0x80000000 and relies on the descriptor's range
    //  check to return zeros for it: the K / V extent of a group must then stay below 2^31 BYTES, or that offset is a real address)
    const int64_t kv_extent_limit = (Dk0 != 0 && Dk0 != D) ? (1ll << 31) / elt_size(dtype) : (1ll << 31);
    const bool wide_ok = !wide_rows && Dk0 != 0 && aligned16(Dk0, elt_size(dtype), ptrs, strides) && L * k_strides[2] < kv_extent_limit && L * v_strides[2] < kv_extent_limit;
    const int64_t Dk = wide_ok ? Dk0 : D;          // the head dim the launched kernel runs at (D itself on the generic path)
    const bool mfma16_ok = wide_ok && (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16);
    const bool f32_ok = wide_ok && dtype == FA_DTYPE_F32 && !options().f32_generic;
    Selection sel;
    const int rc = select_plan(G, L, Dk, dtype, num_chunks_q, num_chunks_kv, softmax_scale, causal, mfma16_ok, f32_ok && !extra, Dk != D, &sel);
    if (rc != FA_OK) return rc;
    const fa_plan_t& plan = sel.plan;
    const bool extra_mfma = extra && mfma16_ok;

    fa::FwdParams p;
    memset(&p.ex, 0, sizeof(p.ex));
    if (ex) p.ex = *ex;
    p.q = q; p.k = k; p.v = v; p.o = out; p.lse = lse;
    p.q_sb = q_strides[0]; p.q_sh = q_strides[1]; p.q_sl = q_strides[2];
    p.k_sb = k_strides[0]; p.k_sh = k_strides[1]; p.k_sl = k_strides[2];
    p.v_sb = v_strides[0]; p.v_sh = v_strides[1]; p.v_sl = v_strides[2];
    p.o_sb = o_strides[0]; p.o_sh = o_strides[1]; p.o_sl = o_strides[2];
    p.G = (int32_t)G;
    p.G8 = (int32_t)((G + 7) / 8 * 8);
    p.H = (int32_t)heads;
    p.kv_div = (int32_t)(heads / heads_kv);
    p.L = (int32_t)L;
    p.D = (int32_t)Dk;
    p.Dr = (int32_t)D;
    p.causal = causal;
    p.nq_tiles = plan.kernel != 1 ? (int32_t)((L + plan.block_q - 1) / plan.block_q) : 0;
    if (extra_mfma) p.nq_tiles = (int32_t)((L + 127) / 128);     // fa_fwd_extra_kernel: 4 waves x 32 rows, exact-max arithmetic
    p.xs = 1;
    p.nq_real = p.nq_tiles;
    if (sel.xs > 1 && plan.kernel != 1) {     // q-tile subsets: the block map runs over (group, subset) pairs (FwdParams::xs)
        p.xs = sel.xs;
        p.G = (int32_t)(G * sel.xs);
        p.G8 = (int32_t)((G * sel.xs + 7) / 8 * 8);
        p.nq_tiles = (p.nq_real + sel.xs - 1) / sel.xs;
    }
    p.sched = plan.sched;
    p.scale = softmax_scale;
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    p.skipmax = plan.skipmax;
    p.perm_n = sel.perm_n;
    if (sel.perm_n > 0) memcpy(p.perm, sel.perm, sizeof(p.perm));
    p.stamps = nullptr;
    if (void* sb = g_stamp_buf.load()) {   // 16 u64 per wave of the launch; too small a buffer = no stamps
        if (plan.grid * plan.waves * 16 * 8 <= g_stamp_bytes.load()) p.stamps = static_cast<unsigned long long*>(sb);
    }
    hipStream_t st = (hipStream_t)stream;
    if (extra_mfma) { p.perm_n = 0; p.skipmax = 0; }
    hipError_t e = extra_mfma         ? fa::launch_fwd_extra(p, dtype, st)
                   : sel.which == 'g' ? fa::launch_fwd_generic(p, dtype, st)
                   : sel.which == 'f' ? fa::launch_fwd_f32(p, st)
                   : sel.which == 'p' ? fa::launch_fwd_pipe(p, dtype, plan.waves, plan.qb, st)
                   : sel.which == 'x' ? fa::launch_fwd_pipe16(p, dtype, plan.waves, plan.qb, st)
                   : sel.which == 'h' ? fa::launch_fwd_kh(p, dtype, sel.split, st)
                                      : fa::launch_fwd_dma(p, dtype, plan.waves / sel.split, plan.qb, sel.split, sel.pair, st);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

int bwd_impl(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
             void* dq, void* dk, void* dv, float* dsum_workspace,
             int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv, const int64_t strides[8][3],
             int32_t dtype, float softmax_scale, int32_t causal, void* stream, const fa::ExtraParams* ex);

}  // namespace

extern "C" {

int fa_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
           int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv,
           const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],
           int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype,
           float softmax_scale, int32_t causal, void* stream) {
    return fwd_impl(q, k, v, out, lse, G, L, D, heads, heads_kv, q_strides, k_strides, v_strides, o_strides,
                    num_chunks_q, num_chunks_kv, dtype, softmax_scale, causal, stream, nullptr);
}

int fa_fwd_ex(const void* q, const void* k, const void* v, void* out, float* lse,
              int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv,
              const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],
              int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype, float softmax_scale, int32_t causal,
              const void* mask, int32_t mask_kind, const int64_t mask_strides[3],
              float dropout_p, uint64_t dropout_seed, uint64_t dropout_offset, const uint64_t* dropout_state, void* stream) {
    if (!(softmax_scale > 0.0f) || !isfinite(softmax_scale)) return fail(FA_ERR_BAD_SCALE, "softmax_scale must be finite and > 0 (got %g)", (double)softmax_scale);
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    fa::ExtraParams ex;
    const int rc = make_extra(mask, mask_kind, mask_strides, dtype, softmax_scale, dropout_p, dropout_seed, dropout_offset, dropout_state, &ex);
    if (rc != FA_OK) return rc;
    return fwd_impl(q, k, v, out, lse, G, L, D, heads, heads_kv, q_strides, k_strides, v_strides, o_strides,
                    num_chunks_q, num_chunks_kv, dtype, softmax_scale, causal, stream, &ex);
}

int fa_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
           void* dq, void* dk, void* dv, float* dsum_workspace,
           int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv, const int64_t strides[8][3],
           int32_t dtype, float softmax_scale, int32_t causal, void* stream) {
    return bwd_impl(q, k, v, out, dout, lse, dq, dk, dv, dsum_workspace, G, L, D, heads, heads_kv, strides, dtype, softmax_scale, causal, stream, nullptr);
}

int fa_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
              void* dq, void* dk, void* dv, float* dsum_workspace,
              int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv, const int64_t strides[8][3],
              int32_t dtype, float softmax_scale, int32_t causal,
              const void* mask, int32_t mask_kind, const int64_t mask_strides[3],
              float dropout_p, uint64_t dropout_seed, uint64_t dropout_offset, const uint64_t* dropout_state, void* stream) {
    if (!(softmax_scale > 0.0f) || !isfinite(softmax_scale)) return fail(FA_ERR_BAD_SCALE, "softmax_scale must be finite and > 0 (got %g)", (double)softmax_scale);
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    fa::ExtraParams ex;
    const int rc = make_extra(mask, mask_kind, mask_strides, dtype, softmax_scale, dropout_p, dropout_seed, dropout_offset, dropout_state, &ex);
    if (rc != FA_OK) return rc;
    return bwd_impl(q, k, v, out, dout, lse, dq, dk, dv, dsum_workspace, G, L, D, heads, heads_kv, strides, dtype, softmax_scale, causal, stream, &ex);
}

}  // extern "C"

namespace {

int bwd_impl(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
           void* dq, void* dk, void* dv, float* dsum_workspace,
           int64_t G, int64_t L, int64_t D, int64_t heads, int64_t heads_kv, const int64_t strides[8][3],
           int32_t dtype, float softmax_scale, int32_t causal, void* stream, const fa::ExtraParams* ex) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !dsum_workspace || !strides)
        return fail(FA_ERR_NULL_POINTER, "fa_bwd: every pointer argument must be non-null");
    if (!(causal == 0 || causal == 1)) return fail(FA_ERR_NOT_CAUSAL, "causal must be 0 or 1 (got %d)", causal);
    if (G < 1 || L < 1 || D < 1) return fail(FA_ERR_BAD_SHAPE, "G, L, D must be >= 1");
    if (!(softmax_scale > 0.0f) || !isfinite(softmax_scale)) return fail(FA_ERR_BAD_SCALE, "softmax_scale must be finite and > 0 (got %g)", (double)softmax_scale);
    if (heads < 1 || G % heads != 0) return fail(FA_ERR_BAD_SHAPE, "G=%lld must be a positive multiple of heads=%lld", (long long)G, (long long)heads);
    if (heads_kv < 1 || heads % heads_kv != 0) return fail(FA_ERR_BAD_SHAPE, "heads=%lld must be a positive multiple of heads_kv=%lld", (long long)heads, (long long)heads_kv);
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    if (L > (1ll << 30) || G > (1ll << 30)) return fail(FA_ERR_TOO_LARGE, "G and L must be < 2^30");
    const void* const ptrs[8] = {q, k, v, out, dout, dq, dk, dv};
    // MFMA kernels: head_dim 64/128 (16-bit: 256 too, and every multiple of 8 below, zero-extended), 16-byte aligned rows - 16-bit dtypes (fa_bwd_kernels.hip) and fp32
    // (fa_bwd_f32.hip, v_mfma_f32_32x32x2_f32); everything else -> generic kernels
    const int64_t Dk0 = kernel_head_dim(D, dtype);
    bool mfma = Dk0 != 0 && !(dtype == FA_DTYPE_F32 && options().f32_generic);
    const int64_t row_align = 16 / elt_size(dtype);
    for (int i = 0; i < 8; ++i) {
        if (reinterpret_cast<uintptr_t>(ptrs[i]) % elt_size(dtype)) return fail(FA_ERR_MISALIGNED, "fa_bwd: tensor %d is not element aligned", i);
        if (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) mfma = false;
        for (int j = 0; j < 3; ++j) {
            if (strides[i][j] < 0 || strides[i][j] > (1ll << 40)) return fail(FA_ERR_BAD_STRIDE, "strides must be in [0, 2^40] elements");
            if (strides[i][j] % row_align) mfma = false;
        }
        // (zero-extended head dims: extents below 2^31 BYTES - the out-of-range LDS-DMA offset 0x80000000 must stay out of range, see fwd_impl)
        if (strides[i][2] > (1ll << 32) || L * strides[i][2] >= ((Dk0 != 0 && Dk0 != D) ? (1ll << 31) / elt_size(dtype) : (1ll << 31))) mfma = false;
    }
    const bool extra = ex && (ex->mask != nullptr || ex->drop_thr != 0u);
    if (extra && dtype == FA_DTYPE_F32) mfma = false;       // (the fp32 MFMA kernels have no mask path: generic kernels)
    if (!mfma && D > 1024) return fail(FA_ERR_BAD_SHAPE, "generic backward supports D <= 1024 (got %lld)", (long long)D);
    if (!mfma && G * L > 0x7fffffffll) return fail(FA_ERR_TOO_LARGE, "grid exceeds 2^31-1 workgroups");
    fa::BwdParams p;
    memset(&p.ex, 0, sizeof(p.ex));
    if (ex) p.ex = *ex;
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
    p.G = (int32_t)G; p.G8 = (int32_t)((G + 7) / 8 * 8); p.H = (int32_t)heads; p.L = (int32_t)L;
    p.D = (int32_t)(mfma ? Dk0 : D);      // the head dim the kernels run at; the operands' own is Dr (zero-extended 16-bit calls)
    p.Dr = (int32_t)D;
    p.Hkv = (int32_t)heads_kv; p.kv_div = (int32_t)(heads / heads_kv);
    p.Gkv = (int32_t)(G / p.kv_div); p.G8kv = (int32_t)((p.Gkv + 7) / 8 * 8);
    p.causal = causal;
    p.nq_tiles = (int32_t)((L + 127) / 128);
    p.nk_tiles = (int32_t)((L + 127) / 128);
    {
        int64_t bunch = (8ll << 20) / (2 * L * D * 2);
        if (options().sched > 0) bunch = options().sched;
        if (bunch < 1) bunch = 1;
        const int64_t ng = p.G8 / 8, ngkv = p.G8kv / 8;
        p.sched = (int32_t)(bunch > ng ? ng : bunch);
        p.sched_kv = (int32_t)(bunch > ngkv ? ngkv : bunch);
    }
    // Balanced dQ units (fa_bwd_kernels.hip BAL): a causal dQ grid of at most one workgroup per CU lasts as long as its heaviest row block - twice the
    // mean.  The heavy half of the row blocks become two units each (halves of the kv sweep; fp32 partial sums parked in the group's dK / dV until the
    // reduce pass, so: query and K/V heads agree, the operands' own head dim), the units go out in descending length.  Measured
    // (profiles/r06_bwd_split/balanced_dq_units_rule.txt, whole backward): up to 128 row blocks in the grid x1.06 (16 blocks a group) .. 1.20 (64 and more),
    // level at 8 a group; at 129 - 256 row blocks x1.07 - 1.08 from 64 blocks a group on and a loss below (32: x0.99, 16: x0.93, 8: x0.82); a loss beyond.
    p.bal = 0;
    if (mfma && !extra && dtype != FA_DTYPE_F32 && causal && (p.D == 64 || p.D == 128) && p.Dr == p.D && p.kv_div == 1 && options().bbal != 0 &&
        p.nq_tiles >= 8 && p.nq_tiles % 4 == 0 && p.dk_sl >= p.D && p.dv_sl >= p.D && dk != dv &&   // (rows of dk / dv that can hold the parked sums)
        (options().bbal > 0 || (p.nq_tiles >= 16 && (G * p.nq_tiles <= 128 || (G * p.nq_tiles <= 256 && p.nq_tiles >= 64))))) {
        p.bal = 1;
        p.nq_tiles = p.nq_tiles + p.nq_tiles / 2;      // units per group
    }
    // block subsets for group counts that leave XCDs idle (as the forward, subsets_for): MFMA kernels only
    p.xs_q = p.xs_kv = 1;
    p.nq_real = p.nq_tiles;
    p.nk_real = p.nk_tiles;
    if (mfma && options().subsets != 0) {
        const int64_t kv_bytes = 2 * L * p.D * elt_size(dtype);
        // (measured, profiles/r05_subsets/bwd_ab.txt: the backward gains where XCDs are IDLE - [2,16384,64] x1.54, [4,8192,128] x1.14 - and loses
        //  where they are merely uneven - [12,4096,64] x0.89, [25,2048,64] x0.82: its grids are bound by the heaviest block's chain, and a group's
        //  Q / dO / K / V streamed into more L2s cost more than the evener spread returns - so: below 8 groups only)
        const int xq = G < 8 ? subsets_for(G, L, kv_bytes) : 1, xk = p.Gkv < 8 ? subsets_for(p.Gkv, L, kv_bytes) : 1;
        const auto cap_sched = [](int32_t sched, int32_t g8) { return sched > g8 / 8 ? g8 / 8 : sched; };
        if (xq > 1) {
            p.xs_q = xq;
            p.G = (int32_t)(G * xq);
            p.G8 = (p.G + 7) / 8 * 8;
            p.nq_tiles = (p.nq_real + xq - 1) / xq;
            p.sched = cap_sched(p.sched, p.G8);
        }
        if (xk > 1) {
            p.xs_kv = xk;
            p.Gkv = p.Gkv * xk;
            p.G8kv = (p.Gkv + 7) / 8 * 8;
            p.nk_tiles = (p.nk_real + xk - 1) / xk;
            p.sched_kv = cap_sched(p.sched_kv, p.G8kv);
        }
    }
    // Split groups inside the workgroups (fa_bwd_kernels.hip QS / QH) for grids that leave the chip short of waves: such a launch lasts as long as
    // its heaviest block's chain - [1,8192,64]: 64 workgroups, the longest sweeping 128 tiles with one wave per SIMD, 0.03 of peak - and the chain
    // shrinks with the groups.  Measured map (profiles/r06_bwd_split/threshold_map_*.txt, in-process A/B over 28 shapes): the split forms win up to
    // 512 workgroups in the grid in question at L >= 2048 (x1.06 .. 1.58) and up to 384 at L = 1024 (x1.05 .. 1.19), and lose from 640 on (x0.82 ..
    // 0.93); three groups (head_dim 64 only: 144 / 168 registers = three waves per SIMD) beat two up to 256 workgroups at L >= 2048, two win elsewhere.
    // At head_dim 128 "two groups" is two kv split groups in dQ and two row-half groups on shared tiles in dK/dV (its sweep fills the register file).
    p.qs_q = p.qs_kv = 1;
    if (mfma && !extra && dtype != FA_DTYPE_F32 && (p.D == 64 || p.D == 128) && options().bsplit != 0) {
        const int qs_max = p.D == 64 ? 3 : 2;
        const int forced = options().bsplit > 0 ? (options().bsplit > qs_max ? qs_max : (int)options().bsplit) : 0;
        const auto groups_for = [&](int64_t wgs) {
            if (L < 1024 || wgs > (L >= 2048 ? 512 : 384)) return 1;
            return (qs_max == 3 && wgs <= 256 && L >= 2048) ? 3 : 2;
        };
        if (forced == 1) { /* (1 = unsplit) */ }
        else if (forced > 1) p.qs_q = p.qs_kv = forced;
        else {
            p.qs_q = groups_for((int64_t)p.G * p.nq_tiles);
            p.qs_kv = groups_for((int64_t)p.Gkv * p.nk_tiles);
        }
    }
    if ((int64_t)p.G8 * p.nq_tiles > 0x7fffffffll) return fail(FA_ERR_TOO_LARGE, "grid exceeds 2^31-1 workgroups");
    p.scale = softmax_scale;
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    hipError_t e = extra && mfma          ? fa::launch_bwd_extra(p, dtype, (hipStream_t)stream)
                   : !mfma                ? fa::launch_bwd_generic(p, dtype, (hipStream_t)stream)
                   : dtype == FA_DTYPE_F32 ? fa::launch_bwd_f32(p, (hipStream_t)stream)
                                           : fa::launch_bwd(p, dtype, (hipStream_t)stream);
    if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

}  // namespace

extern "C" {

int fa_fwd_qkv_packed(const void* qkv, void* out, float* lse, int64_t G, int64_t L, int64_t D,
                      int32_t num_chunks_q, int32_t num_chunks_kv, int32_t dtype, void* stream) {
    if (!qkv || !out) return fail(FA_ERR_NULL_POINTER, "qkv and out must be non-null");
    if (G < 1 || L < 1 || D < 1) return fail(FA_ERR_BAD_SHAPE, "G, L, D must be >= 1");
    if (dtype < 0 || dtype > 2) return fail(FA_ERR_BAD_DTYPE, "dtype must be FA_DTYPE_F16/BF16/F32 (got %d)", dtype);
    // (sizes are bounded BEFORE the pointer arithmetic: G * L * D * es below must not overflow; the whole packed tensor must
    // also be addressable, 3 * G * L * D * es < 2^62)
    if (L > (1ll << 30) || G > (1ll << 30) || D > (1ll << 20)) return fail(FA_ERR_TOO_LARGE, "G and L must be < 2^30, D < 2^20");
    if (G * L > (1ll << 58) / (4 * D)) return fail(FA_ERR_TOO_LARGE, "packed qkv of %lld x %lld x %lld elements per operand is not addressable", (long long)G, (long long)L, (long long)D);
    const int64_t es = elt_size(dtype);
    const int64_t strides[3] = {L * D, 0, D};
    const char* base = static_cast<const char*>(qkv);
    return fa_fwd(base, base + G * L * D * es, base + 2 * G * L * D * es, out, lse, G, L, D, 1, 1,
                  strides, strides, strides, strides, num_chunks_q, num_chunks_kv, dtype, 1.0f, 1, stream);
}

}  // extern "C"
