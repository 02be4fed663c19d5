// C++ caller of the C ABI (no torch, no Python): the counterpart of the reference's on-device C++ test
// (tests/cpp/test_flash_attention.cpp:34-102 - flash vs dense attention, max |delta| bound) and of its
// cycle-count benchmark (:177-240).  Builds with hipcc, links libflash_attention.so, fills a packed qkv on
// the host, runs fa_fwd_qkv_packed, checks the result against a dense fp64 host computation on the same
// 16-bit-rounded inputs, then times a few launches with hipEvents.
//
//   hipcc -O2 -Iinclude tests/cpp/test_capi.cpp -Lflash-attention-ipu_amd/build -lflash_attention \
//         -Wl,-rpath,$PWD/flash-attention-ipu_amd/build -o tests/cpp/test_capi && tests/cpp/test_capi
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "flash_attention_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static uint16_t f2bf(float f) {   // round to nearest even
    uint32_t u; std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

int main() {
    const int64_t G = 4, L = 300, D = 64;   // L deliberately not a multiple of the 64-key tile
    std::vector<uint16_t> qkv(3 * G * L * D);
    uint32_t s = 12345u;
    for (auto& x : qkv) {   // deterministic pseudo-random values in (-1.5, 1.5)
        s = s * 1664525u + 1013904223u;
        x = f2bf(((s >> 8) / 16777216.0f - 0.5f) * 3.0f);
    }
    void *d_qkv = nullptr, *d_out = nullptr; float* d_lse = nullptr;
    HIP_OK(hipMalloc(&d_qkv, qkv.size() * 2));
    HIP_OK(hipMalloc(&d_out, G * L * D * 2));
    HIP_OK(hipMalloc((void**)&d_lse, G * L * 4));
    HIP_OK(hipMemcpy(d_qkv, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice));

    if (fa_abi_version() != FA_ABI_VERSION) { std::printf("ABI mismatch\n"); return 1; }
    fa_plan_t plan;
    if (fa_plan(G, L, D, FA_DTYPE_BF16, 1, 1, 1.0f, 1, &plan) != FA_OK || plan.kernel == 1) { std::printf("plan: %s\n", fa_last_error()); return 1; }   // an MFMA kernel (0 or 2)
    if (fa_fwd_qkv_packed(d_qkv, d_out, d_lse, G, L, D, 7, 1, FA_DTYPE_BF16, nullptr) != FA_ERR_CHUNKS_Q) { std::printf("expected FA_ERR_CHUNKS_Q\n"); return 1; }
    int rc = fa_fwd_qkv_packed(d_qkv, d_out, d_lse, G, L, D, 1, 1, FA_DTYPE_BF16, nullptr);
    if (rc != FA_OK) { std::printf("fa_fwd_qkv_packed: %s\n", fa_last_error()); return 1; }
    HIP_OK(hipDeviceSynchronize());
    std::vector<uint16_t> out(G * L * D);
    std::vector<float> lse(G * L);
    HIP_OK(hipMemcpy(out.data(), d_out, out.size() * 2, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(lse.data(), d_lse, lse.size() * 4, hipMemcpyDeviceToHost));

    // dense causal attention in fp64 on the host (vanilla_attention.cpp:33-66 arithmetic)
    double max_err = 0, max_lse_err = 0;
    const uint16_t *Q = qkv.data(), *K = Q + G * L * D, *V = K + G * L * D;
    std::vector<double> p(L);
    for (int64_t g = 0; g < G; ++g)
        for (int64_t r = 0; r < L; ++r) {
            double m = -1e300;
            for (int64_t c = 0; c <= r; ++c) {
                double acc = 0;
                for (int64_t d = 0; d < D; ++d) acc += (double)bf2f(Q[(g * L + r) * D + d]) * bf2f(K[(g * L + c) * D + d]);
                p[c] = acc; m = std::fmax(m, acc);
            }
            double l = 0;
            for (int64_t c = 0; c <= r; ++c) { p[c] = std::exp(p[c] - m); l += p[c]; }
            max_lse_err = std::fmax(max_lse_err, std::fabs(m + std::log(l) - lse[g * L + r]));
            for (int64_t d = 0; d < D; ++d) {
                double o = 0;
                for (int64_t c = 0; c <= r; ++c) o += p[c] * bf2f(V[(g * L + c) * D + d]);
                max_err = std::fmax(max_err, std::fabs(o / l - bf2f(out[(g * L + r) * D + d])));
            }
        }
    std::printf("max|out - dense| = %.3e   max|lse - dense| = %.3e\n", max_err, max_lse_err);
    // (lse: the denominator is the sum of the bf16-rounded P the PV product uses: <= 2^-8 relative, log -> 3.9e-3)
    if (!(max_err <= 1e-2) || !(max_lse_err <= 6e-3)) { std::printf("FAILED\n"); return 1; }

    // ---- fa_fwd_ex [ABI 4]: a key-padding boolean mask (one [L] row of bytes, broadcast over groups and rows: strides {0, 0, 0}) and
    // dropout, unscaled, not causal, against the same dense host computation with the generator restated here in plain C++
    // (mix32 = lowbias32; keep(g, row, col) as documented in include/flash_attention_amd.h / csrc/fa_extras.h)
    {
        auto mix32 = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
        const float pdrop = 0.25f;
        const uint64_t seed = 0x0123456789ABCDEFull, offset = 4242;
        uint32_t key0 = mix32((uint32_t)seed ^ 0x9E3779B9u);
        key0 = mix32(key0 ^ (uint32_t)(seed >> 32)); key0 = mix32(key0 ^ (uint32_t)offset); key0 = mix32(key0 ^ (uint32_t)(offset >> 32));
        const uint32_t thr = (uint32_t)((double)pdrop * 4294967296.0);
        std::vector<uint8_t> keep_key(L);
        for (int64_t c = 0; c < L; ++c) keep_key[c] = (c % 7 != 3) && c < L - 11;      // some keys hidden, a padded tail
        void* d_mask = nullptr;
        HIP_OK(hipMalloc(&d_mask, L));
        HIP_OK(hipMemcpy(d_mask, keep_key.data(), L, hipMemcpyHostToDevice));
        const int64_t st[3] = {L * D, 0, D}, mst[3] = {0, 0, 0};
        const char* base = static_cast<const char*>(d_qkv);
        rc = fa_fwd_ex(base, base + G * L * D * 2, base + 2 * G * L * D * 2, d_out, d_lse, G, L, D, 1, 1, st, st, st, st, 1, 1, FA_DTYPE_BF16,
                       0.125f, 0, d_mask, FA_MASK_BOOL, mst, pdrop, seed, offset, nullptr, nullptr);
        if (rc != FA_OK) { std::printf("fa_fwd_ex: %s\n", fa_last_error()); return 1; }
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(out.data(), d_out, out.size() * 2, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(lse.data(), d_lse, lse.size() * 4, hipMemcpyDeviceToHost));
        double ex_err = 0, ex_lse = 0;
        for (int64_t g = 0; g < G; ++g) {
            const uint32_t kg = mix32(key0 ^ ((uint32_t)g * 0x9E3779B1U)), kt = mix32(kg + 0u);     // (L < 65536: one window)
            for (int64_t r = 0; r < L; ++r) {
                double m = -1e300;
                for (int64_t c = 0; c < L; ++c) {
                    if (!keep_key[c]) { p[c] = -1e300; continue; }
                    double acc = 0;
                    for (int64_t d = 0; d < D; ++d) acc += (double)bf2f(Q[(g * L + r) * D + d]) * bf2f(K[(g * L + c) * D + d]);
                    p[c] = acc * 0.125; m = std::fmax(m, p[c]);
                }
                double l = 0;
                for (int64_t c = 0; c < L; ++c) { p[c] = keep_key[c] ? std::exp(p[c] - m) : 0.0; l += p[c]; }
                ex_lse = std::fmax(ex_lse, std::fabs(m + std::log(l) - lse[g * L + r]));
                for (int64_t c = 0; c < L; ++c)
                    if (mix32(kt ^ (((uint32_t)r & 0xffffu) << 16 | ((uint32_t)c & 0xffffu))) < thr) p[c] = 0.0;      // dropped
                for (int64_t d = 0; d < D; ++d) {
                    double o = 0;
                    for (int64_t c = 0; c < L; ++c) o += p[c] * bf2f(V[(g * L + c) * D + d]);
                    ex_err = std::fmax(ex_err, std::fabs(o / l / (1.0 - pdrop) - bf2f(out[(g * L + r) * D + d])));
                }
            }
        }
        std::printf("fa_fwd_ex (key-padding mask + dropout 0.25): max|out - dense| = %.3e   max|lse - dense| = %.3e\n", ex_err, ex_lse);
        if (!(ex_err <= 1e-2) || !(ex_lse <= 6e-3)) { std::printf("FAILED\n"); return 1; }
        (void)hipFree(d_mask);
    }

    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < 20; ++i) fa_fwd_qkv_packed(d_qkv, d_out, nullptr, G, L, D, 1, 1, FA_DTYPE_BF16, nullptr);
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("20 launches: %.1f us each\nPASSED\n", ms * 1000 / 20);
    (void)hipFree(d_qkv); (void)hipFree(d_out); (void)hipFree(d_lse);
    return 0;
}
