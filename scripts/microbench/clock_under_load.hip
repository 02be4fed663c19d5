// What core clock does the GPU actually hold while the attention kernel runs?  A one-wave sampler kernel on a second
// stream reads the shader-clock counter (clock64 = s_memtime) against the constant 100 MHz counter (wall_clock64 =
// s_memrealtime) while fa_fwd_qkv_packed launches back to back on the first stream.
//   hipcc --offload-arch=gfx950 -O3 -I include -o /tmp/clk scripts/microbench/clock_under_load.hip \
//         -L flash-attention-ipu_amd/build -lflash_attention -Wl,-rpath,$PWD/flash-attention-ipu_amd/build && /tmp/clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "flash_attention_amd.h"

__global__ void sampler(unsigned long long* out, int spins) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    unsigned long long c1 = c0, w1 = w0;
    for (int i = 0; i < spins; ++i) {
        __builtin_amdgcn_s_sleep(64);
        c1 = clock64();
        w1 = wall_clock64();
    }
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;   // 0: idle, 1: attention with random data, 2: attention with constant data, 3: backward (random data)
    const long D = argc > 2 ? atol(argv[2]) : 64, L = argc > 3 ? atol(argv[3]) : 4096, G = argc > 4 ? atol(argv[4]) : 128;   // [mode] [D] [L] [G]
    const size_t n = 3ul * G * L * D;
    std::vector<unsigned short> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; float u1 = ((s >> 8) + 1) / 16777217.0f; s = s * 1664525u + 1013904223u; float u2 = (s >> 8) / 16777216.0f;
        h[i] = f2bf(mode == 2 ? 0.25f : sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2)); }
    void *qkv, *out; unsigned long long* d;
    (void)hipMalloc(&qkv, n * 2); (void)hipMalloc(&out, n * 2 / 3); (void)hipMalloc(&d, 16);
    (void)hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
    hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    void *dq = nullptr, *dout = nullptr; float *lse = nullptr, *dsum = nullptr;
    const int64_t st[3] = {L * D, 0, D};
    int64_t st8[8][3];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 3; ++j) st8[i][j] = st[j];
    if (mode == 3) {
        (void)hipMalloc(&dq, n * 2); (void)hipMalloc(&dout, n * 2 / 3); (void)hipMalloc((void**)&lse, G * L * 4); (void)hipMalloc((void**)&dsum, G * L * 4);
        (void)hipMemcpy(dout, qkv, n * 2 / 3, hipMemcpyDeviceToDevice);
        fa_fwd_qkv_packed(qkv, out, lse, G, L, D, 8, 8, FA_DTYPE_BF16, s1);
    }
    for (int rep = 0; rep < 3; ++rep) {
        if (mode == 3) {
            const char* b = (const char*)qkv; char* g = (char*)dq; const size_t sz = (size_t)G * L * D * 2;
            for (int i = 0; i < 90; ++i)
                fa_bwd(b, b + sz, b + 2 * sz, out, dout, lse, g, g + sz, g + 2 * sz, dsum, G, L, D, 1, st8, FA_DTYPE_BF16, 1.0f, 1, s1);
        } else if (mode) for (int i = 0; i < 300; ++i) fa_fwd_qkv_packed(qkv, out, nullptr, G, L, D, 8, 8, FA_DTYPE_BF16, s1);   // ~90 ms of load
        if (mode) for (volatile int w = 0; w < 20000000; ++w) {}   // let the load start (host spin ~20 ms)
        hipLaunchKernelGGL(sampler, dim3(1), dim3(64), 0, s2, d, 20000);
        (void)hipStreamSynchronize(s2);
        unsigned long long r[2]; (void)hipMemcpy(r, d, 16, hipMemcpyDeviceToHost);
        (void)hipStreamSynchronize(s1);
        std::printf("mode %d rep %d: shader clock %.0f MHz over %.2f ms\n", mode, rep, (double)r[0] / ((double)r[1] / 100.0), (double)r[1] / 1e5);
    }
    return 0;
}
