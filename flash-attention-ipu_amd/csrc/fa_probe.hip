// Measured matrix-pipe peak of THIS device: a register-only loop of back-to-back v_mfma_f32_32x32x16_{bf16,f16}
// (4 independent accumulator chains per wave, 4 waves per SIMD, every CU busy), timed with hipEvents.
// SURVEY.md 8(d) asks for the roofline denominator to be reported against a measured pure-MFMA number as well
// as the nominal 2.5 PFLOP/s; bench.py prints both.  Not part of the attention path.

#include "fa_device.h"

namespace fa {

template <typename T>
__global__ __launch_bounds__(256) void fa_mfma_peak_kernel(float* out, int iters, int random_data) {
    using E = Elt<T>;
    typename E::vec8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (T)(0.001f * (float)(threadIdx.x + j)); b[j] = (T)(0.002f * (float)(j + 1)); }
    if (random_data) {   // N(0,1)-like operands, different in every lane: realistic switching activity (power)
        unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            a[j] = (T)(((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f);
            s = s * 1664525u + 1013904223u;
            b[j] = (T)(((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f);
        }
    }
    f32x16 c0, c1, c2, c3;
#pragma unroll
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = E::mfma(a, b, c0);
            c1 = E::mfma(b, a, c1);
            c2 = E::mfma(a, a, c2);
            c3 = E::mfma(b, b, c3);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) r += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// returns TFLOP/s (<= 0 on failure); synchronises the device - a bench utility, not a data-path call
double probe_mfma_peak(int dtype, int iters, int random_data) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1.0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1.0;
    const int grid = cus * 4;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    float* buf = nullptr;
    if (hipMalloc((void**)&buf, (size_t)grid * 256 * sizeof(float)) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double best = -1.0;
    for (int rep = 0; rep < 6; ++rep) {   // rep 0 warms the clock up
        (void)hipEventRecord(e0, nullptr);
        if (dtype == 1) hipLaunchKernelGGL(fa_mfma_peak_kernel<__bf16>, dim3(grid), dim3(256), 0, nullptr, buf, iters, random_data);
        else hipLaunchKernelGGL(fa_mfma_peak_kernel<_Float16>, dim3(grid), dim3(256), 0, nullptr, buf, iters, random_data);
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 /*waves*/ * iters * 16 /*MFMA per iteration*/ * (2.0 * 32 * 32 * 16);
        const double tf = flops / (ms * 1e-3) / 1e12;
        if (rep > 0 && tf > best) best = tf;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return best;
}

// One wave that samples the shader clock counter (clock64 = s_memtime) against the constant 100 MHz counter
// (wall_clock64 = s_memrealtime) for `spins` sleeps of ~2 us: launched on a side stream while other work runs, it
// tells which core clock the device actually holds under that load.  out[0] = shader cycles, out[1] = 10 ns ticks.
__global__ void fa_clock_sampler_kernel(unsigned long long* out, int spins) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    unsigned long long c1 = c0, w1 = w0;
    for (int i = 0; i < spins; ++i) {
        __builtin_amdgcn_s_sleep(64);
        c1 = clock64();
        w1 = wall_clock64();
    }
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
}

hipError_t launch_clock_sampler(unsigned long long* out, int spins, hipStream_t stream) {
    void* args[] = {&out, &spins};
    return hipLaunchKernel((const void*)fa_clock_sampler_kernel, dim3(1), dim3(64), args, 0, stream);
}

}  // namespace fa
