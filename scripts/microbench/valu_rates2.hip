// Microbenchmark 2: (a) is the cheap class "VOP2/VOP1 e32 encoding" or "two sources"?  (b) what does an MFMA cost a
// concurrent VALU stream when its accumulator lives in AGPRs instead of VGPRs, and for 16x16x32 vs 32x32x16?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates2 scripts/microbench/valu_rates2.hip && /tmp/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// MFMA: 0 none, 1 32x32x16 acc in VGPR, 2 32x32x16 acc in AGPR, 3 2x 16x16x32 acc in VGPR, 4 2x 16x16x32 acc in AGPR
template <int OP, int MFMA>
__global__ __launch_bounds__(256) void rate(float* out, int iters, float c, float sc) {
    float r[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { r[i] = 0.001f * (threadIdx.x + i); u[i] = threadIdx.x * 2654435761u + i; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (j + 1)); }
    f32x16 acc0, acc1; f32x4 q0, q1, q2, q3;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    for (int e = 0; e < 4; ++e) { q0[e] = q1[e] = q2[e] = q3[e] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MFMA == 1) {
                if (rep & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(b), "v"(a));
            }
            if (MFMA == 2) {
                if (rep & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(b), "v"(a));
            }
            if (MFMA == 3) {
                if (rep & 1) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q0) : "v"(a), "v"(b)); asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q1) : "v"(a), "v"(b)); }
                else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q2) : "v"(b), "v"(a)); asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q3) : "v"(b), "v"(a)); }
            }
            if (MFMA == 4) {
                if (rep & 1) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(q0) : "v"(a), "v"(b)); asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(q1) : "v"(a), "v"(b)); }
                else { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(q2) : "v"(b), "v"(a)); asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(q3) : "v"(b), "v"(a)); }
            }
#define ONE(i)                                                                                                   \
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));                          \
            if (OP == 1) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                          \
            if (OP == 2) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(c));                          \
            if (OP == 3) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(r[i]) : "v"(c));                         \
            if (OP == 4) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                          \
            if (OP == 5) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(r[i]) : "v"(c));                              \
            if (OP == 6) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(r[i]) : "s"(sc));                         \
            if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "s"(sc), "v"(c));                 \
            if (OP == 8) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(r[i]));                                       \
            if (OP == 9) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]), "v"(r[(i + 3) & 7])); \
            if (OP == 10) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(r[i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7])); \
            if (OP == 11) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                         \
            if (OP == 12) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3fb8aa3b" : "+v"(r[i]) : "v"(c));              \
            if (OP == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[i]), "v"(c));
            REP8(ONE)
#undef ONE
        }
    }
    float res = 0.f;
    for (int i = 0; i < 8; ++i) res += r[i] + (float)u[i];
    for (int e = 0; e < 16; ++e) res += acc0[e] + acc1[e];
    for (int e = 0; e < 4; ++e) res += q0[e] + q1[e] + q2[e] + q3[e];
    out[blockIdx.x * 256 + threadIdx.x] = res;
}

template <int OP, int MFMA>
static double run(float* d) {
    const int iters = 4000, grid = 256 * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<OP, MFMA>), dim3(grid), dim3(256), 0, 0, d, 200, 1.0001f, 0.9999f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((rate<OP, MFMA>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e6 / (double(iters) * 4 * 32);   // ns per VALU instruction per SIMD
}

template <int OP>
static void row(const char* name, float* d, double cyc) {
    const double t0 = run<OP, 0>(d), t1 = run<OP, 1>(d), t2 = run<OP, 2>(d), t3 = run<OP, 3>(d), t4 = run<OP, 4>(d);
    std::printf("%-26s alone %.2f | per 8 VALU + MFMA: 32x32 vgpr %.1f  32x32 agpr %.1f  2x16x16 vgpr %.1f  2x16x16 agpr %.1f cycles\n", name, t0 / cyc,
                t1 * 8 / cyc, t2 * 8 / cyc, t3 * 8 / cyc, t4 * 8 / cyc);
}

int main() {
    float* d; (void)hipMalloc(&d, 1024 * 256 * sizeof(float));
    const double cyc = run<0, 0>(d) / 4.0;
    std::printf("clock from v_fma_f32 = 4 cycles: %.3f GHz\n", 1.0 / cyc);
    row<0>("v_fma_f32 (2 regs)", d, cyc);
    row<9>("v_fma_f32 (3 distinct src)", d, cyc);
    row<7>("v_fma_f32 (sgpr src)", d, cyc);
    row<12>("v_fmaak_f32 (literal)", d, cyc);
    row<1>("v_add_f32_e32", d, cyc);
    row<10>("v_add_f32_e32 (2 src, dst)", d, cyc);
    row<2>("v_add_f32_e64", d, cyc);
    row<11>("v_sub_f32_e32", d, cyc);
    row<3>("v_fmac_f32_e32", d, cyc);
    row<4>("v_max_f32_e32", d, cyc);
    row<5>("v_mov_b32_e32", d, cyc);
    row<6>("v_mul_f32_e32 (sgpr)", d, cyc);
    row<8>("v_exp_f32_e32", d, cyc);
    row<13>("v_cvt_pk_bf16_f32", d, cyc);
    (void)hipFree(d);
    return 0;
}
