// Microbenchmark: issue cost of single VALU instructions on gfx950, alone and under a stream of MFMAs.
// Each wave runs an unrolled loop of 8 independent chains of ONE instruction; 4 waves per SIMD on every CU.
// Prints cycles per instruction per SIMD assuming the clock measured from the v_fma_f32 loop = 4 cycles.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/microbench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP, int MFMA>
__global__ __launch_bounds__(256) void rate(float* out, int iters, float c) {
    float r[8]; f32x2 p[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { r[i] = 0.001f * (threadIdx.x + i); p[i] = f32x2{r[i], r[i] * 0.5f}; u[i] = threadIdx.x * 2654435761u + i; }
    f32x2 cc = {c, c * 0.999f};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (j + 1)); }
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MFMA) {   // one MFMA per 8 VALU instructions (the attention kernel has ~1 per 7)
                if (rep & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
            }
#define ONE(i)                                                                                                   \
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));                          \
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cc));                      \
            if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));                          \
            if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));                          \
            if (OP == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));                                           \
            if (OP == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                              \
            if (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[i]), "v"(c));           \
            if (OP == 7) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));                         \
            if (OP == 8) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(u[i]), "v"(u[(i + 1) & 7])); \
            if (OP == 9) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));                                           \
            if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "v"(cc));   \
            if (OP == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                             \
            if (OP == 12) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));             \
            if (OP == 13) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[i]) : "v"(u[i]));                        \
            if (OP == 14) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            REP8(ONE)
#undef ONE
        }
    }
    float res = 0.f;
    for (int i = 0; i < 8; ++i) res += r[i] + p[i][0] + p[i][1] + (float)u[i];
    for (int e = 0; e < 16; ++e) res += acc0[e] + acc1[e];
    out[blockIdx.x * 256 + threadIdx.x] = res;
}

template <int OP, int MFMA>
static double run(float* d) {
    const int iters = 4000, grid = 256 * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<OP, MFMA>), dim3(grid), dim3(256), 0, 0, d, 200, 1.0001f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((rate<OP, MFMA>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // per SIMD: 4 waves x iters x 32 instructions
    return best * 1e6 / (double(iters) * 4 * 32);   // ns per instruction per SIMD
}

int main() {
    float* d; (void)hipMalloc(&d, 1024 * 256 * sizeof(float));
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_exp_f32", "v_add_f32", "v_cvt_pk_bf16_f32",
                           "v_max3_f32", "v_dot2_f32_bf16", "v_exp_f16", "v_pk_fma_f32(op_sel)", "v_mul_f32", "v_pk_add_f16", "v_ldexp_f32", "v_perm_b32"};
    double t[15], tm[15];
    t[0] = run<0, 0>(d); t[1] = run<1, 0>(d); t[2] = run<2, 0>(d); t[3] = run<3, 0>(d); t[4] = run<4, 0>(d); t[5] = run<5, 0>(d); t[6] = run<6, 0>(d);
    t[7] = run<7, 0>(d); t[8] = run<8, 0>(d); t[9] = run<9, 0>(d); t[10] = run<10, 0>(d); t[11] = run<11, 0>(d); t[12] = run<12, 0>(d); t[13] = run<13, 0>(d); t[14] = run<14, 0>(d);
    tm[0] = run<0, 1>(d); tm[1] = run<1, 1>(d); tm[2] = run<2, 1>(d); tm[3] = run<3, 1>(d); tm[4] = run<4, 1>(d); tm[5] = run<5, 1>(d); tm[6] = run<6, 1>(d);
    tm[7] = run<7, 1>(d); tm[8] = run<8, 1>(d); tm[9] = run<9, 1>(d); tm[10] = run<10, 1>(d); tm[11] = run<11, 1>(d); tm[12] = run<12, 1>(d); tm[13] = run<13, 1>(d); tm[14] = run<14, 1>(d);
    const double cyc = t[0] / 4.0;   // ns per cycle if v_fma_f32 = 4 cycles
    std::printf("clock from v_fma_f32 = 4 cycles: %.3f GHz\n", 1.0 / cyc);
    for (int i = 0; i < 15; ++i)
        std::printf("%-22s alone %.2f cycles/instr   with 1 MFMA per 8: %.2f cycles/instr (= +%.1f cycles per MFMA)\n", names[i], t[i] / cyc, tm[i] / cyc,
                    (tm[i] - t[i]) / cyc * 8.0);
    (void)hipFree(d);
    return 0;
}
