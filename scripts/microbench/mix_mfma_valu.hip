// Microbenchmark: how fast can one SIMD retire the instruction mix of one attention wave-tile
// (16 x v_mfma_f32_32x32x16_bf16 + the softmax VALU work on 32 scores per lane) with W waves per SIMD and no
// memory, LDS or barrier traffic?  Prints cycles per wave-tile per SIMD for several variants.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mix scripts/microbench/mix_mfma_valu.hip && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef RANDOM_DATA
#define RANDOM_DATA 0
#endif
__device__ inline float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: MFMA + VALU, 1: MFMA only, 2: VALU only, 3: MFMA + VALU without exp, 4: as 0 with the A operands read from LDS like the kernel (8 ds_read_b128 + 16 ds_read_b64_tr_b16 per tile)
__global__ __launch_bounds__(256) void mix(float* out, int iters, float c, float mc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384 + 16640];
    unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 99u;
    if (MODE == 4) { for (int i = threadIdx.x; i < (16384 + 16640) / 2; i += 256) ((__bf16*)lds)[i] = RANDOM_DATA ? (__bf16)rnd(seed) : (__bf16)(0.001f * i); __syncthreads(); }
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * (j + 1)); }
    if (RANDOM_DATA) for (int j = 0; j < 8; ++j) { a[j] = (__bf16)rnd(seed); b[j] = (__bf16)rnd(seed); }
    f32x16 s0, s1, o0, o1;
    for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; o0[e] = 0.f; o1[e] = 0.f; }
    float l = 0.f, m = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (MODE == 4) {
                    const unsigned char* kb = lds + (it & 1) * 8192 + r * 128 + (((2 * ks + h) ^ sw) << 4);
                    const bf16x8 k0 = *(const bf16x8*)kb, k1 = *(const bf16x8*)(kb + 4096);
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, b, s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, b, s1, 0, 0, 0);
                } else {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s1, 0, 0, 0);
                }
            }
        }
        bf16x8 pf[4];
        if (MODE != 1) {
            float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(fmaxf(mx, s0[e]), s1[e]);
            m = fmaxf(m, mx * 1e-30f);
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x0 = __builtin_fmaf(s0[e], c, -mc), x1 = __builtin_fmaf(s1[e], c, -mc);
                if (MODE == 3) { x0 = __builtin_fmaf(x0, c, mc); x1 = __builtin_fmaf(x1, c, mc); }
                else { x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); }
                s0[e] = x0; s1[e] = x1;
                ls += x0; ls += x1;
            }
            l += ls;
#pragma unroll
            for (int j = 0; j < 8; ++j) { pf[0][j] = (__bf16)s0[j]; pf[1][j] = (__bf16)s0[8 + j]; pf[2][j] = (__bf16)s1[j]; pf[3][j] = (__bf16)s1[8 + j]; }
        } else {
            pf[0] = a; pf[1] = b; pf[2] = a; pf[3] = b;
        }
        if (MODE != 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (MODE == 4) {
                    for (int dt = 0; dt < 2; ++dt) {
                        const unsigned char* vp = lds + 16384 + (it & 1) * 0 + dt * 4160 + kk * 1024 + v_rd;
                        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp));
                        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp + 512));
                        const i16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf[kk], o0, 0, 0, 0);
                        else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf[kk], o1, 0, 0, 0);
                    }
                } else {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[kk], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, pf[kk], o1, 0, 0, 0);
                }
            }
        } else {
            o0[0] += (float)pf[0][0] + (float)pf[3][7];
        }
        if (MODE == 1) { s0[0] = o0[1] * 1e-30f; s1[0] = o1[1] * 1e-30f; }
    }
    float res = l + m;
    for (int e = 0; e < 16; ++e) res += o0[e] + o1[e] + s0[e] + s1[e];
    out[blockIdx.x * 256 + threadIdx.x] = res;
}

// Two 32-row q-blocks per wave sharing every LDS operand read (K fragments and V^T fragments are read once
// and fed to both blocks' MFMAs): 32 MFMA + 2x the softmax VALU per iteration = two wave-tiles.
__global__ __launch_bounds__(256, 2) void mix2(float* out, int iters, float c, float mc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384 + 16640];
    unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 99u;
    for (int i = threadIdx.x; i < (16384 + 16640) / 2; i += 256) ((__bf16*)lds)[i] = RANDOM_DATA ? (__bf16)rnd(seed) : (__bf16)(0.001f * i);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    bf16x8 qa, qb;
    for (int j = 0; j < 8; ++j) { qa[j] = (__bf16)(0.01f * (threadIdx.x + j)); qb[j] = (__bf16)(0.02f * (j + 1)); }
    if (RANDOM_DATA) for (int j = 0; j < 8; ++j) { qa[j] = (__bf16)rnd(seed); qb[j] = (__bf16)rnd(seed); }
    f32x16 sa0, sa1, sb0, sb1, oa0, oa1, ob0, ob1;
    for (int e = 0; e < 16; ++e) { sa0[e] = sa1[e] = sb0[e] = sb1[e] = oa0[e] = oa1[e] = ob0[e] = ob1[e] = 0.f; }
    float la = 0.f, lb = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned char* kb = lds + (it & 1) * 8192 + r * 128 + (((2 * ks + h) ^ sw) << 4);
            const bf16x8 k0 = *(const bf16x8*)kb, k1 = *(const bf16x8*)(kb + 4096);
            sa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qa, sa0, 0, 0, 0);
            sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qa, sa1, 0, 0, 0);
            sb0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qb, sb0, 0, 0, 0);
            sb1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qb, sb1, 0, 0, 0);
        }
        bf16x8 pa[4], pb[4];
#define SOFTMAX(S0, S1, PF, L)                                                                        \
        {                                                                                             \
            float mx = fmaxf(S0[0], S1[0]);                                                           \
            _Pragma("unroll") for (int e = 1; e < 16; ++e) mx = fmaxf(fmaxf(mx, S0[e]), S1[e]);       \
            float ls = mx * 1e-30f;                                                                   \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                          \
                S0[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(S0[e], c, -mc));                        \
                S1[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(S1[e], c, -mc));                        \
                ls += S0[e]; ls += S1[e];                                                             \
            }                                                                                         \
            L += ls;                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { PF[0][j] = (__bf16)S0[j]; PF[1][j] = (__bf16)S0[8 + j]; PF[2][j] = (__bf16)S1[j]; PF[3][j] = (__bf16)S1[8 + j]; } \
        }
        SOFTMAX(sa0, sa1, pa, la)
        SOFTMAX(sb0, sb1, pb, lb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const unsigned char* vp = lds + 16384 + dt * 4160 + kk * 1024 + v_rd;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp + 512));
                const bf16x8 v8 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                if (dt == 0) { oa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v8, pa[kk], oa0, 0, 0, 0); ob0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v8, pb[kk], ob0, 0, 0, 0); }
                else { oa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v8, pa[kk], oa1, 0, 0, 0); ob1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v8, pb[kk], ob1, 0, 0, 0); }
            }
        }
    }
    float res = la + lb;
    for (int e = 0; e < 16; ++e) res += oa0[e] + oa1[e] + ob0[e] + ob1[e] + sa0[e] + sa1[e] + sb0[e] + sb1[e];
    out[blockIdx.x * 256 + threadIdx.x] = res;
}

static void run2(int wgs_per_cu, float* d) {
    const int iters = RANDOM_DATA ? 10000 : 1000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mix2, dim3(grid), dim3(256), 0, 0, d, 50, 1.4427f, 3.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mix2, dim3(grid), dim3(256), 0, 0, d, iters, 1.4427f, 3.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double us = best * 1e3 / (double(iters) * wgs_per_cu * 2);   // two 32-row wave-tiles per iteration
    std::printf("%-28s waves/SIMD=%d  %.3f ms  -> %.3f us per 32-row wave-tile per SIMD\n", "2 q-blocks/wave + LDS reads", wgs_per_cu, best, us);
}

template <int MODE>
static void run(const char* name, int wgs_per_cu, float* d) {
    const int iters = RANDOM_DATA ? 20000 : 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mix<MODE>, dim3(grid), dim3(256), 0, 0, d, 50, 1.4427f, 3.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mix<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.4427f, 3.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // per SIMD: wgs_per_cu waves, each `iters` wave-tiles
    const double us_per_tile = best * 1e3 / (double(iters) * wgs_per_cu);
    std::printf("%-28s waves/SIMD=%d  %.3f ms  -> %.3f us per wave-tile per SIMD (= %.0f cycles at 2.0 GHz)\n",
                name, wgs_per_cu, best, us_per_tile, us_per_tile * 2000.0);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("mfma+valu(exp)", 1, d); run<1>("mfma only", 1, d); run<2>("valu only", 1, d); run<3>("mfma+valu(no exp)", 1, d); }
        if (w == 2) { run<0>("mfma+valu(exp)", 2, d); run<1>("mfma only", 2, d); run<2>("valu only", 2, d); run<3>("mfma+valu(no exp)", 2, d); }
        if (w == 4) { run<0>("mfma+valu(exp)", 4, d); run<1>("mfma only", 4, d); run<2>("valu only", 4, d); run<3>("mfma+valu(no exp)", 4, d); run<4>("mfma+valu(exp)+LDS reads", 4, d); }
        if (w == 2) run<4>("mfma+valu(exp)+LDS reads", 2, d);
    }
    run2(1, d);
    run2(2, d);
    hipFree(d);
    return 0;
}
