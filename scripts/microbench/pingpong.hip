// Microbenchmark: what does a 32-key attention step of a 64-row wave (head_dim 64: 16 v_mfma_f32_32x32x16_bf16 + 4 row-sum
// v_mfma_f32_16x16x32_bf16, 12 LDS operand reads, 32 v_fma + 32 v_exp + 16 v_cvt_pk) cost per SIMD when the two waves of a SIMD
//   MODE 0  run the hand-interleaved mix of fa_fwd_pipe.hip (MFMA, its share of vector items, MFMA, ...) in two independent
//           256-thread workgroups per CU, one barrier per step (today's structure),
//   MODE 1  run in ONE 512-thread workgroup in complementary phases: waves 0-3 issue nothing but the step's MFMAs (+ LDS reads) while
//           waves 4-7 issue nothing but the step's vector work, then swap, an s_barrier between the phases (ping-pong),
//   MODE 2  as 1 with `s_setprio 1` on the matrix phase,
//   MODE 5  as 1 but ONE barrier per step (after each half's second phase): the halves keep their half-step stagger only loosely,
//   MODE 6  as 1 with ALL LDS operand reads in the vector phase (the matrix phase is MFMAs only),
//   MODE 7  as 6 with the row sums as 16 v_dot2c_f32_bf16 (P pair . (1, 1)) in the vector phase instead of 4 small MFMAs in the matrix phase,
//   MODE 3  matrix phases only (both halves), MODE 4 vector phases only: the two floors.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pingpong scripts/microbench/pingpong.hip && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }

constexpr int KH_B = 32 * 128, VBLK_B = 32 * 64 + 64, VH_B = 2 * VBLK_B, NSLOT = 4;

struct State {
    f32x16 s[2], o[2][2];
    f32x4 lsum[2];
    float ld[2];
    bf16x8 p[2][2], q[2][4], asel, pre[4], kpre[4];   // pre: the V^T fragments of the next matrix phase, requested at the end of the vector phase
};

// the matrix phase of one step: O += V^T P (8 MFMA), row sums (4 small MFMA), S = K Q^T (8 MFMA); LDS operands requested 3 ahead
template <bool ALLPRE, bool DOT>
__device__ __forceinline__ void matrix_phase(State& st, const unsigned char* kb, const unsigned char* vb, int k_rd0, int sw, int h, int v_rd) {
    bf16x8 opnd[8];
    auto load = [&](int x) {
        if (x < 4) {   // V^T fragment t = x: (dt = t % 2, kk = t / 2)
            const unsigned char* vp = vb + (x % 2) * VBLK_B + (x / 2) * 1024 + v_rd;
            const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp));
            const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp + 512));
            opnd[x] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        } else {
            const int ks = x - 4;
            opnd[x] = *(const bf16x8*)(kb + k_rd0 + (((2 * ks + h) ^ sw) << 4));
        }
    };
#pragma unroll
    for (int x = 0; x < 4; ++x) opnd[x] = st.pre[x];
    if constexpr (ALLPRE) {
#pragma unroll
        for (int x = 0; x < 4; ++x) opnd[4 + x] = st.kpre[x];
    } else {
        load(4); load(5); load(6);
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        if (!ALLPRE && x == 0) load(7);
        if (x < 4) {
#pragma unroll
            for (int b = 0; b < 2; ++b) st.o[b][x % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opnd[x], st.p[b][x / 2], st.o[b][x % 2], 0, 0, 0);
            if (!DOT && (x == 1 || x == 3)) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    st.lsum[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.asel, st.p[b][x / 2], st.lsum[b], 0, 0, 0);
                }
            }
        } else {
            const int ks = x - 4;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (ks == 0) {
                    f32x16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0.f;
                    st.s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opnd[x], st.q[b][ks], z, 0, 0, 0);
                } else {
                    st.s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opnd[x], st.q[b][ks], st.s[b], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the vector phase: P = 2^(S c - mc), packed to bf16
template <bool ALLPRE, bool DOT>
__device__ __forceinline__ void vector_phase(State& st, float c, float mc, const unsigned char* vb, int v_rd, const unsigned char* kb, int k_rd0, int sw, int h) {
    if constexpr (ALLPRE) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st.kpre[ks] = *(const bf16x8*)(kb + k_rd0 + (((2 * ks + h) ^ sw) << 4));
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const unsigned char* vp = vb + (x % 2) * VBLK_B + (x / 2) * 1024 + v_rd;
        const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp));
        const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp + 512));
        st.pre[x] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st.s[b][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st.s[b][e], c, -mc));
#pragma unroll
        for (int j = 0; j < 8; ++j) { st.p[b][0][j] = (__bf16)st.s[b][j]; st.p[b][1][j] = (__bf16)st.s[b][8 + j]; }
        if constexpr (DOT) {
            const bf16x2 one = {(__bf16)1.f, (__bf16)1.f};
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x2 pr = {st.p[b][k][2 * j], st.p[b][k][2 * j + 1]};
                    st.ld[b] = __builtin_amdgcn_fdot2_f32_bf16(pr, one, st.ld[b], false);
                }
        }
    }
}

// today's structure: each MFMA followed by its share of the vector work of the PREVIOUS scores (sc -> pn), compiler free inside a group
__device__ __forceinline__ void mixed_step(State& st, f32x16 (&sc)[2], const unsigned char* kb, const unsigned char* vb, int k_rd0, int sw, int h, int v_rd, float c, float mc) {
    bf16x8 opnd[8], pn[2][2];
    auto load = [&](int x) {
        if (x < 4) {
            const unsigned char* vp = vb + (x % 2) * VBLK_B + (x / 2) * 1024 + v_rd;
            const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp));
            const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vp + 512));
            opnd[x] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        } else {
            const int ks = x - 4;
            opnd[x] = *(const bf16x8*)(kb + k_rd0 + (((2 * ks + h) ^ sw) << 4));
        }
    };
    load(0); load(1); load(2);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        if (x + 3 < 8) load(x + 3);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (x < 4) {
                st.o[b][x % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opnd[x], st.p[b][x / 2], st.o[b][x % 2], 0, 0, 0);
            } else if (x == 4) {
                f32x16 z;
#pragma unroll
                for (int e = 0; e < 16; ++e) z[e] = 0.f;
                st.s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opnd[x], st.q[b][0], z, 0, 0, 0);
            } else {
                st.s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(opnd[x], st.q[b][x - 4], st.s[b], 0, 0, 0);
            }
            // vector share: MFMA n = 2 x + b of 16 handles score pair n (block n / 8, pair n % 8)
            const int n = 2 * x + b, vb_ = n / 8, e = 2 * (n % 8);
            sc[vb_][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[vb_][e], c, -mc));
            sc[vb_][e + 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[vb_][e + 1], c, -mc));
            pn[vb_][(n % 8) >> 2][2 * (n & 3)] = (__bf16)sc[vb_][e];
            pn[vb_][(n % 8) >> 2][2 * (n & 3) + 1] = (__bf16)sc[vb_][e + 1];
            if ((n & 3) == 3) st.lsum[vb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.asel, pn[vb_][(n % 8) >> 2], st.lsum[vb_], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) { st.p[b][0] = pn[b][0]; st.p[b][1] = pn[b][1]; }
}

template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512, MODE == 0 ? 2 : 1) void pp(float* out, unsigned long long* cyc, int iters, float c, float mc) {
    constexpr int NT = MODE == 0 ? 256 : 512, NG = NT / 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NG * NSLOT * (KH_B + VH_B)];
    unsigned seed = (blockIdx.x * NT + threadIdx.x) * 2654435761u + 99u;
    for (int i = threadIdx.x; i < (int)sizeof(lds) / 2; i += NT) ((__bf16*)lds)[i] = (__bf16)rnd(seed);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave >> 2;
    const int sw = (r >> 1) & 7, k_rd0 = r * 128;
    const int v_rd = (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const unsigned char* kbase = lds + half * NSLOT * (KH_B + VH_B);
    const unsigned char* vbase = kbase + NSLOT * KH_B;
    State st;
    for (int b = 0; b < 2; ++b) {
        for (int e = 0; e < 16; ++e) { st.s[b][e] = rnd(seed); st.o[b][0][e] = 0.f; st.o[b][1][e] = 0.f; }
        st.lsum[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        st.ld[b] = 0.f;
        for (int j = 0; j < 8; ++j) { st.p[b][0][j] = (__bf16)rnd(seed); st.p[b][1][j] = (__bf16)rnd(seed); st.pre[b][j] = (__bf16)rnd(seed); st.pre[2 + b][j] = (__bf16)rnd(seed); }
        for (int ks = 0; ks < 4; ++ks) for (int j = 0; j < 8; ++j) st.q[b][ks][j] = (__bf16)(0.05f * rnd(seed));
    }
    for (int x = 0; x < 4; ++x) for (int j = 0; j < 8; ++j) st.kpre[x][j] = (__bf16)rnd(seed);
    for (int j = 0; j < 8; ++j) st.asel[j] = (__bf16)((lane & 15) == 0 ? 1.f : 0.f);
    f32x16 sc[2];
    for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) sc[b][e] = rnd(seed);
    if (MODE == 2 && half == 1) {}   // (static priority variants are tried through the per-phase flips below)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mat = [&](int it) {
        if (MODE != 4) {
            if (MODE == 2) __builtin_amdgcn_s_setprio(1);
            matrix_phase<MODE == 6 || MODE == 7, MODE == 7>(st, kbase + (it & 3) * KH_B, vbase + (it & 3) * VH_B, k_rd0, sw, h, v_rd);
            if (MODE == 2) __builtin_amdgcn_s_setprio(0);
        }
    };
    auto vec = [&](int it) { if (MODE != 3) vector_phase<MODE == 6 || MODE == 7, MODE == 7>(st, c, mc, vbase + (it & 3) * VH_B, v_rd, kbase + ((it + 1) & 3) * KH_B, k_rd0, sw, h); };
    if constexpr (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            mixed_step(st, sc, kbase + (it & 3) * KH_B, vbase + (it & 3) * VH_B, k_rd0, sw, h, v_rd, c, mc);
#pragma unroll
            for (int b = 0; b < 2; ++b) sc[b] = st.s[b];
            __syncthreads();
        }
    } else if (half == 0) {     // (one branch on the half OUTSIDE the loop: each half runs a straight-line loop of its own)
        for (int it = 0; it < iters; ++it) { mat(it); if (MODE != 5) bar(); vec(it); bar(); }
    } else {
        if (MODE == 5) vec(0);
        for (int it = 0; it < iters; ++it) { if (MODE != 5) { vec(it); bar(); mat(it); } else { mat(it); vec(it); } bar(); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    for (int b = 0; b < 2; ++b) {
        for (int e = 0; e < 16; ++e) res += st.o[b][0][e] + st.o[b][1][e] + st.s[b][e] + sc[b][e];
        res += st.lsum[b][0] + (float)st.p[b][0][0] + st.ld[b];
    }
    out[blockIdx.x * NT + threadIdx.x] = res;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static float run_once(int iters, double* cycles) {
    constexpr int NT = MODE == 0 ? 256 : 512;
    const int blocks = MODE == 0 ? 512 : 256;
    static float* out = nullptr;
    static unsigned long long* cyc = nullptr;
    if (!out) { (void)hipMalloc(&out, sizeof(float) * 512 * 512); (void)hipMalloc(&cyc, sizeof(unsigned long long) * 512); }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    pp<MODE><<<blocks, NT>>>(out, cyc, iters, 0.18f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[512];
    (void)hipMemcpy(hc, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += (double)hc[i];
    *cycles = mean / blocks / (2.0 * iters) * (MODE == 0 ? 2 : 1) / (MODE == 0 ? 2 : 1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const char* names[8] = {"mixed, 2 x 256-thread workgroups per CU", "ping-pong, one 512-thread workgroup", "ping-pong, s_setprio 1 on the matrix phase",
                            "matrix phases only", "vector phases only", "half-step stagger, one barrier per step",
                            "ping-pong, all LDS reads in the vector phase", "ping-pong, LDS reads + dot2 row sums in the vector phase"};
    float ms[8][16];
    double cyc[8];
    for (int r = -1; r < rounds; ++r) {   // (round -1 = warm-up); the modes take turns inside every round: clock state is shared
        float t;
        double c;
        t = run_once<0>(iters, &c); if (r >= 0) { ms[0][r] = t; cyc[0] = c; }
        t = run_once<1>(iters, &c); if (r >= 0) { ms[1][r] = t; cyc[1] = c; }
        t = run_once<2>(iters, &c); if (r >= 0) { ms[2][r] = t; cyc[2] = c; }
        t = run_once<6>(iters, &c); if (r >= 0) { ms[6][r] = t; cyc[6] = c; }
        t = run_once<7>(iters, &c); if (r >= 0) { ms[7][r] = t; cyc[7] = c; }
        t = run_once<5>(iters, &c); if (r >= 0) { ms[5][r] = t; cyc[5] = c; }
        t = run_once<3>(iters, &c); if (r >= 0) { ms[3][r] = t; cyc[3] = c; }
        t = run_once<4>(iters, &c); if (r >= 0) { ms[4][r] = t; cyc[4] = c; }
    }
    for (int m = 0; m < 8; ++m) {
        float* a = ms[m];
        for (int i = 0; i < rounds; ++i) for (int j = i + 1; j < rounds; ++j) if (a[j] < a[i]) { float x = a[i]; a[i] = a[j]; a[j] = x; }
        const double med = a[rounds / 2];
        const double flops = 1024.0 * 2.0 * iters * 16 * 32768.0;   // 1024 SIMDs x 2 waves x iters steps x 16 big MFMAs
        printf("%-58s median %7.3f ms (min %7.3f)  %6.1f cycles per wave-step and SIMD  clock %.2f GHz  %7.1f TFLOP/s\n", names[m], med, a[0], cyc[m],
               cyc[m] * 2.0 * iters / (med * 1e6), flops / med * 1e-9);
    }
    return 0;
}
