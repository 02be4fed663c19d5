// Generic backward (any head_dim <= 1024, any L, fp32/bf16/f16 I/O, fp32 arithmetic): the correctness path
// for everything fa_bwd_kernels.hip does not serve, mirroring fa_fwd_generic_kernel on the forward side.
// Same decomposition (FA.cpp:452-472): P = exp(scale*S - lse), dP = dO V^T, dS = P o (dP - Dsum) * scale,
// dQ = dS K, dK = dS^T Q, dV = P^T dO; one workgroup per query row (dQ) and per key (dK, dV), so no
// gradient is summed across workgroups here either.

#include "fa_device.h"
#include "fa_extras.h"

namespace fa {

constexpr int GB_THREADS = 256;
constexpr int GB_CHUNK = 1024;
constexpr int GB_MAXD_PER_THREAD = 4;   // D <= 1024

template <typename T>
__global__ __launch_bounds__(64) void fa_bwd_generic_dsum_kernel(const BwdParams p) {
    const int64_t row_lin = blockIdx.x;   // over G*L
    const int g = (int)(row_lin / p.L), r = (int)(row_lin - (int64_t)g * p.L);
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* op = (const T*)p.o + (int64_t)bb * p.o_sb + (int64_t)hh * p.o_sh + (int64_t)r * p.o_sl;
    const T* gp = (const T*)p.dout + (int64_t)bb * p.do_sb + (int64_t)hh * p.do_sh + (int64_t)r * p.do_sl;
    float acc = 0.f;
    for (int d = threadIdx.x; d < p.D; d += 64) acc += (float)op[d] * (float)gp[d];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (threadIdx.x == 0) p.dsum[row_lin] = acc;
}

// dQ[g, row] : threads over keys <= row in chunks, then threads over d
template <typename T>
__global__ __launch_bounds__(GB_THREADS) void fa_bwd_generic_dq_kernel(const BwdParams p) {
    __shared__ float dsbuf[GB_CHUNK];
    extern __shared__ float rowbuf[];   // q row then dO row, D floats each
    const int L = p.L, D = p.D;
    const int g = blockIdx.x / L, row = blockIdx.x - g * L;
    const int bb = g / p.H, hh = g - bb * p.H;
    const T* qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hh * p.q_sh + (int64_t)row * p.q_sl;
    const T* gg = (const T*)p.dout + (int64_t)bb * p.do_sb + (int64_t)hh * p.do_sh + (int64_t)row * p.do_sl;
    const T* kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)(hh / p.kv_div) * p.k_sh;   // (grouped-query K/V head)
    const T* vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)(hh / p.kv_div) * p.v_sh;
    const int tid = threadIdx.x;
    float* qs = rowbuf;
    float* gs = rowbuf + D;
    for (int d = tid; d < D; d += GB_THREADS) { qs[d] = (float)qg[d]; gs[d] = (float)gg[d]; }
    __syncthreads();
    const float lse = p.lse[(int64_t)g * L + row], dsum = p.dsum[(int64_t)g * L + row];
    // attention mask / bias and dropout (fa_extras.h), as in fa_fwd_generic_kernel
    const bool has_mask = p.ex.mask != nullptr, dropping = p.ex.drop_thr != 0u;
    const char* const mbase = has_mask ? (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)hh * p.ex.m_sh) * (p.ex.mask_kind == 2 ? (int)sizeof(T) : 1) : nullptr;
    const unsigned drop_kg = drop_group_key(drop_key0_of(p.ex), (unsigned)g);
    float acc[GB_MAXD_PER_THREAD];
#pragma unroll
    for (int i = 0; i < GB_MAXD_PER_THREAD; ++i) acc[i] = 0.f;
    const int nkeys = p.causal ? row + 1 : p.L;
    for (int k0 = 0; k0 < nkeys; k0 += GB_CHUNK) {
        const int kn = nkeys - k0 < GB_CHUNK ? nkeys - k0 : GB_CHUNK;
        for (int c = tid; c < kn; c += GB_THREADS) {
            const T* kp = kg + (int64_t)(k0 + c) * p.k_sl;
            const T* vp = vg + (int64_t)(k0 + c) * p.v_sl;
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < D; ++d) {
                s = __builtin_fmaf(qs[d], (float)kp[d], s);
                dp = __builtin_fmaf(gs[d], (float)vp[d], dp);
            }
            s *= p.scale;
            if (has_mask) s += mask_load1<T>(p.ex.mask_kind, mbase, (int64_t)row * p.ex.m_sl, k0 + c, 1.0f);
            if (dropping) {
                const unsigned col = (unsigned)(k0 + c);
                const bool keep = mix32(drop_tile_key(drop_kg, (unsigned)row, col) ^ ((((unsigned)row & 0xffffu) << 16) | (col & 0xffffu))) >= p.ex.drop_thr;
                dp = keep ? dp * p.ex.drop_rp : 0.0f;
            }
            const float pr = lse == -INFINITY ? 0.0f : __expf(s - lse);      // (a row without any admitted key: P = 0)
            dsbuf[c] = pr * (dp - dsum) * p.scale;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GB_MAXD_PER_THREAD; ++i) {
            const int d = tid + i * GB_THREADS;
            if (d < D) {
                float a = acc[i];
                const T* kp = kg + (int64_t)k0 * p.k_sl + d;
                for (int c = 0; c < kn; ++c) a = __builtin_fmaf(dsbuf[c], (float)kp[(int64_t)c * p.k_sl], a);
                acc[i] = a;
            }
        }
        __syncthreads();
    }
    T* og = (T*)p.dq + (int64_t)bb * p.dq_sb + (int64_t)hh * p.dq_sh + (int64_t)row * p.dq_sl;
#pragma unroll
    for (int i = 0; i < GB_MAXD_PER_THREAD; ++i) {
        const int d = tid + i * GB_THREADS;
        if (d < D) og[d] = (T)acc[i];
    }
}

// dK[g, key], dV[g, key] : threads over query rows >= key in chunks, then threads over d
template <typename T>
__global__ __launch_bounds__(GB_THREADS) void fa_bwd_generic_dkdv_kernel(const BwdParams p) {
    __shared__ float pbuf[GB_CHUNK];
    __shared__ float dsbuf[GB_CHUNK];
    extern __shared__ float rowbuf[];   // k row then v row
    const int L = p.L, D = p.D;
    const int g = blockIdx.x / L, key = blockIdx.x - g * L;       // g: K/V group (batch entry x K/V head)
    const int bb = g / p.Hkv, hh = g - bb * p.Hkv;
    const T* kg = (const T*)p.k + (int64_t)bb * p.k_sb + (int64_t)hh * p.k_sh + (int64_t)key * p.k_sl;
    const T* vg = (const T*)p.v + (int64_t)bb * p.v_sb + (int64_t)hh * p.v_sh + (int64_t)key * p.v_sl;
    const int tid = threadIdx.x;
    float* ks = rowbuf;
    float* vs = rowbuf + D;
    for (int d = tid; d < D; d += GB_THREADS) { ks[d] = (float)kg[d]; vs[d] = (float)vg[d]; }
    __syncthreads();
    float ak[GB_MAXD_PER_THREAD], av[GB_MAXD_PER_THREAD];
#pragma unroll
    for (int i = 0; i < GB_MAXD_PER_THREAD; ++i) { ak[i] = 0.f; av[i] = 0.f; }
    const int row_lo = p.causal ? key : 0;   // causal: rows key .. L-1 can see this key; otherwise every row
    const int nrows = L - row_lo;
    const bool has_mask = p.ex.mask != nullptr, dropping = p.ex.drop_thr != 0u;
    const int mask_esz = p.ex.mask_kind == 2 ? (int)sizeof(T) : 1;
    for (int hq = hh * p.kv_div; hq < (hh + 1) * p.kv_div; ++hq) {   // every query head that reads this K/V head
        const T* qg = (const T*)p.q + (int64_t)bb * p.q_sb + (int64_t)hq * p.q_sh;
        const T* gg = (const T*)p.dout + (int64_t)bb * p.do_sb + (int64_t)hq * p.do_sh;
        const float* lse = p.lse + ((int64_t)bb * p.H + hq) * L;
        const float* dsum = p.dsum + ((int64_t)bb * p.H + hq) * L;
        const char* const mbase = has_mask ? (const char*)p.ex.mask + ((int64_t)bb * p.ex.m_sb + (int64_t)hq * p.ex.m_sh) * mask_esz : nullptr;
        const unsigned drop_kg = drop_group_key(drop_key0_of(p.ex), (unsigned)(bb * p.H + hq));
        for (int r0 = 0; r0 < nrows; r0 += GB_CHUNK) {
            const int rn = nrows - r0 < GB_CHUNK ? nrows - r0 : GB_CHUNK;
            for (int c = tid; c < rn; c += GB_THREADS) {
                const int row = row_lo + r0 + c;
                const T* qp = qg + (int64_t)row * p.q_sl;
                const T* gp = gg + (int64_t)row * p.do_sl;
                float s = 0.f, dp = 0.f;
                for (int d = 0; d < D; ++d) {
                    s = __builtin_fmaf((float)qp[d], ks[d], s);
                    dp = __builtin_fmaf((float)gp[d], vs[d], dp);
                }
                s *= p.scale;
                if (has_mask) s += mask_load1<T>(p.ex.mask_kind, mbase, (int64_t)row * p.ex.m_sl, key, 1.0f);
                const float pr = lse[row] == -INFINITY ? 0.0f : __expf(s - lse[row]);
                float pd = pr;           // what dV sees: P, or with dropout keep ? P / (1 - p) : 0
                if (dropping) {
                    const bool keep = mix32(drop_tile_key(drop_kg, (unsigned)row, (unsigned)key) ^ ((((unsigned)row & 0xffffu) << 16) | ((unsigned)key & 0xffffu))) >= p.ex.drop_thr;
                    dp = keep ? dp * p.ex.drop_rp : 0.0f;
                    pd = keep ? pr * p.ex.drop_rp : 0.0f;
                }
                pbuf[c] = pd;
                dsbuf[c] = pr * (dp - dsum[row]) * p.scale;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < GB_MAXD_PER_THREAD; ++i) {
                const int d = tid + i * GB_THREADS;
                if (d < D) {
                    float a = ak[i], b = av[i];
                    const T* qp = qg + (int64_t)(row_lo + r0) * p.q_sl + d;
                    const T* gp = gg + (int64_t)(row_lo + r0) * p.do_sl + d;
                    for (int c = 0; c < rn; ++c) {
                        a = __builtin_fmaf(dsbuf[c], (float)qp[(int64_t)c * p.q_sl], a);
                        b = __builtin_fmaf(pbuf[c], (float)gp[(int64_t)c * p.do_sl], b);
                    }
                    ak[i] = a; av[i] = b;
                }
            }
            __syncthreads();
        }
    }
    T* dkg = (T*)p.dk + (int64_t)bb * p.dk_sb + (int64_t)hh * p.dk_sh + (int64_t)key * p.dk_sl;
    T* dvg = (T*)p.dv + (int64_t)bb * p.dv_sb + (int64_t)hh * p.dv_sh + (int64_t)key * p.dv_sl;
#pragma unroll
    for (int i = 0; i < GB_MAXD_PER_THREAD; ++i) {
        const int d = tid + i * GB_THREADS;
        if (d < D) { dkg[d] = (T)ak[i]; dvg[d] = (T)av[i]; }
    }
}

template <typename T>
static hipError_t launch_bwd_generic_t(const BwdParams& p, hipStream_t stream) {
    const unsigned rows = (unsigned)((int64_t)p.G * p.L);
    const size_t dyn = 2 * (size_t)p.D * sizeof(float);
    hipError_t e = launch1((const void*)(fa_bwd_generic_dsum_kernel<T>), dim3(rows), dim3(64), 0, stream, p);
    if (e == hipSuccess) e = launch1((const void*)(fa_bwd_generic_dq_kernel<T>), dim3(rows), dim3(GB_THREADS), dyn, stream, p);
    const unsigned kv_rows = (unsigned)((int64_t)p.Gkv * p.L);
    if (e == hipSuccess) e = launch1((const void*)(fa_bwd_generic_dkdv_kernel<T>), dim3(kv_rows), dim3(GB_THREADS), dyn, stream, p);
    return e;
}

hipError_t launch_bwd_generic(const BwdParams& p, int dtype, hipStream_t stream) {
    if (dtype == 1) return launch_bwd_generic_t<__bf16>(p, stream);
    if (dtype == 0) return launch_bwd_generic_t<_Float16>(p, stream);
    return launch_bwd_generic_t<float>(p, stream);
}

}  // namespace fa
