// Device-side helpers shared by the forward kernels (types, MFMA wrappers, half-wave exchange).
#ifndef FA_DEVICE_H
#define FA_DEVICE_H

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "fa_params.h"

namespace fa {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Elt;
template <> struct Elt<__bf16> {
    using vec8 = bf16x8;
    static constexpr bool is_f16 = false;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static constexpr unsigned short one_bits = 0x3F80;   // 1.0
};
template <> struct Elt<_Float16> {
    using vec8 = f16x8;
    static constexpr bool is_f16 = true;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static constexpr unsigned short one_bits = 0x3C00;   // 1.0
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// The build uses -fno-honor-nans: without it fmaxf() on MFMA results costs a canonicalising
// v_max_f32 x,x per operand (IEEE mode must quiet signalling NaNs).  Do NOT replace these by inline
// asm: hipcc does not pad the MFMA-result -> VALU-read hazard for an asm statement, and the max then
// reads stale scores.
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float max2(float a, float b) { return __builtin_fmaxf(a, b); }

// Exchange between the two 32-lane halves of a wave with ONE v_permlane32_swap: called with the same
// value in both operands it returns {lower half's value, upper half's value} in every lane.
// (Copy the two results to scalars before __builtin_bit_cast: bit-casting the vector ELEMENT r[1]
// directly reads element 0 with this clang.)
struct HalfPair { float lo, hi; };
__device__ __forceinline__ HalfPair half_exchange(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    return {__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}
__device__ __forceinline__ float half_swap_max(float x) {
    const HalfPair p = half_exchange(x);
    return max2(p.lo, p.hi);
}
__device__ __forceinline__ float half_swap_sum(float x) {
    const HalfPair p = half_exchange(x);
    return p.lo + p.hi;
}

// ---------------------------------------------------------------------------------------------
// MFMA tile kernel.  T in {__bf16, _Float16}, D in {64, 128}, NWAVES in {1, 2, 4, 8}.
// ---------------------------------------------------------------------------------------------

// ---- LDS-DMA through inline asm -------------------------------------------------------------------
// `buffer_load_dwordx4 ... lds`: 16 bytes per lane from (descriptor base + voffset) straight into
// LDS[lds_base + 16*lane]; no VGPR destination.  Issued by asm on purpose: hipcc tracks an LDS-DMA
// issued through its builtin as a pending LDS write and drains `vmcnt(0)` in front of the first
// ds_read_b64_tr_b16 that follows - i.e. it waits for the NEXT tile's prefetch in the middle of the
// current tile.  An asm statement is invisible to that bookkeeping, so the wait is placed by hand
// (lds_dma_wait_all() right before the workgroup barrier that publishes the tile).
typedef int i32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor over [base, base + num_bytes): stride 0, reads past the end return zeros
__device__ __forceinline__ i32x4 make_raw_rsrc(const void* base, unsigned num_bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
    r[2] = (int)num_bytes;
    r[3] = 0x00020000;
    return r;
}

// lds_base and rsrc must be wave-uniform (they travel in SGPRs; M0 carries the LDS base and is
// written and restored inside the same statement, as the compiler reserves it)
__device__ __forceinline__ void lds_dma_b128(i32x4 rsrc, unsigned lds_base, unsigned voffset) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_base), "v"(voffset), "s"(rsrc)
        : "memory");
}
// the same with a compile-time byte offset (0 .. 4095) added to the lane's: pieces that differ by a constant share one offset register.
// (The instruction's offset field moves BOTH ends of an LDS load - memory address and LDS address: M0 carries lds_base - IMM.)
template <int IMM>
__device__ __forceinline__ void lds_dma_b128_imm(i32x4 rsrc, unsigned lds_base, unsigned voffset) {
    static_assert(IMM >= 0 && IMM < 4096, "12-bit instruction offset");
    lds_base -= IMM;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, 0 offen offset:%4 lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_base), "v"(voffset), "s"(rsrc), "n"(IMM)
        : "memory");
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

// ---- row sums of P on the matrix pipe ---------------------------------------------------------------
// The forward at head_dim 64 is bound by VALU issue, not by the matrix pipe (~7 vector
// instructions per MFMA against a budget of ~5), and a quarter of those instructions are the fp32 adds of the
// softmax denominator.  One v_mfma_f32_16x16x32 with a constant 0/1 selector as A sums the 8 packed P values a
// lane holds as the B operand of the PV product AND the 8 of its partner in the other half-wave, for all 32 rows
// of a query block at once (16 matrix-pipe cycles and one issue slot instead of 8 v_add_f32 = 32 issue cycles):
//   B operand of v_mfma_f32_32x32x16 (what a lane holds): P[k = 8 (lane >> 5) + j][n = lane & 31], j = 0..7
//   the same registers read by v_mfma_f32_16x16x32:       B'[k' = 8 (lane >> 4) + j][n' = lane & 15]
//   so column n' collects lanes n', n'+16, n'+32, n'+48 = rows n' (both halves) and n'+16 (both halves);
//   A[i][k'] = 1 where ((i >> 2) & 1) == ((k' >> 3) & 1) picks the even lane groups for output rows 0-3 / 8-11
//   and the odd ones for rows 4-7 / 12-15: D[i = 4 (lane >> 4) + jj][n'] lands the FULL row sum of query row
//   lane & 31 in every lane (all four result registers equal) - no half-wave exchange needed afterwards.
// The sum is taken over the 16-bit-rounded P, i.e. exactly the weights the PV product uses.
// A 0 in the selector times an overflowed P (+inf) gives NaN, so overflow checks on these sums use
// sum_left_safe_range(): an integer compare that NaN cannot slip through (the build uses -fno-honor-nans).
template <typename T>
__device__ __forceinline__ typename Elt<T>::vec8 rowsum_selector(int lane) {
    const bool one = (((lane & 15) >> 2) & 1) == ((lane >> 4) & 1);
    const unsigned w = one ? (unsigned)Elt<T>::one_bits * 0x10001u : 0u;
    const u32x4 v = {w, w, w, w};
    return __builtin_bit_cast(typename Elt<T>::vec8, v);
}
__device__ __forceinline__ bool sum_left_safe_range(float x) {   // |x| > 2^100, +-inf or NaN
    return (__builtin_bit_cast(unsigned, x) & 0x7fffffffu) > 0x71800000u;
}

// Lazy rescale threshold in log2 units: the running max m is only raised (and O, l rescaled) when a
// tile's row max exceeds it by more than this, so P = 2^((s-m)c) stays <= 2^8 - exact in fp32
// accumulators, harmless in bf16/f16 P (same relative rounding), and normalised away by O/l.
constexpr float RESCALE_THR = 8.0f;

}  // namespace fa
// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) - indices usable in `if constexpr`
template <int B, int E, typename F>
__device__ __forceinline__ void fa_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        fa_static_for<B + 1, E>(f);
    }
}

#endif
