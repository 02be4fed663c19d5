// Attention mask / additive bias and dropout for the MFMA kernels (the two modes of
// scaled_dot_product_attention the reference rejects, scaled_dot_product_attention.py:21-28).
//
// Mask: `kind` 1 = boolean bytes (non-zero = the key takes part, torch semantics), 2 = additive bias in the operand dtype;
// addressed mask[(g / H) * sb + (g % H) * sh + row * sl + key] (strides in ELEMENTS, 0 = broadcast, unit key stride).
// Both become an ADDITIVE term on the raw scores: bias / softmax_scale, or 0 / -inf - the kernels' arithmetic
// (P = 2^((s - m) c), c = scale log2 e) is unchanged.  A row with no admitted key gives l = 0 and O = NaN, as torch does.
//
// Dropout: keep(g, row, col) = hash >= thr with thr = floor(p 2^32): a counter-based generator, a pure function of
// (seed, offset, g, row, col), so the backward kernels regenerate the forward's mask exactly; 1 / (1 - p) is folded into the
// final scaling.  hash = mix32(kt ^ ((row & 0xffff) << 16 | (col & 0xffff))), kt = mix32(kg + ((row >> 16) << 16 | col >> 16)),
// kg = mix32(key0 ^ g * 0x9E3779B1), key0 = the 64-bit seed and offset folded on the host (fa_capi.cpp: drop_key0);
// mix32 = the "lowbias32" integer hash (two multiplies, three xor-shifts; bijective, so distinct (row, col) of one
// 65536 x 65536 window never share an input).  oracle/dropout.py restates it in numpy for the tests.
#ifndef FA_EXTRAS_H
#define FA_EXTRAS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

__host__ __device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// (seed, offset) -> key0: on the host for immediate values (fa_capi.cpp), on the device when the two words arrive in a device buffer
// (ABI 5 `dropout_state`: a launch captured in a HIP graph then sees the offset its caller's captured increment left there, i.e. a
// fresh mask on every replay - with immediate values the graph would replay ONE mask forever)
__host__ __device__ __forceinline__ unsigned drop_fold_key0(unsigned long long seed, unsigned long long offset) {
    unsigned k0 = mix32((unsigned)seed ^ 0x9E3779B9u);
    k0 = mix32(k0 ^ (unsigned)(seed >> 32));
    k0 = mix32(k0 ^ (unsigned)offset);
    k0 = mix32(k0 ^ (unsigned)(offset >> 32));
    return k0;
}
__host__ __device__ __forceinline__ unsigned drop_group_key(unsigned key0, unsigned g) { return mix32(key0 ^ (g * 0x9E3779B1U)); }
// key of the 65536 x 65536 window holding (row, col): uniform over a kernel tile (tiles are 64-aligned)
__host__ __device__ __forceinline__ unsigned drop_tile_key(unsigned kg, unsigned row, unsigned col) {
    return mix32(kg + (((row >> 16) << 16) | (col >> 16)));
}

#if defined(__HIPCC__)   /* device-side loaders (fa_capi.cpp's host-only sanitizer build includes this header for mix32 alone) */
// An additive bias reaches the RAW scores as bias / softmax_scale.  A finite "minus infinity" of the operand dtype - the common
// torch.finfo(dtype).min padding convention, -3.39e38 in bf16 - must stay FINITE on the way: -3.39e38 * 8 is -inf in fp32, and a row
// whose keys are all padded that way would then end with l = 0 and O = NaN where torch returns the uniform average of V (its masked
// scores are all equal).  So the converted term is held at a floor of -2^99 unless the bias is exactly -inf:
//   * far below any real score, and it ABSORBS the raw score (|s| < 2^75): every padded key of a row gets the same logit, as in torch;
//   * a power of two, so that the row constants derived from it are exact: a fully padded row has m = -2^99, m c and
//     lse = m scale (log l is absorbed) without rounding, hence the backward's exponent (s c - lse log2 e) is exactly 0 and its
//     P is 1 - finite, never the 2^(+-1e22) a rounded floor gives (such a row's true P is 1 / l: stock SDPA's fused backward has
//     the same artefact, and a padded row's dO is normally zero);
//   * small enough that (floor - m) scale log2(e) cannot overflow in the exponent arithmetic for any scale up to 2^28.
__device__ __forceinline__ float bias_term(float bias, float inv_scale) {
    const float t = bias * inv_scale;
    return bias == -INFINITY ? t : __builtin_fmaxf(t, -0x1p99f);
}
// key0 of this launch: from the device state when there is one (two uniform 8-byte loads per workgroup), else the folded immediates
template <typename EX>
__device__ __forceinline__ unsigned drop_key0_of(const EX& ex) {
    return ex.drop_state != nullptr ? drop_fold_key0(ex.drop_state[0], ex.drop_state[1]) : ex.drop_key0;
}
// four consecutive keys k0 .. k0+3 of one mask row -> additive terms on the raw scores
template <typename T>
__device__ __forceinline__ void mask_load4(int kind, bool fast, const char* mrow, int k0, int L, float inv_scale, float (&a)[4]) {
    typedef T vec4 __attribute__((ext_vector_type(4)));
    if (kind == 2) {
        const T* mp = (const T*)mrow;
        if (fast) {
            const vec4 v = *(const vec4*)(mp + k0);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = bias_term((float)v[i], inv_scale);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = bias_term((float)mp[k0 + i < L ? k0 + i : L - 1], inv_scale);
        }
    } else {
        const unsigned char* mp = (const unsigned char*)mrow;
        if (fast) {
            const unsigned w = *(const unsigned*)(mp + k0);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = ((w >> (8 * i)) & 0xffu) ? 0.f : -INFINITY;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = mp[k0 + i < L ? k0 + i : L - 1] ? 0.f : -INFINITY;
        }
    }
}
// one element (the key-on-the-lane kernels: consecutive lanes read consecutive keys of one row)
template <typename T>
__device__ __forceinline__ float mask_load1(int kind, const char* mbase, int64_t row_off_elems, int key, float inv_scale) {
    if (kind == 2) return bias_term((float)((const T*)mbase)[row_off_elems + key], inv_scale);
    return ((const unsigned char*)mbase)[row_off_elems + key] ? 0.f : -INFINITY;
}

#endif  // __HIPCC__

}  // namespace fa
#endif
