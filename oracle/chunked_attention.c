/*
 * Chunked online-softmax CPU restatement of the reference forward graph (ORACLE - test
 * infrastructure; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it).
 *
 * Restates the *program* that flash_attention_ipu/cpp/flash_attention_qkv_packed.cpp builds:
 *   - getTriuOffsetSequence            FA.cpp:47-126   -> fa_oracle_triu_offsets()
 *   - makeDynamicCausalMaskBlocks      FA.cpp:133-159  -> mask value at (r,c) of block with offset k
 *   - dynamicAddMask (cyclic counter)  FA.cpp:161-173  -> mask_ctr = (mask_ctr+1) % n_offsets
 *   - flashAttentionQKVPackedWithStash FA.cpp:175-334  -> fa_oracle_chunked_{f32,f64}()
 * Arithmetic order per executed block (FA.cpp:254-300):
 *   S = q_i k_j^T ; [S += mask] ; m_new = max(m_i, rowmax S) ; c = exp(m_i - m_new) ; m_i = m_new ;
 *   P = exp(S - m_new) ; s = rowsum P ; l = l*c + s ; o = o*c + P v_j
 * and per q-chunk (FA.cpp:314-316):  o /= l ; lse = m + log(l).
 * Running max starts at -10000 (FA.cpp:238), masks are additive -10000 (FA.cpp:141); no 1/sqrt(D).
 * All arithmetic is in the element type (no mixed-precision accumulate, FA.cpp:254).
 *
 * The reference never terminates for kv chunk length 1 (FA.cpp:106-124); this restatement returns
 * -2 there instead of hanging.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libfa_oracle.so oracle/chunked_attention.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FA_ORACLE_MASK (-10000.0)

/* FA.cpp:47-126.  Returns the number of offsets written (<= cap), or -2 when the reference's own
 * loop would not terminate (numCols == 1 and numRows > 1 ... tmp never returns to 1). */
int fa_oracle_triu_offsets(uint32_t numRows, uint32_t numCols, int32_t* out, int cap) {
    int n = 0;
    if (cap > 0) out[n] = 1;
    n = 1;
    int tmp = 1;
    int max_offset = (int)numCols - 1;
    int min_offset = 2 - (int)numRows;
    long guard = 0;
    for (;;) {
        tmp += (int)numRows;
        if (tmp > max_offset) tmp -= (int)(numRows + numCols);
        if (tmp == 1) break;
        if (tmp >= min_offset) {
            if (n < cap) out[n] = tmp;
            ++n;
        }
        if (++guard > 4L * ((long)numRows + (long)numCols) + 16) return -2;
    }
    return n;
}

/* mask block with triu offset k, FA.cpp:141-151: filled with -10000, then for each row r the
 * columns [0, min(r + k, n)) are zeroed when r + k > 0. */
static inline double mask_at(int r, int c, int k) { return (c < r + k) ? 0.0 : FA_ORACLE_MASK; }

#define DEFINE_CHUNKED(NAME, T, EXP, LOG, FMAX)                                                      \
    int NAME(const T* qkv, int64_t G, int64_t L, int64_t D, int32_t nq, int32_t nk, T* out, T* lse) { \
        if (nq <= 0 || nk <= 0 || L % nq != 0 || L % nk != 0) return -1; /* FA.cpp:190-191 */         \
        const int64_t Bq = L / nq, Bk = L / nk;                                                        \
        int32_t* offs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(Bq + Bk + 4));                     \
        int noff = fa_oracle_triu_offsets((uint32_t)Bq, (uint32_t)Bk, offs, (int)(Bq + Bk + 4));       \
        if (noff < 0) { free(offs); return -2; }                                                       \
        const T* Q = qkv;                                                                              \
        const T* K = qkv + G * L * D;                                                                  \
        const T* V = qkv + 2 * G * L * D;                                                              \
        T* S = (T*)malloc(sizeof(T) * (size_t)(Bq * Bk));                                              \
        T* m = (T*)malloc(sizeof(T) * (size_t)Bq);                                                     \
        T* l = (T*)malloc(sizeof(T) * (size_t)Bq);                                                     \
        T* o = (T*)malloc(sizeof(T) * (size_t)(Bq * D));                                               \
        /* The reference runs all groups inside each block; groups are independent, and the mask      \
         * counter advances once per masked block for all groups together, so it is a function of     \
         * (i, j) only: replay the counter per group. */                                              \
        for (int64_t g = 0; g < G; ++g) {                                                              \
            int ctr = 0;                                                                               \
            for (int64_t i = 0; i < nq; ++i) {                                                         \
                for (int64_t r = 0; r < Bq; ++r) { m[r] = (T)FA_ORACLE_MASK; l[r] = 0; }               \
                memset(o, 0, sizeof(T) * (size_t)(Bq * D));                                            \
                for (int64_t j = 0; j < nk; ++j) {                                                     \
                    if (!((i + 1) * Bq > j * Bk)) continue; /* doBlock, FA.cpp:244 */                  \
                    const T* qi = Q + (g * L + i * Bq) * D;                                            \
                    const T* kj = K + (g * L + j * Bk) * D;                                            \
                    const T* vj = V + (g * L + j * Bk) * D;                                            \
                    for (int64_t r = 0; r < Bq; ++r)                                                   \
                        for (int64_t c = 0; c < Bk; ++c) {                                             \
                            T acc = 0;                                                                 \
                            for (int64_t d = 0; d < D; ++d) acc += qi[r * D + d] * kj[c * D + d];      \
                            S[r * Bk + c] = acc;                                                       \
                        }                                                                              \
                    if (i * Bq < (j + 1) * Bk - 1) { /* doMask, FA.cpp:268 */                          \
                        int k = offs[ctr];                                                             \
                        for (int64_t r = 0; r < Bq; ++r)                                               \
                            for (int64_t c = 0; c < Bk; ++c) S[r * Bk + c] += (T)mask_at((int)r, (int)c, k); \
                        ctr = (ctr + 1) % noff;                                                        \
                    }                                                                                  \
                    for (int64_t r = 0; r < Bq; ++r) {                                                 \
                        T mx = S[r * Bk];                                                              \
                        for (int64_t c = 1; c < Bk; ++c) mx = FMAX(mx, S[r * Bk + c]);                 \
                        T mnew = FMAX(m[r], mx);                                                       \
                        T cfac = EXP(m[r] - mnew);                                                     \
                        m[r] = mnew;                                                                   \
                        T s = 0;                                                                       \
                        for (int64_t c = 0; c < Bk; ++c) {                                             \
                            S[r * Bk + c] = EXP(S[r * Bk + c] - mnew);                                 \
                            s += S[r * Bk + c];                                                        \
                        }                                                                              \
                        l[r] = l[r] * cfac + s;                                                        \
                        for (int64_t d = 0; d < D; ++d) {                                              \
                            T acc = o[r * D + d] * cfac;                                               \
                            for (int64_t c = 0; c < Bk; ++c) acc += S[r * Bk + c] * vj[c * D + d];     \
                            o[r * D + d] = acc;                                                        \
                        }                                                                              \
                    }                                                                                  \
                }                                                                                      \
                for (int64_t r = 0; r < Bq; ++r) {                                                     \
                    for (int64_t d = 0; d < D; ++d)                                                    \
                        out[(g * L + i * Bq + r) * D + d] = o[r * D + d] / l[r];                       \
                    if (lse) lse[g * L + i * Bq + r] = m[r] + LOG(l[r]);                               \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
        free(S); free(m); free(l); free(o); free(offs);                                                \
        return 0;                                                                                      \
    }

DEFINE_CHUNKED(fa_oracle_chunked_f32, float, expf, logf, fmaxf)
DEFINE_CHUNKED(fa_oracle_chunked_f64, double, exp, log, fmax)
