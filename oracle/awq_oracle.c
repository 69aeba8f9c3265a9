/*
 * awq_oracle.c -- CPU restatement of the AutoAWQ int4 weight-only matmul path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under autoawq_amd/ may include, link, import or execute
 * this file.  Its only consumers are tests/, __graft_entry__.smoke() and the `cpu_baseline`
 * leg of bench.py, and there only as the checker / the reported CPU number, never as the
 * product path.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here against the golden
 * fixtures in the tests/golden .npz files, which were produced by running the reference itself
 * (tests/golden/make_golden.py imports /root/reference): dequantize_gemm, the naive
 * WQLinear_GEMM forward, and the three reference from_linear packers.  The GEMVFast dequant
 * arithmetic and the MoE helpers have no in-tree reference implementation (they live in the
 * un-vendored `autoawq-kernels` package, version unpinned, reference setup.py:53); for those the
 * oracle restates the semantics visible at the reference call sites and says so below.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Plain C99 + OpenMP; fp16 is handled in software so the file builds with any gcc.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AWQ_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- fp16 <-> fp32 (IEEE, RNE) */
static inline float h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | (ax > 0x7F800000u ? (0x200u | ((ax >> 13) & 0x3FFu)) : 0));
    if (ax >= 0x477FF000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7C00u);
    if (ax < 0x33000001u) /* <= 2^-25 rounds to zero (ties-to-even at exactly 2^-25) */
        return sign;
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, hexp;
    if (e < -14) { /* subnormal result */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    /* q carries the implicit bit for normals: (hexp<<10) + q - 0x400 ; carry propagates */
    uint32_t out = (hexp == 0) ? q : ((hexp << 10) + q - 0x400u);
    return (uint16_t)(sign | out);
}

AWQ_API float awq_oracle_h2f(uint16_t h) { return h2f(h); }
AWQ_API uint16_t awq_oracle_f2h(float f) { return f2h(f); }

/* ------------------------------------------------------------- GEMM layout (Appendix A.1) */

/* awq/utils/packing_utils.py:4-5  AWQ_ORDER / AWQ_REVERSE_ORDER.  Logical column 8c+j of a
 * packed word sits in nibble REV[j]. */
static const int AWQ_REV[8] = {0, 4, 1, 5, 2, 6, 3, 7};

/* unpack_awq (packing_utils.py:8-26) + reverse_awq_order (:29-43) + the `& 0xF` of
 * dequantize_gemm (:94-95).  q is [rows, words] int32; out is [rows, 8*words] uint8 in 0..15.
 * The reference shifts arithmetically, truncates to int8 and masks afterwards, which equals a
 * logical shift-and-mask for every 32-bit pattern; that is what is done here. */
AWQ_API void awq_oracle_unpack_gemm(const int32_t* q, int64_t rows, int64_t words, uint8_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const uint32_t* qr = (const uint32_t*)q + r * words;
        uint8_t* o = out + r * words * 8;
        for (int64_t c = 0; c < words; ++c) {
            uint32_t w = qr[c];
            for (int j = 0; j < 8; ++j) o[c * 8 + j] = (uint8_t)((w >> (4 * AWQ_REV[j])) & 0xFu);
        }
    }
}

/* dequantize_gemm (packing_utils.py:87-102): W[k,n] = fp16( (w[k,n] - z[k/g,n]) * s[k/g,n] ).
 * (w - z) is an int8 in [-15,15]; its product with an fp16 scale is exact in fp32 (5+11 bits),
 * so one fp32 multiply followed by one RNE rounding reproduces torch's int8*fp16 -> fp16. */
AWQ_API void awq_oracle_dequant_gemm(const int32_t* qweight, const int32_t* qzeros,
                                     const uint16_t* scales, int64_t K, int64_t N, int64_t g,
                                     uint16_t* W) {
    const int64_t words = N / 8;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < K; ++k) {
        const uint32_t* qw = (const uint32_t*)qweight + k * words;
        const uint32_t* qz = (const uint32_t*)qzeros + (k / g) * words;
        const uint16_t* s = scales + (k / g) * N;
        uint16_t* o = W + k * N;
        for (int64_t c = 0; c < words; ++c) {
            for (int j = 0; j < 8; ++j) {
                int sh = 4 * AWQ_REV[j];
                int wi = (int)((qw[c] >> sh) & 0xFu);
                int zi = (int)((qz[c] >> sh) & 0xFu);
                o[c * 8 + j] = f2h((float)(wi - zi) * h2f(s[c * 8 + j]));
            }
        }
    }
}

/* ------------------------------------------------------------- GEMV layout (Appendix A.3) */

/* awq/modules/linear/gemv.py:12-24 calculate_zeros_width */
AWQ_API int64_t awq_oracle_zeros_width(int64_t in_features, int64_t group_size) {
    int64_t mult = group_size >= 128 ? 1 : (group_size == 64 ? 2 : (group_size == 32 ? 4 : 0));
    if (!mult) return -1;
    int64_t bw = (in_features / group_size + 7) / 8;
    return (bw + mult - 1) / mult * mult;
}

/* Restates the packer awq/modules/linear/gemv.py:94-153: qweight[n, c] nibble i = w[n, 8c+i]
 * (ordinal order, :128); qzeros[n, c] nibble i = z[n, 8c+i]; scales[n, 8*ZW] zero padded.
 * Output W is written in [K, N] orientation so that it compares directly with the GEMM-layout W. */
AWQ_API void awq_oracle_dequant_gemv(const int32_t* qweight, const int32_t* qzeros,
                                     const uint16_t* scales, int64_t K, int64_t N, int64_t g,
                                     uint16_t* W) {
    const int64_t zw = awq_oracle_zeros_width(K, g);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const uint32_t* qw = (const uint32_t*)qweight + n * (K / 8);
        const uint32_t* qz = (const uint32_t*)qzeros + n * zw;
        const uint16_t* s = scales + n * zw * 8;
        for (int64_t k = 0; k < K; ++k) {
            int64_t grp = k / g;
            int wi = (int)((qw[k / 8] >> (4 * (k % 8))) & 0xFu);
            int zi = (int)((qz[grp / 8] >> (4 * (grp % 8))) & 0xFu);
            W[k * N + n] = f2h((float)(wi - zi) * h2f(s[grp]));
        }
    }
}

/* ---------------------------------------------------------- GEMVFast layout (Appendix A.4) */

/* Integer weight at (n,k) of a tensor packed by pack_intweight(interleave=4, kstride=64)
 * (awq/modules/linear/gemv_fast.py:26-65).  Closed form of that sequence of reshapes:
 * qweight[n/4, 64*(k/64) + 16*(n%4) + 8*h + t] nibble j  with  k%64 = 32*h + 8*j + t. */
static inline int fast_nibble(const uint16_t* qweight, int64_t K, int64_t n, int64_t k) {
    int64_t kb = k / 64, kr = k % 64;
    int64_t h = kr / 32, j = (kr % 32) / 8, t = kr % 8;
    uint16_t v = qweight[(n / 4) * K + 64 * kb + 16 * (n % 4) + 8 * h + t];
    return (v >> (4 * j)) & 0xF;
}

AWQ_API void awq_oracle_unpack_gemvfast(const int16_t* qweight, int64_t K, int64_t N, uint8_t* w_kn) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k)
            w_kn[k * N + n] = (uint8_t)fast_nibble((const uint16_t*)qweight, K, n, k);
}

/* GEMVFast stores scales [8*ZW, N] and qzeros = -(s*z) as fp16 [8*ZW, N]
 * (gemv_fast.py:175-181).  The kernel that consumes them is not in the reference tree
 * (awq_v2_ext, call sites gemv_fast.py:191-205), so its dequant arithmetic is UNPINNED there;
 * this oracle defines it as one fused multiply-add rounded once to fp16,
 * W = fp16(w*s + qzeros), the form the stored pre-multiplied zero implies. */
AWQ_API void awq_oracle_dequant_gemvfast(const int16_t* qweight, const uint16_t* scales,
                                         const uint16_t* qzeros, int64_t K, int64_t N, int64_t g,
                                         uint16_t* W) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) {
            int wi = fast_nibble((const uint16_t*)qweight, K, n, k);
            int64_t grp = k / g;
            /* w*s is exact in fp32 (4+11 bits); adding an fp16 is exact in double */
            double v = (double)wi * (double)h2f(scales[grp * N + n]) + (double)h2f(qzeros[grp * N + n]);
            W[k * N + n] = f2h((float)v);
        }
}

/* ------------------------------------------------------------------------------- products */

/* Naive branch awq/modules/linear/gemm.py:71-79: out = x_fp16 @ W_fp16 (+ bias).  The exact
 * product is accumulated in double; y32 receives it before the final rounding, y16 after.
 * (torch accumulates in fp32 in a BLAS-dependent order, so its fp16 result can differ from
 * y16 by an ulp; tests compare with a stated tolerance, not bit-exactly.) */
AWQ_API void awq_oracle_matmul(const uint16_t* x, const uint16_t* W, const uint16_t* bias,
                               int64_t M, int64_t K, int64_t N, float* y32, uint16_t* y16) {
    float* Wf = (float*)malloc(sizeof(float) * (size_t)K * (size_t)N);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < K * N; ++i) Wf[i] = h2f(W[i]);
    for (int64_t m = 0; m < M; ++m) {
        float* xf = (float*)malloc(sizeof(float) * (size_t)K);
        for (int64_t k = 0; k < K; ++k) xf[k] = h2f(x[m * K + k]);
#pragma omp parallel for schedule(static)
        for (int64_t n0 = 0; n0 < N; n0 += 64) {
            double acc[64];
            int64_t nb = N - n0 < 64 ? N - n0 : 64;
            for (int64_t j = 0; j < nb; ++j) acc[j] = 0.0;
            for (int64_t k = 0; k < K; ++k) {
                const float* wr = Wf + k * N + n0;
                double xv = xf[k];
                for (int64_t j = 0; j < nb; ++j) acc[j] += xv * (double)wr[j];
            }
            for (int64_t j = 0; j < nb; ++j) {
                double v = acc[j] + (bias ? (double)h2f(bias[n0 + j]) : 0.0);
                if (y32) y32[m * N + n0 + j] = (float)v;
                if (y16) y16[m * N + n0 + j] = f2h((float)v);
            }
        }
        free(xf);
    }
    free(Wf);
}

/* Whole CPU path of one WQLinear_GEMM call: dequantize_gemm then matmul (gemm.py:76-79). */
AWQ_API void awq_oracle_linear_gemm(const uint16_t* x, const int32_t* qweight, const int32_t* qzeros,
                                    const uint16_t* scales, const uint16_t* bias, int64_t M,
                                    int64_t K, int64_t N, int64_t g, float* y32, uint16_t* y16) {
    uint16_t* W = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)K * (size_t)N);
    awq_oracle_dequant_gemm(qweight, qzeros, scales, K, N, g, W);
    awq_oracle_matmul(x, W, bias, M, K, N, y32, y16);
    free(W);
}

/* ------------------------------------------------------------------------------------ MoE */

/* awq/modules/fused/moe.py:73-76 call site of silu_and_mul(out, gate_up): first half of the
 * last dim is the gate (fuse_linears([w1, w3]), awq/models/mixtral.py:131-138).  Kernel source
 * not in tree -> semantics restated: out = fp16( silu(fp32 gate) * fp32 up ). */
AWQ_API void awq_oracle_silu_and_mul(const uint16_t* gate_up, int64_t rows, int64_t d, uint16_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t i = 0; i < d; ++i) {
            float gt = h2f(gate_up[r * 2 * d + i]);
            float up = h2f(gate_up[r * 2 * d + d + i]);
            float s = gt / (1.0f + expf(-gt));
            out[r * d + i] = f2h(s * up);
        }
}

/* moe_align_block_size semantics, awq/modules/fused/moe.py:94-134 (docstring incl. the worked
 * example :111-119).  topk_ids flat [numel]; sorted_ids has numel + E*(block-1) slots pre-filled
 * with the sentinel `numel` (:129); expert_ids one entry per block; returns padded count. */
AWQ_API int32_t awq_oracle_moe_align(const int32_t* topk_ids, int64_t numel, int32_t num_experts,
                                     int32_t block, int32_t* sorted_ids, int32_t* expert_ids) {
    int64_t cap = numel + (int64_t)num_experts * (block - 1);
    for (int64_t i = 0; i < cap; ++i) sorted_ids[i] = (int32_t)numel;
    int32_t pos = 0, nblk = 0;
    for (int32_t e = 0; e < num_experts; ++e) {
        int32_t cnt = 0;
        for (int64_t i = 0; i < numel; ++i)
            if (topk_ids[i] == e) sorted_ids[pos + cnt++] = (int32_t)i;
        int32_t padded = (cnt + block - 1) / block * block;
        for (int32_t b = 0; b < padded / block; ++b) expert_ids[nblk++] = e;
        pos += padded;
    }
    return pos;
}
