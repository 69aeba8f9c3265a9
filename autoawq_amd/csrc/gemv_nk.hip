// gemv_nk.hip -- decode GEMV / skinny GEMM (1 <= M <= 16) on the GEMV layout, gfx950.
//
// Replaces awq_ext.gemv_forward_cuda(x, qweight, scales, qzeros, group_size) (M <= 8) and
// awq_ext.gemmv2_forward_cuda(..., split_k_iters) (M > 8) as called by
// awq/modules/linear/gemv.py:168-180.  Layout (packer gemv.py:94-153):
//   qweight [N, K/8] int32, nibble i of word c = w[n, 8c+i] (ordinal order);
//   qzeros  [N, ZW]  int32, nibble i of word c = z[n, group 8c+i];
//   scales  [N, 8*ZW] fp16, zero padded past K/g groups.
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 + (K/g)*N/2 (zeros, nibbles actually used)
// + (K/g)*N*2 + M*K*2 + M*N*2.
//
// Why this layout is the natural decode format on MI355X: every output row n is a CONTIGUOUS run
// of K/2 bytes, so a block owns 16 whole rows (one contiguous 32-88 KiB region, read linearly) and
// no cross-CU split-K exchange exists at all -- the K reduction is done by the MFMA itself and by
// one LDS fold across the block's waves.
//
// How a packed word feeds v_mfma_f32_16x16x32_f16: a word IS a lane's B fragment (8 K-adjacent
// weights of one output row).  (q >> 4t) & 0x000F000F pairs nibbles (t, t+4); shifted to mantissa
// position and OR-ed with 0x4C00 that is the exact fp16 pair (16 + w[8c+t], 16 + w[8c+t+4]), one
// shift + one v_and_or per pair.  The A fragment is the activation chunk in the same (0,4,1,5,2,
// 6,3,7) slot order -- permuted ONCE while it is staged in LDS, so the K loop reads it with one
// ds_read_b128 and spends no VALU on it.  A rows are batch rows: M <= 16 costs nothing extra.
// Group factorisation as in gemv_mfma.hip:  y += s[n,g] * (acc - (16 + z[n,g]) * sum_k x), folded
// once per 128 K (= one loop iteration), sum_k x from an MFMA against an all-ones B.
#include "awq_device.h"
#include "awq_internal.h"

namespace {

struct GemvNkParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    half_t* y;
    int M, K, N, g;
    int KW, ZW;   // words per qweight / qzeros row
    int xpitch;   // halfs per staged activation row in LDS (K + 8)
    uint32_t g_magic, xc1_magic, zw_magic;  // (v * magic) >> 32 == v / {g, K/8 + 1, ZW} for the values divided (awq_magic_u32); zw_magic 0: ZW == 1
};

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// Phase timestamps for tools/trace_gemv.py (debug build only: -DAWQ_GEMV_TRACE)
#ifdef AWQ_GEMV_TRACE
__device__ unsigned long long* g_awq_trace_nk = nullptr;
#define NK_STAMP(slot)                                                                                \
    do {                                                                                              \
        if (g_awq_trace_nk && lane == 0)                                                              \
            g_awq_trace_nk[((size_t)blockIdx.x * NWAVES + wave) * 16 + (slot)] = wall_clock64();      \
    } while (0)
#else
#define NK_STAMP(slot) do { } while (0)
#endif

AWQ_DEV float4_t mfma16(u32x4v a, u32x4v b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0,
                                                  0, 0);
}

// fp16 pair (bias_T + nibble t, bias_T + nibble t+4) of a packed word with the nibbles left in
// place: bits 0-3 / 16-19 under exponent 2^10 (bias 1024), bits 4-7 / 20-23 under 2^6 (bias 64),
// nibbles 2, 3, 6, 7 through one shared `q >> 8`: 5 VALU ops per word (see gemv_mfma.hip).
template <int T>
AWQ_DEV uint32_t pairbo(uint32_t q, uint32_t q8) {
    if constexpr (T == 0) return and_or(q, 0x000F000Fu, 0x64006400u);
    else if constexpr (T == 1) return and_or(q, 0x00F000F0u, 0x54005400u);
    else if constexpr (T == 2) return and_or(q8, 0x000F000Fu, 0x64006400u);
    else return and_or(q8, 0x00F000F0u, 0x54005400u);
}

// NG: quantisation groups per 128-K iteration (1: g % 128 == 0, 2: g == 64, 4: g == 32).
// U: iterations (16-byte loads per lane) a wave keeps in flight.
template <int NWAVES, int U, int NG>
__global__ __launch_bounds__(NWAVES * 64) void awq_gemv_nk_kernel(GemvNkParams p) {
    constexpr int NTHR = NWAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: xs[(M+1)][xpitch] fp16 (slot-permuted, row M = zeros) | zq[16][ZW] u32 | zsc[16][8*ZW] fp16
    //      | red[NWAVES][16 m][16 n] fp32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kb = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int M = p.M, K = p.K;
    const int SW = 8 * p.ZW;

    NK_STAMP(0);
    half_t* xs = reinterpret_cast<half_t*>(smem);
    uint32_t* zq = reinterpret_cast<uint32_t*>(smem + (size_t)(M + 1) * p.xpitch * 2);
    half_t* zsc = reinterpret_cast<half_t*>(zq + 16 * p.ZW);
    float* red = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(zsc) + (size_t)16 * SW * 2);

    // ---- stage activations (slot-permuted), zeros and scales of the block's 16 rows
    {
        const int xc = K >> 3;  // 16-byte chunks per activation row
        for (int c = tid; c < (M + 1) * (xc + 1); c += NTHR) {
            const int m = (int)__umulhi((uint32_t)c, p.xc1_magic), cc = c - m * (xc + 1);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < M && cc < xc) {
                const u32x4 d = *reinterpret_cast<const u32x4*>(p.x + (int64_t)m * K + 8 * cc);
                v[0] = __builtin_amdgcn_perm(d[2], d[0], 0x05040100u);  // (x0, x4)
                v[1] = __builtin_amdgcn_perm(d[2], d[0], 0x07060302u);  // (x1, x5)
                v[2] = __builtin_amdgcn_perm(d[3], d[1], 0x05040100u);  // (x2, x6)
                v[3] = __builtin_amdgcn_perm(d[3], d[1], 0x07060302u);  // (x3, x7)
            }
            *reinterpret_cast<u32x4*>(xs + (size_t)m * p.xpitch + 8 * cc) = v;
        }
        for (int c = tid; c < 16 * p.ZW; c += NTHR) {
            const int r = p.zw_magic ? (int)__umulhi((uint32_t)c, p.zw_magic) : c, w = c - r * p.ZW;
            zq[c] = (n0 + r < p.N) ? p.qzeros[(int64_t)(n0 + r) * p.ZW + w] : 0u;
        }
        for (int c = tid; c < 16 * p.ZW; c += NTHR) {  // 8 scales = 16 bytes per chunk
            const int r = p.zw_magic ? (int)__umulhi((uint32_t)c, p.zw_magic) : c, w = c - r * p.ZW;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n0 + r < p.N) v = *reinterpret_cast<const u32x4*>(p.scales + (int64_t)(n0 + r) * SW + 8 * w);
            *reinterpret_cast<u32x4*>(zsc + r * SW + 8 * w) = v;
        }
    }

    NK_STAMP(1);
    const int row = n0 + j;
    const bool active = row < p.N;
    const uint32_t* wrow = p.qweight + (int64_t)(active ? row : 0) * p.KW + 4 * kb;
    const half_t* xlane = xs + (size_t)min(j, M) * p.xpitch + 32 * kb;  // A row = batch row j (zeros past M)
    const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    const int kiter = K >> 7;  // 128 K per iteration (K % 128 == 0)

    float yv[4] = {0.f, 0.f, 0.f, 0.f};
    bool staged = false;
    for (int it0 = wave; it0 < kiter; it0 += NWAVES * U) {
        // ---- request U iterations: lane (j, kb) reads words 16*it + 4*kb .. +3 of its row
        u32x4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * NWAVES;
            q[u] = u32x4{0u, 0u, 0u, 0u};
            if (active && it < kiter) q[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + 16 * it));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!staged) {
            NK_STAMP(2);
            __syncthreads();
            staged = true;
            NK_STAMP(3);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * NWAVES;
            if (it >= kiter) break;
            // B fragment of the zero points in the same slots and biases: (bias_T + z) pairs, so
            // acc - accz = sum_k x * (w - z) with one extra MFMA and no sum-of-x term
            float4_t acc[NG], accz[NG];
            u32x4v bz[NG];
            float scl[NG];
#pragma unroll
            for (int h = 0; h < NG; ++h) {
                acc[h] = accz[h] = float4_t{0.f, 0.f, 0.f, 0.f};
                const int grp = (int)__umulhi((uint32_t)(128 * it + (128 / NG) * h), p.g_magic);
                scl[h] = (float)zsc[j * SW + grp];
                const uint32_t z = (zq[j * p.ZW + (grp >> 3)] >> (4 * (grp & 7))) & 15u;
                const uint32_t zz = z | (z << 16);
                const uint32_t be = 0x64006400u | zz, bo = 0x54005400u | (zz << 4);
                bz[h] = u32x4v{be, bo, be, bo};
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4v a = *reinterpret_cast<const u32x4v*>(xlane + 128 * it + 8 * s);
                const uint32_t w = q[u][s], w8 = w >> 8;
                const u32x4v b = {pairbo<0>(w, w8), pairbo<1>(w, w8), pairbo<2>(w, w8), pairbo<3>(w, w8)};
                if constexpr (NG == 1) {
                    acc[0] = mfma16(a, b, acc[0]);
                    accz[0] = mfma16(a, bz[0], accz[0]);
                } else {
#pragma unroll
                    for (int h = 0; h < NG; ++h) {  // only the K lanes of group h contribute
                        const bool mine = (kb / (4 / NG)) == h;
                        const u32x4v am = mine ? a : u32x4v{0u, 0u, 0u, 0u};
                        acc[h] = mfma16(am, b, acc[h]);
                        accz[h] = mfma16(am, bz[h], accz[h]);
                    }
                }
            }
            // fold: y[m][n] += s[n,g] * sum_k x (w - z); D reg r = batch row 4*kb + r
#pragma unroll
            for (int h = 0; h < NG; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[r] = __builtin_fmaf(scl[h], acc[h][r] - accz[h][r], yv[r]);
        }
    }
    if (!staged) __syncthreads();
    NK_STAMP(4);

    // ---- fold the waves: red[wave][m][n]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * kb + r) * 16 + j] = yv[r];
    __syncthreads();
    for (int e = tid; e < M * 16; e += NTHR) {
        const int m = e >> 4, n = e & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) s += red[(w * 16 + m) * 16 + n];
        if (n0 + n < p.N) p.y[(int64_t)m * p.N + n0 + n] = (half_t)s;
    }
    NK_STAMP(5);
}

// GEMV layout -> fp16 W^T [N, K] (row n = dequantised weights of output n), bit-identical to the
// GEMM-layout dequant of the same (w, z, s) transposed.
__global__ __launch_bounds__(256) void awq_dequant_nk_kernel(const uint32_t* __restrict__ qweight,
                                                             const uint32_t* __restrict__ qzeros,
                                                             const half_t* __restrict__ scales,
                                                             half_t* __restrict__ out, int N, int KW, int ZW, int g) {
    const int64_t total = (int64_t)N * KW;
    for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = (int)(idx / KW), c = (int)(idx % KW);
        const int grp = (8 * c) / g;  // g % 8 == 0: one group per word
        const uint32_t q = qweight[idx];
        const uint32_t z = (qzeros[(int64_t)n * ZW + (grp >> 3)] >> (4 * (grp & 7))) & 15u;
        const half_t s = scales[(int64_t)n * 8 * ZW + grp];
        const half2_t zm = u2h2(0x64006400u | z | (z << 16)), s2 = {s, s};
        u32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // ordinal nibbles (2t, 2t+1)
            const uint32_t pr = ((q >> (8 * t)) & 0xFu) | (((q >> (8 * t + 4)) & 0xFu) << 16) | 0x64006400u;
            o[t] = h22u((u2h2(pr) - zm) * s2);
        }
        *reinterpret_cast<u32x4*>(out + idx * 8) = o;
    }
}

template <int NWAVES, int U>
void launch_nk(const GemvNkParams& p, size_t lds, hipStream_t st) {
    dim3 grid((unsigned)((p.N + 15) / 16));
#define AWQ_NK_LAUNCH(NGV)                                                                              \
    {                                                                                                   \
        static std::atomic<unsigned long long> opted{0};                                                \
        (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_nk_kernel<NWAVES, U, NGV>), opted); \
        hipLaunchKernelGGL((awq_gemv_nk_kernel<NWAVES, U, NGV>), grid, dim3(NWAVES * 64), lds, st, p);   \
    }
    if (p.g % 128 == 0) AWQ_NK_LAUNCH(1)
    else if (p.g == 64) AWQ_NK_LAUNCH(2)
    else AWQ_NK_LAUNCH(4)
#undef AWQ_NK_LAUNCH
}

}  // namespace

#ifdef AWQ_GEMV_TRACE
extern "C" __attribute__((visibility("default"))) void awq_debug_set_trace_nk(void* dev_buf) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_awq_trace_nk), &dev_buf, sizeof(void*));
}
#endif

size_t awq_gemv_nk_lds_bytes(int M, int K, int ZW, int nwaves) {
    return (size_t)(M + 1) * (K + 8) * 2 + (size_t)16 * ZW * 4 + (size_t)16 * 8 * ZW * 2 + (size_t)nwaves * 16 * 16 * 4;
}

bool awq_gemv_nk_supports(int M, int K, int N, int g) {
    if (M < 1 || M > 16 || K % 128 || N < 1) return false;
    if (!(g % 128 == 0 || g == 64 || g == 32)) return false;
    if (K % g) return false;
    return true;
}

int awq_launch_gemv_nk(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                       uint16_t* y, int M, int K, int N, int g, int ZW, int nwaves, int unroll, hipStream_t st) {
    if (!awq_gemv_nk_supports(M, K, N, g)) return AWQ_ERR_UNSUPPORTED;
    const int iters = (K + 127) / 128;
    if (nwaves == 0) nwaves = 8;
    if (unroll == 0) unroll = (iters + nwaves - 1) / nwaves > 4 ? 8 : 4;
    if (awq_gemv_nk_lds_bytes(M, K, ZW, nwaves) > 160 * 1024) return AWQ_ERR_UNSUPPORTED;  // caller chunks M
    GemvNkParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N; p.g = g;
    p.KW = K / 8; p.ZW = ZW;
    p.xpitch = K + 8;
    p.zw_magic = 0;
    if (!awq_magic_u32((uint32_t)g, (uint32_t)K + 128u, &p.g_magic) || !awq_magic_u32((uint32_t)(K / 8 + 1), 17u * (uint32_t)(K / 8 + 1) + 1u, &p.xc1_magic) ||
        (ZW > 1 && !awq_magic_u32((uint32_t)ZW, 16u * (uint32_t)ZW + 1u, &p.zw_magic)))
        return AWQ_ERR_UNSUPPORTED;
    const size_t lds = awq_gemv_nk_lds_bytes(M, K, ZW, nwaves);
    if (nwaves == 4 && unroll == 4) launch_nk<4, 4>(p, lds, st);
    else if (nwaves == 4 && unroll == 8) launch_nk<4, 8>(p, lds, st);
    else if (nwaves == 8 && unroll == 4) launch_nk<8, 4>(p, lds, st);
    else if (nwaves == 8 && unroll == 8) launch_nk<8, 8>(p, lds, st);
    else if (nwaves == 16 && unroll == 4) launch_nk<16, 4>(p, lds, st);
    else return AWQ_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_launch_dequant_nk(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out, int K,
                          int N, int g, int ZW, hipStream_t st) {
    if (K % 8 || g % 8 || N < 0) return AWQ_ERR_BAD_SHAPE;
    const int64_t total = (int64_t)N * (K / 8);
    if (total == 0) return AWQ_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(awq_dequant_nk_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const uint32_t*>(qweight), reinterpret_cast<const uint32_t*>(qzeros),
                       reinterpret_cast<const half_t*>(scales), reinterpret_cast<half_t*>(out), N, K / 8, ZW, g);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
