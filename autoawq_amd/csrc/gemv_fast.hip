// gemv_fast.hip -- decode GEMV / skinny GEMM (1 <= M <= 16) on the GEMVFast layout, gfx950.
//
// Replaces awq_v2_ext.gemv_forward_cuda_decode(x, qweight, scales, qzeros, m, n, k, group_size) and
// (for small M) awq_v2_ext.gemm_forward_cuda_prefill as called by awq/modules/linear/gemv_fast.py:
// 185-208.  Layout (packer gemv_fast.py:26-65,142-181):
//   qweight [N/4, K] int16: element [r, 64*b + 16*i + 8*h + t], nibble j = w[4r+i, 64b + 32h + 8j + t];
//   scales  [8*ZW, N] fp16 (group major);  qzeros [8*ZW, N] fp16 = -(s*z).
// Dequant there is W = w*s + qzeros.
//
// Roofline: HBM; algorithmic bytes K*N/2 + 2*(K/g)*N*2 + M*K*2 + M*N*2.
//
// Same structure as gemv_nk.hip: 4 output rows x 64 K are one contiguous 128-byte run, a block
// owns 16 whole output rows (no split-K exchange), a 32-bit word of the stream is a lane's MFMA B
// fragment: (q >> 4t) & 0x000F000F pairs nibble t of its two int16 = weights (8t + 2d, 8t + 2d + 1)
// of a 32-K half block (d = dword index).  The activations are staged in LDS permuted to exactly
// that slot order (chunk d of a 32-K block = x dwords d, 4+d, 8+d, 12+d), so the K loop spends no
// VALU on them.  Fold per 128 K:  y += s*(acc - 16*sx) + qzeros*sx, with sx = sum of x (ones-MFMA).
#include "awq_device.h"
#include "awq_internal.h"

namespace {

struct GemvFastParams {
    const uint32_t* qweight;  // int16 [N/4, K] viewed as dwords
    const half_t* qzeros;     // [GP, N] fp16 = -(s*z)
    const half_t* scales;     // [GP, N] fp16
    const half_t* x;
    half_t* y;
    int M, K, N, g;
    int GP;       // padded group rows of scales / qzeros (8*ZW)
    uint32_t g_magic, xb_magic;  // (v * magic) >> 32 == v / {g, K/32} for the values divided (awq_magic_u32)
    int xpitch;   // halfs per staged activation row in LDS (K + 8)
};

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// Phase timestamps for tools/trace_gemv.py (debug build only: -DAWQ_GEMV_TRACE)
#ifdef AWQ_GEMV_TRACE
__device__ unsigned long long* g_awq_trace_fast = nullptr;
#define FAST_STAMP(slot)                                                                                \
    do {                                                                                              \
        if (g_awq_trace_fast && lane == 0)                                                              \
            g_awq_trace_fast[((size_t)blockIdx.x * NWAVES + wave) * 16 + (slot)] = wall_clock64();      \
    } while (0)
#else
#define FAST_STAMP(slot) do { } while (0)
#endif

AWQ_DEV float4_t mfma16(u32x4v a, u32x4v b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0,
                                                  0, 0);
}

// fp16 pair (16 + nibble t, 16 + nibble t+4) of a packed word
template <int T>
AWQ_DEV uint32_t pair16o(uint32_t q) {
    constexpr int SH = 6 - 4 * T;
    const uint32_t v = SH >= 0 ? (q << (SH >= 0 ? SH : 0)) : (q >> (SH < 0 ? -SH : 0));
    return and_or(v, 0x03C003C0u, 0x4C004C00u);
}

// NG: quantisation groups per 128-K iteration (1: g % 128 == 0, 2: g == 64, 4: g == 32).
// U: iterations (16-byte loads per lane) a wave keeps in flight.
template <int NWAVES, int U, int NG>
__global__ __launch_bounds__(NWAVES * 64) void awq_gemv_fast_kernel(GemvFastParams p) {
    constexpr int NTHR = NWAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: xs[(M+1)][xpitch] fp16 (slot-permuted, row M = zeros) | zsc[G][16] fp16 | zqz[G][16] fp16
    //      | red[NWAVES][16 m][16 n] fp32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kb = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int M = p.M, K = p.K;
    const int G = (int)__umulhi((uint32_t)K, p.g_magic);

    FAST_STAMP(0);
    half_t* xs = reinterpret_cast<half_t*>(smem);
    half_t* zsc = reinterpret_cast<half_t*>(smem + (size_t)(M + 1) * p.xpitch * 2);
    half_t* zqz = zsc + (size_t)G * 16;
    float* red = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(zqz) + (size_t)G * 16 * 2);

    // ---- stage activations (slot-permuted per 32-K block), scales and pre-multiplied zeros
    {
        const int xb = K >> 5;  // 32-K blocks per activation row
        for (int c = tid; c < (M + 1) * xb; c += NTHR) {
            const int m = (int)__umulhi((uint32_t)c, p.xb_magic), b = c - m * xb;
            u32x4 in[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                in[e] = u32x4{0u, 0u, 0u, 0u};
                if (m < M) in[e] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)m * K + 32 * b + 8 * e);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {  // chunk d = x dwords d, 4+d, 8+d, 12+d of the block
                const u32x4 o = {in[0][d], in[1][d], in[2][d], in[3][d]};
                *reinterpret_cast<u32x4*>(xs + (size_t)m * p.xpitch + 32 * b + 8 * d) = o;
            }
        }
        for (int c = tid; c < (M + 1); c += NTHR)  // zero pad chunk at the end of every row
            *reinterpret_cast<u32x4*>(xs + (size_t)c * p.xpitch + K) = u32x4{0u, 0u, 0u, 0u};
        for (int c = tid; c < G * 2; c += NTHR) {  // 16 columns = two 16-byte chunks per group row
            const int gr = c >> 1, hf = c & 1;
            u32x4 vs = {0u, 0u, 0u, 0u}, vz = vs;
            if (n0 + 8 * hf < p.N) {
                vs = *reinterpret_cast<const u32x4*>(p.scales + (int64_t)gr * p.N + n0 + 8 * hf);
                vz = *reinterpret_cast<const u32x4*>(p.qzeros + (int64_t)gr * p.N + n0 + 8 * hf);
            }
            *reinterpret_cast<u32x4*>(zsc + gr * 16 + 8 * hf) = vs;
            *reinterpret_cast<u32x4*>(zqz + gr * 16 + 8 * hf) = vz;
        }
    }

    FAST_STAMP(1);
    const int row = n0 + j;
    const bool active = row < p.N;
    // lane (j = 4*rr + i, kb): 16 bytes = int16 [16*i + 8*(kb&1) .. +8) of 64-K block 2*it + (kb>>1)
    // of row group (n0/4 + rr); in dwords: rg*K/2 + 32*(2*it + (kb>>1)) + 8*i + 4*(kb&1)
    const uint32_t* wrow = p.qweight + (int64_t)((active ? row : 0) >> 2) * (K >> 1) + 32 * (kb >> 1) + 8 * (j & 3) +
                           4 * (kb & 1);
    const half_t* xlane = xs + (size_t)min(j, M) * p.xpitch + 32 * kb;  // A row = batch row j (zeros past M)
    const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    const int kiter = K >> 7;  // 128 K per iteration (K % 128 == 0)

    float yv[4] = {0.f, 0.f, 0.f, 0.f};
    bool staged = false;
    for (int it0 = wave; it0 < kiter; it0 += NWAVES * U) {
        // ---- request U iterations (128 K each): one 16-byte load per lane and iteration
        u32x4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * NWAVES;
            q[u] = u32x4{0u, 0u, 0u, 0u};
            if (active && it < kiter) q[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + 64 * it));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!staged) {
            FAST_STAMP(2);
            __syncthreads();
            staged = true;
            FAST_STAMP(3);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * NWAVES;
            if (it >= kiter) break;
            float4_t acc[NG], sx[NG];
#pragma unroll
            for (int h = 0; h < NG; ++h) acc[h] = sx[h] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4v a = *reinterpret_cast<const u32x4v*>(xlane + 128 * it + 8 * s);
                const uint32_t w = q[u][s];
                const u32x4v b = {pair16o<0>(w), pair16o<1>(w), pair16o<2>(w), pair16o<3>(w)};
                if constexpr (NG == 1) {
                    acc[0] = mfma16(a, b, acc[0]);
                    sx[0] = mfma16(a, ones, sx[0]);
                } else {
#pragma unroll
                    for (int h = 0; h < NG; ++h) {  // only the K lanes of group h contribute
                        const bool mine = (kb / (4 / NG)) == h;
                        const u32x4v am = mine ? a : u32x4v{0u, 0u, 0u, 0u};
                        acc[h] = mfma16(am, b, acc[h]);
                        sx[h] = mfma16(am, ones, sx[h]);
                    }
                }
            }
            // fold: y[m][n] += s*(acc - 16*sx) + qzeros*sx; D reg r = batch row 4*kb + r
#pragma unroll
            for (int h = 0; h < NG; ++h) {
                const int grp = (int)__umulhi((uint32_t)(128 * it + (128 / NG) * h), p.g_magic);
                const float sc = (float)zsc[grp * 16 + j];
                const float zc = __builtin_fmaf(-16.f, sc, (float)zqz[grp * 16 + j]);
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[r] = __builtin_fmaf(sc, acc[h][r], __builtin_fmaf(zc, sx[h][r], yv[r]));
            }
        }
    }
    if (!staged) __syncthreads();
    FAST_STAMP(4);

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
    FAST_STAMP(5);
}

// GEMVFast layout -> fp16 W^T [N, K]:  W = fp16(w*s + qzeros), computed as one fp32 fma rounded once
// more to fp16 (w*s + qzeros is exact in double and its RN to fp32 equals the fp32 fma).
__global__ __launch_bounds__(256) void awq_dequant_fast_kernel(const uint32_t* __restrict__ qweight,
                                                               const half_t* __restrict__ qzeros,
                                                               const half_t* __restrict__ scales,
                                                               half_t* __restrict__ out, int N, int K, int g) {
    // one thread = one dword of the stream: row group rr, 64-K block b, row i, half h, dword d
    const int64_t total = (int64_t)(N / 4) * (K / 2);
    for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int rr = (int)(idx / (K / 2)), w = (int)(idx % (K / 2));
        const int b = w >> 5, i = (w >> 3) & 3, h = (w >> 2) & 1, d = w & 3;
        const int n = 4 * rr + i;
        const uint32_t q = qweight[idx];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 64 * b + 32 * h + 8 * t + 2 * d + e;
                const float wv = (float)((q >> (4 * t + 16 * e)) & 15u);
                const int grp = k / g;
                const float v = __builtin_fmaf(wv, (float)scales[(int64_t)grp * N + n], (float)qzeros[(int64_t)grp * N + n]);
                out[(int64_t)n * K + k] = (half_t)v;
            }
    }
}

template <int NWAVES, int U>
void launch_fast(const GemvFastParams& p, size_t lds, hipStream_t st) {
    dim3 grid((unsigned)((p.N + 15) / 16));
#define AWQ_FAST_LAUNCH(NGV)                                                                              \
    {                                                                                                     \
        static std::atomic<unsigned long long> opted{0};                                                  \
        (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_fast_kernel<NWAVES, U, NGV>), opted); \
        hipLaunchKernelGGL((awq_gemv_fast_kernel<NWAVES, U, NGV>), grid, dim3(NWAVES * 64), lds, st, p);   \
    }
    if (p.g % 128 == 0) AWQ_FAST_LAUNCH(1)
    else if (p.g == 64) AWQ_FAST_LAUNCH(2)
    else AWQ_FAST_LAUNCH(4)
#undef AWQ_FAST_LAUNCH
}

}  // namespace

#ifdef AWQ_GEMV_TRACE
extern "C" __attribute__((visibility("default"))) void awq_debug_set_trace_fast(void* dev_buf) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_awq_trace_fast), &dev_buf, sizeof(void*));
}
#endif

size_t awq_gemv_fast_lds_bytes(int M, int K, int g, int nwaves) {
    return (size_t)(M + 1) * (K + 8) * 2 + (size_t)2 * (K / g) * 16 * 2 + (size_t)nwaves * 16 * 16 * 4;
}

bool awq_gemv_fast_supports(int M, int K, int N, int g) {
    if (M < 1 || M > 16 || K % 128 || N < 16 || N % 16) return false;
    if (!(g % 128 == 0 || g == 64 || g == 32)) return false;
    if (K % g) return false;
    return true;
}

int awq_launch_gemv_fast(const uint16_t* x, const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros,
                         uint16_t* y, int M, int K, int N, int g, int GP, int nwaves, int unroll, hipStream_t st) {
    if (!awq_gemv_fast_supports(M, K, N, g)) return AWQ_ERR_UNSUPPORTED;
    if (GP < K / g) return AWQ_ERR_BAD_SHAPE;
    const int iters = K / 128;
    if (nwaves == 0) nwaves = 8;
    if (unroll == 0) unroll = (iters + nwaves - 1) / nwaves > 4 ? 8 : 4;
    const size_t lds = awq_gemv_fast_lds_bytes(M, K, g, nwaves);
    if (lds > 160 * 1024) return AWQ_ERR_UNSUPPORTED;  // caller chunks M
    GemvFastParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const half_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N; p.g = g;
    p.GP = GP;
    p.xpitch = K + 8;
    if (!awq_magic_u32((uint32_t)g, (uint32_t)K + 128u, &p.g_magic) || !awq_magic_u32((uint32_t)(K / 32), 17u * (uint32_t)(K / 32) + 1u, &p.xb_magic))
        return AWQ_ERR_UNSUPPORTED;
    if (nwaves == 4 && unroll == 4) launch_fast<4, 4>(p, lds, st);
    else if (nwaves == 4 && unroll == 8) launch_fast<4, 8>(p, lds, st);
    else if (nwaves == 8 && unroll == 4) launch_fast<8, 4>(p, lds, st);
    else if (nwaves == 8 && unroll == 8) launch_fast<8, 8>(p, lds, st);
    else if (nwaves == 16 && unroll == 4) launch_fast<16, 4>(p, lds, st);
    else return AWQ_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_launch_dequant_fast(const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros, uint16_t* out, int K,
                            int N, int g, hipStream_t st) {
    if (K % 64 || N % 4 || g <= 0 || K % g) return AWQ_ERR_BAD_SHAPE;
    const int64_t total = (int64_t)(N / 4) * (K / 2);
    if (total == 0) return AWQ_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(awq_dequant_fast_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const uint32_t*>(qweight), reinterpret_cast<const half_t*>(qzeros),
                       reinterpret_cast<const half_t*>(scales), reinterpret_cast<half_t*>(out), N, K, g);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
