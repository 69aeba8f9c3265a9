// gemv_valu.hip -- batch-1..4 decode GEMV on the GEMM layout (qweight [K, N/8]), wave64 VALU.
//
// Replaces awq_ext.gemm_forward_cuda for M <= 4 (awq/modules/linear/gemm.py:56-58).
//
// Roofline: HBM.  Algorithmic bytes per call:
//     K*N/2 + (K/g)*(N/8)*4 + (K/g)*N*2 + M*K*2 + M*N*2 (+ N*2 with bias).
//
// Work decomposition (GEMM layout is N-contiguous, so lanes run along N):
//   * a lane owns 32 adjacent output columns = one 16-byte load per weight row;
//   * a wave is NL column-lanes x (64/NL) K-lanes; a block is 4 waves stacked along K;
//   * each K-lane walks 8-row chunks (8 independent 16 B loads in flight, one group per chunk:
//     g % 8 == 0), prefetching the next chunk's weights before it computes the current one;
//   * weights are dequantised to the exact fp16 the reference materialises ((w - z) * s, see
//     awq_device.h) and accumulated in fp32 with mixed-precision FMAs;
//   * block partials are folded through LDS; K is split over gridDim.y, slabs are plain fp32
//     stores and a second kernel sums them in FIXED slab order (deterministic).
// This is the reference-order-numerics variant (weights rounded to fp16 exactly as
// dequantize_gemm does); the default decode kernel is gemv_mfma.hip.
#include "awq_device.h"
#include "awq_internal.h"

namespace {

struct GemvParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    const half_t* bias;
    half_t* y;
    float* partial;
    int K, N, g, rows_per_block;
};

template <int M, int NLOG, bool NT>
__global__ __launch_bounds__(256) void awq_gemv_valu_kernel(GemvParams p) {
    constexpr int NL = 1 << NLOG;    // column-lanes per wave
    constexpr int KLW = 64 / NL;     // K-lanes per wave
    constexpr int KLB = KLW * 4;     // K-lanes per block
    constexpr int CT = NL * 32;      // columns per block
    constexpr int STEP = KLB * 8;    // rows covered by one block pass
    __shared__ float red[KLB * CT];  // 32 KiB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & (NL - 1);
    const int kl = (lane >> NLOG) + wave * KLW;
    const int c32 = blockIdx.x * NL + nl;  // 32-column chunk index
    const int NW = p.N >> 3;
    const bool active = c32 * 32 < p.N;
    const int kbeg = blockIdx.y * p.rows_per_block;
    const int kend = min(p.K, kbeg + p.rows_per_block);

    float acc[M][32];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[m][c] = 0.f;

    // Buffer loads through wave-uniform descriptors: one 32-bit lane offset + scalar row offsets.
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    const uint32_t row_bytes = (uint32_t)NW * 4u;
    const rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qweight), 0, (uint32_t)p.K * row_bytes, 0x00020000);
    const rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.qzeros), 0, (uint32_t)(p.K / p.g) * row_bytes, 0x00020000);
    const rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.scales), 0, (uint32_t)(p.K / p.g) * (uint32_t)p.N * 2u, 0x00020000);
    const uint32_t wvoff = (uint32_t)c32 * 16u;

    struct ChunkBuf {
        u32x4 q[8];
        u32x4 qz;
        u32x4 sc[4];
        u32x4 xv[M];
    };

    // A chunk's weights, zeros, scales and activations are all requested together, one chunk
    // ahead of their use (vmcnt retires in order: a load issued at its point of use would drag
    // the whole prefetched stream into its wait).
    auto load_chunk = [&](ChunkBuf& b, int k0) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            b.q[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, wvoff + (uint32_t)(k0 + r) * row_bytes, 0, NT ? 2 : 0));
        const uint32_t grp = (uint32_t)(k0 / p.g);
        b.qz = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(zres, wvoff + grp * row_bytes, 0, 0));
#pragma unroll
        for (int wd = 0; wd < 4; ++wd)
            b.sc[wd] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sres, (uint32_t)c32 * 64u + 16u * wd + grp * (uint32_t)p.N * 2u, 0, 0));
#pragma unroll
        for (int m = 0; m < M; ++m) b.xv[m] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)m * p.K + k0);
    };

    auto compute_chunk = [&](const ChunkBuf& b) {
        half2_t zm[16], sc[16];
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            zm[wd * 4 + 0] = u2h2(awq_pair_magic<0>(b.qz[wd]));
            zm[wd * 4 + 1] = u2h2(awq_pair_magic<1>(b.qz[wd]));
            zm[wd * 4 + 2] = u2h2(awq_pair_magic<2>(b.qz[wd]));
            zm[wd * 4 + 3] = u2h2(awq_pair_magic<3>(b.qz[wd]));
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[wd * 4 + j] = u2h2(b.sc[wd][j]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                const uint32_t w = b.q[r][wd];
                half2_t d[4];
                d[0] = awq_dq_pair<0>(w, zm[wd * 4 + 0], sc[wd * 4 + 0]);
                d[1] = awq_dq_pair<1>(w, zm[wd * 4 + 1], sc[wd * 4 + 1]);
                d[2] = awq_dq_pair<2>(w, zm[wd * 4 + 2], sc[wd * 4 + 2]);
                d[3] = awq_dq_pair<3>(w, zm[wd * 4 + 3], sc[wd * 4 + 3]);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const half2_t xp = u2h2(b.xv[m][r >> 1]);
                    const float xk = (float)xp[r & 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[m][wd * 8 + 2 * j] = __builtin_fmaf((float)d[j][0], xk, acc[m][wd * 8 + 2 * j]);
                        acc[m][wd * 8 + 2 * j + 1] = __builtin_fmaf((float)d[j][1], xk, acc[m][wd * 8 + 2 * j + 1]);
                    }
                }
            }
        }
    };

    if (active) {
        int k0 = kbeg + kl * 8;
        ChunkBuf ba, bb;
        if (k0 < kend) load_chunk(ba, k0);
        while (k0 < kend) {
            int k1 = k0 + STEP;
            if (k1 < kend) load_chunk(bb, k1);
            compute_chunk(ba);
            if (k1 >= kend) break;
            k0 = k1 + STEP;
            if (k0 < kend) load_chunk(ba, k0);
            compute_chunk(bb);
        }
    }

    // ---- fold the KLB K-lanes of the block through LDS, one activation row at a time
    const int S = gridDim.y;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if (m) __syncthreads();
        float* row = red + kl * CT + nl * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // rotate the float4 slot by nl: conflict-free ds_write_b128
            float4_t v = {acc[m][4 * i], acc[m][4 * i + 1], acc[m][4 * i + 2], acc[m][4 * i + 3]};
            *reinterpret_cast<float4_t*>(row + 4 * ((i + nl) & 7)) = v;
        }
        __syncthreads();
        for (int c = tid; c < CT; c += 256) {
            const int cn = c >> 5, ci = (c & 31) >> 2, ce = c & 3;
            const int off = cn * 32 + 4 * ((ci + cn) & 7) + ce;
            float s = 0.f;
#pragma unroll 8
            for (int k = 0; k < KLB; ++k) s += red[k * CT + off];
            const int col = blockIdx.x * CT + c;
            if (col >= p.N) continue;
            if (S > 1) {  // plain fp32 slabs [S][M][N]; awq_splitk_reduce_kernel sums them in slab order
                p.partial[((int64_t)blockIdx.y * M + m) * p.N + col] = s;
                continue;
            }
            if (p.bias) s += (float)p.bias[col];
            p.y[(int64_t)m * p.N + col] = (half_t)s;
        }
    }
}

// y[m, n] = fp16( sum_s partial[s, m, n] + bias[n] )
__global__ __launch_bounds__(256) void awq_splitk_reduce_kernel(const float* __restrict__ partial,
                                                                const half_t* __restrict__ bias,
                                                                half_t* __restrict__ y, int MN, int N, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float s = 0.f;
    for (int sp0 = 0; sp0 < S; sp0 += 8) {  // 8 independent loads in flight, summed in slab order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(sp0 + u < S ? sp0 + u : S - 1) * MN + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (sp0 + u < S) s += v[u];
    }
    if (bias) s += (float)bias[i % N];
    y[i] = (half_t)s;
}

// Checker / odd-shape fallback: one thread per output element, k ascending, same exact-W dequant.
__global__ __launch_bounds__(256) void awq_gemm_naive_kernel(const uint32_t* __restrict__ qweight,
                                                             const uint32_t* __restrict__ qzeros,
                                                             const half_t* __restrict__ scales,
                                                             const half_t* __restrict__ x,
                                                             const half_t* __restrict__ bias, half_t* __restrict__ y,
                                                             int M, int K, int N, int g) {
    const int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int n = (int)(idx % N), m = (int)(idx / N);
    const int NW = N >> 3, c = n >> 3, j = n & 7;
    const int sh = 4 * ((j >> 1) + 4 * (j & 1));  // nibble REV[j]
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int grp = k / g;
        const int wi = (qweight[(int64_t)k * NW + c] >> sh) & 15;
        const int zi = (qzeros[(int64_t)grp * NW + c] >> sh) & 15;
        const half_t w = (half_t)(float)(wi - zi) * scales[(int64_t)grp * N + n];
        acc = __builtin_fmaf((float)w, (float)x[(int64_t)m * K + k], acc);
    }
    if (bias) acc += (float)bias[n];
    y[idx] = (half_t)acc;
}

template <int M, int NLOG, bool NT>
void launch_valu(const GemvParams& p, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL((awq_gemv_valu_kernel<M, NLOG, NT>), grid, dim3(256), 0, stream, p);
}

template <int M>
void dispatch_valu(const GemvParams& p, dim3 grid, int nlog, bool nt, hipStream_t stream) {
    if (nt) {
        if (nlog == 2) launch_valu<M, 2, true>(p, grid, stream);
        else if (nlog == 3) launch_valu<M, 3, true>(p, grid, stream);
        else launch_valu<M, 4, true>(p, grid, stream);
    } else {
        if (nlog == 2) launch_valu<M, 2, false>(p, grid, stream);
        else if (nlog == 3) launch_valu<M, 3, false>(p, grid, stream);
        else launch_valu<M, 4, false>(p, grid, stream);
    }
}

}  // namespace

int awq_gemv_valu_default_split(int K, int N, int nlog) {
    const int CT = 32 << nlog;
    const int step = (64 >> nlog) * 4 * 8;  // rows per block pass
    const int tiles = (N + CT - 1) / CT;
    const int passes = (K + step - 1) / step;
    // aim for ~2 blocks per CU (512 blocks), at least one pass per block
    int s = (512 + tiles - 1) / tiles;
    if (s > passes) s = passes;
    if (s < 1) s = 1;
    return s;
}

int awq_launch_gemv_valu(const AwqGemmArgs& a, int nlog, int splitk, bool nt) {
    if (a.M < 1 || a.M > 4 || a.N % 32 || a.g % 8 || a.K % 8) return AWQ_ERR_UNSUPPORTED;
    if (nlog < 2 || nlog > 4) return AWQ_ERR_UNSUPPORTED;
    const int CT = 32 << nlog;
    const int step = (64 >> nlog) * 4 * 8;
    const int tiles = (a.N + CT - 1) / CT;
    const int passes = (a.K + step - 1) / step;
    if (splitk < 1) splitk = 1;
    if (splitk > 64) splitk = 64;
    if (splitk > passes) splitk = passes;
    const int ppb = (passes + splitk - 1) / splitk;  // passes per block
    splitk = (passes + ppb - 1) / ppb;
    if (splitk > 1) {
        const size_t need = (size_t)splitk * a.M * a.N * sizeof(float);
        if (!a.partial || a.partial_floats * sizeof(float) < need) return AWQ_ERR_WORKSPACE;
    }
    GemvParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.bias = reinterpret_cast<const half_t*>(a.bias);
    p.y = reinterpret_cast<half_t*>(a.y);
    p.partial = a.partial;
    p.K = a.K; p.N = a.N; p.g = a.g;
    p.rows_per_block = ppb * step;
    dim3 grid(tiles, splitk);
    switch (a.M) {
        case 1: dispatch_valu<1>(p, grid, nlog, nt, a.stream); break;
        case 2: dispatch_valu<2>(p, grid, nlog, nt, a.stream); break;
        case 3: dispatch_valu<3>(p, grid, nlog, nt, a.stream); break;
        default: dispatch_valu<4>(p, grid, nlog, nt, a.stream); break;
    }
    if (hipGetLastError() != hipSuccess) return AWQ_ERR_LAUNCH;
    if (splitk > 1) return awq_launch_splitk_reduce(a, splitk);
    return AWQ_OK;
}

int awq_launch_splitk_reduce(const AwqGemmArgs& a, int splitk) {
    const int MN = a.M * a.N;
    hipLaunchKernelGGL(awq_splitk_reduce_kernel, dim3((MN + 255) / 256), dim3(256), 0, a.stream, a.partial,
                       reinterpret_cast<const half_t*>(a.bias), reinterpret_cast<half_t*>(a.y), MN, a.N, splitk);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_launch_gemm_naive(const AwqGemmArgs& a) {
    const int64_t total = (int64_t)a.M * a.N;
    if (total == 0) return AWQ_OK;
    hipLaunchKernelGGL(awq_gemm_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a.stream,
                       reinterpret_cast<const uint32_t*>(a.qweight), reinterpret_cast<const uint32_t*>(a.qzeros),
                       reinterpret_cast<const half_t*>(a.scales), reinterpret_cast<const half_t*>(a.x),
                       reinterpret_cast<const half_t*>(a.bias), reinterpret_cast<half_t*>(a.y), a.M, a.K, a.N, a.g);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
