// dequant.hip -- GEMM-layout int4 -> fp16 weight materialisation and the integer unpack.
//
// Roofline: HBM.  Algorithmic bytes per call = K*N/2 (qweight) + (K/g)*(N/8)*4 (qzeros)
// + (K/g)*N*2 (scales) + K*N*2 (fp16 out): write dominated (4x the packed read).
// Replaces awq_ext.dequantize_weights_cuda (awq/modules/linear/gemm.py:51-53,100-102).
#include "awq_device.h"
#include "awq_internal.h"

// One thread owns one packed word column (8 logical columns) for ROWS consecutive rows of one
// group: zeros/scales are fetched once, every store is a fully coalesced 16 B/lane.
template <int ROWS>
__global__ __launch_bounds__(256) void awq_dequant_kernel(const uint32_t* __restrict__ qweight,
                                                          const uint32_t* __restrict__ qzeros,
                                                          const half_t* __restrict__ scales,
                                                          half_t* __restrict__ out, int K, int NW, int g) {
    const int64_t total = (int64_t)(K / ROWS) * NW;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % NW);
        const int k0 = (int)(idx / NW) * ROWS;
        const int grp = k0 / g;
        const uint32_t qz = qzeros[(int64_t)grp * NW + c];
        const u32x4 sv = *reinterpret_cast<const u32x4*>(scales + ((int64_t)grp * NW + c) * 8);
        const half2_t z0 = u2h2(awq_pair_magic<0>(qz)), z1 = u2h2(awq_pair_magic<1>(qz));
        const half2_t z2 = u2h2(awq_pair_magic<2>(qz)), z3 = u2h2(awq_pair_magic<3>(qz));
        uint32_t q[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) q[r] = __builtin_nontemporal_load(qweight + (int64_t)(k0 + r) * NW + c);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            u32x4 o;
            o[0] = h22u(awq_dq_pair<0>(q[r], z0, u2h2(sv[0])));
            o[1] = h22u(awq_dq_pair<1>(q[r], z1, u2h2(sv[1])));
            o[2] = h22u(awq_dq_pair<2>(q[r], z2, u2h2(sv[2])));
            o[3] = h22u(awq_dq_pair<3>(q[r], z3, u2h2(sv[3])));
            *reinterpret_cast<u32x4*>(out + ((int64_t)(k0 + r) * NW + c) * 8) = o;
        }
    }
}

// out[r, 8c + j] = logical column j of word q[r, c], as a byte 0..15
__global__ __launch_bounds__(256) void awq_unpack_kernel(const uint32_t* __restrict__ q, uint8_t* __restrict__ out,
                                                         int64_t total) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w = q[idx];
        // columns (0,1),(2,3),(4,5),(6,7) = nibbles (0,4),(1,5),(2,6),(3,7)
        const uint32_t lo = (w & 0xFu) | ((w >> 8) & 0xF00u) | ((w << 12) & 0xF0000u) | ((w << 4) & 0xF000000u);
        const uint32_t hi = ((w >> 8) & 0xFu) | ((w >> 16) & 0xF00u) | ((w << 4) & 0xF0000u) | ((w >> 4) & 0xF000000u);
        u32x2 o = {lo, hi};
        *reinterpret_cast<u32x2*>(out + idx * 8) = o;
    }
}

static int grid_for(int64_t total_threads) {
    int64_t blocks = (total_threads + 255) / 256;
    const int64_t cap = 256 * 8 * 2;  // 256 CUs x 8 blocks, x2 for tail balance; grid-stride the rest
    return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

int awq_launch_dequant(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out,
                       int64_t K, int64_t N, int64_t g, hipStream_t stream) {
    const int NW = (int)(N / 8);
    if (K == 0 || N == 0) return AWQ_OK;
    auto qw = reinterpret_cast<const uint32_t*>(qweight);
    auto qz = reinterpret_cast<const uint32_t*>(qzeros);
    auto sc = reinterpret_cast<const half_t*>(scales);
    auto o = reinterpret_cast<half_t*>(out);
    if (g % 8 == 0) {
        hipLaunchKernelGGL(awq_dequant_kernel<8>, dim3(grid_for(K / 8 * NW)), dim3(256), 0, stream, qw, qz, sc, o,
                           (int)K, NW, (int)g);
    } else {
        hipLaunchKernelGGL(awq_dequant_kernel<1>, dim3(grid_for(K * NW)), dim3(256), 0, stream, qw, qz, sc, o, (int)K,
                           NW, (int)g);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_launch_unpack(const int32_t* q, uint8_t* out, int64_t rows, int64_t words, hipStream_t stream) {
    const int64_t total = rows * words;
    if (total == 0) return AWQ_OK;
    hipLaunchKernelGGL(awq_unpack_kernel, dim3(grid_for(total)), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t*>(q), out, total);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
