// repack.hip -- GEMV-layout buffers -> GEMM-layout buffers of the SAME integers and scales, on the device, gfx950 (HBM-bound).
//
// Role: the prefill route of WQLinear_GEMV (awq/modules/linear/gemv.py:168-176 runs awq_ext.gemmv2_forward_cuda on this layout at
// every batch size).  Rounds 3-4 either kept a second, GEMM-layout copy of every matrix resident (dropped: twice the weight memory)
// or dequantised to fp16 and called the vendor's GEMM.  The fused MFMA prefill kernel (gemm_regb.hip) wants K-major words -- a lane
// reads dwords along N -- so a prefill-sized call now transposes the packed nibbles into a TEMPORARY of the call (K N / 2 bytes:
// 22.5 MB at 4096 x 11008, read once + written once) and runs the fused kernel on it: two hand-written launches, no vendor GEMM, no
// fp16 copy of the weights (90 MB at that shape), nothing resident.
//   in : qweight [N, K/8] (nibble i of word c = w[n, 8c+i]), qzeros [N, ZW] (nibble i of word c = z[n, group 8c+i]), scales [N, 8 ZW]
//   out: qweight [K, N/8] (nibble i of word c = w[k, 8c + ORDER[i]], ORDER = 0,2,4,6,1,3,5,7: awq/modules/linear/gemm.py:220-249),
//        qzeros [K/g, N/8] (same packing), scales [K/g, N]
// Bit-exact: pure nibble moves (tests/test_gpu_parity.py compares with utils/convert.py's torch repack, which tests/test_checkpoint.py
// pins against reference-written checkpoints).  Algorithmic bytes: K N / 2 read + K N / 2 written (+ the small tensors).
//
// Tile = 256 output columns (32 words: 128-byte write runs) x 256 k (32 input words: 128-byte read runs) through LDS as [kw][n].
#include "awq_device.h"
#include "awq_internal.h"

namespace {

constexpr int TN = 256, TKW = 32, PITCH = TN + 8;  // LDS row pitch in words: 264 (8 consecutive n of one thread stay 32-byte aligned); 33 KB

struct RepackParams {
    const uint32_t* qw_in;
    const uint32_t* qz_in;
    const half_t* sc_in;
    uint32_t* qw_out;
    uint32_t* qz_out;
    half_t* sc_out;
    int K, N, KW, ZW, SW, NW, G, g;
    int nt, kt;       // weight tiles along N and K
    int main_blocks;  // nt * kt; the blocks after them move the zero points and scales
};

__global__ __launch_bounds__(512) void awq_repack_nk_kernel(RepackParams p) {
    __shared__ uint32_t tile[TKW * PITCH];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < p.main_blocks) {
        const int tn = blockIdx.x % p.nt, tk = blockIdx.x / p.nt;
        const int n0 = tn * TN, kw0 = tk * TKW;
        // ---- in: 256 rows x 32 words, half a wave reads one row's 128 bytes; LDS [kw][n]
#pragma unroll 4
        for (int i = 0; i < TN * TKW / 512; ++i) {
            const int e = i * 512 + tid, r = e >> 5, kw = e & 31;
            uint32_t v = 0;
            if (n0 + r < p.N && kw0 + kw < p.KW) v = __builtin_nontemporal_load(p.qw_in + (size_t)(n0 + r) * p.KW + kw0 + kw);
            tile[kw * PITCH + r] = v;
        }
        __syncthreads();
        // ---- out: item (kw, c): the eight input words of columns 8c .. 8c+7 at word kw -> eight output words (k = 8 kw + j, word c)
#pragma unroll 2
        for (int i = 0; i < TKW * (TN / 8) / 512; ++i) {
            const int e = i * 512 + tid, c = e & 31, kw = e >> 5;
            const u32x4 a = *reinterpret_cast<const u32x4*>(&tile[kw * PITCH + 8 * c]);
            const u32x4 b = *reinterpret_cast<const u32x4*>(&tile[kw * PITCH + 8 * c + 4]);
            const uint32_t in[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            const int cw = (n0 >> 3) + c;
            if (cw < p.NW && kw0 + kw < p.KW) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // nibble i of the output word = column 8c + ORDER[i]: 0,2,4,6,1,3,5,7
                    const uint32_t o = ((in[0] >> (4 * j)) & 15u) | (((in[2] >> (4 * j)) & 15u) << 4) | (((in[4] >> (4 * j)) & 15u) << 8) |
                                       (((in[6] >> (4 * j)) & 15u) << 12) | (((in[1] >> (4 * j)) & 15u) << 16) | (((in[3] >> (4 * j)) & 15u) << 20) |
                                       (((in[5] >> (4 * j)) & 15u) << 24) | (((in[7] >> (4 * j)) & 15u) << 28);
                    p.qw_out[(size_t)(8 * (kw0 + kw) + j) * p.NW + cw] = o;
                }
            }
        }
        return;
    }
    // ---- zero points and scales: item (group, word c): eight rows' zero nibbles of the group -> one word; eight scales
    const int items = p.G * p.NW;
    for (int e = ((int)blockIdx.x - p.main_blocks) * 512 + tid; e < items; e += ((int)gridDim.x - p.main_blocks) * 512) {
        const int c = e % p.NW, gi = e / p.NW;
        uint32_t o = 0;
        half_t s[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = 8 * c + r;
            const uint32_t z = (p.qz_in[(size_t)n * p.ZW + (gi >> 3)] >> (4 * (gi & 7))) & 15u;
            constexpr int POS[8] = {0, 4, 1, 5, 2, 6, 3, 7};  // column r sits in nibble POS[r] (the inverse of ORDER)
            o |= z << (4 * POS[r]);
            s[r] = p.sc_in[(size_t)n * p.SW + gi];
        }
        p.qz_out[(size_t)gi * p.NW + c] = o;
        *reinterpret_cast<u32x4*>(p.sc_out + (size_t)gi * p.N + 8 * c) = *reinterpret_cast<const u32x4*>(s);
    }
}

// ---- GEMVFast words -> GEMM-layout words (round 6): the prefill route of WQLinear_GEMVFast (awq/modules/linear/gemv_fast.py:203-206
// runs awq_v2_ext.gemm_forward_cuda_prefill on this layout).  in: qweight int16 [N/4, K], element [r, 64 b + 16 i + 8 h + t] nibble j =
// w[4 r + i, 64 b + 32 h + 8 j + t] (gemv_fast.py:26-65); out: qweight int32 [K, N/8] in the AWQ nibble order.  The
// format's scales and fp16 zero terms are [>= K/g, N] already: gemm_regb's FZ form reads them as they are.  Pure nibble moves.
// Tile = 256 output columns (64 int16 rows) x 256 k (512-byte read runs, 128-byte write runs) through LDS.
constexpr int FT_R = 64, FT_K = 256, FT_PITCH = FT_K + 8;  // int16 elements; 33 KB

struct RepackFastParams {
    const uint16_t* qw_in;
    uint32_t* qw_out;
    int K, N, NW, R4;  // NW = N / 8 words per output row, R4 = N / 4 input rows
    int nt, kt;
};

__global__ __launch_bounds__(512) void awq_repack_fast_kernel(RepackFastParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[FT_R * FT_PITCH];
    const int tid = threadIdx.x;
    const int tn = blockIdx.x % p.nt, tk = blockIdx.x / p.nt;
    const int r0 = tn * FT_R, k0 = tk * FT_K;
    // ---- in: 64 rows x 256 int16 = 64 x 32 chunks of 16 bytes; 32 consecutive threads read one row's 512 bytes
#pragma unroll
    for (int i = 0; i < FT_R * (FT_K / 8) / 512; ++i) {
        const int e = i * 512 + tid, r = e >> 5, ch = e & 31;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < p.R4 && k0 + 8 * ch < p.K) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.qw_in + (size_t)(r0 + r) * p.K + k0 + 8 * ch));
        *reinterpret_cast<u32x4*>(&tile[r * FT_PITCH + 8 * ch]) = v;
    }
    __syncthreads();
    // ---- out: item (b, h, t, c): the eight int16 elements (rows 2c, 2c + 1; i = 0..3) at 64 b + 16 i + 8 h + t hold, in nibble j, the
    // weights of columns 8c .. 8c + 7 at k = 64 b + 32 h + 8 j + t -> four output words (j = 0..3), word column c
#pragma unroll
    for (int it = 0; it < (FT_K / 64) * 2 * 8 * (FT_R / 2) / 512; ++it) {
        const int e = it * 512 + tid, c = e & 31, rest = e >> 5, t = rest & 7, h = (rest >> 3) & 1, b = rest >> 4;
        uint32_t in[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) in[q] = tile[(2 * c + (q >> 2)) * FT_PITCH + 64 * b + 16 * (q & 3) + 8 * h + t];
        const int cw = (r0 >> 1) + c;
        if (cw < p.NW) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 64 * b + 32 * h + 8 * j + t;
                // nibble i of the output word = column 8c + ORDER[i], ORDER = 0,2,4,6,1,3,5,7
                const uint32_t o = ((in[0] >> (4 * j)) & 15u) | (((in[2] >> (4 * j)) & 15u) << 4) | (((in[4] >> (4 * j)) & 15u) << 8) |
                                   (((in[6] >> (4 * j)) & 15u) << 12) | (((in[1] >> (4 * j)) & 15u) << 16) | (((in[3] >> (4 * j)) & 15u) << 20) |
                                   (((in[5] >> (4 * j)) & 15u) << 24) | (((in[7] >> (4 * j)) & 15u) << 28);
                if (k < p.K) p.qw_out[(size_t)k * p.NW + cw] = o;
            }
        }
    }
}

}  // namespace

int awq_repack_gemvfast_to_gemm(const int16_t* qweight, int32_t* qweight_out, int64_t K, int64_t N, void* stream) {
    if (K <= 0 || N <= 0 || K % 64 || N % 8 || K > INT32_MAX || N > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (!qweight || !qweight_out) return AWQ_ERR_NULL;
    if (((uintptr_t)qweight & 15) || ((uintptr_t)qweight_out & 3)) return AWQ_ERR_BAD_ALIGNMENT;
    RepackFastParams p;
    p.qw_in = reinterpret_cast<const uint16_t*>(qweight);
    p.qw_out = reinterpret_cast<uint32_t*>(qweight_out);
    p.K = (int)K; p.N = (int)N; p.NW = (int)(N / 8); p.R4 = (int)(N / 4);
    p.nt = (p.R4 + FT_R - 1) / FT_R;
    p.kt = (p.K + FT_K - 1) / FT_K;
    hipLaunchKernelGGL(awq_repack_fast_kernel, dim3((unsigned)(p.nt * p.kt)), dim3(512), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_repack_gemv_to_gemm(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, int32_t* qweight_out, uint16_t* scales_out,
                            int32_t* qzeros_out, int64_t K, int64_t N, int64_t group_size, int64_t zeros_width, void* stream) {
    if (K <= 0 || N <= 0 || group_size <= 0 || K % group_size || K % 8 || N % 8 || zeros_width <= 0) return AWQ_ERR_BAD_SHAPE;
    if (zeros_width * 8 < K / group_size || K > INT32_MAX || N > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (!qweight || !scales || !qzeros || !qweight_out || !scales_out || !qzeros_out) return AWQ_ERR_NULL;
    if (((uintptr_t)scales_out & 15) || ((uintptr_t)qweight & 3)) return AWQ_ERR_BAD_ALIGNMENT;
    RepackParams p;
    p.qw_in = reinterpret_cast<const uint32_t*>(qweight);
    p.qz_in = reinterpret_cast<const uint32_t*>(qzeros);
    p.sc_in = reinterpret_cast<const half_t*>(scales);
    p.qw_out = reinterpret_cast<uint32_t*>(qweight_out);
    p.qz_out = reinterpret_cast<uint32_t*>(qzeros_out);
    p.sc_out = reinterpret_cast<half_t*>(scales_out);
    p.K = (int)K; p.N = (int)N; p.KW = (int)(K / 8); p.ZW = (int)zeros_width; p.SW = (int)(8 * zeros_width); p.NW = (int)(N / 8);
    p.g = (int)group_size; p.G = (int)(K / group_size);
    p.nt = (p.N + TN - 1) / TN;
    p.kt = (p.KW + TKW - 1) / TKW;
    p.main_blocks = p.nt * p.kt;
    const int small_items = p.G * p.NW;
    int small_blocks = (small_items + 511) / 512;
    if (small_blocks > 256) small_blocks = 256;
    hipLaunchKernelGGL(awq_repack_nk_kernel, dim3((unsigned)(p.main_blocks + small_blocks)), dim3(512), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
