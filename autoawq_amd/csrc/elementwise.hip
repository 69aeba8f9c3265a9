// elementwise.hip -- the one elementwise op on the fused-MLP / MoE path.
//
// Replaces awq_ext.silu_and_mul(out, gate_up) (awq/modules/fused/moe.py:73-76): gate_up [rows, 2*D]
// fp16 = [gate | up] (mixtral.py:131-138), out [rows, D] fp16 = silu(gate) * up.  HBM-bound:
// 6*D bytes per row.  Arithmetic: fp32 silu (x / (1 + exp(-x))), product in fp32, one rounding --
// (the kernel source is not in the reference tree; semantics from the call site).
#include "awq_device.h"
#include "awq_internal.h"

namespace {
__global__ __launch_bounds__(256) void awq_silu_and_mul_kernel(const half_t* __restrict__ in, half_t* __restrict__ out,
                                                              int64_t rows, int64_t D) {
    const int64_t chunks = D >> 3;  // 8 halfs = 16 bytes per thread step
    const int64_t total = rows * chunks;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks, c = i % chunks;
        const half8_t g = *reinterpret_cast<const half8_t*>(in + r * 2 * D + 8 * c);
        const half8_t u = *reinterpret_cast<const half8_t*>(in + r * 2 * D + D + 8 * c);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)g[e];
            o[e] = (half_t)((x / (1.0f + expf(-x))) * (float)u[e]);
        }
        *reinterpret_cast<half8_t*>(out + r * D + 8 * c) = o;
    }
}
}  // namespace

int awq_launch_silu_and_mul(const uint16_t* in, uint16_t* out, int64_t rows, int64_t D, hipStream_t st) {
    if (rows < 0 || D < 0 || D % 8) return AWQ_ERR_BAD_SHAPE;
    const int64_t total = rows * (D / 8);
    if (total == 0) return AWQ_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(awq_silu_and_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       reinterpret_cast<const half_t*>(in), reinterpret_cast<half_t*>(out), rows, D);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
