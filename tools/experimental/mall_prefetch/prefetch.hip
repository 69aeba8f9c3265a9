// prefetch.hip -- EXPERIMENTAL (round 4, not in the product build, never run on a GPU yet): a reader that pulls a byte range
// through the memory-side cache (the 256 MiB Infinity Cache sits in front of HBM) and throws the data away.  Question for the
// next GPU session (probe.py): does a decode step get faster when the NEXT layer's packed weights (105 MB at Llama-2-7B shapes)
// are pulled in by such a reader on a second graph branch while the current layer's launches run -- i.e. can the HBM stream be
// kept running through the dispatch gaps and ramps of 128 short launches, with the launches themselves fed from the cache?
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each wave reads 1 KiB per instruction, `unroll` instructions in flight; plain (cache-allocating) loads on purpose
__global__ __launch_bounds__(256) void awq_exp_prefetch_kernel(const u32x4* __restrict__ p, size_t n16, uint32_t* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9E3779B9u && sink) *sink = 1;  // never true for real data in practice; keeps the loads
}

extern "C" __attribute__((visibility("default"))) int awq_exp_prefetch(const void* p, size_t bytes, int blocks, void* sink, void* stream) {
    if (!p || bytes < 16 || blocks <= 0) return -1;
    hipLaunchKernelGGL(awq_exp_prefetch_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const u32x4*>(p), bytes / 16, static_cast<uint32_t*>(sink));
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
