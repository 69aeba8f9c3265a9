// stream_probe3.hip -- measurement-only microbenchmark (NOT part of the product library).
//
// Floors for the GEMV-layout decode kernels, hipGraph-replayed (the way bench.py runs them), cold buffers:
//   linear   : grid-stride 16-byte non-temporal reads (what a bare read of the matrix costs per launch)
//   rows     : the access pattern of csrc/gemv_rows.hip: a wave instruction = 1 KiB of ONE row, U rows in flight
//   tile16   : the access pattern a direct-to-MFMA-fragment load needs: lane (n = l % 16, kb = l / 16) reads
//              16 bytes of row n at 16*kb + 64*i  -> a wave instruction touches 16 rows x 64 bytes
//   tile8    : 8 rows x 128 bytes per wave instruction
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe3.hip -o tools/bin/stream_probe3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <bool NT>
__global__ __launch_bounds__(256) void linear_read(const u32x4* __restrict__ w, uint32_t* __restrict__ out, int64_t n16) {
    uint32_t acc = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        u32x4 v = ld<NT>(w + i);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// rows: wk waves side by side on a row (chunks of 64 lanes x 16 B), U rows in flight, rows dealt evenly to waves
template <int U>
__global__ __launch_bounds__(512) void rows_read(const u32x4* __restrict__ w, uint32_t* __restrict__ out, int N, int C, int wk, int rg) {
    const int lane = threadIdx.x & 63, wki = threadIdx.x >> 6, rgi = threadIdx.y;
    const int groups = gridDim.x * rg, gi = blockIdx.x * rg + rgi;
    const int base = N / groups, rem = N % groups;
    const int r0 = gi * base + min(gi, rem), cnt = base + (gi < rem);
    const int c = min(wki * 64 + lane, C - 1);
    uint32_t acc = 0;
    for (int b = 0; b < cnt; b += U) {
        u32x4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) q[u] = ld<true>(w + (int64_t)min(r0 + b + u, r0 + cnt - 1) * C + c);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= q[u][0] ^ q[u][1] ^ q[u][2] ^ q[u][3];
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// tile: R rows per wave instruction (16 or 8), each lane 16 B; a wave owns 16-row tiles and walks K
template <int R, int U>
__global__ __launch_bounds__(256) void tile_read(const u32x4* __restrict__ w, uint32_t* __restrict__ out, int N, int C) {
    constexpr int LPR = 64 / R;  // lanes per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane % R, kb = lane / R;
    const int tiles = (N + R - 1) / R;
    const int nw = gridDim.x * 4, wi = blockIdx.x * 4 + wave;
    uint32_t acc = 0;
    const int steps = C / LPR;  // chunks of a row per lane
    for (int t = wi; t < tiles; t += nw) {
        const int row = min(t * R + j, N - 1);
        const u32x4* rp = w + (int64_t)row * C + kb;
        for (int s0 = 0; s0 < steps; s0 += U) {
            u32x4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] = ld<true>(rp + (int64_t)min(s0 + u, steps - 1) * LPR);
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= q[u][0] ^ q[u][1] ^ q[u][2] ^ q[u][3];
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

struct Shape { int K, N; };

template <typename F>
double graph_us(hipStream_t st, size_t nb, F launch) {
    for (size_t i = 0; i < nb; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (size_t i = 0; i < nb; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 5;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / (reps * nb);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    uint32_t* out; CK(hipMalloc(&out, 64 << 20));
    const Shape shapes[] = {{4096, 4096}, {11008, 4096}, {4096, 12288}, {4096, 22016}};
    for (Shape s : shapes) {
        const size_t bytes = (size_t)s.K * s.N / 2;
        const size_t nb = (700ull << 20) / bytes + 1;
        std::vector<uint32_t*> bufs(nb);
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemsetAsync(b, 0x5A, bytes, st)); }
        CK(hipStreamSynchronize(st));
        const int C = s.K / 32;
        auto report = [&](const char* name, double us) {
            printf("K%5d N%5d %-34s %7.2f us %7.0f GB/s\n", s.K, s.N, name, us, bytes / us / 1e3);
            fflush(stdout);
        };
        char nm[96];
        for (int g : {256, 512, 1024, 2048}) {
            snprintf(nm, sizeof nm, "linear nt grid %d", g);
            report(nm, graph_us(st, nb, [&](size_t i) { hipLaunchKernelGGL(linear_read<true>, dim3(g), dim3(256), 0, st, (const u32x4*)bufs[i], out, (int64_t)(bytes / 16)); }));
        }
        report("linear default-policy grid 2048", graph_us(st, nb, [&](size_t i) { hipLaunchKernelGGL(linear_read<false>, dim3(2048), dim3(256), 0, st, (const u32x4*)bufs[i], out, (int64_t)(bytes / 16)); }));
        const int wk = (C + 63) / 64, rg = 8 / wk < 1 ? 1 : 8 / wk;
        for (int bpc : {1, 2, 4}) {
            const int blocks = 256 * bpc;
            snprintf(nm, sizeof nm, "rows wk%d rg%d bpc%d U4", wk, rg, bpc);
            report(nm, graph_us(st, nb, [&](size_t i) { hipLaunchKernelGGL(rows_read<4>, dim3(blocks), dim3(64 * wk, rg), 0, st, (const u32x4*)bufs[i], out, s.N, C, wk, rg); }));
            snprintf(nm, sizeof nm, "rows wk%d rg%d bpc%d U8", wk, rg, bpc);
            report(nm, graph_us(st, nb, [&](size_t i) { hipLaunchKernelGGL(rows_read<8>, dim3(blocks), dim3(64 * wk, rg), 0, st, (const u32x4*)bufs[i], out, s.N, C, wk, rg); }));
        }
        if (C % 8 == 0)
            for (int g : {256, 512, 1024}) {
                snprintf(nm, sizeof nm, "tile16 (16 rows x 64 B) grid %d U8", g);
                report(nm, graph_us(st, nb, [&](size_t i) { hipLaunchKernelGGL((tile_read<16, 8>), dim3(g), dim3(256), 0, st, (const u32x4*)bufs[i], out, s.N, C); }));
                snprintf(nm, sizeof nm, "tile8  (8 rows x 128 B) grid %d U8", g);
                report(nm, graph_us(st, nb, [&](size_t i) { hipLaunchKernelGGL((tile_read<8, 8>), dim3(g), dim3(256), 0, st, (const u32x4*)bufs[i], out, s.N, C); }));
            }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
