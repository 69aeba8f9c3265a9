// stream_probe2.hip -- measurement-only microbenchmark (NOT part of the product library).
//
// Question: can a block OWN a narrow column chunk over ALL of K (no split-K exchange) and still
// stream a GEMM-layout int4 matrix fast?  Lanes run along K (one row each), each reading LB bytes
// of its row; NL lanes side by side along N.  Rows are N/2 bytes apart, so a wave instruction
// touches 64/NL distinct lines and uses LB*NL bytes of each: L2->L1 traffic is amplified, HBM
// traffic is not IF the blocks sharing a line hit the same L2 (same XCD, XMAP=1) or the MALL.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int LB> struct Vec;
template <> struct Vec<4> { typedef uint32_t T; };
template <> struct Vec<8> { typedef u32x2 T; };
template <> struct Vec<16> { typedef u32x4 T; };
template <int LB> __device__ __forceinline__ uint32_t fold(typename Vec<LB>::T v) {
    if constexpr (LB == 4) return v;
    else if constexpr (LB == 8) return v[0] ^ v[1];
    else return v[0] ^ v[1] ^ v[2] ^ v[3];
}

// block = 256 threads; NL lanes along N (LB bytes each), 256/NL lanes along K; R rows in flight per lane
template <int LB, int NLOG, int R, bool XMAP>
__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ w, uint32_t* __restrict__ out, int K, int NW,
                                             int rows_per_block, int tiles) {
    typedef typename Vec<LB>::T V;
    constexpr int NL = 1 << NLOG, KL = 256 / NL, WPL = LB / 4;
    int bx = blockIdx.x % tiles, by = blockIdx.x / tiles;
    int tile = bx;
    if constexpr (XMAP) {  // blocks with equal (blockIdx % 8) get CONSECUTIVE tiles (they share lines)
        const int per = tiles / 8;
        if (tiles % 8 == 0) tile = (blockIdx.x % 8) * per + (bx / 8);
    }
    const int nl = threadIdx.x & (NL - 1), kl = threadIdx.x >> NLOG;
    const int colw = (tile * NL + nl) * WPL;
    const int kbeg = by * rows_per_block, kend = min(K, kbeg + rows_per_block);
    uint32_t acc = 0;
    if (colw < NW) {
        const uint32_t* base = w + colw;
        for (int k0 = kbeg + kl; k0 < kend; k0 += KL * R) {
            V q[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = k0 + r * KL;
                const V* p = reinterpret_cast<const V*>(base + (int64_t)(row < kend ? row : k0) * NW);
                q[r] = __builtin_nontemporal_load(p);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc ^= fold<LB>(q[r]);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void linear_read(const u32x4* __restrict__ w, uint32_t* __restrict__ out, int64_t n16) {
    uint32_t acc = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        u32x4 v = __builtin_nontemporal_load(w + i);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void empty_kernel(uint32_t* out) { if (threadIdx.x == 999) out[0] = 1; }

struct Shape { int K, N; };

template <int LB, int NLOG, int R, bool XMAP>
void run(const std::vector<uint32_t*>& bufs, uint32_t* out, Shape s, int splitk, hipStream_t st) {
    constexpr int NL = 1 << NLOG, WPL = LB / 4;
    const int NW = s.N / 8;
    const int tile_words = NL * WPL;
    const int tiles = (NW + tile_words - 1) / tile_words;
    int rpb = (s.K + splitk - 1) / splitk;
    dim3 grid(tiles * splitk);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 3;
    for (auto b : bufs) hipLaunchKernelGGL((probe<LB, NLOG, R, XMAP>), grid, dim3(256), 0, st, b, out, s.K, NW, rpb, tiles);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        for (auto b : bufs) hipLaunchKernelGGL((probe<LB, NLOG, R, XMAP>), grid, dim3(256), 0, st, b, out, s.K, NW, rpb, tiles);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * bufs.size());
    const double bytes = (double)s.K * s.N / 2;
    printf("K%5d N%5d LB%-2d NL%-2d R%-2d %s splitk %2d (%5d blk, %3d B/row/blk) %7.2f us %7.0f GB/s\n", s.K, s.N, LB, NL, R,
           XMAP ? "xmap" : "rrob", splitk, tiles * splitk, LB * NL, us, bytes / us / 1e3);
    fflush(stdout);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    uint32_t* out; CK(hipMalloc(&out, 64 << 20));
    {   // kernel boundary: empty kernels back to back
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int g : {1, 256, 1024}) {
            for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, st, out);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, st, out);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("empty kernel grid %d: %.2f us per launch (eager, same stream)\n", g, ms);
        }
    }
    const Shape shapes[] = {{4096, 4096}, {11008, 4096}, {4096, 12288}, {4096, 22016}};
    for (Shape s : shapes) {
        const size_t bytes = (size_t)s.K * s.N / 2;
        const int nb = (int)((700ull << 20) / bytes) + 1;
        std::vector<uint32_t*> bufs(nb);
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemsetAsync(b, 0x5A, bytes, st)); }
        CK(hipStreamSynchronize(st));
        {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int g : {1024, 2048}) {
                for (auto b : bufs) hipLaunchKernelGGL(linear_read, dim3(g), dim3(256), 0, st, (const u32x4*)b, out, (int64_t)(bytes / 16));
                CK(hipEventRecord(e0, st));
                for (auto b : bufs) hipLaunchKernelGGL(linear_read, dim3(g), dim3(256), 0, st, (const u32x4*)b, out, (int64_t)(bytes / 16));
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("K%5d N%5d linear 16B read grid %d: %7.2f us %7.0f GB/s\n", s.K, s.N, g, ms * 1e3 / nb, bytes / (ms * 1e3 / nb) / 1e3);
            }
        }
        for (int sk : {1, 2, 4, 8}) {
#define RUN(LB, NLOG, R) run<LB, NLOG, R, false>(bufs, out, s, sk, st); run<LB, NLOG, R, true>(bufs, out, s, sk, st);
            RUN(8, 0, 16)   //  8 B/row/block
            RUN(16, 0, 16)  // 16
            RUN(16, 1, 16)  // 32
            RUN(16, 2, 16)  // 64
            RUN(16, 3, 16)  // 128
            RUN(16, 0, 8)
            RUN(16, 1, 8)
        }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
