// stream_probe.hip -- measurement-only microbenchmark (NOT part of the product library).
//
// Question it answers: how fast can gfx950 stream a GEMM-layout int4 matrix (row-major [K, N/8]
// int32, lanes along N, rows strided by N/2 bytes) as a function of the access decomposition?
// Every variant reads each byte exactly once, XOR-folds it (so nothing is dead) and writes one
// word per lane at the end.  Build + run:  hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip
// -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LB>
struct Vec;
template <>
struct Vec<8> { typedef u32x2 T; };
template <>
struct Vec<16> { typedef u32x4 T; };

template <int LB>
__device__ __forceinline__ uint32_t fold(typename Vec<LB>::T v) {
    if constexpr (LB == 8) return v[0] ^ v[1];
    else return v[0] ^ v[1] ^ v[2] ^ v[3];
}

// LB: bytes per lane per load; NLOG: log2 lanes along N per wave; R: rows per lane per batch;
// WN: waves of the block placed side by side along N (the other 4/WN stack along K);
// PF: issue the next batch before consuming the current one.
template <int LB, int NLOG, int R, int WN, bool PF, bool NT>
__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ w, uint32_t* __restrict__ out, int K, int NW,
                                             int rows_per_block) {
    typedef typename Vec<LB>::T V;
    constexpr int NL = 1 << NLOG, KLW = 64 / NL, WK = 4 / WN;
    constexpr int WPL = LB / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wk = wave / WN;
    const int nl = lane & (NL - 1), kl = (lane >> NLOG) + wk * KLW;  // K-lane within block
    const int colw = ((blockIdx.x * WN + wn) * NL + nl) * WPL;
    const int kbeg = blockIdx.y * rows_per_block, kend = min(K, kbeg + rows_per_block);
    constexpr int STEP = KLW * WK * R;  // rows per block pass
    uint32_t acc = 0;
    if (colw < NW) {
        const uint32_t* base = w + colw;
        auto ld = [&](int row) -> V {
            const V* p = reinterpret_cast<const V*>(base + (int64_t)row * NW);
            if constexpr (NT) return __builtin_nontemporal_load(p);
            else return *p;
        };
        int k0 = kbeg + kl * R;
        if constexpr (!PF) {
            for (; k0 < kend; k0 += STEP) {
                V q[R];
#pragma unroll
                for (int r = 0; r < R; ++r) q[r] = ld(k0 + r);
#pragma unroll
                for (int r = 0; r < R; ++r) acc ^= fold<LB>(q[r]);
            }
        } else {
            V qa[R], qb[R];
            if (k0 < kend) {
#pragma unroll
                for (int r = 0; r < R; ++r) qa[r] = ld(k0 + r);
            }
            while (k0 < kend) {
                const int k1 = k0 + STEP;
                const int kk1 = k1 < kend ? k1 : k0;  // clamped: branch-free prefetch
#pragma unroll
                for (int r = 0; r < R; ++r) qb[r] = ld(kk1 + r);
#pragma unroll
                for (int r = 0; r < R; ++r) acc ^= fold<LB>(qa[r]);
                if (k1 >= kend) break;
                const int k2 = k1 + STEP;
                const int kk2 = k2 < kend ? k2 : k1;
#pragma unroll
                for (int r = 0; r < R; ++r) qa[r] = ld(kk2 + r);
#pragma unroll
                for (int r = 0; r < R; ++r) acc ^= fold<LB>(qb[r]);
                k0 = k2;
            }
        }
    }
    out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc;
}

// contiguous reference: plain linear read of the same bytes (what a copy-style kernel gets)
__global__ __launch_bounds__(256) void linear_read(const u32x4* __restrict__ w, uint32_t* __restrict__ out, int64_t n16) {
    uint32_t acc = 0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        u32x4 v = __builtin_nontemporal_load(w + i);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

struct Shape { int K, N; };

template <int LB, int NLOG, int R, int WN, bool PF, bool NT>
void run(const char* name, const std::vector<uint32_t*>& bufs, uint32_t* out, Shape s, int splitk, hipStream_t st, int lds = 0) {
    constexpr int NL = 1 << NLOG, KLW = 64 / NL, WK = 4 / WN, WPL = LB / 4;
    const int NW = s.N / 8;
    const int tile_words = WN * NL * WPL;
    const int tiles = (NW + tile_words - 1) / tile_words;
    const int step = KLW * WK * R;
    int passes = (s.K + step - 1) / step;
    if (splitk > passes) splitk = passes;
    int ppb = (passes + splitk - 1) / splitk;
    splitk = (passes + ppb - 1) / ppb;
    dim3 grid(tiles, splitk);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 3;
    for (auto b : bufs) hipLaunchKernelGGL((probe<LB, NLOG, R, WN, PF, NT>), grid, dim3(256), lds, st, b, out, s.K, NW, ppb * step);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        for (auto b : bufs) hipLaunchKernelGGL((probe<LB, NLOG, R, WN, PF, NT>), grid, dim3(256), lds, st, b, out, s.K, NW, ppb * step);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * bufs.size());
    const double bytes = (double)s.K * s.N / 2;
    printf("K%5d N%5d %-34s lds %6d grid %4dx%-3d (%5d blk) %7.2f us %7.0f GB/s\n", s.K, s.N, name, lds, tiles, splitk,
           tiles * splitk, us, bytes / us / 1e3);
    fflush(stdout);
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // allow up to 160 KiB dynamic LDS (used only to cap occupancy)
    const Shape shapes[] = {{4096, 22016}, {4096, 4096}, {11008, 4096}};
    uint32_t* out;
    CK(hipMalloc(&out, 64 << 20));
    for (Shape s : shapes) {
        const size_t bytes = (size_t)s.K * s.N / 2;
        const int nb = (int)((700ull << 20) / bytes) + 1;
        std::vector<uint32_t*> bufs(nb);
        for (auto& b : bufs) {
            CK(hipMalloc(&b, bytes));
            CK(hipMemsetAsync(b, 0x5A, bytes, st));
        }
        CK(hipStreamSynchronize(st));
        {  // linear reference
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int g : {1024, 2048, 4096}) {
                for (auto b : bufs) hipLaunchKernelGGL(linear_read, dim3(g), dim3(256), 0, st, (const u32x4*)b, out, (int64_t)(bytes / 16));
                CK(hipEventRecord(e0, st));
                for (auto b : bufs) hipLaunchKernelGGL(linear_read, dim3(g), dim3(256), 0, st, (const u32x4*)b, out, (int64_t)(bytes / 16));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("K%5d N%5d linear 16B read grid %d: %7.2f us %7.0f GB/s\n", s.K, s.N, g, ms * 1e3 / nb, bytes / (ms * 1e3 / nb) / 1e3);
            }
        }
        for (int lds : {0, 40000, 80000, 160000}) {  // blocks per CU: 8, 4, 2, 1
            for (int sk : {2, 3, 4, 6, 8, 16}) {
                char nm[96];
#define RUN(LB, NLOG, R, WN, PF, NT)                                                             \
    snprintf(nm, sizeof nm, "LB%d NL%d R%d WN%d %s %s", LB, 1 << NLOG, R, WN, PF ? "pf" : "--", NT ? "nt" : "pl"); \
    run<LB, NLOG, R, WN, PF, NT>(nm, bufs, out, s, sk, st, lds);
                RUN(16, 3, 8, 1, false, true)
                RUN(16, 3, 8, 1, true, true)
                RUN(16, 3, 16, 1, false, true)
                RUN(16, 3, 16, 1, true, true)
                RUN(8, 4, 8, 1, true, true)
                RUN(8, 4, 16, 1, true, true)
                RUN(8, 4, 32, 1, false, true)
                RUN(8, 4, 32, 1, true, true)
            }
        }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
