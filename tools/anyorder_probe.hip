// anyorder_probe.hip -- can a kernel on gfx950 / ROCm 7.2 start before its stream predecessor ends?
//
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/bin/anyorder_probe
//
// Decode is a chain of dependent GEMVs whose WEIGHTS do not depend on the activations.  If launch
// i+1 may become resident while launch i still runs (AQL packet without the barrier bit:
// hipExtAnyOrderLaunch), it can request its weights first and wait for its activations on a
// device-side tag instead of on the kernel boundary.  This probe measures, with wall_clock64
// (100 MHz), whether that happens (a) eagerly, (b) when captured in a hipGraph, (c) on two streams,
// whether the CP places ALL blocks of launch i before any of launch i+1, and what a chain of
// dependent launches costs per link in each mode.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out: [0] min start, [1] max start, [2] max end, [3] arrivals
__global__ void k_spin(u64* out, uint32_t* flag, uint32_t epoch, u64 delay) {
    const u64 t0 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin(&out[0], t0);
        atomicMax(&out[1], t0);
    }
    while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 t1 = wall_clock64();
        atomicMax(&out[2], t1);
        __threadfence();
        const u64 n = atomicAdd(&out[3], 1ull);
        if (n == gridDim.x - 1) st_sc1(flag, epoch);
    }
}

// out: [0] min start, [1] max start, [2] max time the flag was seen, [3] timeouts
__global__ void k_wait(u64* out, const uint32_t* flag, uint32_t epoch, u64 timeout) {
    const u64 t0 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin(&out[0], t0);
        atomicMax(&out[1], t0);
        bool ok = false;
        while (wall_clock64() - t0 < timeout) {
            if (ld_sc1(flag) == epoch) { ok = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        atomicMax(&out[2], wall_clock64());
        if (!ok) atomicAdd(&out[3], 1ull);
    }
}

// chain link: every block waits for tag_in == epoch (skipped when tag_in == nullptr), works `work` ticks,
// the last block to finish publishes tag_out = epoch and zeroes its arrival counter.
__global__ void k_link(const uint32_t* tag_in, uint32_t* tag_out, uint32_t* arrivals, uint32_t* err, uint32_t epoch,
                       u64 work, u64 timeout) {
    if (threadIdx.x == 0 && tag_in) {
        const u64 t0 = wall_clock64();
        bool ok = false;
        while (wall_clock64() - t0 < timeout) {
            if (ld_sc1(tag_in) == epoch) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) atomicAdd(err, 1u);
    }
    __syncthreads();
    const u64 t1 = wall_clock64();
    while (wall_clock64() - t1 < work) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = atomicAdd(arrivals, 1u);
        if (n == gridDim.x - 1) {
            atomicExch(arrivals, 0u);
            st_sc1(tag_out, epoch);
        }
    }
}

static void reset(u64* d) {
    u64 h[8] = {~0ull, 0, 0, 0, ~0ull, 0, 0, 0};
    CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
}

static void report(const char* name, u64* d) {
    u64 h[8];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const double us = 0.01;  // 100 MHz
    const double base = (double)h[0];
    printf("%-34s A.start[min,max]=[0, %.2f] A.end=%.2f | B.start[min,max]=[%.2f, %.2f] B.seen=%.2f timeouts=%llu  => %s\n",
           name, ((double)h[1] - base) * us, ((double)h[2] - base) * us, ((double)h[4] - base) * us,
           ((double)h[5] - base) * us, ((double)h[6] - base) * us, h[7],
           h[4] < h[2] ? (h[4] >= h[1] ? "OVERLAP (B after all A blocks placed)" : "OVERLAP (B before A fully placed)")
                       : "serialized");
}

int main() {
    CK(hipSetDevice(0));
    u64* out;
    uint32_t* flag;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&flag, 4096));
    CK(hipMemset(flag, 0, 4096));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const u64 delay = 5000;     // 50 us
    const u64 timeout = 50000;  // 500 us
    uint32_t epoch = 1;

    for (int gridA : {256, 2048, 16384}) {
        printf("---- A grid %d x 256 threads, spin 50 us; B grid 512\n", gridA);
        // 1. plain, same stream
        reset(out);
        ++epoch;
        hipLaunchKernelGGL(k_spin, dim3(gridA), dim3(256), 0, s1, out, flag, epoch, delay);
        hipLaunchKernelGGL(k_wait, dim3(512), dim3(256), 0, s1, out + 4, flag, epoch, timeout);
        CK(hipStreamSynchronize(s1));
        report("same stream, plain", out);
        // 2. any-order, same stream
        reset(out);
        ++epoch;
        hipLaunchKernelGGL(k_spin, dim3(gridA), dim3(256), 0, s1, out, flag, epoch, delay);
        hipExtLaunchKernelGGL(k_wait, dim3(512), dim3(256), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, out + 4, flag,
                              epoch, timeout);
        CK(hipStreamSynchronize(s1));
        report("same stream, hipExtAnyOrderLaunch", out);
        // 3. two streams
        reset(out);
        ++epoch;
        hipLaunchKernelGGL(k_spin, dim3(gridA), dim3(256), 0, s1, out, flag, epoch, delay);
        hipLaunchKernelGGL(k_wait, dim3(512), dim3(256), 0, s2, out + 4, flag, epoch, timeout);
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        report("two streams", out);
        // 4. captured in a graph: any-order on one stream
        {
            reset(out);
            ++epoch;
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            hipLaunchKernelGGL(k_spin, dim3(gridA), dim3(256), 0, s1, out, flag, epoch, delay);
            hipExtLaunchKernelGGL(k_wait, dim3(512), dim3(256), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, out + 4,
                                  flag, epoch, timeout);
            CK(hipStreamEndCapture(s1, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s1));
            CK(hipStreamSynchronize(s1));
            report("graph: any-order captured", out);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
        // 5. graph with two parallel branches (fork / join by events)
        {
            reset(out);
            ++epoch;
            hipGraph_t g;
            hipGraphExec_t ge;
            hipEvent_t fork, join;
            CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            CK(hipEventRecord(fork, s1));
            CK(hipStreamWaitEvent(s2, fork, 0));
            hipLaunchKernelGGL(k_spin, dim3(gridA), dim3(256), 0, s1, out, flag, epoch, delay);
            hipLaunchKernelGGL(k_wait, dim3(512), dim3(256), 0, s2, out + 4, flag, epoch, timeout);
            CK(hipEventRecord(join, s2));
            CK(hipStreamWaitEvent(s1, join, 0));
            CK(hipStreamEndCapture(s1, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s1));
            CK(hipStreamSynchronize(s1));
            report("graph: two parallel branches", out);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }

    // ---- chain cost per link: 64 links of 512 blocks x 256 threads, 2 us of "work" each
    {
        const int L = 64, GRID = 512;
        const u64 work = 200;  // 2 us
        uint32_t *tags, *arr, *err;
        CK(hipMalloc(&tags, (L + 1) * 256));
        CK(hipMalloc(&arr, (L + 1) * 256));
        CK(hipMalloc(&err, 4));
        CK(hipMemset(tags, 0, (L + 1) * 256));
        CK(hipMemset(arr, 0, (L + 1) * 256));
        CK(hipMemset(err, 0, 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        auto tag = [&](int i) { return tags + i * 64; };
        auto run_chain = [&](int mode, hipStream_t st, uint32_t ep) {
            // mode 0: plain launches (boundary = dependency), 1: any-order + tag waits, 2: plain + tag waits
            for (int i = 0; i < L; ++i) {
                const uint32_t* tin = (mode != 0 && i > 0) ? tag(i - 1) : nullptr;
                if (mode == 1 && i > 0)
                    hipExtLaunchKernelGGL(k_link, dim3(GRID), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, tin,
                                          tag(i), arr + i * 64, err, ep, work, (u64)100000);
                else
                    hipLaunchKernelGGL(k_link, dim3(GRID), dim3(256), 0, st, tin, tag(i), arr + i * 64, err, ep, work,
                                       (u64)100000);
            }
        };
        const char* names[3] = {"plain boundaries", "any-order + tags", "plain + tags"};
        for (int mode = 0; mode < 3; ++mode) {
            for (int graph = 0; graph < 2; ++graph) {
                float best = 1e9f;
                hipGraph_t g = nullptr;
                hipGraphExec_t ge = nullptr;
                for (int rep = 0; rep < 6; ++rep) {
                    ++epoch;
                    if (graph) {
                        // the epoch is baked into the captured arguments: re-capture per repetition
                        if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
                        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
                        run_chain(mode, s1, epoch);
                        CK(hipStreamEndCapture(s1, &g));
                        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                        CK(hipStreamSynchronize(s1));
                        CK(hipEventRecord(e0, s1));
                        CK(hipGraphLaunch(ge, s1));
                        CK(hipEventRecord(e1, s1));
                    } else {
                        CK(hipStreamSynchronize(s1));
                        CK(hipEventRecord(e0, s1));
                        run_chain(mode, s1, epoch);
                        CK(hipEventRecord(e1, s1));
                    }
                    CK(hipStreamSynchronize(s1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                uint32_t herr;
                CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                printf("chain of %d links (%d blocks, 2 us work): %-18s %-6s  %.1f us total = %.2f us / link  (wait timeouts so far %u)\n",
                       L, GRID, names[mode], graph ? "graph" : "eager", best * 1e3, best * 1e3 / L, herr);
            }
        }
    }
    printf("done\n");
    return 0;
}
