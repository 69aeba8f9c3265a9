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
            o[e] = (half_t)awq_silu_mul_f32((float)g[e], (float)u[e]);
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

// ---- MoE routing in ONE launch: softmax + top-k (+ renormalise) + block alignment.
//
// Replaces awq_ext.topk_softmax + awq_ext.moe_alig_block_size (awq/modules/fused/moe.py:94-171,
// two kernels plus ~20 torch glue kernels in the sync-free torch restatement) for decode-sized
// batches: the whole routing state of a step fits one workgroup.  Semantics: routing weights =
// softmax(logits.float()) in fp32, top-k by value (lowest index first on ties), optional
// renormalisation; sorted_token_ids = pair indices grouped by expert in pair order, each expert's
// run padded to a multiple of `block` with the sentinel num_pairs; expert_ids[b] = expert of block b.
namespace {
constexpr int ROUTE_MAX_E = 64, ROUTE_MAX_K = 8;

__global__ __launch_bounds__(256) void awq_moe_route_kernel(const float* __restrict__ logits, float* __restrict__ topk_w,
                                                           int* __restrict__ topk_ids, int* __restrict__ sorted_ids,
                                                           int* __restrict__ expert_ids, int* __restrict__ num_post_pad,
                                                           int T, int E, int k, int renorm, int block, int cap_sorted,
                                                           int cap_blocks, int e0, int nl, int routed) {
    __shared__ int counts[ROUTE_MAX_E], pad_start[ROUTE_MAX_E + 1];
    const int tid = threadIdx.x;
    const int P = T * k;
    if (tid < E) counts[tid] = 0;
    __syncthreads();
    // phase 1, one WAVE per token, lane e holds expert e's logit (round 6: the first version kept a token's E values in a
    // run-time indexed private array -- scratch memory: 10 us for four tokens).  The arithmetic keeps its order: the softmax
    // denominator is summed over e = 0 .. E - 1 in sequence, the k selections take the first maximum.
    const int lane = tid & 63, wv = tid >> 6;
    // (routing: a grid of ceil(T / 4) blocks, one token per wave.  With the alignment pass ONE block does both for up to 8 tokens;
    // beyond, the launcher runs the routing as its own grid first and this block only counts the ids it wrote: routed < 0)
    if (routed < 0) {
        for (int i = tid; i < P; i += 256) {
            const int e = topk_ids[i];
            if (e >= e0 && e < e0 + nl) atomicAdd(&counts[e - e0], 1);
        }
    }
    for (int t = 4 * blockIdx.x + wv; t < T && routed >= 0; t += 4 * gridDim.x) {
        const float lg = lane < E ? logits[(int64_t)t * E + lane] : -INFINITY;
        float mx = lg;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float v = lane < E ? expf(lg - mx) : -3.f;  // (lanes past E never win a selection: every real value is >= 0)
        float sum = 0.f;
        for (int e = 0; e < E; ++e) sum += __shfl(v, e, 64);
        const float inv = 1.0f / sum;
        float wsum = 0.f, mine = 0.f;
        int mine_id = 0;
        for (int j = 0; j < k; ++j) {
            float bv = v;
            int best = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(best, o, 64);
                if (ov > bv || (ov == bv && oi < best)) {
                    bv = ov;
                    best = oi;
                }
            }
            if (lane == best) v = -2.f;  // taken
            const float wj = bv * inv;
            wsum += wj;
            if (lane == j) {
                mine = wj;
                mine_id = best;
            }
        }
        if (lane < k) {
            topk_w[(int64_t)t * k + lane] = renorm ? mine / wsum : mine;
            topk_ids[(int64_t)t * k + lane] = mine_id;
            if (mine_id >= e0 && mine_id < e0 + nl) atomicAdd(&counts[mine_id - e0], 1);  // pairs of experts outside [e0, e0 + nl) are not placed
        }
    }
    if (block <= 0) return;  // routing only (the row-streaming MoE decode path runs pairs, not aligned blocks)
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        for (int e = 0; e < nl; ++e) {
            pad_start[e] = pos;
            pos += (counts[e] + block - 1) / block * block;
        }
        pad_start[nl] = pos;
        *num_post_pad = pos;
    }
    __syncthreads();
    for (int i = tid; i < cap_sorted; i += 256) sorted_ids[i] = P;
    for (int b = tid; b < cap_blocks; b += 256) {
        int e = 0;
        while (e < nl - 1 && b * block >= pad_start[e + 1]) ++e;
        expert_ids[b] = e;  // relative to e0
    }
    __syncthreads();  // the fill above and the id tables written in phase 1 are visible to the block
    if (tid < nl) {  // stable placement: one thread per (local) expert walks the pairs in order
        int pos = pad_start[tid];
        for (int p = 0; p < P; ++p)
            if (topk_ids[p] == e0 + tid) sorted_ids[pos++] = p;
    }
}
}  // namespace

int awq_launch_moe_route(const float* logits, float* topk_w, int* topk_ids, int* sorted_ids, int* expert_ids,
                         int* num_post_pad, int T, int E, int k, int renorm, int block, int first_expert, int num_local,
                         hipStream_t st) {
    if (T < 1 || E < 1 || E > ROUTE_MAX_E || k < 1 || k > ROUTE_MAX_K || k > E || block < 0) return AWQ_ERR_UNSUPPORTED;
    if (first_expert < 0 || num_local < 1 || first_expert + num_local > E) return AWQ_ERR_BAD_SHAPE;
    const int P = T * k;
    const int cap_sorted = P + num_local * (block - 1), cap_blocks = P + num_local;
    if (block == 0 || T > 8) {  // the routing as a grid of its own (a wave per token), then -- if asked for -- one block aligns
        hipLaunchKernelGGL(awq_moe_route_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, logits, topk_w, topk_ids, sorted_ids, expert_ids,
                           num_post_pad, T, E, k, renorm, 0, 0, 0, first_expert, num_local, 0);
        if (block > 0)
            hipLaunchKernelGGL(awq_moe_route_kernel, dim3(1), dim3(256), 0, st, logits, topk_w, topk_ids, sorted_ids, expert_ids,
                               num_post_pad, T, E, k, renorm, block, cap_sorted, cap_blocks, first_expert, num_local, -1);
    } else {
        hipLaunchKernelGGL(awq_moe_route_kernel, dim3(1), dim3(256), 0, st, logits, topk_w, topk_ids, sorted_ids, expert_ids,
                           num_post_pad, T, E, k, renorm, block, cap_sorted, cap_blocks, first_expert, num_local, 0);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}


// ---- MoE prefill: the (token, expert) pairs sorted by expert as an INDEX LIST, in one launch (round 6).
//
// Replaces the torch glue of the prefill-sized path of apply_moe_weights (awq/modules/fused/moe.py:45-91 at token counts where an
// expert sees GEMM-sized batches): argsort(stable) + scatter_add + cumsum + index_select.  order [P] = the pair indices grouped by
// expert, pair order kept inside an expert (a stable counting sort: identical to torch.argsort(topk_ids.flatten(), stable=True));
// seg [E + 1] = the row range of each expert.  Pairs whose id is outside [0, E) are not placed (seg[E] < P then).
// One block of 1024 threads walks the pairs in chunks of 1024: a histogram pass (LDS atomics), then per chunk the rank of a pair
// inside its expert = pairs of that expert in earlier waves of the chunk + earlier lanes of its wave (ballots).  E <= 64.
namespace {
__global__ __launch_bounds__(1024) void awq_moe_sort_kernel(const int* __restrict__ ids, int P, int E, int* __restrict__ order,
                                                            int* __restrict__ seg) {
    __shared__ int cnt[ROUTE_MAX_E], cursor[ROUTE_MAX_E], wcnt[16][ROUTE_MAX_E];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < ROUTE_MAX_E) cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < P; i += 1024) {
        const int e = ids[i];
        if ((unsigned)e < (unsigned)E) atomicAdd(&cnt[e], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int pos = 0;
        for (int e = 0; e < E; ++e) {
            cursor[e] = pos;
            seg[e] = pos;
            pos += cnt[e];
        }
        seg[E] = pos;
    }
    __syncthreads();
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int c0 = 0; c0 < P; c0 += 1024) {
        const int i = c0 + tid;
        const int e = i < P ? ids[i] : -1;
        const bool valid = (unsigned)e < (unsigned)E;
        int rank = 0;
        for (int x = 0; x < E; ++x) {
            const unsigned long long m = __ballot(valid && e == x);
            if (lane == 0) wcnt[wave][x] = __popcll(m);
            if (valid && e == x) rank = __popcll(m & lt);
        }
        __syncthreads();
        if (valid) {
            int off = cursor[e];
            for (int w = 0; w < wave; ++w) off += wcnt[w][e];
            order[off + rank] = i;
        }
        __syncthreads();
        if (tid < E) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wcnt[w][tid];
            cursor[tid] += tot;
        }
        __syncthreads();
    }
}
}  // namespace

int awq_launch_moe_sort(const int* ids, int P, int E, int* order, int* seg, hipStream_t st) {
    if (P < 1 || E < 1 || E > ROUTE_MAX_E) return AWQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(awq_moe_sort_kernel, dim3(1), dim3(1024), 0, st, ids, P, E, order, seg);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
