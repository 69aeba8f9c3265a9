// awq_internal.h -- host-side launcher prototypes shared between the .hip translation units
// and capi.hip.  Not part of the public ABI (that is include/awq_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/awq_hip.h"
#include <atomic>

// The opt-in to more than 64 KB of dynamic LDS is an attribute of a kernel ON A DEVICE: asked for once per (kernel, device)
// -- the call sites keep one bit per device ordinal -- thread-safe, idempotent, no other process-wide state (a
// function-local `static const bool` did it once per PROCESS, i.e. for the first device only).
inline bool awq_lds_opt_in(const void* kernel, std::atomic<unsigned long long>& done, int bytes = 160 * 1024) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

int awq_launch_dequant(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out,
                       int64_t K, int64_t N, int64_t g, hipStream_t stream);
int awq_launch_unpack(const int32_t* q, uint8_t* out, int64_t rows, int64_t words, hipStream_t stream);

// Division by a run-time constant without an integer divide on the device: (x * magic) >> 32 == x / d for every x < limit.
// magic = floor(2^32 / d) + 1 is exact while x * (magic * d - 2^32) < 2^32.  False if d == 1 or the range is too large.
inline bool awq_magic_u32(uint32_t d, uint32_t limit, uint32_t* out) {
    if (d < 2) return false;
    const uint64_t m = ((uint64_t)1 << 32) / d + 1;
    const uint64_t e = m * d - ((uint64_t)1 << 32);  // 1 .. d
    if ((uint64_t)limit * e >= ((uint64_t)1 << 32)) return false;
    *out = (uint32_t)m;
    return true;
}

struct AwqGemmArgs {
    const uint16_t* x;       // [M, K] fp16
    const int32_t* qweight;  // [K, N/8]
    const uint16_t* scales;  // [K/g, N]
    const int32_t* qzeros;   // [K/g, N/8]
    const uint16_t* bias;    // [N] or null
    uint16_t* y;             // [M, N]
    int M, K, N, g;
    int x_gated = 0;  // decode kernel only: 1 = x is [M, 2K] = [gate | up], stage silu(gate) * up; 2 = stage rmsnorm(x + res_in)
    const uint16_t* res_in = nullptr;   // mode 2: optional residual rows [M, K]
    uint16_t* res_out = nullptr;        // mode 2: fp16(x + res_in) is written here (must not alias res_in)
    const uint16_t* norm_w = nullptr;   // mode 2: norm weight [K]
    float norm_eps = 0.f;
    const float* ssq_in = nullptr;      // mode 2: [M, ssq_in_tiles] row sums of squares from the producing call
    int ssq_in_tiles = 0;
    const uint16_t* add_res = nullptr;  // epilogue (decode kernel, M <= 4): y = fp16(fp16(x W + bias) + add_res)
    float* ssq_out = nullptr;           // epilogue: [M, ceil(N / 256)] sums of squares of y per 256-column tile
    int* counters;   // control words (word 0 = error flag), zero on entry
    float* exchange;  // in-launch split-K exchange region: all-ones sentinel on entry AND on exit
    size_t exchange_bytes;
    float* partial;  // scratch fp32 slabs for the two-pass reducers (no invariant between calls)
    size_t partial_floats;
    hipStream_t stream;
};

int awq_launch_gemm_naive(const AwqGemmArgs& a);
// Reference-order-numerics GEMV (M <= 4).  nlog: log2 of column-lanes per wave (2..4), splitk >= 1.
int awq_launch_gemv_valu(const AwqGemmArgs& a, int nlog, int splitk, bool nt);
int awq_gemv_valu_default_split(int K, int N, int nlog);
int awq_launch_splitk_reduce(const AwqGemmArgs& a, int splitk);
// MFMA decode GEMV / skinny GEMM, M <= 16.  wpl: packed words per lane (2|4); nwaves: waves per
// block (2|4|8); unit: 16-row sets per wave iteration (2|4|8); splitk: K slices.  0 = auto each.
bool awq_gemv_mfma_supports(int M, int K, int N, int g, int wpl);
int awq_launch_gemv_mfma(const AwqGemmArgs& a, int wpl, int nwaves, int unit, int splitk, bool two_pass);
// Fused dequant + LDS-tiled MFMA GEMM, any M (meant for M > 16).  bn: block tile width (128|256, 0 = auto).
bool awq_gemm_tiled_supports(int M, int K, int N, int g);
int awq_launch_gemm_tiled(const AwqGemmArgs& a, int bn, int splitk);
// Prefill GEMM with the weight operand decoded in registers (gemm_regb.hip).  bm: rows per block tile (128|256, 0 = auto).
bool awq_gemm_regb_supports(int M, int K, int N, int g);
int awq_launch_gemm_regb(const AwqGemmArgs& a, int bm);
// Batched-decode GEMM, 17 <= M <= 64, weights decoded in registers, no barrier in the K loop (gemm_skinny.hip).
bool awq_gemm_skinny_supports(int M, int K, int N, int g);
int awq_launch_gemm_skinny(const AwqGemmArgs& a, int splitk);
// GEMV layout (qweight [N, K/8], qzeros [N, ZW], scales [N, 8*ZW]): MFMA GEMV, M <= 16, and the
// bit-exact dequant to W^T [N, K].  nwaves (4|8|16) / unroll (4|8): 0 = auto.
bool awq_gemv_nk_supports(int M, int K, int N, int g);
size_t awq_gemv_nk_lds_bytes(int M, int K, int ZW, int nwaves);
int awq_launch_gemv_nk(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                       uint16_t* y, int M, int K, int N, int g, int ZW, int nwaves, int unroll, hipStream_t st);
// GEMV layout, row-streaming VALU kernel (gemv_rows.hip), M <= 4: waves per block (<= 8), super-units in flight per wave
// (1|2), blocks per CU, 1-KiB slots of a row per wave (1|2|3|4|6|8); 0 = auto each.
bool awq_gemv_rows_supports(int M, int K, int N, int g);
struct AwqRowsFx {  // decoder-block prologue / epilogue of the row-streaming kernel (batch 1): see awq_gemv_forward_ex
    const uint16_t* norm_w = nullptr;
    float norm_eps = 0.f;
    const uint16_t* res = nullptr;
    bool pairs = false;
    // grouped form (MoE decode): one virtual batch-1 call per (token, expert) pair over expert stacks [E, N, ...]; M must be 1,
    // x holds the activation rows (pair i reads row i / x_div), y [num_pairs, N (N / 2 with pairs)]
    const int32_t* pair_expert = nullptr;  // [num_pairs] on the device; a value outside [0, num_experts) skips the pair
    const float* pair_scale = nullptr;     // [num_pairs] routing weights folded into the epilogue, or null
    int num_pairs = 0, num_experts = 0, x_div = 1;
    int first_expert = 0;                  // pair_expert holds global ids; the stack holds experts [first_expert, first_expert + num_experts)
    int parts = 0;                         // blocks one expert matrix is dealt over (0 = auto)
};
int awq_launch_gemv_rows(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                         uint16_t* y, int M, int K, int N, int g, int ZW, int waves, int depth, int bpc, int sl, hipStream_t st,
                         const AwqRowsFx* fx = nullptr);
// GEMV layout, 2 <= M <= 16 while the activations fit LDS as MFMA fragments (M K <= 32768; M <= 8 at K = 4096): weights stream
// through LDS by DMA into v_mfma_f32_16x16x32_f16 (gemv_lds.hip).  ks: waves per tile (1|2|4), depth: pieces in flight; 0 = auto.
bool awq_gemv_lds_supports(int M, int K, int N, int g);
int awq_launch_gemv_lds(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* y,
                        int M, int K, int N, int g, int ZW, int ks, int depth, hipStream_t st);
// GEMV layout, 1 <= M <= 32 per launch, group_size 128: activations as MFMA A fragments in registers, the K range of a 16-row tile
// split over the waves of one block, weights by LDS-DMA (gemv_batch.hip).  form: activations through a wave-private LDS staging area
// (1) or by direct fragment loads (2); depth: pieces in flight per wave (1..3); 0 = auto.
bool awq_gemv_batch_supports(int M, int K, int N, int g);
int awq_launch_gemv_batch(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* y,
                          int M, int K, int N, int g, int ZW, int form, int depth, hipStream_t st);
// the same kernel on the GEMVFast layout's buffers (N % 16 == 0)
bool awq_gemv_batch_fast_supports(int M, int K, int N, int g);
int awq_launch_gemv_batch_fast(const uint16_t* x, const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros, uint16_t* y,
                               int M, int K, int N, int g, int group_rows, int depth, hipStream_t st);
// GEMV layout, prefill-sized batches: the register-decoded MFMA GEMM reading the layout's own buffers (gemm_regb.hip, NK form)
bool awq_gemm_regb_nk_supports(int M, int K, int N, int g, int ZW);
int awq_launch_gemm_regb_nk(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                            const uint16_t* bias, uint16_t* y, int M, int K, int N, int g, int ZW, int bm, hipStream_t st);
// GEMM-layout words + the GEMVFast format's scales / fp16 zero terms [>= K/g, N] (gemm_regb.hip, FZ form: W = fp16(w s + qzeros))
int awq_launch_gemm_regb_fz(const uint16_t* x, const int32_t* qweight_kn, const uint16_t* scales, const uint16_t* qzeros_f16, uint16_t* y,
                            int M, int K, int N, int g, int bm, hipStream_t st);
// MoE prefill: the same kernel over a token list sorted by expert (device-side row offsets), GEMM-layout expert stacks
int awq_launch_gemm_regb_grouped(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                 uint16_t* y, const int32_t* seg, int P, int E, int K, int N, int g, int bm, hipStream_t st,
                                 const int32_t* row_map = nullptr, int x_div = 1, const float* pair_w = nullptr, int gather = 0,
                                 int scatter = 0);
int awq_launch_moe_sort(const int* ids, int P, int E, int* order, int* seg, hipStream_t st);
int awq_launch_dequant_nk(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out, int K,
                          int N, int g, int ZW, hipStream_t st);
// Grouped (MoE) GEMM over stacked expert tensors (awq/modules/fused/moe.py:60-89), M = 16-row token blocks.
size_t awq_grouped_workspace_bytes_impl(int max_blocks, int K, int N);
int awq_launch_grouped_gemm(const AwqGemmArgs& a, const int* sorted_ids, const int* expert_ids, const int* num_post_pad,
                            const float* pair_weights, int num_pairs, int x_div, int max_blocks, int64_t expert_qw_words,
                            int64_t expert_z_words, int64_t expert_s_halfs);
// out[r, d] = silu(in[r, d]) * in[r, D + d]   (awq_ext.silu_and_mul, moe.py:73-76)
int awq_launch_silu_and_mul(const uint16_t* in, uint16_t* out, int64_t rows, int64_t D, hipStream_t st);
// GEMVFast layout (qweight int16 [N/4, K], scales / qzeros fp16 [8*ZW, N], qzeros = -(s*z)).
bool awq_gemv_fast_supports(int M, int K, int N, int g);
size_t awq_gemv_fast_lds_bytes(int M, int K, int g, int nwaves);
int awq_launch_gemv_fast(const uint16_t* x, const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros,
                         uint16_t* y, int M, int K, int N, int g, int GP, int nwaves, int unroll, hipStream_t st);
int awq_launch_dequant_fast(const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros, uint16_t* out, int K,
                            int N, int g, hipStream_t st);
// MoE routing (softmax + top-k + block alignment) in one launch; T tokens, E <= 64 experts, k <= 8.
int awq_launch_moe_route(const float* logits, float* topk_w, int* topk_ids, int* sorted_ids, int* expert_ids,
                         int* num_post_pad, int T, int E, int k, int renorm, int block, int first_expert, int num_local,
                         hipStream_t st);

// decoder.hip: RMSNorm (+ residual add when `residual` is non-null), RoPE + KV-cache append, single-query attention
int awq_launch_rmsnorm(const uint16_t* x, uint16_t* residual, const uint16_t* w, uint16_t* out, int64_t M, int64_t H,
                       float eps, hipStream_t st);
int awq_launch_rope_kv_append(const uint16_t* qkv, uint16_t* q_out, uint16_t* k_cache, uint16_t* v_cache, const float* cos_t,
                              const float* sin_t, const int32_t* pos_dev, int start_pos, int B, int S, int Hq, int Hkv, int D,
                              int rot, int Tmax, hipStream_t st);
size_t awq_decode_attention_workspace_bytes_impl(int B, int Hq, int max_splits);
int awq_launch_decode_attention(const uint16_t* q, uint16_t* k_cache, uint16_t* v_cache, uint16_t* out,
                                const int32_t* len_dev, int seq_len, int max_len, int B, int Hq, int Hkv, int D, int Tmax,
                                float scale, void* workspace, size_t workspace_bytes, const float* cos_t,
                                const float* sin_t, hipStream_t st, float softcap = 0.f, const float* alibi_slopes = nullptr);
