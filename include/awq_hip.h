/*
 * awq_hip.h -- C ABI of libawq_hip.so, the MI355X (gfx950) AWQ int4 weight-only matmul library.
 *
 * This is the drop-in boundary.  Each entry point replaces one function of
 * the pybind11 extension modules `awq_ext` / `awq_v2_ext` that the reference calls but does not
 * ship (they live in the un-vendored `autoawq-kernels` package, reference setup.py:53).  The
 * reference call site each one serves is cited next to it (paths relative to /root/reference).
 *
 * Conventions
 *   - plain pointers and sizes only; fp16 travels as uint16_t bit patterns; no torch types.
 *   - every pointer is a DEVICE pointer owned by the caller, including outputs and workspaces;
 *     the library never allocates, frees or synchronises (hipGraph-capturable).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - return value: AWQ_OK (0) or a negative AWQ_ERR_* code; awq_hip_error_string() names it.
 *   - stateless and re-entrant; layout facts asserted by the reference modules are re-checked
 *     (N % 8 == 0, K % group == 0 -- awq/modules/linear/gemm.py:132-133).
 *   - workspaces: awq_gemm_workspace_bytes() gives the size, awq_gemm_workspace_init() prepares a
 *     fresh allocation ONCE (control words zero, split-K exchange region filled with the all-ones
 *     sentinel); every call restores that state.  One workspace serves one stream at a time.
 */
#ifndef AWQ_HIP_H
#define AWQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWQ_HIP_ABI_VERSION 1

#if defined(__GNUC__)
#define AWQ_EXPORT __attribute__((visibility("default")))
#else
#define AWQ_EXPORT
#endif

enum {
    AWQ_OK = 0,
    AWQ_ERR_BAD_SHAPE = -1,     /* N % 8, K % group, group % 8 ... violated */
    AWQ_ERR_BAD_ALIGNMENT = -2, /* a pointer is not 16-byte aligned */
    AWQ_ERR_UNSUPPORTED = -3,   /* valid AWQ tensor, but no kernel for it yet */
    AWQ_ERR_WORKSPACE = -4,     /* workspace too small / NULL where one is required */
    AWQ_ERR_LAUNCH = -5,        /* hipGetLastError() != hipSuccess after the launch */
    AWQ_ERR_NULL = -6           /* a required pointer is NULL */
};

#define AWQ_WS_COUNTER_BYTES 16384 /* control words at the head of a GEMM workspace (int32 word 0 = error flag) */

AWQ_EXPORT int awq_hip_abi_version(void);
AWQ_EXPORT const char* awq_hip_error_string(int code);
/* Name of the kernel variant the last awq_gemm_forward() call on this thread dispatched to
 * (diagnostics / tests; static string). */
AWQ_EXPORT const char* awq_hip_last_kernel(void);

/* ---- GEMM layout: qweight [K, N/8] i32, qzeros [K/g, N/8] i32, scales [K/g, N] f16 ------- */

/* Integer unpack only: out[r, 8c+j] = nibble of q[r, c] holding logical column 8c+j (0..15).
 * Reference: unpack_awq + reverse_awq_order + `& 0xF`, awq/utils/packing_utils.py:8-43,94-95. */
AWQ_EXPORT int awq_unpack_int4(const int32_t* q, uint8_t* out, int64_t rows, int64_t words, void* stream);

/* Replaces awq_ext.dequantize_weights_cuda(qweight, scales, qzeros, split_k_iters, thx, thy, dbg)
 * (awq/modules/linear/gemm.py:51-53 forward, :100-102 backward; tests/test_dequantization.py:41-49).
 * out [K, N] fp16 = (w - z) * s, bit-identical to dequantize_gemm (packing_utils.py:87-102). */
AWQ_EXPORT int awq_dequantize_weights(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                           uint16_t* out, int64_t K, int64_t N, int64_t group_size, void* stream);

/* Replaces awq_ext.gemm_forward_cuda(x2d, qweight, scales, qzeros, split_k_iters)
 * (awq/modules/linear/gemm.py:56-58; awq/modules/fused/mlp.py:41,49-62).
 * y [M, N] fp16 = x [M, K] fp16 @ dequant(qweight) (+ bias [N] fp16 if bias != NULL), fp32
 * accumulation, one rounding of the result.  Any M >= 0.  `workspace` may be NULL only if
 * awq_gemm_workspace_bytes() returned 0 for the shape.  `flags` = 0 selects the tuned kernel;
 * AWQ_GEMM_FLAG_* force a variant (tests / tuning sweeps). */
AWQ_EXPORT size_t awq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t group_size);
AWQ_EXPORT int awq_gemm_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
AWQ_EXPORT int awq_gemm_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales,
                     const int32_t* qzeros, const uint16_t* bias, uint16_t* y, int64_t M, int64_t K,
                     int64_t N, int64_t group_size, void* workspace, size_t workspace_bytes,
                     uint32_t flags, void* stream);

/* The kernels never hang: a split-K reducer that gives up waiting for its partial sums raises control word 0 of the
 * workspace and uses what is there.  awq_gemm_workspace_status copies that word to the host (synchronises `stream`);
 * non-zero = results since the last check are unreliable: call awq_gemm_workspace_init again. */
AWQ_EXPORT int awq_gemm_workspace_status(const void* workspace, void* stream, int32_t* err_out);
/* Which AWQ_GEMM_KERNEL_* the AUTO dispatch of awq_gemm_forward takes for this shape (-1: invalid shape).  Host code that
 * keeps its own alternative for large M (dequantise + vendor GEMM, awq/modules/linear/gemm.py:48-54) asks this to know
 * whether the fused prefill kernel (AWQ_GEMM_KERNEL_REGB) would run. */
AWQ_EXPORT int awq_gemm_auto_kernel(int64_t M, int64_t K, int64_t N, int64_t group_size);

/* flags for awq_gemm_forward: bits 0-3 kernel family, bits 4-7 lane geometry, bits 8-15 split-K */
#define AWQ_GEMM_KERNEL_AUTO 0u
#define AWQ_GEMM_KERNEL_NAIVE 1u     /* one thread per output, reference-order loop (checker / odd shapes) */
#define AWQ_GEMM_KERNEL_VALU 2u      /* wave64 VALU GEMV, reference-order fp16 dequant + fp32 FMA, M <= 4 */
#define AWQ_GEMM_KERNEL_MFMA_GEMV 3u /* MFMA 16x16x32 streaming GEMV / skinny GEMM, M <= 16 */
#define AWQ_GEMM_KERNEL_TILED 4u     /* LDS-tiled MFMA GEMM with fused dequant, large M */
#define AWQ_GEMM_KERNEL_REGB 5u      /* MFMA GEMM, weights decoded in registers, activations by LDS-DMA: prefill (NLOG: 1 = 128-, 2 = 256-row tile) */
#define AWQ_GEMM_KERNEL_SKINNY 6u    /* batched decode, 17 <= M <= 64: weights decoded in registers, activations of the K slice in LDS, no barrier in the K loop */
#define AWQ_GEMM_FLAG_KERNEL(f) ((f)&0xFu)
#define AWQ_GEMM_FLAG_NLOG(f) (((f) >> 4) & 0xFu)   /* 0 = auto; VALU: log2 column lanes (2..4); MFMA_GEMV: words per lane (2|4); TILED: 1 = 128-, 2 = 256-column tile */
#define AWQ_GEMM_FLAG_SPLITK(f) (((f) >> 8) & 0xFFu) /* 0 = auto */
#define AWQ_GEMM_FLAG_TWO_PASS (1u << 16) /* MFMA_GEMV: split-K reduce in a second kernel instead of in-launch */
#define AWQ_GEMM_FLAG_NO_NT (1u << 17)    /* plain (temporal) weight loads */
/* x is [M, 2K] = [gate | up] and the kernel multiplies silu(gate) * up (fp32, one rounding -- exactly
 * awq_silu_and_mul) while it stages the activations: the down projection of a fused MLP
 * (awq/modules/fused/mlp.py:46-70) without the separate elementwise launch.  M <= 16 only
 * (AWQ_ERR_UNSUPPORTED otherwise: run awq_silu_and_mul first). */
#define AWQ_GEMM_FLAG_X_GATED_SILU (1u << 18)
#define AWQ_GEMM_FLAG_UNIT(f) (((f) >> 20) & 0xFu)  /* MFMA_GEMV: 16-row sets per wave iteration (2|4|8), 0 = auto */
#define AWQ_GEMM_FLAG_WAVES(f) (((f) >> 24) & 0xFu) /* MFMA_GEMV: waves per block (2|4|8), 0 = auto */

/* ---- fused MLP / MoE (GEMM layout, experts stacked on a leading dim: awq/models/mixtral.py:130-158) */

/* Replaces awq_ext.silu_and_mul(out, gate_up) (awq/modules/fused/moe.py:73-76):
 * gate_up [rows, 2d] fp16 = [gate | up], out [rows, d] = silu(gate) * up. */
AWQ_EXPORT int awq_silu_and_mul(const uint16_t* gate_up, uint16_t* out, int64_t rows, int64_t d, void* stream);

/* Replaces awq_ext.topk_softmax + awq_ext.moe_alig_block_size (moe.py:94-171) in one launch:
 * gating_logits [T, E] fp32 -> topk_weights [T, k] fp32 (softmax, optionally renormalised),
 * topk_ids [T, k], sorted_token_ids [T*k + E*(block_rows-1)] (sentinel T*k), expert_ids [T*k + E],
 * num_tokens_post_padded [1].  E <= 64, k <= 8.  block_rows = 0: routing only -- the three alignment outputs are not written and
 * may be NULL (the row-streaming MoE decode path, awq_grouped_gemv_forward, runs pairs, not aligned blocks). */
AWQ_EXPORT int awq_moe_route(const float* gating_logits, float* topk_weights, int32_t* topk_ids,
                             int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_padded,
                             int64_t num_tokens, int64_t num_experts, int64_t topk, int renormalize,
                             int64_t block_rows, void* stream);

/* Expert-parallel form (new: the reference keeps every expert on one device, awq/models/mixtral.py:130-158): the routing
 * (topk_weights, topk_ids: GLOBAL expert ids) is computed over all num_experts, but only the pairs of experts
 * [first_expert, first_expert + num_local) are placed: sorted_token_ids [T*k + num_local*(block_rows-1)], expert_ids
 * [T*k + num_local] RELATIVE to first_expert, num_tokens_post_padded counts the local blocks only.  The rows of the other
 * pairs are never written by awq_grouped_gemm_forward: zero its output first (autoawq_amd/ep.py). */
AWQ_EXPORT int awq_moe_route_local(const float* gating_logits, float* topk_weights, int32_t* topk_ids,
                                   int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_padded,
                                   int64_t num_tokens, int64_t num_experts, int64_t topk, int renormalize,
                                   int64_t block_rows, int64_t first_expert, int64_t num_local, void* stream);

/* Replaces awq_ext.grouped_gemm_forward(x, qweight, scales, qzeros, topk_weights, sorted_token_ids,
 * expert_ids, num_tokens_post_padded, mul_weights, split_k_iters) (moe.py:60-89).
 * qweight [E, K, N/8], qzeros [E, K/g, N/8], scales [E, K/g, N]; sorted_token_ids / expert_ids /
 * num_tokens_post_padded as produced by moe_align_block_size with block `block_rows` = 16 (the
 * reference's value, moe.py:55) or 8 (decode: selector-row MFMA kernel), all on the device (no host
 * read: capturable).  Row i of y [num_pairs, N] (pair = token*topk + slot) =
 * x[i / x_div] @ W[expert of pair i], times pair_weights[i] if that pointer is non-NULL.
 * max_blocks = capacity of expert_ids in 16-row blocks; the workspace is prepared once with
 * awq_gemm_workspace_init like a GEMM workspace. */
AWQ_EXPORT size_t awq_grouped_gemm_workspace_bytes(int64_t max_blocks, int64_t K, int64_t N);
AWQ_EXPORT int awq_grouped_gemm_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales,
                                        const int32_t* qzeros, uint16_t* y, const int32_t* sorted_token_ids,
                                        const int32_t* expert_ids, const int32_t* num_tokens_post_padded,
                                        const float* pair_weights, int64_t num_pairs, int64_t x_div,
                                        int64_t block_rows, int64_t max_blocks, int64_t num_experts, int64_t K, int64_t N,
                                        int64_t group_size, void* workspace, size_t workspace_bytes, void* stream);
/* The same with flags: AWQ_GEMM_FLAG_X_GATED_SILU -- the rows of x are [gate | up] of 2 K halves (the output of the w1|w3
 * grouped GEMM) and silu(gate) * up is applied while a block stages its activations, bit-identical to awq_silu_and_mul: the
 * w2 grouped GEMM of apply_moe_weights (awq/modules/fused/moe.py:73-89) then needs no separate activation launch. */
AWQ_EXPORT int awq_grouped_gemm_forward_ex(const uint16_t* x, const int32_t* qweight, const uint16_t* scales,
                                        const int32_t* qzeros, uint16_t* y, const int32_t* sorted_token_ids,
                                        const int32_t* expert_ids, const int32_t* num_tokens_post_padded,
                                        const float* pair_weights, int64_t num_pairs, int64_t x_div,
                                        int64_t block_rows, int64_t max_blocks, int64_t num_experts, int64_t K, int64_t N,
                                        int64_t group_size, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* MoE PREFILL (round 4; awq/modules/fused/moe.py:45-91 at prefill-sized token counts): x [P, K] fp16 holds the (token, expert)
 * pairs' activation rows SORTED BY EXPERT, seg_offsets [E + 1] int32 ON THE DEVICE the row range of each expert
 * (seg[e] .. seg[e + 1]); qweight / scales / qzeros are the stacked GEMM-layout expert tensors [E, K, N/8] / [E, K/g, N] /
 * [E, K/g, N/8].  y [P, N] fp16 row r = x row r through ITS expert's matrix.  One launch of the register-decoded MFMA GEMM
 * (csrc/gemm_regb.hip) whose M tiles are dealt over the experts from the device-side offsets: nothing about the routing is
 * read back to the host, so the whole MoE block is hipGraph-capturable at every token count.  K % 64 == 0,
 * group_size % 64 == 0, N % 8 == 0.  flags: AWQ_GEMM_FLAG_NLOG = 2 -> 256-row tiles. */
AWQ_EXPORT int awq_grouped_gemm_prefill(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                        uint16_t* y, const int32_t* seg_offsets, int64_t P, int64_t num_experts, int64_t K,
                                        int64_t N, int64_t group_size, uint32_t flags, void* stream);

/* The same launch with the sort kept as an INDEX LIST (round 6: no gathered copy of x, no scatter pass over y, one rounding):
 *   row_map [P] int32 on the device = pair index (token * topk + slot) of sorted row r, as awq_moe_sort_pairs writes it;
 *   AWQ_GROUPED_PREFILL_GATHER_X   x is NOT sorted: sorted row r reads x row row_map[r] / x_div (x_div = topk: x = the tokens);
 *   AWQ_GROUPED_PREFILL_SCATTER_Y  sorted row r is written to y row row_map[r] (y in pair order [T * topk, N]);
 *   pair_weights != NULL           y row = fp16(fp32 product * pair_weights[row_map[r]]) (mul_routed_weight, moe.py:84-88).
 * Rows of y that no sorted row maps to (pairs of foreign experts in an expert-parallel shard) are not written. */
#define AWQ_GROUPED_PREFILL_GATHER_X (1u << 19)
#define AWQ_GROUPED_PREFILL_SCATTER_Y (1u << 20)
AWQ_EXPORT int awq_grouped_gemm_prefill_ex(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                           uint16_t* y, const int32_t* seg_offsets, const int32_t* row_map, const float* pair_weights,
                                           int64_t P, int64_t x_div, int64_t num_experts, int64_t K, int64_t N, int64_t group_size,
                                           uint32_t flags, void* stream);
/* The (token, expert) pairs sorted by expert as an index list, one launch: order [num_pairs] = pair indices grouped by expert,
 * pair order kept inside an expert (== torch.argsort(topk_ids.flatten(), stable=True)), seg_offsets [E + 1] = each expert's row
 * range; ids outside [0, E) are not placed.  Replaces the argsort / scatter_add / cumsum glue of the prefill-sized path of
 * apply_moe_weights (moe.py:45-91).  E <= 64. */
AWQ_EXPORT int awq_moe_sort_pairs(const int32_t* topk_ids, int32_t* order, int32_t* seg_offsets, int64_t num_pairs,
                                  int64_t num_experts, void* stream);

/* ---- GEMV layout: qweight [N, K/8] i32 (ordinal nibbles), qzeros [N, ZW] i32, scales [N, 8*ZW] f16
 *      (awq/modules/linear/gemv.py:45-69; ZW = calculate_zeros_width, gemv.py:12-24) -------------- */

/* Replaces awq_ext.gemv_forward_cuda(x, qweight, scales, qzeros, group_size) (M <= 8) and
 * awq_ext.gemmv2_forward_cuda(..., group_size, split_k_iters) (awq/modules/linear/gemv.py:168-180).
 * y [M, N] fp16 = x [M, K] fp16 @ dequant(qweight)^T, fp32 accumulation; no bias (the reference adds it afterwards,
 * gemv.py:185).  AUTO: from M = 5 the batched kernel AWQ_GEMV_KERNEL_BATCH (any M in one call) wherever it takes the shape
 * (group_size 128); the older decode kernels below serve 1 <= M <= 16 per call with awq_gemv_lds_bytes(M, K, ZW) <= 160 KiB
 * (the host wrapper chunks).  M <= 2 (and M <= 4 for K <= 6144) takes the row-streaming kernel
 * (gemv_rows.hip: a wave instruction reads 1 KiB of one row, activations in registers, no cross-CU exchange; needs
 * group_size % 128 == 0 and K <= 65536); 5 <= M with N >= 8192 and M K <= 32768 the LDS-streaming MFMA kernel (gemv_lds.hip:
 * weights by LDS-DMA, 1 KiB of two rows per instruction, into v_mfma_f32_16x16x32_f16); everything else up to M = 16 the 16-row
 * MFMA tile kernel (gemv_nk.hip).  flags: AWQ_GEMM_FLAG_KERNEL = AWQ_GEMV_KERNEL_*; tuning: _WAVES (waves per
 * block), _UNIT (TILE16: unroll; ROWS: super-units in flight per wave, 1 | 2), _SPLITK (ROWS: blocks per CU), _NLOG (ROWS:
 * 1-KiB slots of a row per wave, 1 | 2 | 3 | 4 | 6 | 8). */
#define AWQ_GEMV_KERNEL_AUTO 0u
#define AWQ_GEMV_KERNEL_TILE16 1u /* 16 rows per block through v_mfma_f32_16x16x32_f16, M <= 16 */
#define AWQ_GEMV_KERNEL_ROWS 2u   /* row-streaming kernel (1 KiB of one row per wave instruction, activations in registers), M <= 4 */
#define AWQ_GEMV_KERNEL_LDS 3u    /* weights through LDS by DMA into MFMA 16x16x32, 2 <= M <= 16 with M K <= 32768; _SPLITK: waves per tile */
#define AWQ_GEMV_KERNEL_BATCH 5u  /* round 5: ANY M in one call (round 6: launches of <= 128 rows), group_size 128, K % 128 == 0: activations
                                     as MFMA A fragments in registers, a tile's K range split over the eight waves of ONE block (no exchange,
                                     no workspace), weights by LDS-DMA in row-contiguous pieces (gemv_batch.hip).  AUTO takes it from M = 5.
                                     Above 32 rows a launch works in ROW PARTS of <= 32 rows, each part in its own block, the blocks of a
                                     tile list residents of one XCD (the matrix leaves HBM once).
                                     _UNIT: how the activations reach the registers (1 = coalesced LDS-DMA into a wave-private staging
                                     area, 2 = direct 16-byte fragment loads; 0 = auto: staged when it fits), _SPLITK: ring slots per wave
                                     (1 .. 3), _WAVES: the row parts (0 = auto, 1 = as wave groups inside one block -- round 6's first
                                     form --, 2 .. 4 = that many parts across blocks; refused where a part would exceed 32 rows) */
#define AWQ_GEMV_KERNEL_PREFILL 4u /* any M in ONE call: the register-decoded MFMA GEMM on this layout's own buffers (gemm_regb.hip, NK
                                     form; K % 64 == 0, group_size % 64 == 0, N % 4 == 0); _NLOG = 2: 256-row tiles.  EXPLICIT only:
                                     measured 0.29 of the MFMA peak at M = 16384 (dequantise + dense GEMM: 0.42) and latency-bound
                                     below ~2000 rows (190 us at M = 32 for 4096 x 11008), so AUTO does not take it */
/* Which AWQ_GEMV_KERNEL_* the AUTO dispatch of awq_gemv_forward takes for this shape (host only; -1: none takes it).  A pure
 * function of its arguments -- what awq_hip_last_kernel (a per-thread diagnostic) reports after the call. */
AWQ_EXPORT int awq_gemv_auto_kernel(int64_t M, int64_t K, int64_t N, int64_t group_size);
AWQ_EXPORT int awq_gemv_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales,
                                const int32_t* qzeros, uint16_t* y, int64_t M, int64_t K, int64_t N,
                                int64_t group_size, int64_t zeros_width, uint32_t flags, void* stream);
AWQ_EXPORT size_t awq_gemv_lds_bytes(int64_t M, int64_t K, int64_t zeros_width);
/* out [N, K] fp16 = dequantised W^T, bit-identical to the GEMM-layout dequant of the same
 * integers transposed (used for M > 16: dequant + fp16 GEMM, and by the tests). */
AWQ_EXPORT int awq_dequantize_weights_gemv(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                           uint16_t* out, int64_t K, int64_t N, int64_t group_size,
                                           int64_t zeros_width, void* stream);

/* GEMV-layout buffers -> GEMM-layout buffers of the same integers and scales (csrc/repack.hip): qweight_out [K, N/8] i32 (AWQ nibble
 * order, awq/modules/linear/gemm.py:220-249), scales_out [K/g, N] f16, qzeros_out [K/g, N/8] i32 -- caller-owned temporaries.  The
 * prefill route of WQLinear_GEMV (the reference runs awq_ext.gemmv2_forward_cuda there, gemv.py:168-176): repack into a temporary of
 * the call, then awq_gemm_forward's fused MFMA kernel on it.  Bit-exact nibble moves; N % 8 == 0. */
AWQ_EXPORT int awq_repack_gemv_to_gemm(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, int32_t* qweight_out,
                                       uint16_t* scales_out, int32_t* qzeros_out, int64_t K, int64_t N, int64_t group_size,
                                       int64_t zeros_width, void* stream);

/* ---- GEMVFast layout: qweight [N/4, K] i16 (4-row interleave, awq/modules/linear/gemv_fast.py:26-65),
 *      scales [8*ZW, N] f16, qzeros [8*ZW, N] f16 = -(scale*zero) (gemv_fast.py:86-118,175-181) ------- */

/* Replaces awq_v2_ext.gemv_forward_cuda_decode(x, qweight, scales, qzeros, m, n, k, group_size) and, for
 * small M, awq_v2_ext.gemm_forward_cuda_prefill (gemv_fast.py:185-208).  y [M, N] = x [M, K] @ W^T with
 * W = w*s + qzeros; N % 16 == 0, K % 128 == 0.  group_rows = rows of scales / qzeros (8*ZW).  AUTO (AWQ_GEMM_FLAG_KERNEL 0): from
 * M = 5 the batched kernel (gemv_batch.hip in its GEMVFast form, group_size 128: ANY M in one call, launches of <= 128 rows;
 * also AWQ_GEMV_KERNEL_BATCH explicitly, _SPLITK = ring slots 1 | 2, _WAVES = the row parts as above); below that, or with AWQ_GEMM_FLAG_KERNEL = 1, the 16-row kernel
 * (gemv_fast.hip: 1 <= M <= 16 per call, the host wrapper chunks). */
AWQ_EXPORT int awq_gemv_fast_forward(const uint16_t* x, const int16_t* qweight, const uint16_t* scales,
                                     const uint16_t* qzeros, uint16_t* y, int64_t M, int64_t K, int64_t N,
                                     int64_t group_size, int64_t group_rows, uint32_t flags, void* stream);
AWQ_EXPORT size_t awq_gemv_fast_lds_bytes_c(int64_t M, int64_t K, int64_t group_size);
/* Replaces awq_v2_ext.gemm_forward_cuda_prefill(x, qweight, scales, qzeros) for prefill-sized token counts
 * (awq/modules/linear/gemv_fast.py:203-206): y [M, N] = x [M, K] @ W^T, W = fp16(w*s + qzeros).  Two hand-written launches, no
 * vendor GEMM: awq_repack_gemvfast_to_gemm transposes the packed words into `qweight_tmp` (caller-owned, K*N/2 bytes, 16-byte
 * aligned, a temporary of the call), then the register-decoded MFMA GEMM (gemm_regb.hip) runs on it with this format's own scales
 * and fp16 zero terms.  K % 64 == 0, group_size % 64 == 0, N % 8 == 0; flags: AWQ_GEMM_FLAG_NLOG = 2 -> 256-row tiles. */
AWQ_EXPORT int awq_gemv_fast_prefill(const uint16_t* x, const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros,
                                     uint16_t* y, int32_t* qweight_tmp, int64_t M, int64_t K, int64_t N, int64_t group_size,
                                     int64_t group_rows, uint32_t flags, void* stream);
/* GEMVFast words int16 [N/4, K] -> GEMM-layout words int32 [K, N/8] (AWQ nibble order) of the same integers; bit-exact. */
AWQ_EXPORT int awq_repack_gemvfast_to_gemm(const int16_t* qweight, int32_t* qweight_out, int64_t K, int64_t N, void* stream);
/* out [N, K] fp16 = dequantised W^T (W = fp16 of the fp32 fma w*s + qzeros). */
AWQ_EXPORT int awq_dequantize_weights_gemv_fast(const int16_t* qweight, const uint16_t* scales,
                                                const uint16_t* qzeros, uint16_t* out, int64_t K, int64_t N,
                                                int64_t group_size, void* stream);

/* ---- fused decoder block around the Linears ------------------------------- */

/* Replaces awq_ext.layernorm_forward_cuda(x, weight, out, eps) (awq/modules/fused/norm.py:33-36):
 * out [M, H] = fp16( x * rsqrt(mean(x^2) + eps) * weight ), fp32 arithmetic.  H % 8 == 0, H <= 16384.
 * If `residual` is non-NULL it is first updated in place, residual = fp16(residual + x), and the
 * updated row is what gets normalised (`h = hidden_states + attn_output` followed by the next norm,
 * awq/modules/fused/block.py:108-119, in one launch). */
AWQ_EXPORT int awq_rmsnorm_forward(const uint16_t* x, uint16_t* residual, const uint16_t* weight, uint16_t* out,
                                   int64_t M, int64_t H, float eps, void* stream);

/* RoPE.forward (awq/modules/fused/attn.py:54-87) + WindowedCache.update_kv (cache.py:40-45) on the
 * fused qkv output: qkv [B*S, (n_heads + 2*n_kv_heads) * head_dim] fp16 (q heads | k heads | v heads).
 * Rotates pairs (i, i + rotary_dim/2) of every q and k head by the angle of position start_pos + s
 * (cos / sin [max_pos, rotary_dim/2] fp32, built like RoPE.precompute_freqs_cis), writes q to
 * q_out [B*S, n_heads, head_dim] and k / v into k_cache / v_cache [B, max_seq, n_kv_heads, head_dim]
 * at row start_pos + s.  If pos_dev is non-NULL the start position is read from it on the device
 * (one hipGraph can then be replayed for every decode step). */
AWQ_EXPORT int awq_rope_kv_append(const uint16_t* qkv, uint16_t* q_out, uint16_t* k_cache, uint16_t* v_cache,
                                  const float* cos_table, const float* sin_table, const int32_t* pos_dev,
                                  int64_t start_pos, int64_t B, int64_t S, int64_t n_heads, int64_t n_kv_heads,
                                  int64_t head_dim, int64_t rotary_dim, int64_t max_seq, void* stream);

/* Replaces flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens, causal=True) for ONE query
 * token per sequence (attn.py:286-302): out [B, n_heads, 128] = softmax(q k^T * scale) v over cache
 * rows [0, seq_len).  head_dim = 128, n_heads / n_kv_heads in {1, 2, 4, 8}.  If len_dev is non-NULL
 * the length is read on the device and max_len (>= the length, <= max_seq) sizes the launch.
 * workspace: awq_decode_attention_workspace_bytes(B, n_heads) bytes of scratch (split-sequence
 * partials), no state between calls. */
AWQ_EXPORT size_t awq_decode_attention_workspace_bytes(int64_t B, int64_t n_heads);
AWQ_EXPORT int awq_decode_attention(const uint16_t* q, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* out,
                                    const int32_t* len_dev, int64_t seq_len, int64_t max_len, int64_t B,
                                    int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t max_seq, float scale,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* awq_decode_attention with the two score modifiers flash_attn_with_kvcache takes at attn.py:286-302: softcap > 0 applies
 * s := softcap * tanh(s / softcap) to the scaled scores (`softcap=self.attn_logit_softcapping`), alibi_slopes != NULL
 * ([n_heads] fp32, ALiBi.slopes of attn.py:89-125) then adds slope_h * (t - (len - 1)) for key row t.  0 / NULL = plain. */
AWQ_EXPORT int awq_decode_attention_ex(const uint16_t* q, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* out,
                                       const int32_t* len_dev, int64_t seq_len, int64_t max_len, int64_t B,
                                       int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t max_seq, float scale,
                                       float softcap, const float* alibi_slopes, void* workspace, size_t workspace_bytes,
                                       void* stream);

/* Replaces flash_attn_func(xq, keys, values, causal=True, alibi_slopes=..., softcap=...) -- the PREFILL step of the fused attention
 * (awq/modules/fused/attn.py:269-277): out [B, S, n_heads, 128] fp16 = causal attention of the S new query rows q [B, S, n_heads, 128]
 * (after RoPE) over the caches [>= B, max_seq, n_kv_heads, 128], which already hold rows 0 .. start_pos + S - 1 of every sequence
 * (awq_rope_kv_append appended them); query row s sees cache rows <= start_pos + s (start_pos > 0: chunked prefill).  softcap /
 * alibi_slopes as in awq_decode_attention_ex (the bias of key row t for query position p is slope_h * (t - p)).  Flash style: no
 * workspace, fp32 online softmax, MFMA 16x16x32 (csrc/prefill_attn.hip).  AWQ_ERR_UNSUPPORTED: head_dim != 128, n_heads % n_kv_heads,
 * tensors beyond 32-bit byte offsets per sequence. */
AWQ_EXPORT int awq_prefill_attention(const uint16_t* q, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* out, int64_t B,
                                     int64_t S, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t max_seq,
                                     int64_t start_pos, float scale, float softcap, const float* alibi_slopes, void* stream);

/* awq_rmsnorm_forward folded into the projection that follows it (decode, M <= 4, GEMM layout):
 * y = rmsnorm(x (+ residual_in)) * norm_weight @ W.  Every block of the decode kernel recomputes the
 * row statistic from the L2-resident row while its first weight requests are in flight and normalises
 * its K slice as it stages it; if residual_in is given, fp16(x + residual_in) -- the residual stream --
 * is written to residual_out (a DIFFERENT buffer: all blocks read residual_in, one writes).
 * Same formulas and roundings as awq_rmsnorm_forward + awq_gemm_forward; the fp32 sum of squares is
 * reduced in a different order.  AWQ_ERR_UNSUPPORTED for M > 4 or shapes the decode kernel does not take. */
AWQ_EXPORT int awq_gemm_forward_normed(const uint16_t* x, const uint16_t* residual_in, uint16_t* residual_out,
                                       const uint16_t* norm_weight, float eps, const int32_t* qweight,
                                       const uint16_t* scales, const int32_t* qzeros, const uint16_t* bias, uint16_t* y,
                                       int64_t M, int64_t K, int64_t N, int64_t group_size, void* workspace,
                                       size_t workspace_bytes, uint32_t flags, void* stream);

/* awq_gemm_forward with the prologue / epilogue a decoder block wants around a decode-sized projection
 * (M <= 4, GEMM layout, shapes the decode kernel takes; AWQ_ERR_UNSUPPORTED otherwise).  Unused fields
 * are 0 / NULL.  Prologue: norm_weight != NULL normalises x while staging (RMSNorm, fp32, the
 * roundings of awq_rmsnorm_forward); the row statistic comes either from ssq_in -- [M, ssq_in_tiles]
 * partial sums of squares written by the PRODUCING call's ssq_out, no statistic pass at all -- or, if
 * ssq_in is NULL, from a pass over x (+ residual_in, written to residual_out: the
 * awq_gemm_forward_normed form).  Epilogue: add_residual != NULL stores
 * y = fp16(fp16(x W + bias) + add_residual) -- the two roundings of `h + proj(x)` in torch;
 * ssq_out != NULL receives [M, awq_gemm_ex_ssq_tiles(N)] sums of squares of the stored y values, one
 * per 256-column tile (fixed summation order).  flags as for awq_gemm_forward. */
typedef struct AwqGemmEx {
    uint32_t struct_bytes; /* sizeof(AwqGemmEx) */
    uint32_t flags;
    const uint16_t* x;
    const int32_t* qweight;
    const uint16_t* scales;
    const int32_t* qzeros;
    const uint16_t* bias;
    uint16_t* y;
    int64_t M, K, N, group_size;
    void* workspace;
    size_t workspace_bytes;
    void* stream;
    const uint16_t* norm_weight;
    float norm_eps;
    const uint16_t* residual_in;
    uint16_t* residual_out;
    const float* ssq_in;
    int64_t ssq_in_tiles;
    const uint16_t* add_residual;
    float* ssq_out;
} AwqGemmEx;
AWQ_EXPORT int64_t awq_gemm_ex_ssq_tiles(int64_t N);
AWQ_EXPORT int awq_gemm_forward_ex(const AwqGemmEx* args);

/* The GEMV-layout decode projection (awq_gemv_forward, batch 1) with the decoder block's prologue / epilogue in the same
 * launch -- what awq/modules/fused/block.py:108-120 runs as norm -> projection -> add / silu * mul launches
 * (awq_ext.layernorm_forward_cuda norm.py:33-36, awq_ext.gemv_forward_cuda gemv.py:178-180, mlp.py:64-66):
 *   norm_weight != NULL   x is RMS-normalised while it is brought into registers: fp16(x * rsqrt(mean(x^2) + eps) * w), the
 *                         arithmetic of awq_rmsnorm_forward (the row statistic is summed in a different order);
 *   add_residual != NULL  y = fp16(fp16(W x) + add_residual)  [N]  -- the two roundings of `h + proj(x)` in torch;
 *   AWQ_GEMV_EX_SILU_PAIRS (flags)  rows (2 i, 2 i + 1) of the matrix are (gate_i, up_i) (the caller interleaved the gate and
 *                         up projections' rows); y [N / 2] = fp16(silu(fp16 gate) * fp16 up), == awq_silu_and_mul on the
 *                         unfused outputs.  Not together with add_residual.
 * Served by the row-streaming kernel only: M == 1 and K <= 16384 (a wave covers whole rows; K <= 12288 with norm_weight),
 * group_size % 128 == 0, and with
 * add_residual at most 64 rows per wave (N <= 65536); AWQ_ERR_UNSUPPORTED otherwise -- the caller then runs the separate
 * launches.  Unused fields are 0 / NULL. */
#define AWQ_GEMV_EX_SILU_PAIRS 1u
typedef struct AwqGemvEx {
    uint32_t struct_bytes; /* sizeof(AwqGemvEx) */
    uint32_t flags;
    const uint16_t* x;
    const int32_t* qweight;
    const uint16_t* scales;
    const int32_t* qzeros;
    uint16_t* y;
    int64_t M, K, N, group_size, zeros_width;
    void* stream;
    const uint16_t* norm_weight;
    float norm_eps;
    const uint16_t* add_residual;
} AwqGemvEx;
AWQ_EXPORT int awq_gemv_forward_ex(const AwqGemvEx* args);

/* MoE DECODE on GEMV-layout expert stacks (round 6; the decode-sized calls of apply_moe_weights, awq/modules/fused/moe.py:60-89:
 * awq_ext.grouped_gemm_forward on the stacked experts of awq/models/mixtral.py:130-158).  The reference keeps the experts in the
 * GEMM layout only; this entry point reads the SAME weights repacked to the GEMV layout and stacked on a leading dim --
 * qweight [E, N, K/8] int32, qzeros [E, N, ZW] int32, scales [E, N, 8 ZW] fp16 -- and runs every (token, expert) pair as one
 * batch-1 call of the row-streaming kernel (csrc/gemv_rows.hip), all pairs in ONE launch:
 *   y[i] = x[i / x_div] @ W[pair_experts[i]]^T            i = 0 .. num_pairs - 1, pair i = token * topk + slot
 * pair_experts [num_pairs] int32 ON THE DEVICE (= topk_ids flattened; nothing is read back: capturable); the stack holds the experts
 * [first_expert, first_expert + num_experts) of those GLOBAL ids (first_expert = 0: all of them) and a pair of any other expert
 * leaves row i of y untouched (expert-parallel shards: zero y first).  x_div = topk for the w1|w3 call (x = the tokens),
 * 1 for the w2 call (x = one row per pair).
 *   pair_weights != NULL              y[i] = fp16(fp32 product * pair_weights[i]) (mul_routed_weight, moe.py:84-88)
 *   AWQ_GEMV_EX_SILU_PAIRS (flags)    rows (2 j, 2 j + 1) of every expert are (gate_j, up_j); y [num_pairs, N / 2] =
 *                                     fp16(silu(fp16 gate) * fp16 up) == awq_silu_and_mul (moe.py:73-76).  Not with pair_weights.
 * The blocks that stream one part of the matrices of ALL pairs are residents of one XCD: pairs that share an expert meet in its
 * L2, so HBM traffic follows the DISTINCT experts.  parts: blocks one expert matrix is dealt over (0 = auto).
 * AWQ_ERR_UNSUPPORTED: K > 16384, group_size % 128, num_pairs > 8191 -- the caller keeps the GEMM-layout grouped kernel. */
AWQ_EXPORT int awq_grouped_gemv_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                        uint16_t* y, const int32_t* pair_experts, const float* pair_weights, int64_t num_pairs,
                                        int64_t x_div, int64_t num_experts, int64_t first_expert, int64_t K, int64_t N,
                                        int64_t group_size, int64_t zeros_width, uint32_t flags, int64_t parts, void* stream);

/* awq_rope_kv_append + awq_decode_attention in ONE launch for a decode step (S = 1, full rotary,
 * head_dim = 128): qkv [B, (n_heads + 2*n_kv_heads) * 128] is the fused projection's output; query
 * heads are rotated in registers, the new token's rotated k and its v are used from registers and
 * appended to the caches at row start_pos (read from pos_dev when non-NULL; max_len then sizes
 * the launch, >= position + 1).  Cache rows bit-identical to awq_rope_kv_append; output equal to the
 * two separate calls up to fp32 contraction order. */
AWQ_EXPORT int awq_decode_attention_rope(const uint16_t* qkv, uint16_t* k_cache, uint16_t* v_cache,
                                         const float* cos_table, const float* sin_table, uint16_t* out,
                                         const int32_t* pos_dev, int64_t start_pos, int64_t max_len, int64_t B,
                                         int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t max_seq,
                                         float scale, void* workspace, size_t workspace_bytes, void* stream);

/* ---- one-shot small-message all-reduce for tensor-parallel decode (no reference counterpart: SURVEY.md 2.3 / 8e) ------
 * out[i] = sum over ranks of in_r[i], fp16 with fp32 accumulation in rank order (bitwise identical on every rank), for the
 * [M, hidden] outputs of row-parallel projections (n_halfs % 4 == 0, n_halfs <= max_halfs; <= 64 KiB is what it is built for).
 * One launch per rank, no host involvement, hipGraph-replayable.  Protocol (csrc/allreduce.hip; round 5): a PUSH -- every rank
 * stores its slice as 8-byte {two fp16 values, epoch tag} granules straight into every peer's staging (one xGMI store hop, no
 * flag, no fence), then sums what arrived in its OWN staging on tag match; no remote read.  peer_flags is unused since round 5
 * (the argument stays in the ABI).
 * Setup (once): every rank allocates awq_allreduce_staging_bytes(max_halfs) of staging (2 parities x AWQ_AR_MAX_RANKS sources x
 * max_halfs / 2 granules) and awq_allreduce_flag_bytes() of flags -- both ZEROED, both mapped by every peer (hipIpc / P2P), both
 * FINE-GRAINED / UNCACHED device memory when the peers are other GPUs (awq_allreduce_alloc below: a kernel that polls memory a
 * peer GPU writes is only guaranteed to see the store in such memory) -- and awq_allreduce_state_bytes() of private, zeroed state.
 * peer_staging[r] / peer_flags[r] are THIS process's addresses of rank r's buffers (entry `rank` = its own); every rank must
 * pass the same max_halfs and issue the same sequence of calls.  A peer that never arrives raises a sticky error word
 * (state[1] != 0) after a bounded spin instead of hanging the GPU, and the slices that gave up are written as NaN: a late
 * rank can not produce a plausible-looking wrong sum. */
#define AWQ_AR_MAX_RANKS 8
#define AWQ_AR_BLOCKS 16
#define AWQ_AR_IPC_HANDLE_BYTES 64
AWQ_EXPORT size_t awq_allreduce_staging_bytes(int64_t max_halfs);
AWQ_EXPORT size_t awq_allreduce_flag_bytes(void);
AWQ_EXPORT size_t awq_allreduce_state_bytes(void);
AWQ_EXPORT int awq_allreduce_oneshot(const void* const* peer_staging, void* const* peer_flags, int64_t rank, int64_t world,
                                     const uint16_t* in, uint16_t* out, int64_t n_halfs, int64_t max_halfs, void* state,
                                     void* stream);
/* All `world` ranks of a SINGLE-PROCESS group (every buffer on one device) in ONE launch: ins / outs / states are arrays of
 * `world` pointers.  Same kernel, same protocol; the ranks are guaranteed co-resident (streams of one process may share a
 * hardware queue and then serialise).  Used by the single-GPU tests of the protocol and by single-process multi-"rank" setups. */
AWQ_EXPORT int awq_allreduce_oneshot_group(const void* const* peer_staging, void* const* peer_flags, int64_t world,
                                           const uint16_t* const* ins, uint16_t* const* outs, int64_t n_halfs,
                                           int64_t max_halfs, void* const* states, void* stream);
/* Setup-time helpers (the ONLY entry points that allocate; nothing on the launch path does): uncached fine-grained device
 * memory on the current device, zeroed and synchronised; its IPC handle (AWQ_AR_IPC_HANDLE_BYTES opaque bytes a peer process
 * opens with awq_allreduce_ipc_open -- it then owns a mapping it closes with awq_allreduce_ipc_close). */
AWQ_EXPORT int awq_allreduce_alloc(void** ptr, size_t bytes);
AWQ_EXPORT int awq_allreduce_free(void* ptr);
AWQ_EXPORT int awq_allreduce_ipc_export(void* ptr, void* handle64);
AWQ_EXPORT int awq_allreduce_ipc_open(const void* handle64, void** ptr);
AWQ_EXPORT int awq_allreduce_ipc_close(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* AWQ_HIP_H */
