/* awq_engine.h -- C ABI of the experimental persistent decode engine (gemv_engine.hip in this directory; NOT part of libawq_hip.so).
 *
 * The reference runs the dependent projections of a decoder layer as one awq_ext.gemv_forward_cuda launch each
 * (awq/modules/linear/gemv.py:177-180 from awq/modules/fused/block.py:108-119, fused/mlp.py:46-62).  Here op i computes
 * y_i [N_i] = W_i x_i with x_0 = x and x_i = the first K_i elements of y_(i-1) (K_i <= N_(i-1)): one block per CU for the whole
 * chain, a loader wave streaming the CU's rows of every op through an LDS ring by LDS-DMA (it runs ahead across the op -> op
 * edges: the weights depend on nothing), consumer waves doing gemv_rows.hip's arithmetic on the ring slots, and every op's
 * output handed to all CUs as 8-byte {2 x fp16, epoch} granules.  group_size 128, K <= 12288, N even.
 *   awq_engine_describe   fills ONE 64-byte descriptor (host memory) for an op; the caller copies the table to the device.
 *                         `y` (plain fp16 [N], may be NULL) also receives the op's output; granule_offset = byte offset of the op's
 *                         granules (awq_engine_granule_bytes(N) bytes, 8-byte aligned, distinct per op) in the granule buffer.
 *   awq_engine_forward    ONE launch.  granules: ZEROED once by the caller; ctrl: awq_engine_ctrl_bytes() bytes, word 0 = 1
 *                         (the epoch), the rest 0, set once by the caller -- the launch re-arms itself (hipGraph-replayable).
 *                         ctrl word 2 != 0 afterwards: a bounded spin gave up (sticky); words 8-9: debug trace buffer pointer.
 *                         flags: bits 0-3 ring slots in flight (1..3, 0 = auto), bits 4-7 consumer waves (3 | 6, 0 = auto),
 *                         bits 8-11 ring slots (0 = as many as fit), bits 12-15 switch-off experiments, bit 16 no loader thinning. */
#pragma once
#include <stddef.h>
#include <stdint.h>

#define AWQ_ENGINE_OP_BYTES 64
#ifdef __cplusplus
extern "C" {
#endif
__attribute__((visibility("default"))) int awq_engine_describe(void* desc, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                                               uint16_t* y, int64_t K, int64_t N, int64_t group_size, int64_t zeros_width,
                                                               uint64_t granule_offset);
__attribute__((visibility("default"))) size_t awq_engine_granule_bytes(int64_t N);
__attribute__((visibility("default"))) size_t awq_engine_ctrl_bytes(void);
__attribute__((visibility("default"))) int awq_engine_forward(const uint16_t* x, const void* ops_dev, int64_t n_ops, int64_t max_N, int64_t max_K,
                                                              void* granules, void* ctrl, uint32_t flags, void* stream);
#ifdef __cplusplus
}
#endif
