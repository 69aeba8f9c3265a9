// capi.hip -- the extern "C" surface of libawq_hip.so (declared in include/awq_hip.h).
// Validation + kernel selection only; kernels live in the other translation units.
#include "awq_internal.h"

namespace {
thread_local const char* g_last_kernel = "none";

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_gemm_layout(int64_t K, int64_t N, int64_t g) {
    // awq/modules/linear/gemm.py:132-133
    if (K < 0 || N < 0 || g <= 0) return AWQ_ERR_BAD_SHAPE;
    if (N % 8 != 0) return AWQ_ERR_BAD_SHAPE;
    if (K % g != 0) return AWQ_ERR_BAD_SHAPE;
    if (K > INT32_MAX || N > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    return AWQ_OK;
}
}  // namespace

extern "C" {

int awq_hip_abi_version(void) { return AWQ_HIP_ABI_VERSION; }

const char* awq_hip_error_string(int code) {
    switch (code) {
        case AWQ_OK: return "ok";
        case AWQ_ERR_BAD_SHAPE: return "bad shape (need N % 8 == 0, K % group_size == 0)";
        case AWQ_ERR_BAD_ALIGNMENT: return "pointer not 16-byte aligned";
        case AWQ_ERR_UNSUPPORTED: return "no kernel for this shape/variant";
        case AWQ_ERR_WORKSPACE: return "workspace missing or too small";
        case AWQ_ERR_LAUNCH: return "HIP launch error";
        case AWQ_ERR_NULL: return "required pointer is NULL";
        default: return "unknown error";
    }
}

const char* awq_hip_last_kernel(void) { return g_last_kernel; }

int awq_unpack_int4(const int32_t* q, uint8_t* out, int64_t rows, int64_t words, void* stream) {
    if (rows < 0 || words < 0) return AWQ_ERR_BAD_SHAPE;
    if (rows * words == 0) return AWQ_OK;
    if (!q || !out) return AWQ_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(out) & 7u) || (reinterpret_cast<uintptr_t>(q) & 3u)) return AWQ_ERR_BAD_ALIGNMENT;
    return awq_launch_unpack(q, out, rows, words, static_cast<hipStream_t>(stream));
}

int awq_dequantize_weights(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out,
                           int64_t K, int64_t N, int64_t group_size, void* stream) {
    int rc = check_gemm_layout(K, N, group_size);
    if (rc) return rc;
    if (K * N == 0) return AWQ_OK;
    if (!qweight || !scales || !qzeros || !out) return AWQ_ERR_NULL;
    if (!aligned16(scales) || !aligned16(out)) return AWQ_ERR_BAD_ALIGNMENT;
    return awq_launch_dequant(qweight, scales, qzeros, out, K, N, group_size, static_cast<hipStream_t>(stream));
}

namespace {
// workspace = [control AWQ_WS_COUNTER_BYTES][exchange region E][scratch region E]
size_t region_bytes(int64_t M, int64_t N) {
    const int64_t m = M < 16 ? M : 16;
    size_t e = (size_t)(64 * m * (N + 512)) * 4;  // up to 64 slabs of [min(M,16), N rounded up to a tile]
    if (e > ((size_t)32 << 20)) e = (size_t)32 << 20;  // the launchers lower the split to what fits
    return (e + 255) & ~(size_t)255;
}
}  // namespace

size_t awq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t group_size) {
    (void)group_size;
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return (size_t)AWQ_WS_COUNTER_BYTES + 2 * region_bytes(M, N);
}

int awq_gemm_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) return AWQ_ERR_NULL;
    if (workspace_bytes < AWQ_WS_COUNTER_BYTES) return AWQ_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(workspace, 0, AWQ_WS_COUNTER_BYTES, st) != hipSuccess) return AWQ_ERR_LAUNCH;
    if (workspace_bytes > AWQ_WS_COUNTER_BYTES &&
        hipMemsetAsync(static_cast<char*>(workspace) + AWQ_WS_COUNTER_BYTES, 0xFF, workspace_bytes - AWQ_WS_COUNTER_BYTES, st) != hipSuccess)
        return AWQ_ERR_LAUNCH;
    return AWQ_OK;
}

namespace {
struct NormArgs {  // prologue / epilogue extras of awq_gemm_forward_normed / _ex
    const uint16_t* res_in;
    uint16_t* res_out;
    const uint16_t* weight;  // NULL: no norm prologue
    float eps;
    const float* ssq_in = nullptr;
    int ssq_in_tiles = 0;
    const uint16_t* add_res = nullptr;
    float* ssq_out = nullptr;
};
int gemm_forward_impl(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                      const uint16_t* bias, uint16_t* y, int64_t M, int64_t K, int64_t N, int64_t group_size,
                      void* workspace, size_t workspace_bytes, uint32_t flags, void* stream, const NormArgs* nrm);
}  // namespace

int awq_gemm_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                     const uint16_t* bias, uint16_t* y, int64_t M, int64_t K, int64_t N, int64_t group_size,
                     void* workspace, size_t workspace_bytes, uint32_t flags, void* stream) {
    return gemm_forward_impl(x, qweight, scales, qzeros, bias, y, M, K, N, group_size, workspace, workspace_bytes, flags,
                             stream, nullptr);
}

int awq_gemm_forward_normed(const uint16_t* x, const uint16_t* residual_in, uint16_t* residual_out,
                            const uint16_t* norm_weight, float eps, const int32_t* qweight, const uint16_t* scales,
                            const int32_t* qzeros, const uint16_t* bias, uint16_t* y, int64_t M, int64_t K, int64_t N,
                            int64_t group_size, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream) {
    if (!norm_weight) return AWQ_ERR_NULL;
    if ((residual_in == nullptr) != (residual_out == nullptr)) return AWQ_ERR_NULL;
    if (residual_in && residual_in == residual_out) return AWQ_ERR_BAD_ALIGNMENT;  // every block reads it, one block writes
    if (!aligned16(norm_weight) || (residual_in && (!aligned16(residual_in) || !aligned16(residual_out))))
        return AWQ_ERR_BAD_ALIGNMENT;
    if (M > 4 || (flags & AWQ_GEMM_FLAG_X_GATED_SILU)) return AWQ_ERR_UNSUPPORTED;
    const NormArgs nrm{residual_in, residual_out, norm_weight, eps};
    return gemm_forward_impl(x, qweight, scales, qzeros, bias, y, M, K, N, group_size, workspace, workspace_bytes, flags,
                             stream, &nrm);
}

int64_t awq_gemm_ex_ssq_tiles(int64_t N) { return N > 0 ? (N + 255) / 256 : 0; }

int awq_gemm_forward_ex(const AwqGemmEx* e) {
    if (!e) return AWQ_ERR_NULL;
    if (e->struct_bytes != sizeof(AwqGemmEx)) return AWQ_ERR_BAD_SHAPE;
    NormArgs n{e->residual_in, e->residual_out, e->norm_weight, e->norm_eps};
    n.ssq_in = e->ssq_in; n.ssq_in_tiles = (int)e->ssq_in_tiles; n.add_res = e->add_residual; n.ssq_out = e->ssq_out;
    if ((e->residual_in == nullptr) != (e->residual_out == nullptr)) return AWQ_ERR_NULL;
    if (e->residual_in && (!e->norm_weight || e->residual_in == e->residual_out)) return AWQ_ERR_BAD_SHAPE;
    if (e->ssq_in && (!e->norm_weight || e->residual_in || e->ssq_in_tiles < 1 || e->ssq_in_tiles > 64)) return AWQ_ERR_BAD_SHAPE;
    if (e->add_residual && e->add_residual == e->y) return AWQ_ERR_BAD_SHAPE;  // other tiles' blocks may still read it
    const uint16_t* ptrs[] = {e->norm_weight, e->residual_in, e->residual_out, e->add_residual};
    for (const uint16_t* q : ptrs)
        if (q && !aligned16(q)) return AWQ_ERR_BAD_ALIGNMENT;
    return gemm_forward_impl(e->x, e->qweight, e->scales, e->qzeros, e->bias, e->y, e->M, e->K, e->N, e->group_size,
                             e->workspace, e->workspace_bytes, e->flags, e->stream, &n);
}

namespace {
// M > 16.  The register-decoded prefill kernel wins once its 128 x 256 tiles give every CU a block (r02 sweep,
// profiles/r02_regb_by_m.txt: 4096 x 11008 from M = 768, 11008 x 4096 from M = 2048); below that the LDS-tiled kernel with
// split-K (smaller tiles, in-launch combine) fills the chip better.
bool skinny_before_decode(int64_t M, int64_t K, int64_t N) { return M >= 9 && M <= 16 && K < 8192 && (N + 255) / 256 < 64; }

unsigned auto_kernel_large(int M, int K, int N, int g) {
    if (M <= 64 && awq_gemm_skinny_supports(M, K, N, g)) return AWQ_GEMM_KERNEL_SKINNY;
    if (awq_gemm_regb_supports(M, K, N, g) && (int64_t)((M + 127) / 128) * ((N + 255) / 256) >= 256) return AWQ_GEMM_KERNEL_REGB;
    if (awq_gemm_tiled_supports(M, K, N, g)) return AWQ_GEMM_KERNEL_TILED;
    return AWQ_GEMM_KERNEL_NAIVE;  // odd shapes
}

int gemm_forward_impl(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                      const uint16_t* bias, uint16_t* y, int64_t M, int64_t K, int64_t N, int64_t group_size,
                      void* workspace, size_t workspace_bytes, uint32_t flags, void* stream, const NormArgs* nrm) {
    int rc = check_gemm_layout(K, N, group_size);
    if (rc) return rc;
    if (M < 0 || M > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (M == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales) || !aligned16(qzeros) || !aligned16(y))
        return AWQ_ERR_BAD_ALIGNMENT;
    if (workspace && !aligned16(workspace)) return AWQ_ERR_BAD_ALIGNMENT;

    AwqGemmArgs a;
    a.x = x; a.qweight = qweight; a.scales = scales; a.qzeros = qzeros; a.bias = bias; a.y = y;
    a.M = (int)M; a.K = (int)K; a.N = (int)N; a.g = (int)group_size;
    a.stream = static_cast<hipStream_t>(stream);
    a.counters = nullptr; a.partial = nullptr; a.partial_floats = 0; a.exchange = nullptr; a.exchange_bytes = 0;
    if (workspace && workspace_bytes > AWQ_WS_COUNTER_BYTES + 512) {
        const size_t half = ((workspace_bytes - AWQ_WS_COUNTER_BYTES) / 2) & ~(size_t)255;
        a.counters = static_cast<int*>(workspace);
        a.exchange = reinterpret_cast<float*>(static_cast<char*>(workspace) + AWQ_WS_COUNTER_BYTES);
        a.exchange_bytes = half;
        a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + AWQ_WS_COUNTER_BYTES + half);
        a.partial_floats = half / sizeof(float);
    }

    a.x_gated = (flags & AWQ_GEMM_FLAG_X_GATED_SILU) ? 1 : 0;
    if (nrm) {
        if (nrm->weight) {
            if (a.x_gated) return AWQ_ERR_UNSUPPORTED;
            a.x_gated = 2;
            a.res_in = nrm->res_in; a.res_out = nrm->res_out; a.norm_w = nrm->weight; a.norm_eps = nrm->eps;
            a.ssq_in = nrm->ssq_in; a.ssq_in_tiles = nrm->ssq_in_tiles;
        }
        a.add_res = nrm->add_res; a.ssq_out = nrm->ssq_out;
    }
    const bool extras = nrm != nullptr;  // decode kernel only, in-launch combine only
    if ((a.x_gated || extras) && (M > 16 || !awq_gemv_mfma_supports(a.M, a.K, a.N, a.g, 2))) return AWQ_ERR_UNSUPPORTED;
    if (extras && (M > 4 || (flags & AWQ_GEMM_FLAG_TWO_PASS))) return AWQ_ERR_UNSUPPORTED;
    unsigned kern = AWQ_GEMM_FLAG_KERNEL(flags);
    if ((a.x_gated || extras) && kern != AWQ_GEMM_KERNEL_AUTO && kern != AWQ_GEMM_KERNEL_MFMA_GEMV) return AWQ_ERR_UNSUPPORTED;
    if (extras && AWQ_GEMM_FLAG_NLOG(flags) == 4) return AWQ_ERR_UNSUPPORTED;  // the epilogue assumes 256-column tiles
    int nlog = (int)AWQ_GEMM_FLAG_NLOG(flags);
    int splitk = (int)AWQ_GEMM_FLAG_SPLITK(flags);
    const int waves = (int)AWQ_GEMM_FLAG_WAVES(flags);
    const bool two_pass = (flags & AWQ_GEMM_FLAG_TWO_PASS) != 0;
    const bool nt = (flags & AWQ_GEMM_FLAG_NO_NT) == 0;
    if ((int64_t)K * N / 2 >= ((int64_t)1 << 31)) return AWQ_ERR_UNSUPPORTED;  // 32-bit buffer offsets

    if (kern == AWQ_GEMM_KERNEL_AUTO) {
        // 9 .. 16 rows: the register-decoded batched kernel is already ahead of the decode kernel on matrices of up to 8191
        // rows and fewer than 64 column tiles (plain calls only; r02: 13.8 vs 15.6 us at 4096 x 11008, M = 16)
        const bool skinny_first = skinny_before_decode(M, K, N) && !a.x_gated && !extras && awq_gemm_skinny_supports(a.M, a.K, a.N, a.g);
        if (M <= 16 && !skinny_first && awq_gemv_mfma_supports(a.M, a.K, a.N, a.g, 2)) {
            rc = awq_launch_gemv_mfma(a, 0, 0, 0, 0, false);
            if (rc != AWQ_ERR_UNSUPPORTED) {
                g_last_kernel = "gemv_mfma";
                return rc;
            }
        }
        if (a.x_gated || extras) return AWQ_ERR_UNSUPPORTED;
        kern = (M > 16 || skinny_first) ? auto_kernel_large(a.M, a.K, a.N, a.g) : AWQ_GEMM_KERNEL_NAIVE;  // M <= 16 that the decode kernel refused: odd shapes
    }
    switch (kern) {
        case AWQ_GEMM_KERNEL_NAIVE:
            g_last_kernel = "naive";
            return awq_launch_gemm_naive(a);
        case AWQ_GEMM_KERNEL_VALU: {
            if (nlog == 0) nlog = 3;
            if (splitk == 0) splitk = awq_gemv_valu_default_split(a.K, a.N, nlog);
            while (splitk > 1 && (size_t)splitk * a.M * a.N * 4 > a.partial_floats * sizeof(float)) --splitk;
            g_last_kernel = "gemv_valu";
            return awq_launch_gemv_valu(a, nlog, splitk, nt);
        }
        case AWQ_GEMM_KERNEL_MFMA_GEMV: {
            if (M > 16) return AWQ_ERR_UNSUPPORTED;
            g_last_kernel = "gemv_mfma";
            return awq_launch_gemv_mfma(a, nlog, waves, (int)AWQ_GEMM_FLAG_UNIT(flags), splitk, two_pass);
        }
        case AWQ_GEMM_KERNEL_TILED: {
            g_last_kernel = "gemm_tiled";
            return awq_launch_gemm_tiled(a, nlog == 2 ? 256 : (nlog == 1 ? 128 : 0), splitk);
        }
        case AWQ_GEMM_KERNEL_SKINNY: {
            rc = awq_launch_gemm_skinny(a, splitk);
            if (rc != AWQ_ERR_UNSUPPORTED || AWQ_GEMM_FLAG_KERNEL(flags) != AWQ_GEMM_KERNEL_AUTO || !awq_gemm_tiled_supports(a.M, a.K, a.N, a.g)) {
                g_last_kernel = "gemm_skinny";
                return rc;
            }
            g_last_kernel = "gemm_tiled";  // AUTO: a slice that does not fit (very wide matrices at M > 32): the LDS-tiled kernel
            return awq_launch_gemm_tiled(a, 0, 0);
        }
        case AWQ_GEMM_KERNEL_REGB: {
            g_last_kernel = "gemm_regb";
            return awq_launch_gemm_regb(a, nlog == 2 ? 256 : (nlog == 1 ? 128 : 0));
        }
        default:
            return AWQ_ERR_UNSUPPORTED;
    }
}

}  // namespace

int awq_gemm_auto_kernel(int64_t M, int64_t K, int64_t N, int64_t group_size) {
    if (M <= 0 || K <= 0 || N <= 0 || M > INT32_MAX || check_gemm_layout(K, N, group_size)) return -1;
    if (skinny_before_decode(M, K, N) && awq_gemm_skinny_supports((int)M, (int)K, (int)N, (int)group_size)) return AWQ_GEMM_KERNEL_SKINNY;
    if (M <= 16 && awq_gemv_mfma_supports((int)M, (int)K, (int)N, (int)group_size, 2)) return AWQ_GEMM_KERNEL_MFMA_GEMV;
    return M > 16 ? (int)auto_kernel_large((int)M, (int)K, (int)N, (int)group_size) : (int)AWQ_GEMM_KERNEL_NAIVE;
}

int awq_gemm_workspace_status(const void* workspace, void* stream, int32_t* err_out) {
    if (!workspace || !err_out) return AWQ_ERR_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int32_t word = 0;
    if (hipMemcpyAsync(&word, workspace, sizeof(word), hipMemcpyDeviceToHost, st) != hipSuccess) return AWQ_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return AWQ_ERR_LAUNCH;
    *err_out = word;
    return AWQ_OK;
}

/* ---- fused MLP / MoE ------------------------------------------------------------------- */

int awq_silu_and_mul(const uint16_t* gate_up, uint16_t* out, int64_t rows, int64_t d, void* stream) {
    if (rows < 0 || d < 0 || d % 8) return AWQ_ERR_BAD_SHAPE;
    if (rows * d == 0) return AWQ_OK;
    if (!gate_up || !out) return AWQ_ERR_NULL;
    if (!aligned16(gate_up) || !aligned16(out)) return AWQ_ERR_BAD_ALIGNMENT;
    return awq_launch_silu_and_mul(gate_up, out, rows, d, static_cast<hipStream_t>(stream));
}

int awq_moe_route_local(const float* gating_logits, float* topk_weights, int32_t* topk_ids, int32_t* sorted_token_ids,
                        int32_t* expert_ids, int32_t* num_tokens_post_padded, int64_t num_tokens, int64_t num_experts,
                        int64_t topk, int renormalize, int64_t block_rows, int64_t first_expert, int64_t num_local,
                        void* stream) {
    if (num_tokens < 0 || num_experts < 1 || topk < 1 || block_rows < 0) return AWQ_ERR_BAD_SHAPE;
    if (first_expert < 0 || num_local < 1 || first_expert + num_local > num_experts) return AWQ_ERR_BAD_SHAPE;
    if (num_tokens == 0) return AWQ_OK;
    // block_rows == 0: routing only (softmax, top-k, renormalisation) -- the three alignment outputs are not touched and may be NULL
    if (!gating_logits || !topk_weights || !topk_ids || (block_rows > 0 && (!sorted_token_ids || !expert_ids || !num_tokens_post_padded)))
        return AWQ_ERR_NULL;
    if (num_tokens * topk > (1 << 24)) return AWQ_ERR_UNSUPPORTED;
    return awq_launch_moe_route(gating_logits, topk_weights, topk_ids, sorted_token_ids, expert_ids,
                                num_tokens_post_padded, (int)num_tokens, (int)num_experts, (int)topk, renormalize,
                                (int)block_rows, (int)first_expert, (int)num_local, static_cast<hipStream_t>(stream));
}

int awq_moe_route(const float* gating_logits, float* topk_weights, int32_t* topk_ids, int32_t* sorted_token_ids,
                  int32_t* expert_ids, int32_t* num_tokens_post_padded, int64_t num_tokens, int64_t num_experts,
                  int64_t topk, int renormalize, int64_t block_rows, void* stream) {
    return awq_moe_route_local(gating_logits, topk_weights, topk_ids, sorted_token_ids, expert_ids, num_tokens_post_padded,
                               num_tokens, num_experts, topk, renormalize, block_rows, 0, num_experts, stream);
}

size_t awq_grouped_gemm_workspace_bytes(int64_t max_blocks, int64_t K, int64_t N) {
    if (max_blocks <= 0 || K <= 0 || N <= 0) return 0;
    return awq_grouped_workspace_bytes_impl((int)max_blocks, (int)K, (int)N);
}

int awq_grouped_gemm_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                             uint16_t* y, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                             const int32_t* num_tokens_post_padded, const float* pair_weights, int64_t num_pairs,
                             int64_t x_div, int64_t block_rows, int64_t max_blocks, int64_t num_experts, int64_t K,
                             int64_t N, int64_t group_size, void* workspace, size_t workspace_bytes, void* stream) {
    return awq_grouped_gemm_forward_ex(x, qweight, scales, qzeros, y, sorted_token_ids, expert_ids, num_tokens_post_padded,
                                       pair_weights, num_pairs, x_div, block_rows, max_blocks, num_experts, K, N, group_size,
                                       workspace, workspace_bytes, 0u, stream);
}

int awq_grouped_gemm_forward_ex(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                uint16_t* y, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                                const int32_t* num_tokens_post_padded, const float* pair_weights, int64_t num_pairs,
                                int64_t x_div, int64_t block_rows, int64_t max_blocks, int64_t num_experts, int64_t K,
                                int64_t N, int64_t group_size, void* workspace, size_t workspace_bytes, uint32_t flags,
                                void* stream) {
    if (flags & ~AWQ_GEMM_FLAG_X_GATED_SILU) return AWQ_ERR_UNSUPPORTED;
    int rc = check_gemm_layout(K, N, group_size);
    if (rc) return rc;
    if (!(block_rows == 8 || block_rows == 16)) return AWQ_ERR_BAD_SHAPE;
    if (num_pairs < 0 || x_div < 1 || max_blocks < 0 || num_experts < 1) return AWQ_ERR_BAD_SHAPE;
    if (num_pairs == 0 || max_blocks == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y || !sorted_token_ids || !expert_ids || !num_tokens_post_padded)
        return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales) || !aligned16(qzeros) || !aligned16(y))
        return AWQ_ERR_BAD_ALIGNMENT;
    if ((int64_t)num_experts * K * N / 2 >= ((int64_t)1 << 40)) return AWQ_ERR_UNSUPPORTED;
    if ((int64_t)K * N / 2 >= ((int64_t)1 << 31)) return AWQ_ERR_UNSUPPORTED;
    AwqGemmArgs a;
    a.x = x; a.qweight = qweight; a.scales = scales; a.qzeros = qzeros; a.bias = nullptr; a.y = y;
    a.M = (int)block_rows; a.K = (int)K; a.N = (int)N; a.g = (int)group_size;
    a.x_gated = (flags & AWQ_GEMM_FLAG_X_GATED_SILU) ? 1 : 0;
    a.stream = static_cast<hipStream_t>(stream);
    a.counters = nullptr; a.partial = nullptr; a.partial_floats = 0; a.exchange = nullptr; a.exchange_bytes = 0;
    if (workspace && workspace_bytes > AWQ_WS_COUNTER_BYTES) {  // [control][exchange]: no scratch half here
        a.counters = static_cast<int*>(workspace);
        a.exchange = reinterpret_cast<float*>(static_cast<char*>(workspace) + AWQ_WS_COUNTER_BYTES);
        a.exchange_bytes = workspace_bytes - AWQ_WS_COUNTER_BYTES;
    }
    const int64_t NW = N / 8, G = K / group_size;
    g_last_kernel = "gemv_mfma_grouped";
    return awq_launch_grouped_gemm(a, sorted_token_ids, expert_ids, num_tokens_post_padded, pair_weights, (int)num_pairs,
                                   (int)x_div, (int)max_blocks, K * NW, G * NW, G * N);
}

int awq_grouped_gemm_prefill(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                             uint16_t* y, const int32_t* seg_offsets, int64_t P, int64_t num_experts, int64_t K, int64_t N,
                             int64_t group_size, uint32_t flags, void* stream) {
    int rc = check_gemm_layout(K, N, group_size);
    if (rc) return rc;
    if (P < 0 || P > INT32_MAX || num_experts < 1 || num_experts > 4096) return AWQ_ERR_BAD_SHAPE;
    if (P == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y || !seg_offsets) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales) || !aligned16(qzeros) || !aligned16(y)) return AWQ_ERR_BAD_ALIGNMENT;
    g_last_kernel = "gemm_regb_grouped";
    return awq_launch_gemm_regb_grouped(x, qweight, scales, qzeros, y, seg_offsets, (int)P, (int)num_experts, (int)K, (int)N,
                                        (int)group_size, AWQ_GEMM_FLAG_NLOG(flags) == 2 ? 256 : 0, static_cast<hipStream_t>(stream));
}

int awq_moe_sort_pairs(const int32_t* topk_ids, int32_t* order, int32_t* seg_offsets, int64_t num_pairs, int64_t num_experts,
                       void* stream) {
    if (num_pairs < 0 || num_pairs > (1 << 24) || num_experts < 1) return AWQ_ERR_BAD_SHAPE;
    if (!seg_offsets || (num_pairs && (!topk_ids || !order))) return AWQ_ERR_NULL;
    if (num_experts > 64) return AWQ_ERR_UNSUPPORTED;
    if (num_pairs == 0) return hipMemsetAsync(seg_offsets, 0, (size_t)(num_experts + 1) * 4, static_cast<hipStream_t>(stream)) == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
    return awq_launch_moe_sort(topk_ids, (int)num_pairs, (int)num_experts, order, seg_offsets, static_cast<hipStream_t>(stream));
}

int awq_grouped_gemm_prefill_ex(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                uint16_t* y, const int32_t* seg_offsets, const int32_t* row_map, const float* pair_weights,
                                int64_t P, int64_t x_div, int64_t num_experts, int64_t K, int64_t N, int64_t group_size,
                                uint32_t flags, void* stream) {
    int rc = check_gemm_layout(K, N, group_size);
    if (rc) return rc;
    if (P < 0 || P > INT32_MAX || num_experts < 1 || num_experts > 4096 || x_div < 1) return AWQ_ERR_BAD_SHAPE;
    if (flags & ~(0xF0u | AWQ_GROUPED_PREFILL_GATHER_X | AWQ_GROUPED_PREFILL_SCATTER_Y)) return AWQ_ERR_BAD_SHAPE;
    if (P == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y || !seg_offsets) return AWQ_ERR_NULL;
    const bool gather = flags & AWQ_GROUPED_PREFILL_GATHER_X, scatter = flags & AWQ_GROUPED_PREFILL_SCATTER_Y;
    if ((gather || scatter || pair_weights) && !row_map) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales) || !aligned16(qzeros) || !aligned16(y)) return AWQ_ERR_BAD_ALIGNMENT;
    g_last_kernel = "gemm_regb_grouped";
    return awq_launch_gemm_regb_grouped(x, qweight, scales, qzeros, y, seg_offsets, (int)P, (int)num_experts, (int)K, (int)N,
                                        (int)group_size, AWQ_GEMM_FLAG_NLOG(flags) == 2 ? 256 : 0, static_cast<hipStream_t>(stream), row_map,
                                        (int)x_div, pair_weights, gather ? 1 : 0, scatter ? 1 : 0);
}

/* ---- GEMV layout ------------------------------------------------------------------------- */

// Which kernel awq_gemv_forward's AUTO dispatch takes (host only, no launch): the row-streaming kernel at batches 1 and 2, and at
// batches 3 .. 4 while K <= 6144 (profiles/r03_gemv_rows_sweep.txt); the LDS-streaming MFMA kernel from five
// batch rows on matrices of 8192 rows and more (4096 x 11008, M = 8: 11.6 us vs 19.0 for the tile kernel; 4096 x 22016: 16.7 vs
// 35.7; below that the tile kernel is level or ahead); the 16-row MFMA tile kernel otherwise.  -1: no kernel takes the shape.
int awq_gemv_auto_kernel(int64_t M, int64_t K, int64_t N, int64_t group_size) {
    if (M <= 0 || K <= 0 || N <= 0 || group_size <= 0 || K % group_size || K % 8 || M > INT32_MAX || K > INT32_MAX || N > INT32_MAX) return -1;
    const int m = (int)M, k = (int)K, n = (int)N, g = (int)group_size;
    // round 5: from five rows the batched kernel (gemv_batch.hip: activations in registers, the K range split over the waves of a
    // block, weights by LDS-DMA) -- up to 32 rows per launch, any M in one call (balanced chunks); it replaces gemv_lds / gemv_nk
    // wherever it takes the shape (group_size 128): 4096 x 11008, M = 8: see profiles/r05_*; AWQ_GEMV_KERNEL_PREFILL is explicit only
    // ... and already at four rows while K > 2048, at three where the row-streaming kernel is not the choice (K > 6144): it costs the
    // same at 1 .. 5 rows while gemv_rows / gemv_nk grow by ~1 us per row (profiles/r05_sweep_small_batch.txt, M = 4: 9.1 vs 10.3 us at
    // 4096 x 11008, 14.4 vs 15.3 at 4096 x 22016, 12.7 vs 17.6 at 8192 x 7168; level at K = 11008; behind at K = 1024)
    if ((M >= 5 || (M == 4 && K > 2048) || (M == 3 && K > 6144)) && awq_gemv_batch_supports((int)(M > 32 ? 32 : M), k, n, g))
        return (int)AWQ_GEMV_KERNEL_BATCH;
    if (M > 16) return -1;  // the older decode kernels serve 16 rows per call (the host wrapper chunks)
    // round 4: batch 2 at every K and batches 3 .. 4 while K <= 6144 also run the row-streaming kernel -- it is ahead of the
    // 16-row tile kernel there on all four 7B shapes (profiles/r03_gemv_rows_sweep.txt: M = 2 4.65 / 7.05 / 10.77 / 8.57 us vs
    // 6.18 / 9.96 / 16.65 / 10.73; M = 4 at K = 4096 5.79 / 9.79 / 14.39 vs 6.59 / 10.40 / 18.29; at K = 11008 12.25 vs 11.79)
    if ((M <= 2 || (M <= 4 && K <= 6144)) && awq_gemv_rows_supports(m, k, n, g)) return (int)AWQ_GEMV_KERNEL_ROWS;
    if (M >= 5 && N >= 8192 && awq_gemv_lds_supports(m, k, n, g)) return (int)AWQ_GEMV_KERNEL_LDS;
    return awq_gemv_nk_supports(m, k, n, g) ? (int)AWQ_GEMV_KERNEL_TILE16 : -1;
}

int awq_gemv_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                     uint16_t* y, int64_t M, int64_t K, int64_t N, int64_t group_size, int64_t zeros_width,
                     uint32_t flags, void* stream) {
    if (K <= 0 || N < 0 || group_size <= 0 || K % group_size || K % 8 || zeros_width <= 0) return AWQ_ERR_BAD_SHAPE;
    if (zeros_width * 8 < K / group_size) return AWQ_ERR_BAD_SHAPE;
    if (M < 0 || M > INT32_MAX || K > INT32_MAX || N > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (M == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales)) return AWQ_ERR_BAD_ALIGNMENT;
    const uint32_t kern = AWQ_GEMM_FLAG_KERNEL(flags);
    if (kern == AWQ_GEMV_KERNEL_PREFILL) {
        if (!aligned16(y)) return AWQ_ERR_BAD_ALIGNMENT;
        g_last_kernel = "gemm_regb_nk";
        return awq_launch_gemm_regb_nk(x, qweight, scales, qzeros, nullptr, y, (int)M, (int)K, (int)N, (int)group_size, (int)zeros_width,
                                       AWQ_GEMM_FLAG_NLOG(flags) == 2 ? 256 : 0, static_cast<hipStream_t>(stream));
    }
    const int auto_k = kern == AWQ_GEMV_KERNEL_AUTO ? awq_gemv_auto_kernel(M, K, N, group_size) : -1;
    if (kern == AWQ_GEMV_KERNEL_BATCH || auto_k == (int)AWQ_GEMV_KERNEL_BATCH) {
        // one launch per <= 128 rows (round 6: up to four row parts of <= 32 rows, each in its own block, the blocks of a tile list
        // residents of one XCD so that the matrix is fetched from HBM once); where the parts do not fit the LDS budget (very long
        // tile lists per block) launches of fewer rows
        int64_t cap = 128;
        while (cap > 32 && !awq_gemv_batch_supports((int)(M > cap ? cap : M), (int)K, (int)N, (int)group_size)) cap /= 2;
        if (!awq_gemv_batch_supports((int)(M > cap ? cap : M), (int)K, (int)N, (int)group_size)) return AWQ_ERR_UNSUPPORTED;
        g_last_kernel = "gemv_batch";
        const int64_t nchunk = (M + cap - 1) / cap, rows = (M + nchunk - 1) / nchunk;  // balanced chunks of at most `cap` rows
        for (int64_t m0 = 0; m0 < M; m0 += rows) {
            const int mm = (int)(M - m0 < rows ? M - m0 : rows);
            const int rc = awq_launch_gemv_batch(x + m0 * K, qweight, scales, qzeros, y + m0 * N, mm, (int)K, (int)N, (int)group_size,
                                                 (int)zeros_width, (int)(AWQ_GEMM_FLAG_UNIT(flags) | (AWQ_GEMM_FLAG_WAVES(flags) << 4)),
                                                 (int)AWQ_GEMM_FLAG_SPLITK(flags), static_cast<hipStream_t>(stream));
            if (rc != AWQ_OK) return rc;
        }
        return AWQ_OK;
    }
    if (M > 16) return AWQ_ERR_UNSUPPORTED;  // the older decode kernels take at most 16 rows per call
    if ((kern == AWQ_GEMV_KERNEL_ROWS && awq_gemv_rows_supports((int)M, (int)K, (int)N, (int)group_size)) ||
        auto_k == (int)AWQ_GEMV_KERNEL_ROWS) {
        g_last_kernel = "gemv_rows";
        return awq_launch_gemv_rows(x, qweight, scales, qzeros, y, (int)M, (int)K, (int)N, (int)group_size, (int)zeros_width,
                                    (int)AWQ_GEMM_FLAG_WAVES(flags), (int)AWQ_GEMM_FLAG_UNIT(flags),
                                    (int)AWQ_GEMM_FLAG_SPLITK(flags), (int)AWQ_GEMM_FLAG_NLOG(flags),
                                    static_cast<hipStream_t>(stream));
    }
    if (kern == AWQ_GEMV_KERNEL_ROWS) return AWQ_ERR_UNSUPPORTED;
    if ((kern == AWQ_GEMV_KERNEL_LDS && awq_gemv_lds_supports((int)M, (int)K, (int)N, (int)group_size)) ||
        auto_k == (int)AWQ_GEMV_KERNEL_LDS) {
        g_last_kernel = "gemv_lds";
        return awq_launch_gemv_lds(x, qweight, scales, qzeros, y, (int)M, (int)K, (int)N, (int)group_size, (int)zeros_width,
                                   (int)AWQ_GEMM_FLAG_SPLITK(flags), (int)AWQ_GEMM_FLAG_UNIT(flags), static_cast<hipStream_t>(stream));
    }
    if (kern == AWQ_GEMV_KERNEL_LDS) return AWQ_ERR_UNSUPPORTED;
    if (!awq_gemv_nk_supports((int)M, (int)K, (int)N, (int)group_size)) return AWQ_ERR_UNSUPPORTED;
    g_last_kernel = "gemv_nk";
    return awq_launch_gemv_nk(x, qweight, scales, qzeros, y, (int)M, (int)K, (int)N, (int)group_size, (int)zeros_width,
                              (int)AWQ_GEMM_FLAG_WAVES(flags), (int)AWQ_GEMM_FLAG_UNIT(flags),
                              static_cast<hipStream_t>(stream));
}

int awq_gemv_forward_ex(const AwqGemvEx* e) {
    if (!e || e->struct_bytes != sizeof(AwqGemvEx)) return AWQ_ERR_BAD_SHAPE;
    const int64_t M = e->M, K = e->K, N = e->N, g = e->group_size, ZW = e->zeros_width;
    if (K <= 0 || N < 0 || g <= 0 || K % g || K % 8 || ZW <= 0 || ZW * 8 < K / g) return AWQ_ERR_BAD_SHAPE;
    if (M < 0 || M > INT32_MAX || K > INT32_MAX || N > INT32_MAX || (e->flags & ~AWQ_GEMV_EX_SILU_PAIRS)) return AWQ_ERR_BAD_SHAPE;
    const bool pairs = e->flags & AWQ_GEMV_EX_SILU_PAIRS;
    if (pairs && (e->add_residual || N % 2)) return AWQ_ERR_BAD_SHAPE;
    if (M == 0 || N == 0) return AWQ_OK;
    if (!e->x || !e->qweight || !e->scales || !e->qzeros || !e->y) return AWQ_ERR_NULL;
    if (!aligned16(e->x) || !aligned16(e->qweight) || !aligned16(e->scales) || (e->norm_weight && !aligned16(e->norm_weight)))
        return AWQ_ERR_BAD_ALIGNMENT;
    if (M != 1) return AWQ_ERR_UNSUPPORTED;
    const AwqRowsFx fx{e->norm_weight, e->norm_eps, e->add_residual, pairs};
    g_last_kernel = "gemv_rows";
    return awq_launch_gemv_rows(e->x, e->qweight, e->scales, e->qzeros, e->y, (int)M, (int)K, (int)N, (int)g, (int)ZW, 0, 0, 0, 0,
                                static_cast<hipStream_t>(e->stream), &fx);
}

int awq_grouped_gemv_forward(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                             uint16_t* y, const int32_t* pair_experts, const float* pair_weights, int64_t num_pairs,
                             int64_t x_div, int64_t num_experts, int64_t first_expert, int64_t K, int64_t N, int64_t group_size,
                             int64_t zeros_width, uint32_t flags, int64_t parts, void* stream) {
    const int64_t g = group_size, ZW = zeros_width;
    if (first_expert < 0 || first_expert > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (K <= 0 || N < 0 || g <= 0 || K % g || K % 8 || ZW <= 0 || ZW * 8 < K / g) return AWQ_ERR_BAD_SHAPE;
    if (num_pairs < 0 || num_experts < 1 || x_div < 1 || parts < 0 || K > INT32_MAX || N > INT32_MAX || num_experts > INT32_MAX ||
        (flags & ~AWQ_GEMV_EX_SILU_PAIRS))
        return AWQ_ERR_BAD_SHAPE;
    const bool pairs = flags & AWQ_GEMV_EX_SILU_PAIRS;
    if (pairs && (pair_weights || N % 2)) return AWQ_ERR_BAD_SHAPE;
    if (num_pairs == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y || !pair_experts) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales)) return AWQ_ERR_BAD_ALIGNMENT;
    if (num_pairs > 8191 || parts > 65535 || x_div > num_pairs) return AWQ_ERR_UNSUPPORTED;
    // the per-expert strides must keep every expert's tensors 16-byte aligned (whole rows of 16-byte multiples)
    if ((N * (K / 8) * 4) % 16 || (N * ZW * 16) % 16) return AWQ_ERR_UNSUPPORTED;
    AwqRowsFx fx;
    fx.pairs = pairs;
    fx.pair_expert = pair_experts;
    fx.pair_scale = pair_weights;
    fx.num_pairs = (int)num_pairs;
    fx.num_experts = (int)num_experts;
    fx.first_expert = (int)first_expert;
    fx.x_div = (int)x_div;
    fx.parts = (int)parts;
    g_last_kernel = "gemv_rows_grouped";
    return awq_launch_gemv_rows(x, qweight, scales, qzeros, y, 1, (int)K, (int)N, (int)g, (int)ZW, 0, 0, 0, 0,
                                static_cast<hipStream_t>(stream), &fx);
}

size_t awq_gemv_lds_bytes(int64_t M, int64_t K, int64_t zeros_width) {
    return awq_gemv_nk_lds_bytes((int)M, (int)K, (int)zeros_width, 8);
}

int awq_dequantize_weights_gemv(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out,
                                int64_t K, int64_t N, int64_t group_size, int64_t zeros_width, void* stream) {
    if (K < 0 || N < 0 || group_size <= 0 || K % 8 || group_size % 8 || (K && K % group_size)) return AWQ_ERR_BAD_SHAPE;
    if (K * N == 0) return AWQ_OK;
    if (!qweight || !scales || !qzeros || !out) return AWQ_ERR_NULL;
    if (!aligned16(out)) return AWQ_ERR_BAD_ALIGNMENT;
    return awq_launch_dequant_nk(qweight, scales, qzeros, out, (int)K, (int)N, (int)group_size, (int)zeros_width,
                                 static_cast<hipStream_t>(stream));
}

/* ---- GEMVFast layout --------------------------------------------------------------------- */

int awq_gemv_fast_forward(const uint16_t* x, const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros,
                          uint16_t* y, int64_t M, int64_t K, int64_t N, int64_t group_size, int64_t group_rows,
                          uint32_t flags, void* stream) {
    if (K <= 0 || N < 0 || group_size <= 0 || K % group_size || N % 4 || K % 64) return AWQ_ERR_BAD_SHAPE;
    if (group_rows < K / group_size) return AWQ_ERR_BAD_SHAPE;
    if (M < 0 || M > INT32_MAX || K > INT32_MAX || N > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (M == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales) || !aligned16(qzeros)) return AWQ_ERR_BAD_ALIGNMENT;
    // round 5: from five rows the batched kernel (gemv_batch.hip, GEMVFast form: any M in launches of <= 32 rows); AWQ_GEMM_FLAG_KERNEL:
    // 0 = auto, 1 = the 16-row kernel (gemv_fast.hip, M <= 16), AWQ_GEMV_KERNEL_BATCH = the batched kernel
    const uint32_t kern = AWQ_GEMM_FLAG_KERNEL(flags);
    int64_t fcap = 128;  // rows per launch (round 6: row parts across blocks, see gemv_batch.hip)
    while (fcap > 32 && !awq_gemv_batch_fast_supports((int)(M > fcap ? fcap : M), (int)K, (int)N, (int)group_size)) fcap /= 2;
    const bool batch_ok = awq_gemv_batch_fast_supports((int)(M > fcap ? fcap : M), (int)K, (int)N, (int)group_size);
    // below five rows AUTO takes it where ONE pass of eight waves covers K (2048 < K <= 4096): there it is ahead of the 16-row kernel at
    // every M (profiles/r05_sweep_small_batch.txt, M = 1: 5.4 / 9.3 / 9.6 / 15.0 us vs 6.2 / 9.7 / 10.1 / 16.3 at N = 4096 / 11008 /
    // 12288 / 22016), behind it at K = 1024 (fewer waves per tile) and from K = 8192 (two passes)
    const bool small_ok = K > 2048 && K <= 4096;
    if (kern == AWQ_GEMV_KERNEL_BATCH || (kern == 0 && (M >= 5 || small_ok) && batch_ok)) {
        if (!batch_ok) return AWQ_ERR_UNSUPPORTED;
        g_last_kernel = "gemv_batch_fast";
        const int64_t nchunk = (M + fcap - 1) / fcap, rows = (M + nchunk - 1) / nchunk;
        for (int64_t m0 = 0; m0 < M; m0 += rows) {
            const int mm = (int)(M - m0 < rows ? M - m0 : rows);
            const int rc = awq_launch_gemv_batch_fast(x + m0 * K, qweight, scales, qzeros, y + m0 * N, mm, (int)K, (int)N, (int)group_size,
                                                      (int)group_rows, (int)((AWQ_GEMM_FLAG_SPLITK(flags) & 15u) | (AWQ_GEMM_FLAG_WAVES(flags) << 4)),
                                                      static_cast<hipStream_t>(stream));
            if (rc != AWQ_OK) return rc;
        }
        return AWQ_OK;
    }
    if (M > 16) return AWQ_ERR_UNSUPPORTED;  // the 16-row kernel (the host wrapper chunks)
    if (!awq_gemv_fast_supports((int)M, (int)K, (int)N, (int)group_size)) return AWQ_ERR_UNSUPPORTED;
    g_last_kernel = "gemv_fast";
    return awq_launch_gemv_fast(x, qweight, scales, qzeros, y, (int)M, (int)K, (int)N, (int)group_size, (int)group_rows,
                                (int)AWQ_GEMM_FLAG_WAVES(flags), (int)AWQ_GEMM_FLAG_UNIT(flags),
                                static_cast<hipStream_t>(stream));
}

// Prefill-sized batches on the GEMVFast layout: the packed words are transposed into the caller's temporary (repack.hip), then the
// register-decoded MFMA GEMM runs on it with the format's own scales / fp16 zero terms (gemm_regb.hip, FZ form).  Two launches.
int awq_gemv_fast_prefill(const uint16_t* x, const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros, uint16_t* y,
                          int32_t* qweight_tmp, int64_t M, int64_t K, int64_t N, int64_t group_size, int64_t group_rows, uint32_t flags,
                          void* stream) {
    if (K <= 0 || N < 0 || group_size <= 0 || K % group_size || N % 4 || K % 64) return AWQ_ERR_BAD_SHAPE;
    if (group_rows < K / group_size) return AWQ_ERR_BAD_SHAPE;
    if (M < 0 || M > INT32_MAX || K > INT32_MAX || N > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (M == 0 || N == 0) return AWQ_OK;
    if (!x || !qweight || !scales || !qzeros || !y || !qweight_tmp) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(scales) || !aligned16(qzeros) || !aligned16(y) || !aligned16(qweight_tmp))
        return AWQ_ERR_BAD_ALIGNMENT;
    if (N % 8 || !awq_gemm_regb_supports((int)M, (int)K, (int)N, (int)group_size)) return AWQ_ERR_UNSUPPORTED;
    int rc = awq_repack_gemvfast_to_gemm(qweight, qweight_tmp, K, N, stream);
    if (rc != AWQ_OK) return rc;
    g_last_kernel = "repack_fast+gemm_regb_fz";
    return awq_launch_gemm_regb_fz(x, qweight_tmp, scales, qzeros, y, (int)M, (int)K, (int)N, (int)group_size,
                                   AWQ_GEMM_FLAG_NLOG(flags) == 2 ? 256 : 0, static_cast<hipStream_t>(stream));
}

size_t awq_gemv_fast_lds_bytes_c(int64_t M, int64_t K, int64_t group_size) {
    return awq_gemv_fast_lds_bytes((int)M, (int)K, (int)group_size, 8);
}

int awq_dequantize_weights_gemv_fast(const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros,
                                     uint16_t* out, int64_t K, int64_t N, int64_t group_size, void* stream) {
    if (K < 0 || N < 0 || group_size <= 0 || K % 64 || N % 4 || (K && K % group_size)) return AWQ_ERR_BAD_SHAPE;
    if (K * N == 0) return AWQ_OK;
    if (!qweight || !scales || !qzeros || !out) return AWQ_ERR_NULL;
    return awq_launch_dequant_fast(qweight, scales, qzeros, out, (int)K, (int)N, (int)group_size,
                                   static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---- fused decoder block (decoder.hip)
int awq_rmsnorm_forward(const uint16_t* x, uint16_t* residual, const uint16_t* weight, uint16_t* out, int64_t M, int64_t H,
                        float eps, void* stream) {
    if (M < 0 || H <= 0) return AWQ_ERR_BAD_SHAPE;
    if (M == 0) return AWQ_OK;
    if (!x || !weight || !out) return AWQ_ERR_NULL;
    if (!aligned16(x) || !aligned16(weight) || !aligned16(out) || (residual && !aligned16(residual))) return AWQ_ERR_BAD_ALIGNMENT;
    return awq_launch_rmsnorm(x, residual, weight, out, M, H, eps, static_cast<hipStream_t>(stream));
}

int awq_rope_kv_append(const uint16_t* qkv, uint16_t* q_out, uint16_t* k_cache, uint16_t* v_cache, const float* cos_table,
                       const float* sin_table, const int32_t* pos_dev, int64_t start_pos, int64_t B, int64_t S,
                       int64_t n_heads, int64_t n_kv_heads, int64_t head_dim, int64_t rotary_dim, int64_t max_seq,
                       void* stream) {
    if (B < 0 || S < 0 || B * S > INT32_MAX / 1024 || max_seq < 1 || max_seq > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (B * S == 0) return AWQ_OK;
    if (!qkv || !q_out || !k_cache || !v_cache || (rotary_dim > 0 && (!cos_table || !sin_table))) return AWQ_ERR_NULL;
    return awq_launch_rope_kv_append(qkv, q_out, k_cache, v_cache, cos_table, sin_table, pos_dev, (int)start_pos, (int)B,
                                     (int)S, (int)n_heads, (int)n_kv_heads, (int)head_dim, (int)rotary_dim, (int)max_seq,
                                     static_cast<hipStream_t>(stream));
}

size_t awq_decode_attention_workspace_bytes(int64_t B, int64_t n_heads) {
    if (B <= 0 || n_heads <= 0) return 0;
    return awq_decode_attention_workspace_bytes_impl((int)B, (int)n_heads, 64);
}

int awq_decode_attention(const uint16_t* q, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* out,
                         const int32_t* len_dev, int64_t seq_len, int64_t max_len, int64_t B, int64_t n_heads,
                         int64_t n_kv_heads, int64_t head_dim, int64_t max_seq, float scale, void* workspace,
                         size_t workspace_bytes, void* stream) {
    if (B < 0 || B > 65535 || n_heads < 1 || n_kv_heads < 1 || n_kv_heads > 65535 || max_seq > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (B == 0) return AWQ_OK;
    if (!q || !k_cache || !v_cache || !out) return AWQ_ERR_NULL;
    if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out)) return AWQ_ERR_BAD_ALIGNMENT;
    return awq_launch_decode_attention(q, const_cast<uint16_t*>(k_cache), const_cast<uint16_t*>(v_cache), out, len_dev,
                                       (int)seq_len, (int)max_len, (int)B, (int)n_heads, (int)n_kv_heads, (int)head_dim,
                                       (int)max_seq, scale, workspace, workspace_bytes, nullptr, nullptr,
                                       static_cast<hipStream_t>(stream));
}

int awq_decode_attention_ex(const uint16_t* q, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* out,
                            const int32_t* len_dev, int64_t seq_len, int64_t max_len, int64_t B, int64_t n_heads,
                            int64_t n_kv_heads, int64_t head_dim, int64_t max_seq, float scale, float softcap,
                            const float* alibi_slopes, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || B > 65535 || n_heads < 1 || n_kv_heads < 1 || n_kv_heads > 65535 || max_seq > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (!(softcap >= 0.f)) return AWQ_ERR_BAD_SHAPE;
    if (B == 0) return AWQ_OK;
    if (!q || !k_cache || !v_cache || !out) return AWQ_ERR_NULL;
    if (!aligned16(q) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out)) return AWQ_ERR_BAD_ALIGNMENT;
    return awq_launch_decode_attention(q, const_cast<uint16_t*>(k_cache), const_cast<uint16_t*>(v_cache), out, len_dev,
                                       (int)seq_len, (int)max_len, (int)B, (int)n_heads, (int)n_kv_heads, (int)head_dim,
                                       (int)max_seq, scale, workspace, workspace_bytes, nullptr, nullptr,
                                       static_cast<hipStream_t>(stream), softcap, alibi_slopes);
}

int awq_decode_attention_rope(const uint16_t* qkv, uint16_t* k_cache, uint16_t* v_cache, const float* cos_table,
                              const float* sin_table, uint16_t* out, const int32_t* pos_dev, int64_t start_pos,
                              int64_t max_len, int64_t B, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim,
                              int64_t max_seq, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 0 || B > 65535 || n_heads < 1 || n_kv_heads < 1 || n_kv_heads > 65535 || max_seq > INT32_MAX) return AWQ_ERR_BAD_SHAPE;
    if (B == 0) return AWQ_OK;
    if (!qkv || !k_cache || !v_cache || !out || !cos_table || !sin_table) return AWQ_ERR_NULL;
    if (!aligned16(qkv) || !aligned16(k_cache) || !aligned16(v_cache) || !aligned16(out)) return AWQ_ERR_BAD_ALIGNMENT;
    if (!pos_dev && (start_pos < 0 || start_pos >= max_seq)) return AWQ_ERR_BAD_SHAPE;
    return awq_launch_decode_attention(qkv, k_cache, v_cache, out, pos_dev, (int)start_pos, (int)max_len, (int)B, (int)n_heads,
                                       (int)n_kv_heads, (int)head_dim, (int)max_seq, scale, workspace, workspace_bytes,
                                       cos_table, sin_table, static_cast<hipStream_t>(stream));
}
