// gemv_mfma.hip -- decode GEMV / skinny GEMM (1 <= M <= 16) on the GEMM layout, gfx950.
//
// Replaces awq_ext.gemm_forward_cuda for small M (awq/modules/linear/gemm.py:56-58,
// awq/modules/fused/mlp.py:41,49-62) and is the engine of the MoE grouped GEMM's 16-row token
// blocks (awq/modules/fused/moe.py:60-89).
//
// Roofline: HBM.  Algorithmic bytes per call:
//     K*N/2 + (K/g)*(N/8)*4 + (K/g)*N*2 + M*K*2 + M*N*2 (+ N*2 with bias).
// The matrix cores are used for their plumbing, not their flops: they take the multiply, the
// accumulate AND the cross-lane K reduction off the VALU, whose whole budget then goes to
// decoding int4 (one shift + one v_and_or per column PAIR).
//
// How a packed word feeds v_mfma_f32_16x16x32_f16 with no transposition and no shuffles
//   GEMM layout packs 8 N-adjacent weights of ONE row k per int32; nibbles (J, J+4) are logical
//   columns (2J, 2J+1) (awq/utils/packing_utils.py:4-43).  ((q >> 4J) << 6) & 0x03C003C0 |
//   0x4C004C00 is the fp16 pair (16 + w[k, 2J], 16 + w[k, 2J+1]), exact.  Four such pairs from
//   four rows k0..k3 fill a lane's B fragment; its 8 K slots are (k0:a, k0:b, k1:a, ... k3:b).
//   The A fragment carries the activations.  Row i of A is a SELECTOR: row 2m+e holds x[m, k_r]
//   in the slots of column e of every pair and 0 in the others, so D[2m+e][j] is the dot product
//   of batch row m with column (2J+e) of lane j's word -- one MFMA yields both columns of the
//   pair for up to 8 batch rows.  For 9..16 rows A row i is batch row i and two MFMAs (even-slot
//   and odd-slot A) are issued per B fragment.
//   Group factorisation keeps it exact in fp32: with sx = sum of x over the rows since the last
//   fold (one extra MFMA against an all-ones B),  y += s * (acc - (16 + z) * sx).  A fold
//   happens at every group end and at least every 128 rows (bounds the cancellation).
//
// Decomposition
//   lane (j = l & 15, kb = l >> 4) owns WPL packed words (8*WPL columns); a wave covers
//   CW = 128*WPL columns; one MFMA set = 16 consecutive rows (lane rows 4*kb .. 4*kb+3); one loop
//   iteration = SETS sets (128 rows when g % 128 == 0, else 64) = 4*SETS independent loads per
//   lane, ALL issued back to back before any is consumed (no software pipeline inside a wave: the
//   2-4 co-resident waves of a SIMD overlap each other's memory and MFMA phases), then FOLDS
//   group folds.
//   The NWAVES waves of a block take consecutive row ranges of one column tile and are folded
//   through LDS; K is further split over S blocks per tile and combined in-launch in ONE fabric
//   hop through self-validating write-through slabs (see the combine code): producers store and
//   exit, the block of the last K slice polls, sums in slice order (bitwise reproducible), writes
//   fp16 and re-arms.  (A drain + ticket + last-arriver design measured three serial fabric round
//   trips, ~3 us, per call: profiles/ notes.)
#include "awq_device.h"
#include "awq_internal.h"
#include "awq_mfma_decode.h"

namespace {

struct GemvMfmaParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    int x_gated;              // activation mode: 0 plain, 1 rows are [gate | up] (stage silu(gate) * up), 2 RMSNorm
    const half_t* res_in;     // mode 2: optional residual rows added to x before the norm ...
    half_t* res_out;          // ... and where fp16(x + residual) is written (by ONE block; must not alias res_in)
    const half_t* norm_w;     // mode 2: norm weight [K]
    float norm_eps;
    const float* ssq_in;      // mode 2: [M, ssq_in_tiles] partial row sums of squares handed over by the producer
    int ssq_in_tiles;         //         (replaces the statistic pass; x already is the stream)
    const half_t* add_res;    // epilogue: y = fp16(fp16(x W + bias) + add_res)   [M, N]
    float* ssq_out;           // epilogue: [M, tiles] sums of squares of the y values of each 256-column tile
    const half_t* bias;
    half_t* y;
    float* slabs;    // in-launch exchange region [S-1][tiles][M][CW] fp32: all-ones sentinel on entry and on exit
    float* scratch;  // two-pass mode: plain fp32 slabs [S][M][N]
    int* err;        // set to 1 if a reducer gave up waiting
    int M, K, N, g;
    int tiles, S;
    int rows_per_block;  // K slice per block: a whole number of 16*UNIT-row units
    int ng_max;          // groups a slice can touch (sizes the LDS staging area)
    int combine;         // split-K combine: 0 one reducer block per tile, 1 two-pass (plain slabs, a second kernel reduces),
                         // 2 / 3: two / four reducer blocks per tile
    // ---- grouped (MoE) mode: one 16-row token block per blockIdx / (tiles*S), each with its own
    // expert's weights (awq/modules/fused/moe.py:60-89).  All null / 0 in the plain mode.
    const int* sorted_ids;    // [nblk*16] (token, expert) pair index per row, >= num_pairs = padding
    const int* expert_ids;    // [nblk] expert of each 16-row block
    const int* num_post_pad;  // device scalar: rows in use (multiple of 16)
    const float* pair_weights;  // [num_pairs] routing weights, applied to the output if non-null
    int num_pairs, x_div;     // activation row of pair i = i / x_div
    int64_t expert_qw_words, expert_z_words, expert_s_halfs;  // per-expert strides
    // (k * g_magic) >> 32 == k / g and (c * xc_magic) >> 32 == c / (rows_per_block / 8) for every k < K, c <= 17 * rows_per_block / 8
    // (checked by the launcher): no integer division in the prologue, the staging or the K loop.  Last in the struct: see above.
    uint32_t g_magic, xc_magic;
};

template <int WPL>
struct Words;
template <>
struct Words<2> { typedef u32x2 T; };
template <>
struct Words<4> { typedef u32x4 T; };

template <int WPL, int AUX>
AWQ_DEV typename Words<WPL>::T ld_words(rsrc_t r, uint32_t voff, uint32_t soff) {
    if constexpr (WPL == 2)
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    else
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}

// fp16 pair (16 + col 2J, 16 + col 2J+1) of a packed word
template <int J>
AWQ_DEV uint32_t pair16(uint32_t q) {
    constexpr int SH = 6 - 4 * J;
    const uint32_t t = SH >= 0 ? (q << (SH >= 0 ? SH : 0)) : (q >> (SH < 0 ? -SH : 0));
    return and_or(t, 0x03C003C0u, 0x4C004C00u);
}

// Phase timestamps for tools/trace_gemv.py (debug build only: -DAWQ_GEMV_TRACE)
#ifdef AWQ_GEMV_TRACE
__device__ unsigned long long* g_awq_trace = nullptr;
#define AWQ_STAMP(slot)                                                                                  \
    do {                                                                                                 \
        if (g_awq_trace && lane == 0)                                                                    \
            g_awq_trace[((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NWAVES + wave) * 16 + (slot)] = wall_clock64();             \
    } while (0)
#else
#define AWQ_STAMP(slot) do { } while (0)
#endif

// SEL: selector-row A (M <= 8, one MFMA per B fragment); !SEL: M <= 16, two MFMAs.
// NREG: live D registers per lane (2 when M == 1, else 4).  UNIT: 16-row sets a wave streams per
// loop iteration (4*UNIT loads per lane in flight); FOLDS: group folds per unit (UNIT*16/FOLDS
// rows each: a divisor of g, <= 128).
// XMODE (a template parameter, not a runtime flag: the plain instantiations must not carry the extra
// registers): 1 = x rows are [gate | up] of 2K halves and silu(gate) * up is applied while staging;
// 2 = x (+ residual) is RMS-normalised while staging -- every block recomputes the row statistic from
// the (L2-resident) row, M <= 4 -- and block (tile 0, slice 0) writes fp16(x + residual) out.
template <int WPL, int NWAVES, int UNIT, bool SEL, int NREG, int FOLDS, bool NT, bool MOE = false, int XMODE = 0>
__global__ __launch_bounds__(NWAVES * 64) void awq_gemv_mfma_kernel(GemvMfmaParams p) {
    constexpr bool GATED = XMODE == 1, NORM = XMODE == 2;
    typedef typename Words<WPL>::T WV;
    constexpr int CPL = 8 * WPL;         // columns per lane
    constexpr int CW = 16 * CPL;         // columns per wave == per block tile
    constexpr int CWP = CW + 16;         // padded LDS row (one float per lane)
    constexpr int NACC = 4 * WPL;        // (word, J) pairs per lane
    constexpr int NA = SEL ? 1 : 2;      // accumulators per pair
    constexpr int AUXW = NT ? 2 : 0;
    constexpr int NTHR = NWAVES * 64;
    constexpr int SPF = UNIT / FOLDS;    // sets per fold
    // dynamic LDS: [xs: (M+1) x RS fp16][zq: ng x CW/8 u32][zsc: ng x CW fp16] during the K loop,
    // re-used as red[NWAVES][M][CWP] fp32 afterwards
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kb = lane >> 4;
    // grid = (tiles, K slices, token blocks): the hardware hands over all three indices, no integer division in the prologue
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int tblk = MOE ? blockIdx.z : 0;  // 16-row token block (grouped mode)
    if constexpr (MOE) {
        if (p.M * tblk >= *p.num_post_pad) return;  // uniform for every block of this token block (p.M rows each)
        const int64_t e = p.expert_ids[tblk];
        p.qweight += e * p.expert_qw_words;
        p.qzeros += e * p.expert_z_words;
        p.scales += e * p.expert_s_halfs;
    }
    const int NW = p.N >> 3;
    const int colw = (tile * 16 + j) * WPL;  // first packed word of this lane
    const bool active = colw < NW;
    const int M = p.M;
    AWQ_STAMP(0);

    // K slice of this block: rows [r0, r1), a whole number of units
    const int r0 = slice * p.rows_per_block;
    const int r1 = min(p.K, r0 + p.rows_per_block);
    const int RS = p.rows_per_block;             // LDS row pitch of xs (multiple of 16*UNIT)
    const int g0 = (int)__umulhi((uint32_t)r0, p.g_magic);                    // first group of the slice
    const int ng = (int)__umulhi((uint32_t)(r1 - 1), p.g_magic) - g0 + 1;     // groups touched by the slice
    half_t* xs = reinterpret_cast<half_t*>(smem);
    uint32_t* zq = reinterpret_cast<uint32_t*>(smem + (size_t)(M + 1) * RS * 2);
    half_t* zsc = reinterpret_cast<half_t*>(reinterpret_cast<unsigned char*>(zq) + (size_t)p.ng_max * (CW / 2));

    const uint32_t row_bytes = (uint32_t)NW * 4u;
    const rsrc_t wres = mk_rsrc(p.qweight, (uint32_t)p.K * row_bytes);

    // ---- stage this block's activations, zeros and scales in LDS (block-cooperative, 16-byte
    // chunks): the K loop then uses the vector-memory path for packed weights ONLY.
    // Decode-sized blocks need at most one chunk of each kind per thread: those are REQUESTED here
    // into registers and written to LDS only after the wave has issued its first unit of weight
    // loads (below), so the staging round trip overlaps the weight stream instead of preceding it.
    constexpr int QC = CW / 32;  // 16-byte chunks of packed zeros per group row of the tile
    constexpr int SC = CW / 8;   // 16-byte chunks of scales per group row of the tile
    const int xchunks = RS >> 3;  // 16-byte chunks per activation row
    const bool reg_staged = !NORM && (M + 1) * xchunks <= NTHR && ng * SC <= NTHR;
    u32x4 st_x = {0u, 0u, 0u, 0u}, st_q = st_x, st_s = st_x, st_u = st_x;
    auto silu_mul = [](u32x4 gate, u32x4 upv) -> u32x4 {  // fp32, one rounding: == awq_silu_and_mul_kernel
        const half8_t gt = __builtin_bit_cast(half8_t, gate), uu = __builtin_bit_cast(half8_t, upv);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = (half_t)awq_silu_mul_f32((float)gt[e], (float)uu[e]);
        }
        return __builtin_bit_cast(u32x4, o);
    };
    auto x_chunk = [&](int c, bool up = false) -> u32x4 {
        const int m = (int)__umulhi((uint32_t)c, p.xc_magic), cc = c - m * xchunks;
        const int row = r0 + 8 * cc;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m < M && row < r1) {
            int64_t xrow = m;
            bool ok = true;
            if constexpr (MOE) {  // gather: row m of the block is pair sorted_ids[M*tblk + m]
                const int pid = p.sorted_ids[p.M * tblk + m];
                ok = pid < p.num_pairs;
                xrow = pid / p.x_div;
            }
            // x_gated (fused-MLP down projection): rows are [gate | up] of 2K halves; `up` selects the half
            if (ok) v = *reinterpret_cast<const u32x4*>(p.x + xrow * (GATED ? 2 * p.K : p.K) + (up ? p.K : 0) + row);
        }
        return v;
    };
    auto q_chunk = [&](int c) -> u32x4 {
        const int gl = c / QC, cc = c % QC;
        const int w0 = tile * (CW / 8) + 4 * cc;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (w0 < NW) v = *reinterpret_cast<const u32x4*>(p.qzeros + (int64_t)(g0 + gl) * NW + w0);  // N % 32 == 0
        return v;
    };
    auto s_chunk = [&](int c) -> u32x4 {
        const int gl = c / SC, cc = c % SC;
        const int col = tile * CW + 8 * cc;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (col < p.N) v = *reinterpret_cast<const u32x4*>(p.scales + (int64_t)(g0 + gl) * p.N + col);
        return v;
    };
    auto x_store = [&](int c, u32x4 v) {
        const int m = (int)__umulhi((uint32_t)c, p.xc_magic);
        *reinterpret_cast<u32x4*>(xs + (size_t)m * RS + 8 * (c - m * xchunks)) = v;
    };
    auto q_store = [&](int c, u32x4 v) { *reinterpret_cast<u32x4*>(zq + (c / QC) * (CW / 8) + 4 * (c % QC)) = v; };
    auto s_store = [&](int c, u32x4 v) { *reinterpret_cast<u32x4*>(zsc + (c / SC) * CW + 8 * (c % SC)) = v; };
    constexpr bool LATE = NREG == 4 || NORM;  // batch > 1 (the plain M = 1 instantiations keep their register budget)
    if (reg_staged) {
        if (tid < (M + 1) * xchunks) {
            st_x = x_chunk(tid);
            if (GATED) st_u = x_chunk(tid, true);  // the activation itself waits until the weights are requested
        }
        if (tid < ng * QC) st_q = q_chunk(tid);
        if (tid < ng * SC) st_s = s_chunk(tid);
    } else if (!LATE) {
        for (int c = tid; c < (M + 1) * xchunks; c += NTHR) x_store(c, GATED ? silu_mul(x_chunk(c), x_chunk(c, true)) : x_chunk(c));
        for (int c = tid; c < ng * QC; c += NTHR) q_store(c, q_chunk(c));
        for (int c = tid; c < ng * SC; c += NTHR) s_store(c, s_chunk(c));
    }
    // Larger batches (several chunks per thread) stage through the same point of the schedule --
    // AFTER the first unit's weight requests are in flight -- so the staging round trip is hidden
    // behind the weight stream instead of preceding it (M = 16, 4096 x 11008: 23.7 -> 18.1 us,
    // which also brought the M > 8 instantiations back to two waves per SIMD).
    auto finish_staging = [&]() {
        if (reg_staged) {
            if (tid < (M + 1) * xchunks) x_store(tid, GATED ? silu_mul(st_x, st_u) : st_x);
            if (tid < ng * QC) q_store(tid, st_q);
            if (tid < ng * SC) s_store(tid, st_s);
        } else if (LATE) {
            if constexpr (NORM) {
                __shared__ float nrm_part[4][8];  // [row][wave]
                __shared__ float nrm_inv[4];
                auto load_h = [&](int m, int col) -> half8_t {  // fp16(x + residual), the value the stream carries
                    half8_t h = *reinterpret_cast<const half8_t*>(p.x + (int64_t)m * p.K + col);
                    if (p.res_in) {
                        const half8_t rr = *reinterpret_cast<const half8_t*>(p.res_in + (int64_t)m * p.K + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (half_t)((float)h[e] + (float)rr[e]);
                    }
                    return h;
                };
                const bool writer = p.res_out != nullptr && tile == 0 && slice == 0;
                const int kch = p.K >> 3;
                if (p.ssq_in) {  // the producing projection handed the statistic over: wave m sums row m's tile partials
                    if (wave < M) {
                        float part = lane < p.ssq_in_tiles ? p.ssq_in[wave * p.ssq_in_tiles + lane] : 0.f;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                        if (lane == 0) nrm_part[wave][0] = part;
                    }
                } else
                for (int m = 0; m < M; ++m) {  // M <= 4: the row statistic, recomputed by every block
                    float ss = 0.f;
                    for (int c = tid; c < kch; c += NTHR) {
                        const half8_t h = load_h(m, 8 * c);
                        if (writer) *reinterpret_cast<half8_t*>(p.res_out + (int64_t)m * p.K + 8 * c) = h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss += (float)h[e] * (float)h[e];
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                    if (lane == 0) nrm_part[m][wave] = ss;
                }
                __syncthreads();
                if (tid < M) {
                    float tot = 0.f;
                    for (int w = 0; w < (p.ssq_in ? 1 : NWAVES); ++w) tot += nrm_part[tid][w];
                    nrm_inv[tid] = rsqrtf(tot / (float)p.K + p.norm_eps);
                }
                __syncthreads();
                for (int c = tid; c < (M + 1) * xchunks; c += NTHR) {
                    const int m = (int)__umulhi((uint32_t)c, p.xc_magic), row = r0 + 8 * (c - m * xchunks);
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (m < M && row < r1) {
                        const half8_t h = load_h(m, row);
                        const half8_t gw = *reinterpret_cast<const half8_t*>(p.norm_w + row);
                        const float inv = nrm_inv[m];
                        half8_t o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)h[e] * inv * (float)gw[e]);  // == awq_rmsnorm_kernel
                        v = __builtin_bit_cast(u32x4, o);
                    }
                    x_store(c, v);
                }
            } else {
                for (int c = tid; c < (M + 1) * xchunks; c += NTHR) x_store(c, GATED ? silu_mul(x_chunk(c), x_chunk(c, true)) : x_chunk(c));
            }
            for (int c = tid; c < ng * QC; c += NTHR) q_store(c, q_chunk(c));
            for (int c = tid; c < ng * SC; c += NTHR) s_store(c, s_chunk(c));
        }
        __syncthreads();
    };

    float yv[NACC][NA][NREG];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < NREG; ++r) yv[c][a][r] = 0.f;

    {
        const uint32_t wvoff = active ? (uint32_t)colw * 4u + (uint32_t)(4 * kb) * row_bytes : OOB;
        // A row of this lane: SEL -> batch row j >> 1, column parity j & 1; else batch row j.
        // Lanes whose A row is >= M read the all-zero row M of xs.
        const int arow = min(SEL ? (j >> 1) : j, M);
        const half_t* xlane = xs + (size_t)arow * RS + 4 * kb;
        const uint32_t sel_lo = (j & 1) ? 0x01000C0Cu : 0x0C0C0100u;  // low half of a dword -> slot (j & 1)
        const uint32_t sel_hi = (j & 1) ? 0x03020C0Cu : 0x0C0C0302u;  // high half
        const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
        const int nunits = (r1 - r0) / (16 * UNIT);
        bool staged = false;
        AWQ_STAMP(7);

        for (int u = wave; u < nunits; u += NWAVES) {
            // ---- request the unit's packed weights: 4*UNIT independent loads per lane
            WV q[UNIT][4];
            const uint32_t urow = (uint32_t)__builtin_amdgcn_readfirstlane(r0 + u * 16 * UNIT);
            uint32_t soff = urow * row_bytes;
#pragma unroll
            for (int t = 0; t < UNIT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[t][r] = ld_words<WPL, AUXW>(wres, wvoff, soff);
                    soff += row_bytes;
                }
                soff += 12u * row_bytes;
            }
            __builtin_amdgcn_sched_barrier(0);  // all requests are issued before anything is consumed
            if (u == wave) AWQ_STAMP(1);
            if (!staged) {  // first unit: complete the staging, make it visible block-wide
                finish_staging();
                staged = true;
            }

            // ---- consume
            const int lrow = (int)urow - r0;  // slice-local first row of the unit
#pragma unroll
            for (int f = 0; f < FOLDS; ++f) {
                float4_t acc[NACC][NA];
                float4_t accsx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NACC; ++c)
#pragma unroll
                    for (int a = 0; a < NA; ++a) acc[c][a] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = f * SPF; t < (f + 1) * SPF; ++t) {
                    const u32x2 xq = *reinterpret_cast<const u32x2*>(xlane + lrow + 16 * t);
                    const uint32_t x01 = xq[0], x23 = xq[1];
                    u32x4v a0, a1;
                    if constexpr (SEL) {
                        a0 = u32x4v{__builtin_amdgcn_perm(0u, x01, sel_lo), __builtin_amdgcn_perm(0u, x01, sel_hi),
                                    __builtin_amdgcn_perm(0u, x23, sel_lo), __builtin_amdgcn_perm(0u, x23, sel_hi)};
                        a1 = a0;
                    } else {
                        a0 = u32x4v{x01 & 0xFFFFu, x01 >> 16, x23 & 0xFFFFu, x23 >> 16};
                        a1 = u32x4v{x01 << 16, x01 & 0xFFFF0000u, x23 << 16, x23 & 0xFFFF0000u};
                    }
                    accsx = mfma16(a0, ones, accsx);
#pragma unroll
                    for (int wd = 0; wd < WPL; ++wd) {
                        const uint32_t q0 = q[t][0][wd], q1 = q[t][1][wd], q2 = q[t][2][wd], q3 = q[t][3][wd];
                        const uint32_t h0 = q0 >> 8, h1 = q1 >> 8, h2 = q2 >> 8, h3 = q3 >> 8;
#define AWQ_MMA_J(J)                                                                                     \
    {                                                                                                    \
        const u32x4v bf = {pairb<J>(q0, h0), pairb<J>(q1, h1), pairb<J>(q2, h2), pairb<J>(q3, h3)};      \
        acc[wd * 4 + J][0] = mfma16(a0, bf, acc[wd * 4 + J][0]);                                         \
        if constexpr (!SEL) acc[wd * 4 + J][1] = mfma16(a1, bf, acc[wd * 4 + J][1]);                     \
    }
                        AWQ_MMA_J(0)
                        AWQ_MMA_J(1)
                        AWQ_MMA_J(2)
                        AWQ_MMA_J(3)
#undef AWQ_MMA_J
                    }
                }
                // y += s * (acc - (16 + z) * sx) over the rows of this fold (inside one group)
                const int gl = (int)__umulhi(urow + 16u * SPF * f, p.g_magic) - g0;
                const WV qzv = *reinterpret_cast<const WV*>(zq + gl * (CW / 8) + j * WPL);
#pragma unroll
                for (int wd = 0; wd < WPL; ++wd) {
                    const u32x4 sv = *reinterpret_cast<const u32x4*>(zsc + gl * CW + j * CPL + 8 * wd);
                    const uint32_t zw = qzv[wd];
                    const uint32_t zw8 = zw >> 8;  // (bias_J + z) pairs, the same biases as the weights
                    const uint32_t zp[4] = {pairb<0>(zw, zw8), pairb<1>(zw, zw8), pairb<2>(zw, zw8), pairb<3>(zw, zw8)};
#pragma unroll
                    for (int J = 0; J < 4; ++J) {
                        const half2_t z2 = u2h2(zp[J]), s2 = u2h2(sv[J]);
                        const int c = wd * 4 + J;
#pragma unroll
                        for (int a = 0; a < NA; ++a)
#pragma unroll
                            for (int r = 0; r < NREG; ++r) {
                                const int e = SEL ? (r & 1) : a;  // column parity this register belongs to
                                const float raw = __builtin_fmaf(-(float)z2[e], accsx[r], acc[c][a][r]);
                                yv[c][a][r] = __builtin_fmaf((float)s2[e], raw, yv[c][a][r]);
                            }
                    }
                }
            }
            if (u == wave) AWQ_STAMP(8);
        }
        if (!staged) finish_staging();  // waves without a unit still take part in the staging
    }
    __syncthreads();  // every wave is done with xs / zq / zsc: the LDS becomes red[]
    float* red = reinterpret_cast<float*>(smem);

    AWQ_STAMP(2);
    // ---- fold the waves of the block through LDS: red[wave][m][col]
    {
        float* mine = red + wave * M * CWP;
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    const int i = 4 * kb + r;  // D row
                    const int m = SEL ? (i >> 1) : i;
                    const int e = SEL ? (i & 1) : a;
                    const int col = j * CPL + (c >> 2) * 8 + 2 * (c & 3) + e;
                    if (m < M) mine[m * CWP + col + j] = yv[c][a][r];
                }
    }
    __syncthreads();
    AWQ_STAMP(3);

    const int quads = M * (CW / 4);  // float4 groups of the block's [M][CW] partial tile
    const int col0 = tile * CW;
    auto block_sum4 = [&](int qd) -> float4_t {
        const int m = qd / (CW / 4), c4 = (qd % (CW / 4)) * 4;
        float4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            const float* src = red + (w * M + m) * CWP + c4 + (c4 / CPL);
            s += float4_t{src[0], src[1], src[2], src[3]};
        }
        return s;
    };
    auto emit4 = [&](int qd, float4_t s) -> float {  // returns the sum of squares of the four stored values
        const int m = qd / (CW / 4), c4 = (qd % (CW / 4)) * 4;
        const int col = col0 + c4;
        if (col >= p.N) return 0.f;  // N % 8 == 0: a quad is all in or all out
        int64_t orow = m;
        if constexpr (MOE) {  // scatter to the pair's row, optionally scaled by its routing weight
            const int pid = p.sorted_ids[p.M * tblk + m];
            if (pid >= p.num_pairs) return 0.f;
            orow = pid;
            if (p.pair_weights) s *= p.pair_weights[pid];
        }
        if (p.bias) {
            const half4_t b4 = *reinterpret_cast<const half4_t*>(p.bias + col);
            s += float4_t{(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
        }
        half4_t o = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
        if (p.add_res) {  // the residual stream: fp16(fp16(projection) + residual), the two roundings torch makes
            const half4_t r4 = *reinterpret_cast<const half4_t*>(p.add_res + orow * p.N + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)o[e] + (float)r4[e]);
        }
        *reinterpret_cast<half4_t*>(p.y + orow * p.N + col) = o;
        return (float)o[0] * (float)o[0] + (float)o[1] * (float)o[1] + (float)o[2] * (float)o[2] + (float)o[3] * (float)o[3];
    };
    // ssq_out (M <= 4, 256-column tiles): quads of row m are threads 64 m .. 64 m + 63 = wave m, so one
    // butterfly per wave gives the tile's partial sum of squares of that row in a fixed order
    auto emit_ssq = [&](float ss) {
        if (p.ssq_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (lane == 0 && wave < M) p.ssq_out[wave * p.tiles + tile] = ss;
        }
    };

    const int S = p.S;
    if (S == 1) {
        float ss = 0.f;
        for (int qd = tid; qd < quads; qd += NWAVES * 64) ss += emit4(qd, block_sum4(qd));
        emit_ssq(ss);
        return;
    }
    if (NREG == 4 ? p.combine == 1 : p.combine != 0) {  // plain fp32 slabs [S][M][N]; a second kernel reduces
        for (int qd = tid; qd < quads; qd += NWAVES * 64) {
            const int m = qd / (CW / 4), col = col0 + (qd % (CW / 4)) * 4;
            if (col < p.N)
                *reinterpret_cast<float4_t*>(p.scratch + ((int64_t)slice * M + m) * p.N + col) = block_sum4(qd);
        }
        return;
    }

    // ---- in-launch combine, ONE fabric hop: self-validating slabs.
    // The exchange region is kept filled with a sentinel (all bits set, a NaN no arithmetic here
    // produces).  A producer block stores its fp32 partial tile with 16-byte write-through stores
    // and exits: no drain, no flag, no ticket -- every aligned 4-byte word is its own "ready"
    // signal, so torn 16-byte stores are harmless.  The blocks of the LAST R K slices of a tile are
    // its reducers (R = 1 up to four batch rows, 2 or 4 above), each for 1/R of the tile's quads: a
    // reducer stores its partial for the quads the others own, polls the other slices' words of ITS
    // quads with agent-scope (sc1) loads until none is the sentinel, adds all S partials in slice
    // order (bitwise reproducible, and independent of R), writes fp16 and re-arms the words it
    // consumed.  One reducer per tile spent 2.8 us (M = 8) to 5 us (M = 16) of serial polling over
    // 512 / 1024 quads (profiles/r01_gemv_phase_trace_by_m.txt); R of them take one round trip each.
    // Producers never wait; the reducers are the last blocks in dispatch order and wait only for
    // producers and for reducers dispatched BEFORE or right after them (R x tiles <= 256 blocks, so
    // every one of them gets a slot while the others spin); every spin is bounded and raises *err
    // instead of hanging.  The R = 1 path is kept as its own, older code: routed through the general
    // loop below the M = 1 launches lose 5 % (producer phase 0.24 -> 0.40 us, r02 trace A/B).
    constexpr uint32_t SENT = 0xFFFFFFFFu, QNAN = 0x7FC00000u;
    const uint32_t slab_bytes = (uint32_t)quads * 16u;
    // NOTE: the statements of the one-reducer path below are deliberately left exactly as they were tuned: this
    // kernel's M = 1 time moves by 1-2 % with source-neutral rearrangements here (r02 A/B: an extra `if` around it,
    // S instead of S - 1 in the two lines below: 866 -> 846 / 854 tok/s on the headline), code placement, not work.
    if (NREG != 4 || p.combine == 0) {  // ---- one reducer: the block of the tile's LAST K slice (its own partial never leaves the block)
        const uint32_t tb0 = (uint32_t)tblk * (uint32_t)(S - 1) * (uint32_t)p.tiles;  // this token block's slabs: [S-1][tiles]
        const rsrc_t slres = mk_rsrc(p.slabs, (uint32_t)(gridDim.z * (S - 1)) * (uint32_t)p.tiles * slab_bytes);
        if (slice != S - 1) {
            for (int qd = tid; qd < quads; qd += NWAVES * 64) {
                u32x4 b = __builtin_bit_cast(u32x4, block_sum4(qd));
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (b[e] == SENT) ? QNAN : b[e];
                // soffset must stay the constant 0 on 16-byte buffer stores (see gemm_tiled.hip: with
                // an SGPR soffset no wait states are inserted before the data VGPRs are rewritten)
                __builtin_amdgcn_raw_buffer_store_b128(b, slres, (uint32_t)qd * 16u + (tb0 + (uint32_t)(slice * p.tiles + tile)) * slab_bytes,
                                                       0, 16 /* sc1 */);
            }
            AWQ_STAMP(4);
            return;
        }
        const u32x4 sent4 = {SENT, SENT, SENT, SENT};
        float ssq_acc = 0.f;
        for (int qd = tid; qd < quads; qd += NWAVES * 64) {
            const float4_t own = block_sum4(qd);
            float4_t s = {0.f, 0.f, 0.f, 0.f};
            for (int sl0 = 0; sl0 < S - 1; sl0 += 8) {  // 8 independent 16-byte loads in flight per poll
                u32x4 v[8];
                for (unsigned spins = 0;; ++spins) {
                    uint32_t pending = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)  // slices past S-1 are requested out of range: zeros, no traffic
                        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                             slres, (sl0 + u < S - 1) ? (uint32_t)qd * 16u : OOB,
                                                             (tb0 + (uint32_t)((sl0 + u) * p.tiles + tile)) * slab_bytes, 16));
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        pending |= (v[u][0] == SENT) | (v[u][1] == SENT) | (v[u][2] == SENT) | (v[u][3] == SENT);
                    if (!pending) break;
                    if (spins > (1u << 18)) {  // give up: flag the error, use what is there
                        *p.err = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += __builtin_bit_cast(float4_t, v[u]);  // fixed order: bitwise reproducible
#pragma unroll
                for (int u = 0; u < 8; ++u)  // re-arm (write-through; also drops the line from this XCD's L2)
                    if (sl0 + u < S - 1)
                        __builtin_amdgcn_raw_buffer_store_b128(sent4, slres, (uint32_t)qd * 16u + (tb0 + (uint32_t)((sl0 + u) * p.tiles + tile)) * slab_bytes,
                                                               0, 16);
            }
            ssq_acc += emit4(qd, s + own);
        }
        emit_ssq(ssq_acc);
        AWQ_STAMP(5);
        return;
    }
    if constexpr (NREG == 4) {  // batch rows > 1 only: the M = 1 instantiations do not carry this code
        const uint32_t tb0 = (uint32_t)tblk * (uint32_t)S * (uint32_t)p.tiles;  // this token block's slabs: [S][tiles]
        const rsrc_t slres = mk_rsrc(p.slabs, (uint32_t)(gridDim.z * S) * (uint32_t)p.tiles * slab_bytes);
        // Reducer ranges are whole 64-quad blocks (quads = 64 M with 256-column tiles), so a wave is either all
        // "store" or all "reduce" in an iteration; a reducer walks the quads starting AFTER its own range, i.e.
        // stores everything the other reducers wait for before it starts waiting itself.
        const int rlog = p.combine - 1;           // 2 or 4 reducers (a shift: an integer division here costs every block ~0.3 us)
        const int ridx = slice - (S - (1 << rlog));   // >= 0: this block reduces quads [q0, q1)
        const int q64 = quads >> 6;
        const int q0 = ridx >= 0 ? ((q64 * ridx) >> rlog) << 6 : quads;
        const int q1 = ridx >= 0 ? (ridx + 1 == (1 << rlog) ? quads : ((q64 * (ridx + 1)) >> rlog) << 6) : quads;
        const int start = q1 >= quads ? 0 : q1;
        const uint32_t my_slab = (tb0 + (uint32_t)(slice * p.tiles + tile)) * slab_bytes;
        const u32x4 sent4 = {SENT, SENT, SENT, SENT};
        float ssq_acc = 0.f;
        for (int i = tid; i < quads; i += NWAVES * 64) {
            int qd = start + i;
            if (qd >= quads) qd -= quads;
            u32x4 own = __builtin_bit_cast(u32x4, block_sum4(qd));
            if (qd < q0 || qd >= q1) {  // somebody else reduces this quad
#pragma unroll
                for (int e = 0; e < 4; ++e) own[e] = (own[e] == SENT) ? QNAN : own[e];
                // soffset must stay the constant 0 on 16-byte buffer stores (see gemm_tiled.hip: with
                // an SGPR soffset no wait states are inserted before the data VGPRs are rewritten)
                __builtin_amdgcn_raw_buffer_store_b128(own, slres, (uint32_t)qd * 16u + my_slab, 0, 16 /* sc1 */);
                continue;
            }
            float4_t s = {0.f, 0.f, 0.f, 0.f};
            // the S - 1 OTHER slices, k = 0 .. S-2 -> slice k + (k >= own): no poll round is spent on this block's own slot
            for (int k0 = 0; k0 < S - 1; k0 += 8) {  // 8 independent 16-byte loads in flight per poll
                u32x4 v[8];
                uint32_t soff[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)  // wave-uniform by construction; say so, or the loads are wrapped in waterfall loops
                    soff[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((tb0 + (uint32_t)((k0 + u + (k0 + u >= slice ? 1 : 0)) * p.tiles + tile)) * slab_bytes));
                for (unsigned spins = 0;; ++spins) {
                    uint32_t pending = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)  // past the last slice: requested out of range, zeros and no traffic
                        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(slres, (k0 + u < S - 1) ? (uint32_t)qd * 16u : OOB, soff[u], 16));
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        pending |= (v[u][0] == SENT) | (v[u][1] == SENT) | (v[u][2] == SENT) | (v[u][3] == SENT);
                    if (!pending) break;
                    if (spins > (1u << 18)) {  // give up: flag the error, use what is there
                        *p.err = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // slice order, this block's own partial at its place: bitwise reproducible for every R
                    if (k0 + u == slice && k0 + u < S - 1) s += __builtin_bit_cast(float4_t, own);
                    s += __builtin_bit_cast(float4_t, v[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)  // re-arm (write-through; also drops the line from this XCD's L2)
                    if (k0 + u < S - 1) __builtin_amdgcn_raw_buffer_store_b128(sent4, slres, (uint32_t)qd * 16u + soff[u], 0, 16);
            }
            if (slice == S - 1) s += __builtin_bit_cast(float4_t, own);
            ssq_acc += emit4(qd, s);
        }
        if (ridx < 0) {
            AWQ_STAMP(4);
            return;
        }
        emit_ssq(ssq_acc);
        AWQ_STAMP(5);
    }
}

// y[m, n] = fp16( sum_s slabs[s, m, n] + bias[n] )   (two-pass mode)
__global__ __launch_bounds__(256) void awq_gemv_mfma_reduce_kernel(const float* __restrict__ slabs,
                                                                   const half_t* __restrict__ bias,
                                                                   half_t* __restrict__ y, int MN, int N, int S) {
    const int i4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= MN) return;
    float4_t s = {0.f, 0.f, 0.f, 0.f};
    for (int sp0 = 0; sp0 < S; sp0 += 8) {  // 8 independent loads in flight, summed in slab order
        float4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sp = sp0 + u < S ? sp0 + u : S - 1;
            v[u] = *reinterpret_cast<const float4_t*>(slabs + (int64_t)sp * MN + i4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (sp0 + u < S) s += v[u];
    }
    if (bias) {
        const half4_t b4 = *reinterpret_cast<const half4_t*>(bias + (i4 % N));
        s += float4_t{(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
    }
    const half4_t o = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
    *reinterpret_cast<half4_t*>(y + i4) = o;
}

template <int WPL, int NWAVES, int UNIT, bool SEL, int NREG, int FOLDS, int GATED = 0>
void launch6(const GemvMfmaParams& p, dim3 grid, size_t lds, hipStream_t st) {
    // dynamic LDS above 64 KiB needs the opt-in once per kernel (host-side attribute, no sync)
    static std::atomic<unsigned long long> opted{0};
    (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_mfma_kernel<WPL, NWAVES, UNIT, SEL, NREG, FOLDS, true, false, GATED>), opted,
                         (160 - (GATED == 2 ? 1 : 0)) * 1024);  // mode 2 has 144 static bytes
    hipLaunchKernelGGL((awq_gemv_mfma_kernel<WPL, NWAVES, UNIT, SEL, NREG, FOLDS, true, false, GATED>), grid,
                       dim3(NWAVES * 64), lds, st, p);
}

template <int WPL, int NWAVES, int UNIT>
bool launch3(const GemvMfmaParams& p, dim3 grid, size_t lds, hipStream_t st) {
    constexpr int UROWS = 16 * UNIT;
    if (p.x_gated) {  // instantiated for the configurations the dispatcher picks by itself
        if constexpr (WPL == 2 && NWAVES >= 4 && UNIT <= 4) {
            if (p.g % UROWS) return false;
            if (p.x_gated == 2) {  // RMSNorm staging: decode batches only
                if (p.M == 1) launch6<WPL, NWAVES, UNIT, true, 2, 1, 2>(p, grid, lds, st);
                else if (p.M <= 4) launch6<WPL, NWAVES, UNIT, true, 4, 1, 2>(p, grid, lds, st);
                else return false;
                return true;
            }
            if (p.M == 1) launch6<WPL, NWAVES, UNIT, true, 2, 1, 1>(p, grid, lds, st);
            else if (p.M <= 8) launch6<WPL, NWAVES, UNIT, true, 4, 1, 1>(p, grid, lds, st);
            else if constexpr (NWAVES <= 4) launch6<WPL, NWAVES, UNIT, false, 4, 1, 1>(p, grid, lds, st);
            else return false;
            return true;
        }
        return false;
    }
    if (p.g % UROWS == 0) {  // a unit lies inside one group: one fold per unit
        if (p.M == 1) launch6<WPL, NWAVES, UNIT, true, 2, 1>(p, grid, lds, st);
        else if (p.M <= 8) launch6<WPL, NWAVES, UNIT, true, 4, 1>(p, grid, lds, st);
        else if constexpr (WPL == 2 && NWAVES <= 4) launch6<WPL, NWAVES, UNIT, false, 4, 1>(p, grid, lds, st);
        else return false;
        return true;
    }
    if constexpr (UNIT == 2 && WPL == 2 && NWAVES == 4) {
        if (p.g == 16) {  // two groups per 32-row unit
            if (p.M == 1) launch6<2, 4, 2, true, 2, 2>(p, grid, lds, st);
            else if (p.M <= 8) launch6<2, 4, 2, true, 4, 2>(p, grid, lds, st);
            else launch6<2, 4, 2, false, 4, 2>(p, grid, lds, st);
            return true;
        }
    }
    return false;
}

}  // namespace

#ifdef AWQ_GEMV_TRACE
extern "C" __attribute__((visibility("default"))) void awq_debug_set_trace(void* dev_buf) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_awq_trace), &dev_buf, sizeof(void*));
}
#endif

bool awq_gemv_mfma_supports(int M, int K, int N, int g, int wpl) {
    if (M < 1 || M > 16) return false;
    if (K % 32 || N % 32) return false;  // 32-row units; 16-byte zeros / scales chunks
    if (wpl == 4 && M > 8) return false;
    if (!(g % 32 == 0 || g == 16)) return false;
    if (g % 32 == 0 && g < 128 && (g & (g - 1))) return false;  // 32, 64 or a multiple of 128
    if (g > 128 && g % 128) return false;
    return true;
}

namespace {
struct GemvCfg {
    int wpl, nwaves, unit, S, rows_per_block, ng_max;
    size_t lds;
};

// log2 of the reducer blocks per tile.  Reducers wait for each other, so all of them must be able to be resident at the same
// time whatever else runs: at most one per CU of the smallest part this may run on.  The per-tile sum of squares
// (ssq_out) wants the whole tile in one block.
int reducers_per_tile(int S, int tile_blocks, int M, bool whole_tile) {
    // up to 4 batch rows one block reduces a tile in a single round (256 quads) and extra reducers only add stores and
    // waiting: 4096 x 11008, M = 1: 8.9 -> 10.1 us with four; M = 8: 13.2 -> 12.7, M = 16: 18.0 -> 15.5
    if (whole_tile || M <= 4 || S < 2 || tile_blocks < 1) return 0;
    int rlog = 0;
    while (rlog < 2 && (2 << rlog) <= S && (2 << rlog) * tile_blocks <= 256) ++rlog;
    return rlog;  // log2 of the count
}

bool set_magics(GemvMfmaParams& p) {
    return awq_magic_u32((uint32_t)p.g, (uint32_t)p.K + 64u, &p.g_magic) &&
           awq_magic_u32((uint32_t)(p.rows_per_block >> 3), 17u * (uint32_t)(p.rows_per_block >> 3) + 1u, &p.xc_magic);
}

size_t gemv_lds_bytes(int M, int CW, int nwaves, int rows_per_block, int ng_max) {
    const size_t staging = (size_t)(M + 1) * rows_per_block * 2 + (size_t)ng_max * (CW / 2 + 2 * CW);
    const size_t red = (size_t)nwaves * M * (CW + 16) * 4;
    return (staging > red ? staging : red) + 16;
}

// Fill in what the caller left at 0 and make the decomposition legal.  Returns false if no legal one.
bool gemv_config(const AwqGemmArgs& a, bool two_pass, GemvCfg& c) {
    const int M = a.M, K = a.K, N = a.N, g = a.g;
    if (c.wpl == 0) c.wpl = 2;
    if (c.wpl == 4 && M > 8) return false;
    const int CW = 128 * c.wpl;
    const int tiles = (N + CW - 1) / CW;
    // Defaults from the r21 sweep (profiles/): ~2048-2752 waves in flight, i.e. every wave streams
    // one or two units and the whole matrix is requested in the first microsecond; narrow
    // matrices (< 32 column tiles) use 8-wave blocks and a 16-way K split.
    const bool narrow = tiles < 32;
    if (c.nwaves == 0) c.nwaves = (narrow && M <= 8 && c.wpl == 2) ? 8 : 4;
    if (M > 8 && c.nwaves > 4) c.nwaves = 4;
    if (c.wpl == 4 && c.nwaves > 4) c.nwaves = 4;
    int S = c.S;
    if (S == 0) S = narrow ? 16 : ((640 + tiles - 1) / tiles > 8 ? 8 : (640 + tiles - 1) / tiles);
    if (c.S == 0 && !narrow && M <= 4) {
        // ~768 blocks = three per CU, evenly: 48 tiles x 16 slices beats x 8 (1.5 per CU: half the
        // chip carries two blocks, half one) by 8 % on 4096 x 12288 (profiles/r01_gemv_sweep.txt, r76)
        S = (768 + tiles / 2) / tiles;
        if (S > 16) S = 16;
    }
    if (c.S == 0 && M > 8) {  // 16-row slabs: fewer, fatter slices (r62 sweep: 22016 wide 28 -> 24 us at S = 4)
        const int cap = K >= 8192 ? 16 : 8;
        S = (320 + tiles - 1) / tiles > cap ? cap : (320 + tiles - 1) / tiles;
    }
    // unit: the largest power of two <= 8 sets (requested, or 4 / 2 so that every wave gets one)
    // that divides g and K
    int unit = c.unit ? c.unit : ((K / 64) >= S * c.nwaves ? 4 : 2);
    if (!c.unit && tiles >= 64 && M <= 4 && S <= 8) unit = 2;  // very wide (gate|up): 4 short units per wave, 3 % (r76/r77 sweeps)
    if (g == 16) unit = 2;
    while (unit > 2 && ((g % (16 * unit)) || (K % (16 * unit)))) unit >>= 1;
    if (K % (16 * unit)) return false;
    if (g != 16 && g % (16 * unit)) return false;
    if (c.wpl == 4 && unit == 8) unit = 4;  // register budget
    c.unit = unit;
    const int units = K / (16 * unit);
    if (c.S == 0) {  // every wave at least one unit
        const int max_s = (units + c.nwaves - 1) / c.nwaves;
        if (S > max_s) S = max_s;
    }
    if (S > 64) S = 64;
    if (S > units) S = units;
    if (S < 1) S = 1;
    for (;; ++S) {  // grow the split until the staging area fits the LDS
        const int upb = (units + S - 1) / S;
        c.rows_per_block = upb * 16 * unit;
        c.ng_max = c.rows_per_block / g + 2;
        c.lds = gemv_lds_bytes(M, CW, c.nwaves, c.rows_per_block, c.ng_max);
        c.S = (units + upb - 1) / upb;
        if (c.lds <= 160 * 1024) break;
        if (S >= units || S >= 64) return false;
    }
    if (c.S > 1) {  // keep the slabs inside the workspace the caller gave us
        const size_t per_slice = (two_pass ? (size_t)M * N : (size_t)tiles * M * CW) * sizeof(float);
        const size_t have = two_pass ? a.partial_floats * sizeof(float) : a.exchange_bytes;
        const size_t fit = have / per_slice;  // every slice has a slab (the reducers store what the others reduce)
        if ((size_t)c.S > fit) return false;
    }
    return true;
}
}  // namespace

namespace {
template <int UNIT, bool SEL, int XMODE>
void launch_moe_x(const GemvMfmaParams& p, dim3 grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> opted{0};
    (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_mfma_kernel<2, 4, UNIT, SEL, 4, 1, true, true, XMODE>), opted);
    hipLaunchKernelGGL((awq_gemv_mfma_kernel<2, 4, UNIT, SEL, 4, 1, true, true, XMODE>), grid, dim3(256), lds, st, p);
}
// x_gated: the pair rows are [gate | up] of 2K halves (the w1|w3 output) and silu(gate) * up is applied while the block stages
// its activations -- the w2 grouped GEMM of a MoE block then needs no separate silu_and_mul launch (moe.py:73-76)
template <int UNIT, bool SEL>
void launch_moe(const GemvMfmaParams& p, dim3 grid, size_t lds, hipStream_t st) {
    if (p.x_gated) launch_moe_x<UNIT, SEL, 1>(p, grid, lds, st);
    else launch_moe_x<UNIT, SEL, 0>(p, grid, lds, st);
}
}  // namespace

// Grouped (MoE) GEMM: `max_blocks` token blocks of a.M (8 or 16) rows, block b multiplies the gathered
// rows sorted_ids[M*b .. M*b+M-1] with expert expert_ids[b]'s weights; blocks past *num_post_pad exit.
// 8-row blocks use the selector-row kernel (one MFMA per fragment): the decode case.
// The block path is the decode-sized one: past 1 GiB of split-K exchange (thousands of token blocks) the launcher refuses
// and the caller runs its per-expert GEMM path (moe.py PREFILL_MIN_PAIRS sits far below this).
static const size_t AWQ_GROUPED_MAX_EXCHANGE = (size_t)1 << 30;

size_t awq_grouped_workspace_bytes_impl(int max_blocks, int K, int N) {
    (void)K;
    const size_t tiles = (size_t)(N + 255) / 256;
    size_t need = (size_t)max_blocks * 8 * tiles * 16 * 256 * 4;  // S <= 8
    if (need > AWQ_GROUPED_MAX_EXCHANGE) need = AWQ_GROUPED_MAX_EXCHANGE;
    return (size_t)AWQ_WS_COUNTER_BYTES + need;
}

int awq_launch_grouped_gemm(const AwqGemmArgs& a, const int* sorted_ids, const int* expert_ids, const int* num_post_pad,
                            const float* pair_weights, int num_pairs, int x_div, int max_blocks, int64_t expert_qw_words,
                            int64_t expert_z_words, int64_t expert_s_halfs) {
    if (!(a.M == 16 || a.M == 8) || max_blocks < 1 || a.x_gated > 1) return AWQ_ERR_BAD_SHAPE;
    if (max_blocks > 65535) return AWQ_ERR_UNSUPPORTED;  // token blocks ride in gridDim.z; prefill-sized routings take the per-expert path
    if (!awq_gemv_mfma_supports(a.M, a.K, a.N, a.g, 2) || a.g % 32) return AWQ_ERR_UNSUPPORTED;
    GemvCfg c{2, 4, 0, 0, 0, 0, 0};
    AwqGemmArgs probe = a;
    probe.exchange_bytes = (size_t)-1 >> 1;  // the split is bounded below instead
    if (!gemv_config(probe, false, c)) return AWQ_ERR_UNSUPPORTED;
    if (a.g % (16 * c.unit)) return AWQ_ERR_UNSUPPORTED;
    const int CW = 256;
    const int tiles = (a.N + CW - 1) / CW;
    // re-balance the K split for max_blocks x tiles blocks already in the grid (cap S at 8)
    const int units = a.K / (16 * c.unit);
    int S = (512 + tiles * max_blocks - 1) / (tiles * max_blocks);
    if (S > 8) S = 8;
    if (S > (units + 3) / 4) S = (units + 3) / 4;
    if (S < 1) S = 1;
    for (;; ++S) {
        const int upb = (units + S - 1) / S;
        c.rows_per_block = upb * 16 * c.unit;
        c.ng_max = c.rows_per_block / a.g + 2;
        c.lds = gemv_lds_bytes(a.M, CW, 4, c.rows_per_block, c.ng_max);
        c.S = (units + upb - 1) / upb;
        // Only the token blocks of experts that were hit do any work (5 of 16 in the Mixtral bs = 4
        // case), so the grid size says little; what matters is that three blocks fit a CU: split K
        // (up to the 8 slices the workspace is sized for) until the staging area is <= 48 KB.
        // One 95 KB block per CU streamed w1|w3 at 3.1 TB/s (profiles/r01_moe_mixtral_bs4.txt).
        const size_t floor_lds = (size_t)4 * a.M * (CW + 16) * 4 + 16;  // the fold area does not shrink with S
        if (c.lds <= (floor_lds > 48 * 1024 ? floor_lds : 48 * 1024) || ((S >= 8 || S >= units) && c.lds <= 160 * 1024)) break;
        if (S >= units || S >= 64) return AWQ_ERR_UNSUPPORTED;
    }
    if (c.S > 1) {
        const size_t need = (size_t)max_blocks * c.S * tiles * a.M * CW * sizeof(float);
        if (need > AWQ_GROUPED_MAX_EXCHANGE) return AWQ_ERR_UNSUPPORTED;
        if (!a.exchange || !a.counters || a.exchange_bytes < need) return AWQ_ERR_WORKSPACE;
    }
    GemvMfmaParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.x_gated = a.x_gated;
    p.res_in = reinterpret_cast<const half_t*>(a.res_in);
    p.res_out = reinterpret_cast<half_t*>(a.res_out);
    p.norm_w = reinterpret_cast<const half_t*>(a.norm_w);
    p.norm_eps = a.norm_eps;
    p.ssq_in = a.ssq_in; p.ssq_in_tiles = a.ssq_in_tiles;
    p.add_res = reinterpret_cast<const half_t*>(a.add_res);
    p.ssq_out = a.ssq_out;
    p.bias = nullptr;
    p.y = reinterpret_cast<half_t*>(a.y);
    p.M = a.M; p.K = a.K; p.N = a.N; p.g = a.g;
    p.tiles = tiles; p.S = c.S;
    const int rlog = reducers_per_tile(c.S, tiles * max_blocks, a.M, a.ssq_out != nullptr);
    p.rows_per_block = c.rows_per_block;
    p.ng_max = c.ng_max;
    if (!set_magics(p)) return AWQ_ERR_UNSUPPORTED;
    p.combine = rlog ? 1 + rlog : 0;
    p.slabs = a.exchange;
    p.scratch = nullptr;
    p.err = a.counters;
    p.sorted_ids = sorted_ids; p.expert_ids = expert_ids; p.num_post_pad = num_post_pad;
    p.pair_weights = pair_weights;
    p.num_pairs = num_pairs; p.x_div = x_div;
    p.expert_qw_words = expert_qw_words; p.expert_z_words = expert_z_words; p.expert_s_halfs = expert_s_halfs;
    dim3 grid((unsigned)tiles, (unsigned)c.S, (unsigned)max_blocks);
    if (a.M == 8) {
        if (c.unit == 2) launch_moe<2, true>(p, grid, c.lds, a.stream);
        else if (c.unit == 4) launch_moe<4, true>(p, grid, c.lds, a.stream);
        else launch_moe<8, true>(p, grid, c.lds, a.stream);
    } else {
        if (c.unit == 2) launch_moe<2, false>(p, grid, c.lds, a.stream);
        else if (c.unit == 4) launch_moe<4, false>(p, grid, c.lds, a.stream);
        else launch_moe<8, false>(p, grid, c.lds, a.stream);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_launch_gemv_mfma(const AwqGemmArgs& a, int wpl, int nwaves, int unit, int splitk, bool two_pass) {
    if (!(wpl == 0 || wpl == 2 || wpl == 4) || !(nwaves == 0 || nwaves == 2 || nwaves == 4 || nwaves == 8) ||
        !(unit == 0 || unit == 2 || unit == 4 || unit == 8))
        return AWQ_ERR_UNSUPPORTED;
    if (!awq_gemv_mfma_supports(a.M, a.K, a.N, a.g, wpl ? wpl : 2)) return AWQ_ERR_UNSUPPORTED;
    GemvCfg c{wpl, nwaves, unit, splitk, 0, 0, 0};
    if (!gemv_config(a, two_pass, c)) return AWQ_ERR_UNSUPPORTED;
    const int CW = 128 * c.wpl;
    const int tiles = (a.N + CW - 1) / CW;
    GemvMfmaParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.x_gated = a.x_gated;
    p.res_in = reinterpret_cast<const half_t*>(a.res_in);
    p.res_out = reinterpret_cast<half_t*>(a.res_out);
    p.norm_w = reinterpret_cast<const half_t*>(a.norm_w);
    p.norm_eps = a.norm_eps;
    p.ssq_in = a.ssq_in; p.ssq_in_tiles = a.ssq_in_tiles;
    p.add_res = reinterpret_cast<const half_t*>(a.add_res);
    p.ssq_out = a.ssq_out;
    p.bias = reinterpret_cast<const half_t*>(a.bias);
    p.y = reinterpret_cast<half_t*>(a.y);
    p.M = a.M; p.K = a.K; p.N = a.N; p.g = a.g;
    p.tiles = tiles; p.S = c.S;
    const int rlog = reducers_per_tile(c.S, tiles, a.M, a.ssq_out != nullptr);
    p.rows_per_block = c.rows_per_block;
    p.ng_max = c.ng_max;
    if (!set_magics(p)) return AWQ_ERR_UNSUPPORTED;
    p.combine = two_pass ? 1 : (rlog ? 1 + rlog : 0);
    p.slabs = a.exchange;
    p.scratch = a.partial;
    p.err = a.counters;
    p.sorted_ids = nullptr; p.expert_ids = nullptr; p.num_post_pad = nullptr; p.pair_weights = nullptr;
    p.num_pairs = 0; p.x_div = 1; p.expert_qw_words = p.expert_z_words = p.expert_s_halfs = 0;
    if (c.S > 1 && !two_pass && (!a.exchange || !a.counters)) return AWQ_ERR_WORKSPACE;
    if (c.S > 1 && two_pass && !a.partial) return AWQ_ERR_WORKSPACE;
    dim3 grid((unsigned)tiles, (unsigned)c.S, 1u);
    bool ok = false;
#define AWQ_GEMV_CASE(W, V, U) \
    if (c.wpl == W && c.nwaves == V && c.unit == U) ok = launch3<W, V, U>(p, grid, c.lds, a.stream);
    AWQ_GEMV_CASE(2, 2, 2) AWQ_GEMV_CASE(2, 2, 4) AWQ_GEMV_CASE(2, 2, 8)
    AWQ_GEMV_CASE(2, 4, 2) AWQ_GEMV_CASE(2, 4, 4) AWQ_GEMV_CASE(2, 4, 8)
    AWQ_GEMV_CASE(2, 8, 2) AWQ_GEMV_CASE(2, 8, 4) AWQ_GEMV_CASE(2, 8, 8)
    AWQ_GEMV_CASE(4, 2, 2) AWQ_GEMV_CASE(4, 2, 4)
    AWQ_GEMV_CASE(4, 4, 2) AWQ_GEMV_CASE(4, 4, 4)
#undef AWQ_GEMV_CASE
    if (!ok) return AWQ_ERR_UNSUPPORTED;
    if (hipGetLastError() != hipSuccess) return AWQ_ERR_LAUNCH;
    if (c.S > 1 && two_pass) {
        const int MN = a.M * a.N;
        hipLaunchKernelGGL(awq_gemv_mfma_reduce_kernel, dim3((MN / 4 + 255) / 256), dim3(256), 0, a.stream, a.partial,
                           reinterpret_cast<const half_t*>(a.bias), reinterpret_cast<half_t*>(a.y), MN, a.N, c.S);
        if (hipGetLastError() != hipSuccess) return AWQ_ERR_LAUNCH;
    }
    return AWQ_OK;
}
