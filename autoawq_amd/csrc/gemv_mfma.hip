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
//   through LDS; K is further split over S blocks per tile.  Slabs are combined in-launch:
//   write-through 16-byte slab stores, drain, one relaxed ticket; the last arriver sums the slabs
//   in slab order (bitwise reproducible), writes fp16 and re-arms the ticket
//   (cdna_hip_programming.md section 5, split-K recipe with sc1 slabs).
#include "awq_device.h"
#include "awq_internal.h"

namespace {

struct GemvMfmaParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    const half_t* bias;
    half_t* y;
    float* slabs;       // [S][tiles][M][CW] fp32 (in-launch) or [S][M][N] (two-pass)
    unsigned* tickets;  // [tiles], zero on entry, zero on exit
    int M, K, N, g;
    int tiles, S;
    int iters_per_block;  // loop iterations (SETS sets each) per K slice
    int two_pass;
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

AWQ_DEV rsrc_t mk_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

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

AWQ_DEV float4_t mfma16(u32x4v a, u32x4v b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0,
                                                  0, 0);
}

constexpr uint32_t OOB = 0x80000000u;  // lane offset beyond every descriptor: returns 0, no traffic

// SEL: selector-row A (M <= 8, one MFMA per B fragment); !SEL: M <= 16, two MFMAs.
// NREG: live D registers per lane (2 when M == 1, else 4).  SETS: 16-row sets per loop iteration
// (8 or 4); FOLDS: group folds per iteration (SETS*16/FOLDS rows each: a divisor of g, <= 128).
template <int WPL, int NWAVES, bool SEL, int NREG, int SETS, int FOLDS, bool NT>
__global__ __launch_bounds__(NWAVES * 64) void awq_gemv_mfma_kernel(GemvMfmaParams p) {
    typedef typename Words<WPL>::T WV;
    constexpr int CPL = 8 * WPL;         // columns per lane
    constexpr int CW = 16 * CPL;         // columns per wave == per block tile
    constexpr int CWP = CW + 16;         // padded LDS row (one float per lane)
    constexpr int NACC = 4 * WPL;        // (word, J) pairs per lane
    constexpr int NA = SEL ? 1 : 2;      // accumulators per pair
    constexpr int AUXW = NT ? 2 : 0;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [NWAVES][M][CWP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kb = lane >> 4;
    const int tile = blockIdx.x % p.tiles, slice = blockIdx.x / p.tiles;
    const int NW = p.N >> 3;
    const int colw = (tile * 16 + j) * WPL;  // first packed word of this lane
    const bool active = colw < NW;

    // this wave's loop iterations [ws, we), in units of SETS sets
    const int nsets = p.K >> 4;
    const int niter = (nsets + SETS - 1) / SETS;
    const int bs = slice * p.iters_per_block;
    const int be = min(niter, bs + p.iters_per_block);
    const int per_wave = (be - bs + NWAVES - 1) / NWAVES;
    const int ws = min(be, bs + wave * per_wave);
    const int we = min(be, ws + per_wave);

    float yv[NACC][NA][NREG];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < NREG; ++r) yv[c][a][r] = 0.f;

    if (ws < we) {
        const uint32_t row_bytes = (uint32_t)NW * 4u;
        const uint32_t ngroups = (uint32_t)(p.K / p.g);
        const rsrc_t wres = mk_rsrc(p.qweight, (uint32_t)p.K * row_bytes);
        const rsrc_t zres = mk_rsrc(p.qzeros, ngroups * row_bytes);
        const rsrc_t sres = mk_rsrc(p.scales, ngroups * (uint32_t)p.N * 2u);
        const rsrc_t xres = mk_rsrc(p.x, (uint32_t)p.M * (uint32_t)p.K * 2u);
        const uint32_t wvoff = active ? (uint32_t)colw * 4u + (uint32_t)(4 * kb) * row_bytes : OOB;
        const uint32_t zvoff = active ? (uint32_t)colw * 4u : OOB;
        const uint32_t svoff = active ? (uint32_t)colw * 16u : OOB;
        // A row of this lane: SEL -> batch row j >> 1, column parity j & 1; else batch row j
        const int arow = SEL ? (j >> 1) : j;
        const uint32_t xvoff = (arow < p.M) ? ((uint32_t)arow * (uint32_t)p.K + 4u * kb) * 2u : OOB;
        const uint32_t sel_lo = (j & 1) ? 0x01000C0Cu : 0x0C0C0100u;  // low half of a dword -> slot (j & 1)
        const uint32_t sel_hi = (j & 1) ? 0x03020C0Cu : 0x0C0C0302u;  // high half
        const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
        constexpr int SPF = SETS / FOLDS;  // sets per fold

        for (int it = ws; it < we; ++it) {
            // ---- request the whole iteration: weights, activations, zeros/scales of its groups
            WV q[SETS][4];
            u32x2 xq[SETS];
            WV qz[FOLDS];
            u32x4 sc[FOLDS][WPL];
            const uint32_t uset = (uint32_t)__builtin_amdgcn_readfirstlane(it) * SETS;
#pragma unroll
            for (int t = 0; t < SETS; ++t) {
                const bool valid = (int)(uset + t) < nsets;  // only the K tail can be short
                const uint32_t srow = (uset + t) * 16u * row_bytes;
                const uint32_t wv = valid ? wvoff : OOB;
#pragma unroll
                for (int r = 0; r < 4; ++r) q[t][r] = ld_words<WPL, AUXW>(wres, wv, srow + (uint32_t)r * row_bytes);
                xq[t] = __builtin_bit_cast(
                    u32x2, __builtin_amdgcn_raw_buffer_load_b64(xres, valid ? xvoff : OOB, (uset + t) * 32u, 0));
            }
#pragma unroll
            for (int f = 0; f < FOLDS; ++f) {
                uint32_t grp = ((uset + SPF * f) * 16u) / (uint32_t)p.g;
                grp = grp < ngroups ? grp : ngroups - 1;
                qz[f] = ld_words<WPL, 0>(zres, zvoff, grp * row_bytes);
#pragma unroll
                for (int wd = 0; wd < WPL; ++wd) sc[f][wd] = __builtin_bit_cast(
                    u32x4, __builtin_amdgcn_raw_buffer_load_b128(sres, svoff + 16u * wd, grp * (uint32_t)p.N * 2u, 0));
            }
            __builtin_amdgcn_sched_barrier(0);  // every request above is issued before anything is consumed

            // ---- consume
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
                    const uint32_t x01 = xq[t][0], x23 = xq[t][1];
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
#define AWQ_MMA_J(J)                                                                                     \
    {                                                                                                    \
        const u32x4v bf = {pair16<J>(q0), pair16<J>(q1), pair16<J>(q2), pair16<J>(q3)};                  \
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
                // y += s * (acc - (16 + z) * sx) over the rows of this fold (one group, <= 128 rows)
#pragma unroll
                for (int wd = 0; wd < WPL; ++wd) {
                    const u32x4 sv = sc[f][wd];
                    const uint32_t zw = qz[f][wd];
                    const uint32_t zp[4] = {pair16<0>(zw), pair16<1>(zw), pair16<2>(zw), pair16<3>(zw)};
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
        }
    }

    // ---- fold the waves of the block through LDS: red[wave][m][col]
    const int M = p.M;
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
    auto emit4 = [&](int qd, float4_t s) {
        const int m = qd / (CW / 4), c4 = (qd % (CW / 4)) * 4;
        const int col = col0 + c4;
        if (col >= p.N) return;  // N % 8 == 0: a quad is all in or all out
        if (p.bias) {
            const half4_t b4 = *reinterpret_cast<const half4_t*>(p.bias + col);
            s += float4_t{(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
        }
        const half4_t o = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
        *reinterpret_cast<half4_t*>(p.y + (int64_t)m * p.N + col) = o;
    };

    const int S = p.S;
    if (S == 1) {
        for (int qd = tid; qd < quads; qd += NWAVES * 64) emit4(qd, block_sum4(qd));
        return;
    }
    if (p.two_pass) {  // plain fp32 slabs [S][M][N]; a second kernel reduces
        for (int qd = tid; qd < quads; qd += NWAVES * 64) {
            const int m = qd / (CW / 4), col = col0 + (qd % (CW / 4)) * 4;
            if (col < p.N)
                *reinterpret_cast<float4_t*>(p.slabs + ((int64_t)slice * M + m) * p.N + col) = block_sum4(qd);
        }
        return;
    }

    // ---- in-launch combine: write-through slab, drain, ticket; last arriver reduces
    const uint32_t slab_bytes = (uint32_t)quads * 16u;
    const rsrc_t slres = mk_rsrc(p.slabs, (uint32_t)S * (uint32_t)p.tiles * slab_bytes);
    for (int qd = tid; qd < quads; qd += NWAVES * 64) {
        const float4_t s = block_sum4(qd);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), slres, (uint32_t)qd * 16u,
                                               (uint32_t)(slice * p.tiles + tile) * slab_bytes, 16 /* sc1 */);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's slab stores are drained; red[] is free again
    unsigned* flag = reinterpret_cast<unsigned*>(red);
    if (tid == 0)
        *flag = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != (unsigned)(S - 1)) return;
    for (int qd = tid; qd < quads; qd += NWAVES * 64) {
        float4_t s = {0.f, 0.f, 0.f, 0.f};
        for (int sl0 = 0; sl0 < S; sl0 += 8) {  // 8 independent 16-byte loads in flight, summed in slab order
            float4_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)  // slabs past S are requested out of range: zeros, no traffic
                v[u] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                        slres, (sl0 + u < S) ? (uint32_t)qd * 16u : OOB,
                                                        (uint32_t)((sl0 + u) * p.tiles + tile) * slab_bytes, 16));
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];  // fixed order: bitwise reproducible
        }
        emit4(qd, s);
    }
    if (tid == 0) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

template <int WPL, int NWAVES, bool SEL, int NREG, int SETS, int FOLDS>
void launch6(const GemvMfmaParams& p, dim3 grid, size_t lds, bool nt, hipStream_t st) {
    // dynamic LDS above 64 KiB needs the opt-in once per kernel (host-side attribute, no sync)
    static const bool lds_opt_in = [] {
        (void)hipFuncSetAttribute(
            reinterpret_cast<const void*>(&awq_gemv_mfma_kernel<WPL, NWAVES, SEL, NREG, SETS, FOLDS, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(
            reinterpret_cast<const void*>(&awq_gemv_mfma_kernel<WPL, NWAVES, SEL, NREG, SETS, FOLDS, false>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)lds_opt_in;
    if (nt)
        hipLaunchKernelGGL((awq_gemv_mfma_kernel<WPL, NWAVES, SEL, NREG, SETS, FOLDS, true>), grid, dim3(NWAVES * 64), lds, st, p);
    else
        hipLaunchKernelGGL((awq_gemv_mfma_kernel<WPL, NWAVES, SEL, NREG, SETS, FOLDS, false>), grid, dim3(NWAVES * 64), lds, st, p);
}

template <int WPL, int NWAVES, bool SEL, int NREG>
void launch4(const GemvMfmaParams& p, dim3 grid, size_t lds, bool nt, hipStream_t st) {
    if (p.g % 128 == 0) launch6<WPL, NWAVES, SEL, NREG, 8, 1>(p, grid, lds, nt, st);
    else if (p.g % 64 == 0) launch6<WPL, NWAVES, SEL, NREG, 4, 1>(p, grid, lds, nt, st);
    else if (p.g == 32) launch6<WPL, NWAVES, SEL, NREG, 4, 2>(p, grid, lds, nt, st);
    else launch6<WPL, NWAVES, SEL, NREG, 4, 4>(p, grid, lds, nt, st);
}

template <int WPL, int NWAVES>
void launch2(const GemvMfmaParams& p, dim3 grid, size_t lds, bool nt, hipStream_t st) {
    if (p.M == 1) launch4<WPL, NWAVES, true, 2>(p, grid, lds, nt, st);
    else if (p.M <= 8) launch4<WPL, NWAVES, true, 4>(p, grid, lds, nt, st);
    else if constexpr (WPL == 2 && NWAVES <= 4) launch4<WPL, NWAVES, false, 4>(p, grid, lds, nt, st);
}

int sets_per_iter(int g) { return g % 128 == 0 ? 8 : 4; }

}  // namespace

bool awq_gemv_mfma_supports(int M, int K, int N, int g, int wpl) {
    if (M < 1 || M > 16) return false;
    if (K % 16 || N % (8 * wpl)) return false;
    if (!(g % 64 == 0 || g == 32 || g == 16)) return false;
    return true;
}

// Default decomposition: about one block per CU, every wave at least one loop iteration.
void awq_gemv_mfma_default_config(int M, int K, int N, int g, int* wpl, int* nwaves, int* splitk) {
    (void)M;
    if (*wpl == 0) *wpl = 2;
    const int CW = 128 * *wpl;
    const int tiles = (N + CW - 1) / CW;
    const int niter = (K / 16 + sets_per_iter(g) - 1) / sets_per_iter(g);
    if (*nwaves == 0) *nwaves = (tiles * niter >= 1024) ? 4 : 2;
    if (*splitk == 0) {
        int s = (256 + tiles - 1) / tiles;
        const int max_s = (niter + *nwaves - 1) / *nwaves;
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
        *splitk = s;
    }
}

int awq_launch_gemv_mfma(const AwqGemmArgs& a, int wpl, int nwaves, int splitk, bool two_pass, bool nt) {
    awq_gemv_mfma_default_config(a.M, a.K, a.N, a.g, &wpl, &nwaves, &splitk);
    if (!awq_gemv_mfma_supports(a.M, a.K, a.N, a.g, wpl)) return AWQ_ERR_UNSUPPORTED;
    if (!(wpl == 2 || wpl == 4) || !(nwaves == 2 || nwaves == 4 || nwaves == 8)) return AWQ_ERR_UNSUPPORTED;
    if ((wpl == 4 && (a.M > 8 || nwaves == 8)) || (a.M > 8 && nwaves == 8)) return AWQ_ERR_UNSUPPORTED;  // register budget
    const int CW = 128 * wpl;
    const int tiles = (a.N + CW - 1) / CW;
    const int niter = (a.K / 16 + sets_per_iter(a.g) - 1) / sets_per_iter(a.g);
    int S = splitk < 1 ? 1 : splitk;
    if (S > 64) S = 64;
    if (S > niter) S = niter;
    {  // keep the slabs inside the workspace the caller gave us
        const size_t per_slice = (two_pass ? (size_t)a.M * a.N : (size_t)tiles * a.M * CW) * sizeof(float);
        const size_t fit = a.partial ? (a.partial_floats * sizeof(float)) / per_slice : 0;
        if (S > 1 && (size_t)S > fit) S = fit < 1 ? 1 : (int)fit;
    }
    const int ipb = (niter + S - 1) / S;
    S = (niter + ipb - 1) / ipb;
    GemvMfmaParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.bias = reinterpret_cast<const half_t*>(a.bias);
    p.y = reinterpret_cast<half_t*>(a.y);
    p.M = a.M; p.K = a.K; p.N = a.N; p.g = a.g;
    p.tiles = tiles; p.S = S;
    p.iters_per_block = ipb;
    p.two_pass = two_pass ? 1 : 0;
    p.slabs = a.partial;
    p.tickets = reinterpret_cast<unsigned*>(a.counters);
    if (S > 1) {
        const size_t need = two_pass ? (size_t)S * a.M * a.N * sizeof(float)
                                     : (size_t)S * tiles * a.M * CW * sizeof(float);
        if (!a.partial || a.partial_floats * sizeof(float) < need || !a.counters) return AWQ_ERR_WORKSPACE;
        if ((size_t)tiles * sizeof(unsigned) > AWQ_WS_COUNTER_BYTES) return AWQ_ERR_WORKSPACE;
    }
    const size_t lds = (size_t)nwaves * a.M * (CW + 16) * sizeof(float);
    if (lds > 160 * 1024) return AWQ_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(tiles * S));
    if (wpl == 2) {
        if (nwaves == 2) launch2<2, 2>(p, grid, lds, nt, a.stream);
        else if (nwaves == 4) launch2<2, 4>(p, grid, lds, nt, a.stream);
        else launch2<2, 8>(p, grid, lds, nt, a.stream);
    } else {
        if (nwaves == 2) launch2<4, 2>(p, grid, lds, nt, a.stream);
        else launch2<4, 4>(p, grid, lds, nt, a.stream);
    }
    if (hipGetLastError() != hipSuccess) return AWQ_ERR_LAUNCH;
    if (S > 1 && two_pass) {
        const int MN = a.M * a.N;
        hipLaunchKernelGGL(awq_gemv_mfma_reduce_kernel, dim3((MN / 4 + 255) / 256), dim3(256), 0, a.stream, a.partial,
                           reinterpret_cast<const half_t*>(a.bias), reinterpret_cast<half_t*>(a.y), MN, a.N, S);
        if (hipGetLastError() != hipSuccess) return AWQ_ERR_LAUNCH;
    }
    return AWQ_OK;
}
