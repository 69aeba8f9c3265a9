// gemm_skinny.hip -- M <= 16 decode GEMM on the GEMM layout with MFMA 16x16x32 f16.
//
// Replaces awq_ext.gemm_forward_cuda (awq/modules/linear/gemm.py:56-58) for 1 <= M <= 16 and is
// the engine of the MoE grouped GEMM (awq/modules/fused/moe.py:60-89, 16-row token blocks).
//
// Roofline: HBM (AI <= 61 flop/B at M = 16, ridge ~312).  The matrix cores are used not for
// flops but to take the multiply-accumulate AND the cross-lane K reduction off the VALU, whose
// whole budget goes to decoding int4: one shift + one v_and_or per column PAIR and nothing else.
//
// How a packed word feeds an MFMA without any transposition
//   GEMM layout packs 8 N-adjacent weights of ONE k per int32; the B fragment of
//   v_mfma_f32_16x16x32_f16 wants 8 K-adjacent values of ONE n per lane.  Nibbles (J, J+4) of a
//   word are logical columns (2J, 2J+1); ((q >> 4J) << 6) & 0x03C003C0 | 0x4C004C00 is the fp16
//   pair (16 + w[k, a], 16 + w[k, b]) -- exact.  Four such pairs from FOUR rows fill a B fragment
//   whose K slots are (k0:a, k0:b, k1:a, k1:b, ...).  Multiplying it once by an A fragment that
//   carries x in the even slots and 0 in the odd ones gives column a, and once by the
//   odd-slot twin gives column b.  Twice the MFMA issues (the pipe is idle anyway) and zero
//   cross-lane shuffles.
//   Group factorisation keeps it exact: with sx = sum_k x[m,k] over the rows of a group
//   (obtained from one extra MFMA against an all-ones B), y[m,n] += s[g,n] * (acc - (16 + z) * sx),
//   all in fp32 -- more accurate than multiplying fp16-rounded weights.
//
// Decomposition: lane (nl = l & 15, kb = l >> 4) owns WPL packed words (8*WPL columns) of rows
// 8*kb .. 8*kb+7 of each 32-row K-step; a wave covers 128*WPL columns; the 4 waves of a block
// take consecutive K ranges of one column tile and are folded through LDS; blocks along
// gridDim.y split K further and are combined in-launch with tagged granules (awq_combine.h).
#include "awq_combine.h"
#include "awq_device.h"
#include "awq_internal.h"

namespace {

struct SkinnyParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    const half_t* bias;
    half_t* y;
    awq_granule_t* granules;
    int* err;
    int M, K, N, g, steps_per_block;
    unsigned long long* trace;  // diagnostics: 8 timestamps (100 MHz) per wave, or null
};

template <int WPL>
struct WordVec;
template <>
struct WordVec<2> { typedef u32x2 type; };
template <>
struct WordVec<4> { typedef u32x4 type; };

// Buffer loads through a wave-uniform descriptor: ONE 32-bit per-lane byte offset for the whole
// kernel plus a scalar (SGPR) row offset per load, instead of a 64-bit address per row.
typedef __amdgpu_buffer_rsrc_t rsrc_t;

AWQ_DEV rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <int WPL, bool NT>
AWQ_DEV typename WordVec<WPL>::type buf_words(rsrc_t r, uint32_t voff, uint32_t soff) {
    constexpr int AUX = NT ? 2 : 0;  // bit 1 = nt (streamed-once weights)
    if constexpr (WPL == 2)
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    else
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}

AWQ_DEV u32x4 buf_b128(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// fp16 pair (16 + col 2J, 16 + col 2J+1) of a packed word
template <int J>
AWQ_DEV uint32_t pair16(uint32_t q) {
    constexpr int SH = 6 - 4 * J;
    const uint32_t t = SH >= 0 ? (q << (SH >= 0 ? SH : 0)) : (q >> (SH < 0 ? -SH : 0));
    return and_or(t, 0x03C003C0u, 0x4C004C00u);
}

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

AWQ_DEV float4_t mfma16(u32x4v a, u32x4v b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0,
                                                  0, 0);
}

#define AWQ_TRACE(slot)                                                                                   \
    if constexpr (TRACE) {                                                                                \
        if (lane == 0) p.trace[(((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + (slot)] = wall_clock64(); \
    }

// FPQ = group folds per quad of K-steps: 1 when g % 128 == 0, 2 for g = 64, 4 for g = 32.
template <int WPL, int MR, bool NT, int FPQ, bool TRACE = false>
__global__ __launch_bounds__(256, ((MR <= 2 && FPQ == 1) ? 2 : 1)) void awq_skinny_kernel(SkinnyParams p) {
    typedef typename WordVec<WPL>::type WV;
    constexpr int CPL = 8 * WPL;  // columns per lane
    constexpr int CW = 16 * CPL;  // columns per wave == per block tile
    constexpr uint32_t OOB = 0x80000000u;  // lane offset past every descriptor: load returns 0, no traffic
    __shared__ float red[4 * 4 * CW];  // [wave][kb][CW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR)
    const int nl = lane & 15, kb = lane >> 4;
    const int NW = p.N >> 3;
    const int colw = (blockIdx.x * 16 + nl) * WPL;  // first packed-word column of this lane
    const bool active = colw < NW;
    const int nsteps = p.K >> 5;
    const int bs = blockIdx.y * p.steps_per_block;
    const int be = min(nsteps, bs + p.steps_per_block);
    const int per_wave = (be - bs + 3) >> 2;
    const int ws = bs + wave * per_wave;  // this wave's K-steps [ws, we)
    const int we = min(be, ws + per_wave);
    const int spg = p.g >> 5;  // K-steps per quantisation group
    AWQ_TRACE(0)

    float yv[CPL][MR];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int r = 0; r < MR; ++r) yv[c][r] = 0.f;

    if (ws < we) {
        const uint32_t row_bytes = (uint32_t)NW * 4u;
        const rsrc_t wres = make_rsrc(p.qweight, (uint32_t)p.K * row_bytes);
        const rsrc_t zres = make_rsrc(p.qzeros, (uint32_t)(p.K / p.g) * row_bytes);
        const rsrc_t sres = make_rsrc(p.scales, (uint32_t)(p.K / p.g) * (uint32_t)p.N * 2u);
        const rsrc_t xres = make_rsrc(p.x, (uint32_t)p.M * (uint32_t)p.K * 2u);
        // inactive lanes (ragged last tile) and rows >= M of the 16-row A tile read out of range = 0
        const uint32_t wvoff = active ? (uint32_t)colw * 4u + (uint32_t)(8 * kb) * row_bytes : OOB;
        const uint32_t zvoff = active ? (uint32_t)colw * 4u : OOB;
        const uint32_t svoff = active ? (uint32_t)colw * 16u : OOB;
        const uint32_t xvoff = (nl < p.M) ? ((uint32_t)nl * (uint32_t)p.K + 8u * kb) * 2u : OOB;
        const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};

        float4_t acc[CPL];
        float4_t accsx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = float4_t{0.f, 0.f, 0.f, 0.f};

        // Request one K-step: the 8 activations of this lane's rows, then its 8 weight rows.  Steps
        // outside [ws, we) are requested out of range: they cost no memory traffic, return zeros
        // and (x = 0) contribute nothing, so the pipeline below needs no conditional loads --
        // outstanding-load counts stay static and hipcc's waits stay partial (vmcnt(N)).
        auto issue_step = [&](WV(&q)[8], u32x4v& xv, int step) {
            const bool valid = (step >= ws) && (step < we);
            const uint32_t ustep = (uint32_t)__builtin_amdgcn_readfirstlane(valid ? step : ws);
            const uint32_t srow = ustep * 32u * row_bytes;
            xv = buf_b128(xres, valid ? xvoff : OOB, ustep * 64u);
            const uint32_t wv = valid ? wvoff : OOB;
#pragma unroll
            for (int r = 0; r < 8; ++r) q[r] = buf_words<WPL, NT>(wres, wv, srow + (uint32_t)r * row_bytes);
        };

        struct ZS {
            WV qz[FPQ];
            u32x4 sc[FPQ][WPL];
        };
        // zeros / scales of the FPQ groups a quad touches (clamped to the last group)
        auto issue_zs = [&](ZS& zs, int quad) {
            const int last_grp = p.K / p.g - 1;
#pragma unroll
            for (int f = 0; f < FPQ; ++f) {
                int grp = (quad * 4 + f * (4 / FPQ)) / spg;
                grp = grp > last_grp ? last_grp : grp;
                const uint32_t ugrp = (uint32_t)__builtin_amdgcn_readfirstlane(grp);
                zs.qz[f] = buf_words<WPL, false>(zres, zvoff, ugrp * row_bytes);
#pragma unroll
                for (int wd = 0; wd < WPL; ++wd)
                    zs.sc[f][wd] = buf_b128(sres, svoff + 16u * wd, ugrp * (uint32_t)p.N * 2u);
            }
        };

        auto mma_step = [&](const WV(&q)[8], const u32x4v& xv) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // rows 4h..4h+3 of this lane: x halves (4h, 4h+1 | 4h+2, 4h+3) = words 2h, 2h+1
                const uint32_t x01 = xv[2 * h], x23 = xv[2 * h + 1];
                const u32x4v a_even = {x01 & 0xFFFFu, x01 >> 16, x23 & 0xFFFFu, x23 >> 16};
                const u32x4v a_odd = {x01 << 16, x01 & 0xFFFF0000u, x23 << 16, x23 & 0xFFFF0000u};
                accsx = mfma16(a_even, ones, accsx);
#pragma unroll
                for (int wd = 0; wd < WPL; ++wd) {
                    const uint32_t q0 = q[4 * h + 0][wd], q1 = q[4 * h + 1][wd];
                    const uint32_t q2 = q[4 * h + 2][wd], q3 = q[4 * h + 3][wd];
                    {
                        const u32x4v b = {pair16<0>(q0), pair16<0>(q1), pair16<0>(q2), pair16<0>(q3)};
                        acc[wd * 8 + 0] = mfma16(a_even, b, acc[wd * 8 + 0]);
                        acc[wd * 8 + 1] = mfma16(a_odd, b, acc[wd * 8 + 1]);
                    }
                    {
                        const u32x4v b = {pair16<1>(q0), pair16<1>(q1), pair16<1>(q2), pair16<1>(q3)};
                        acc[wd * 8 + 2] = mfma16(a_even, b, acc[wd * 8 + 2]);
                        acc[wd * 8 + 3] = mfma16(a_odd, b, acc[wd * 8 + 3]);
                    }
                    {
                        const u32x4v b = {pair16<2>(q0), pair16<2>(q1), pair16<2>(q2), pair16<2>(q3)};
                        acc[wd * 8 + 4] = mfma16(a_even, b, acc[wd * 8 + 4]);
                        acc[wd * 8 + 5] = mfma16(a_odd, b, acc[wd * 8 + 5]);
                    }
                    {
                        const u32x4v b = {pair16<3>(q0), pair16<3>(q1), pair16<3>(q2), pair16<3>(q3)};
                        acc[wd * 8 + 6] = mfma16(a_even, b, acc[wd * 8 + 6]);
                        acc[wd * 8 + 7] = mfma16(a_odd, b, acc[wd * 8 + 7]);
                    }
                }
            }
        };

        // y += s * (acc - (16 + z) * sx) for the group that just ended; reset the accumulators
        auto fold_group = [&](const WV& qz, const u32x4(&svs)[WPL]) {
#pragma unroll
            for (int wd = 0; wd < WPL; ++wd) {
                const u32x4 sv = svs[wd];
                const uint32_t zp[4] = {pair16<0>(qz[wd]), pair16<1>(qz[wd]), pair16<2>(qz[wd]), pair16<3>(qz[wd])};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half2_t z2 = u2h2(zp[j]), s2 = u2h2(sv[j]);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = wd * 8 + 2 * j + e;
#pragma unroll
                        for (int r = 0; r < MR; ++r) {
                            const float raw = __builtin_fmaf(-(float)z2[e], accsx[r], acc[c][r]);
                            yv[c][r] = __builtin_fmaf((float)s2[e], raw, yv[c][r]);
                        }
                        acc[c] = float4_t{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            accsx = float4_t{0.f, 0.f, 0.f, 0.f};
        };

        // ---- deep software pipeline: a ring of 4 K-step buffers (one "quad" = 128 rows).  While
        // step i of quad Q is multiplied, step i of quad Q+1 is already being requested into the
        // buffer step i-1 just released: 3-4 steps (>= 192 B per lane) are always in flight.
        WV w0[8], w1[8], w2[8], w3[8];
        u32x4v x0, x1, x2, x3;
        ZS zsa, zsb;
        const int qbeg = ws >> 2, qend = (we + 3) >> 2;
        const int qpg = spg >> 2;  // quads per group (FPQ == 1 only)
        issue_zs(zsa, qbeg);
        issue_step(w0, x0, 4 * qbeg + 0);
        issue_step(w1, x1, 4 * qbeg + 1);
        issue_step(w2, x2, 4 * qbeg + 2);
        issue_step(w3, x3, 4 * qbeg + 3);
        AWQ_TRACE(1)

        auto quad_body = [&](ZS& zcur, ZS& znext, int quad) {
            const int nb = 4 * (quad + 1);
            issue_zs(znext, quad + 1 < qend ? quad + 1 : quad);
            mma_step(w0, x0);
            issue_step(w0, x0, nb + 0);
            if constexpr (FPQ == 4) fold_group(zcur.qz[0], zcur.sc[0]);
            mma_step(w1, x1);
            issue_step(w1, x1, nb + 1);
            if constexpr (FPQ == 4) fold_group(zcur.qz[1], zcur.sc[1]);
            if constexpr (FPQ == 2) fold_group(zcur.qz[0], zcur.sc[0]);
            mma_step(w2, x2);
            issue_step(w2, x2, nb + 2);
            if constexpr (FPQ == 4) fold_group(zcur.qz[2], zcur.sc[2]);
            mma_step(w3, x3);
            issue_step(w3, x3, nb + 3);
            if constexpr (FPQ == 4) fold_group(zcur.qz[3], zcur.sc[3]);
            if constexpr (FPQ == 2) fold_group(zcur.qz[1], zcur.sc[1]);
            if constexpr (FPQ == 1) {
                if ((quad + 1) % qpg == 0 || quad + 1 >= qend) fold_group(zcur.qz[0], zcur.sc[0]);
            }
        };

        for (int quad = qbeg; quad < qend; quad += 2) {
            quad_body(zsa, zsb, quad);
            if constexpr (TRACE) { if (quad == qbeg) { AWQ_TRACE(2) } }
            if (quad + 1 >= qend) break;
            quad_body(zsb, zsa, quad + 1);
        }
    }

    AWQ_TRACE(3)
    // ---- fold the 4 waves through LDS; D-fragment register r of lane (nl, kb) is row 4*kb + r
    const int S = gridDim.y;
    const bool reducer = (S > 1) && (blockIdx.y == S - 1);
    const int64_t slab = (int64_t)p.M * p.N;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        if (r) __syncthreads();
        float* row = red + (wave * 4 + kb) * CW + nl * CPL;
#pragma unroll
        for (int i = 0; i < CPL / 4; ++i) {
            float4_t v = {yv[4 * i][r], yv[4 * i + 1][r], yv[4 * i + 2][r], yv[4 * i + 3][r]};
            *reinterpret_cast<float4_t*>(row + 4 * ((i + nl) % (CPL / 4))) = v;
        }
        __syncthreads();
        if (r == 0) { AWQ_TRACE(4) }
        for (int e = tid; e < 4 * CW; e += 256) {
            const int ekb = e / CW, c = e % CW;
            const int m = 4 * ekb + r;
            const int cn = c / CPL, ci = (c % CPL) >> 2, ce = c & 3;
            const int off = ekb * CW + cn * CPL + 4 * ((ci + cn) % (CPL / 4)) + ce;
            const int col = blockIdx.x * CW + c;
            if (m >= p.M || col >= p.N) continue;
            float s = red[off] + red[4 * CW + off] + red[8 * CW + off] + red[12 * CW + off];
            if (S > 1) {
                awq_granule_t* g = p.granules + (int64_t)m * p.N + col;
                if (!reducer) {
                    awq_publish(g + (int64_t)blockIdx.y * slab, s);
                    continue;
                }
                float others;
                if (!awq_collect<64>(g, slab, S - 1, others)) {
                    *p.err = 1;
                    others = 0.f;
                }
                awq_clear(g, slab, S - 1);
                s = others + s;
            }
            if (p.bias) s += (float)p.bias[col];
            p.y[(int64_t)m * p.N + col] = (half_t)s;
        }
    }
    AWQ_TRACE(5)
}

template <int WPL, int MR, int FPQ>
void launch_skinny2(const SkinnyParams& p, dim3 grid, bool nt, hipStream_t st) {
    if (nt)
        hipLaunchKernelGGL((awq_skinny_kernel<WPL, MR, true, FPQ>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((awq_skinny_kernel<WPL, MR, false, FPQ>), grid, dim3(256), 0, st, p);
}

template <int WPL, int MR>
void launch_skinny(const SkinnyParams& p, dim3 grid, bool nt, hipStream_t st) {
    if (p.g % 128 == 0) launch_skinny2<WPL, MR, 1>(p, grid, nt, st);
    else if (p.g == 64) launch_skinny2<WPL, MR, 2>(p, grid, nt, st);
    else launch_skinny2<WPL, MR, 4>(p, grid, nt, st);
}

}  // namespace

int awq_skinny_default_split(int K, int N, int wpl) {
    const int CW = 128 * wpl;
    const int tiles = (N + CW - 1) / CW;
    const int nsteps = K / 32;
    int s = (512 + tiles - 1) / tiles;  // ~2 blocks per CU
    const int max_s = (nsteps + 3) / 4;  // at least one K-step per wave
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}

// M <= 16.  wpl: packed words per lane (2 = 8-byte loads, 4 = 16-byte loads).
int awq_launch_gemm_skinny(const AwqGemmArgs& a, int wpl, int splitk, bool nt, void* trace) {
    if (a.M < 1 || a.M > 16) return AWQ_ERR_UNSUPPORTED;
    if (wpl != 2) return AWQ_ERR_UNSUPPORTED;  // 16-byte-per-lane variant (wpl 4) needs 2x the accumulators: retired
    if (a.K % 32 || a.N % (8 * wpl)) return AWQ_ERR_UNSUPPORTED;
    if (!(a.g % 128 == 0 || a.g == 64 || a.g == 32)) return AWQ_ERR_UNSUPPORTED;
    const int CW = 128 * wpl;
    const int tiles = (a.N + CW - 1) / CW;
    const int nsteps = a.K / 32;
    if (splitk < 1) splitk = 1;
    if (splitk > 64) splitk = 64;
    if (splitk > nsteps) splitk = nsteps;
    int spb = (nsteps + splitk - 1) / splitk;  // K-steps per block
    splitk = (nsteps + spb - 1) / spb;
    if (splitk > 1) {
        const size_t need = (size_t)(splitk - 1) * a.M * a.N * sizeof(awq_granule_t);
        if (!a.partial || a.partial_floats * sizeof(float) < need || !a.counters) return AWQ_ERR_WORKSPACE;
    }
    SkinnyParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.bias = reinterpret_cast<const half_t*>(a.bias);
    p.y = reinterpret_cast<half_t*>(a.y);
    p.granules = reinterpret_cast<awq_granule_t*>(a.partial);
    p.err = a.counters ? a.counters + (AWQ_WS_COUNTER_BYTES / 4 - 1) : nullptr;
    p.M = a.M; p.K = a.K; p.N = a.N; p.g = a.g;
    p.steps_per_block = spb;
    p.trace = static_cast<unsigned long long*>(trace);
    dim3 grid(tiles, splitk);
    const int mr = a.M >= 4 ? 4 : a.M;
    if (trace) {
        if (wpl != 2 || mr != 1) return AWQ_ERR_UNSUPPORTED;
        if (a.g % 128) return AWQ_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((awq_skinny_kernel<2, 1, true, 1, true>), grid, dim3(256), 0, a.stream, p);
        return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
    }
    switch (mr) {
        case 1: launch_skinny<2, 1>(p, grid, nt, a.stream); break;
        case 2: launch_skinny<2, 2>(p, grid, nt, a.stream); break;
        case 3: launch_skinny<2, 3>(p, grid, nt, a.stream); break;
        default: launch_skinny<2, 4>(p, grid, nt, a.stream); break;
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
