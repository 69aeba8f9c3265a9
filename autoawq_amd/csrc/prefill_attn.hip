// prefill_attn.hip -- causal attention of S new query rows over a KV cache, flash style, gfx950 (MFMA-bound).
//
// Replaces the reference's prefill call `flash_attn_func(xq, keys, values, causal=True, alibi_slopes=..., softcap=...)`
// (awq/modules/fused/attn.py:269-277).  Rounds 1-4 ran the vendor's scaled_dot_product_attention there (and an fp32 matmul softmax
// when the scores carry ALiBi or a soft cap); measured beside it on the same inputs (profiles/r05_first_call/prefill_attn.txt):
// 0.099 vs 0.141 ms at B 1 x S 2048 x 32 heads, 0.493 vs 0.806 ms at B 8, 0.714 vs 1.280 ms at S 8192, 0.416 vs 0.980 ms at 64 / 8 heads.
// Roofline: MFMA (causal flops = 2 * 2 * B * Hq * 128 * S * (S + 1) / 2 + the chunk's rectangle); algorithmic bytes are negligible.
//
// Shapes: q / out [B, S, Hq, 128] fp16 (after RoPE), caches [Bc, Tmax, Hkv, 128] fp16 already holding rows 0 .. start + S - 1;
// query row s sees cache rows <= start + s.  One block = 128 query rows of one head, four waves of 32 rows; the KV rows are
// walked in tiles of 64 through LDS (two buffers, one barrier per tile, global loads requested a whole iteration ahead).
//
// MFMA mapping (v_mfma_f32_16x16x32_f16: A lane (j, kb) = row j, k 8 kb .. 8 kb + 7; B the same with column j; C lane (j, kb),
// e = row 4 kb + e, column j -- the layouts every GEMM kernel of csrc/ uses):
//   S^T = K Q^T   A = K rows (16 B of a cache row per lane), B = Q rows (16 B of a query row per lane, loaded once)
//                 -> lane (j, kb) holds scores of QUERY j for KV rows 16 t + 4 kb + e: a query row lives in four lanes
//                 (j, j + 16, j + 32, j + 48); max / sum are 15 in-lane ops + two cross-lane steps per tile.
//   O^T = V^T P^T B = P: the lane's own 16 probabilities, converted in place (k slot i of step u = KV row 16 (2 u + i / 4) +
//                 4 kb + i % 4 -- A and B only have to agree on the slot order); A = V^T[d][those rows] = two 8-byte reads of
//                 a TRANSPOSED LDS image of V (built when the tile is written: four cache rows per thread, 8 x v_perm x 2).
//                 -> lane (j, kb) holds O[query j][d = 16 dt + 4 kb + e]: the online-softmax rescale is a lane-local factor.
// LDS images (both conflict-free for the reads; derivation in the probe's README section):
//   K  [64 rows][16 chunks of 16 B], chunk c of row r at slot c ^ (r & 15)
//   V^T [128 d][16 chunks of 8 B = 4 KV rows], chunk c of row d at slot c ^ g(d), g(d) = (d ^ (d >> 3)) & 15
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>

#include "awq_hip.h"
#include "awq_internal.h"
#include "awq_mfma_decode.h"

namespace {

constexpr int HD = 128;        // head dimension
constexpr int BQ = 128;        // query rows per block (32 per wave)
constexpr int BKV = 64;        // KV rows per tile
constexpr int K_TILE = BKV * HD * 2;   // 16 KiB
constexpr int V_TILE = HD * BKV * 2;   // 16 KiB
constexpr int LDS_BYTES = 2 * (K_TILE + V_TILE);

struct PrefillAttnParams {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    half_t* out;
    const float* alibi;  // [Hq] or null
    int B, S, Hq, Hkv, Tmax, start;
    int qblocks, per_group, groups;  // blocks of 128 rows; blocks per (batch, kv head); number of (batch, kv head) pairs
    float scale_log2;    // scale * log2(e)   (no soft cap)
    float scale, softcap;  // soft cap: cap * tanh(s * scale / cap)
};

AWQ_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32: exp2(-inf) = 0

AWQ_DEV uint32_t pack_h2(float a, float b) {
    half2_t h = {(half_t)a, (half_t)b};
    return h22u(h);
}

template <bool MODS>  // MODS: ALiBi slopes and / or a soft cap on the scores
__global__ __launch_bounds__(256, 2) void awq_prefill_attn_kernel(PrefillAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kb = lane >> 4;

    // ---- which (batch, kv head, query head, row block): the blocks of one (batch, kv head) share an XCD (blockIdx % 8), so
    // its K / V rows are fetched into ONE L2; the heaviest row blocks (the last ones: most KV tiles) first
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = xcd + 8 * (slot / p.per_group);
    if (group >= p.groups) return;
    const int within = slot % p.per_group;
    const int b = group / p.Hkv, kvh = group % p.Hkv;
    const int hq_per = p.Hq / p.Hkv;
    const int qb = p.qblocks - 1 - within / hq_per;
    const int h = kvh * hq_per + within % hq_per;
    const int q0 = qb * BQ;

    const uint32_t q_row_bytes = (uint32_t)p.Hq * HD * 2u, kv_row_bytes = (uint32_t)p.Hkv * HD * 2u;
    const int kv_len = p.start + p.S;
    const rsrc_t qres = mk_rsrc(p.q + (size_t)b * p.S * p.Hq * HD, (uint32_t)p.S * q_row_bytes);
    const rsrc_t kres = mk_rsrc(p.k + (size_t)b * p.Tmax * p.Hkv * HD, (uint32_t)kv_len * kv_row_bytes);  // rows >= kv_len read 0
    const rsrc_t vres = mk_rsrc(p.v + (size_t)b * p.Tmax * p.Hkv * HD, (uint32_t)kv_len * kv_row_bytes);

    // ---- Q fragments: query 16 qt + j of this wave, 16 B at d = 32 ks + 8 kb (rows >= S: zeros, never stored)
    const int qw0 = q0 + 32 * wave;
    u32x4v qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const uint32_t row = (uint32_t)(qw0 + 16 * qt + j);
        const uint32_t off = row < (uint32_t)p.S ? row * q_row_bytes + (uint32_t)h * (HD * 2u) + 16u * (uint32_t)kb : OOB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qt][ks] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(qres, off, 64 * ks, 0));
    }

    // ---- tile staging: K thread (row tid / 16 + 16 i, chunk tid % 16); V thread (rows 4 (tid / 16) + r, d block tid % 16)
    const int sc = tid & 15, sr = tid >> 4;
    const uint32_t kv_col = (uint32_t)kvh * (HD * 2u) + 16u * (uint32_t)sc;
    // Two register sets, each requested a whole iteration before it is written to LDS: at the top of iteration `it` the K rows of
    // tile it + 1 (requested at the top of it - 1) go to the other LDS buffer -- free since the previous barrier -- and tile it + 2 is
    // requested; the V rows likewise after S^T = K Q^T.
    u32x4v kst[4], vst[4];
    // The staging loads are inline asm with hand-counted waits, like the GEMM kernels of csrc/ (vector-memory operations retire
    // in order: "four newer requests may be pending" is exact).  hipcc's own bookkeeping ends the loop's first wait at vmcnt(0),
    // i.e. it also waits for the V rows requested half an iteration ago (measured: +3 .. 10 % for the counted form).
    auto srd4 = [](const void* base, uint32_t bytes) -> u32x4 {
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
    };
    const u32x4 ksrd = srd4(p.k + (size_t)b * p.Tmax * p.Hkv * HD, (uint32_t)kv_len * kv_row_bytes);
    const u32x4 vsrd = srd4(p.v + (size_t)b * p.Tmax * p.Hkv * HD, (uint32_t)kv_len * kv_row_bytes);
#define PATTN_LOAD4(R, o0, o1, o2, o3, rs)                                                                                          \
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %8, 0 offen\n\tbuffer_load_dwordx4 %1, %5, %8, 0 offen\n\t"                 \
                 "buffer_load_dwordx4 %2, %6, %8, 0 offen\n\tbuffer_load_dwordx4 %3, %7, %8, 0 offen"                                 \
                 : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3])  /* early-clobber: never the register of an address operand */ \
                 : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rs))
#define PATTN_WAIT4(R, newer) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]) : "n"(newer))
    auto load_k = [&](int kv0) {
        const uint32_t o = (uint32_t)(kv0 + sr) * kv_row_bytes + kv_col, d = 16u * kv_row_bytes;
        PATTN_LOAD4(kst, o, o + d, o + 2 * d, o + 3 * d, ksrd);
    };
    auto load_v = [&](int kv0) {
        const uint32_t o = (uint32_t)(kv0 + 4 * sr) * kv_row_bytes + kv_col, d = kv_row_bytes;
        PATTN_LOAD4(vst, o, o + d, o + 2 * d, o + 3 * d, vsrd);
    };
    // (LDS addresses are recomputed from lane coordinates made opaque once per iteration: left loop-invariant, the compiler
    //  keeps all ~50 of them in registers across the tile loop and spills the accumulators instead)
    auto write_k = [&](int buf, int sc, int sr) {
        char* ks_ = smem + buf * (K_TILE + V_TILE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = sr + 16 * i;
            *reinterpret_cast<u32x4v*>(ks_ + r * 256 + 16 * (sc ^ (r & 15))) = kst[i];
        }
    };
    auto write_v = [&](int buf, int sc, int sr) {
        char* vt_ = smem + buf * (K_TILE + V_TILE) + K_TILE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // d = 8 sc + i: the four KV rows 4 sr .. 4 sr + 3 of that column, one 8-byte chunk
            const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;
            u32x2 t;
            t[0] = __builtin_amdgcn_perm(vst[1][i >> 1], vst[0][i >> 1], sel);
            t[1] = __builtin_amdgcn_perm(vst[3][i >> 1], vst[2][i >> 1], sel);
            const int d = 8 * sc + i;
            const int g = (d ^ (d >> 3)) & 15;
            *reinterpret_cast<u32x2*>(vt_ + d * 128 + 8 * (sr ^ g)) = t;
        }
    };

    // ---- per-query state (query qt: 16 qt + j; the four lanes of a query hold the same m, their own part of l)
    float4_t oacc[2][8];
    float m_run[2], l_run[2];
    // The row sums come from the matrix pipe (measured +2 .. 8 % over 32 v_add_f32 per lane and tile) -- one more MFMA per (query
    // tile, k step) with an all-ones A operand: C[any row][query j] = sum over the step's 32 KV slots of P (the fp16 values the
    // numerator uses), already summed over the query's four lanes -- instead of 32 v_add_f32 per lane and tile.
    float4_t lacc[2] = {float4_t{0.f, 0.f, 0.f, 0.f}, float4_t{0.f, 0.f, 0.f, 0.f}};
    const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) oacc[qt][dt] = float4_t{0.f, 0.f, 0.f, 0.f};
    }
    int qpos[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) qpos[qt] = p.start + qw0 + 16 * qt + j;
    const int qpos_wave_min = p.start + qw0;
    float slope = 0.f;
    if constexpr (MODS) slope = p.alibi ? p.alibi[h] * 1.44269504088896f : 0.f;

    const int last_row = min(q0 + BQ, p.S) - 1;
    const int ntiles = (p.start + last_row) / BKV + 1;

    load_k(0);
    load_v(0);
    PATTN_WAIT4(kst, 0);
    PATTN_WAIT4(vst, 0);
    // the compiler's own wait for the Q fragments has to happen HERE: left pending, its bookkeeping carries them into the loop and
    // puts a vmcnt(0) in front of the first MFMAs of every iteration -- behind the K rows just requested
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qt][ks]));
    write_k(0, sc, sr);
    write_v(0, sc, sr);
    load_k(BKV);
    load_v(BKV);
    __syncthreads();

    // The iteration is straight-line on purpose (no "is there a next tile", no "does this wave see the tile"): the tile after the
    // last one lies outside the descriptors (zeros, no traffic) and lands in the LDS buffer nobody reads again; a tile a wave's
    // rows do not see is all -inf to it (p = 0).  Conditional staging made the compiler keep two copies of the staging registers.
    for (int it = 0; it < ntiles; ++it) {
        const int buf = it & 1, kv0 = it * BKV;
        const char* ks_ = smem + buf * (K_TILE + V_TILE);
        const char* vt_ = ks_ + K_TILE;
        int jo = j, kbo = kb, sco = sc, sro = sr;
        asm volatile("" : "+v"(jo), "+v"(kbo), "+v"(sco), "+v"(sro));
        PATTN_WAIT4(kst, 4);          // (asm variant) the four V requests issued after these may still be pending
        write_k(buf ^ 1, sco, sro);   // tile it + 1, requested an iteration ago
        load_k(kv0 + 2 * BKV);

        // ---- S^T = K Q^T
        float4_t sacc[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc[qt][t] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4v kf = *reinterpret_cast<const u32x4v*>(ks_ + (16 * t + jo) * 256 + 16 * ((4 * ks + kbo) ^ jo));
                sacc[0][t] = mfma16(kf, qf[0][ks], sacc[0][t]);
                sacc[1][t] = mfma16(kf, qf[1][ks], sacc[1][t]);
            }
            __builtin_amdgcn_sched_barrier(0);  // keeps the LDS reads of later KV sub-tiles from being hoisted (register budget)
        }
        PATTN_WAIT4(vst, 4);
        write_v(buf ^ 1, sco, sro);
        load_v(kv0 + 2 * BKV);

        // ---- online softmax in the log2 domain; lane (j, kb) holds KV rows kv0 + 16 t + 4 kb + e of query 16 qt + j
        if constexpr (MODS) {  // scores -> log2 units with the cap and the position bias applied (sc2 = 1 below)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float sv = sacc[qt][t][e] * p.scale;
                        if (p.softcap > 0.f) sv = p.softcap * tanhf(sv / p.softcap);
                        sacc[qt][t][e] = sv * 1.44269504088896f + slope * (float)(kv0 + 16 * t + 4 * kbo + e - qpos[qt]);
                    }
        }
        if (kv0 + BKV - 1 > qpos_wave_min) {  // (wave-uniform) the tile reaches past some query of this wave
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kv0 + 16 * t + 4 * kbo + e > qpos[qt]) sacc[qt][t][e] = -INFINITY;
        }
        const float sc2 = MODS ? 1.0f : p.scale_log2;  // > 0: the maximum commutes with it
        u32x4v pf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mloc = fmaxf(fmaxf(sacc[qt][0][0], sacc[qt][0][1]), fmaxf(sacc[qt][0][2], sacc[qt][0][3]));
#pragma unroll
            for (int t = 1; t < 4; ++t)
                mloc = fmaxf(mloc, fmaxf(fmaxf(sacc[qt][t][0], sacc[qt][t][1]), fmaxf(sacc[qt][t][2], sacc[qt][t][3])));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m_run[qt], mloc * sc2);  // finite from the first tile on: KV row 0 is visible to every query
            const float alpha = fast_exp2(m_run[qt] - m_new);
            m_run[qt] = m_new;
            float psum = 0.f;
            float pr[16];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = fast_exp2(fmaf(sacc[qt][t][e], sc2, -m_new));
                    pr[4 * t + e] = pv;
                }
            l_run[qt] = l_run[qt] * alpha + psum;
            if (!__all(alpha == 1.0f)) {  // (wave-uniform) the running maximum moved for some query of the wave
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) oacc[qt][dt] *= alpha;
                lacc[qt] *= alpha;
            }
            // B fragments of the two k steps: slots 0-3 = sub-tile 2 u (e = 0 .. 3), slots 4-7 = sub-tile 2 u + 1
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pf[qt][u][0] = pack_h2(pr[8 * u + 0], pr[8 * u + 1]);
                pf[qt][u][1] = pack_h2(pr[8 * u + 2], pr[8 * u + 3]);
                pf[qt][u][2] = pack_h2(pr[8 * u + 4], pr[8 * u + 5]);
                pf[qt][u][3] = pack_h2(pr[8 * u + 6], pr[8 * u + 7]);
                lacc[qt] = mfma16(ones, pf[qt][u], lacc[qt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            // g(16 dt + j) = (j ^ (j >> 3)) ^ 2 dt: slot = (compile-time constant) ^ (lane constant)
            const int lane_x = 8 * (kbo ^ jo ^ (jo >> 3));
            const char* row = vt_ + (16 * dt + jo) * 128;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x2 lo = *reinterpret_cast<const u32x2*>(row + (lane_x ^ (8 * ((8 * u) ^ (2 * dt)))));
                const u32x2 hi = *reinterpret_cast<const u32x2*>(row + (lane_x ^ (8 * ((8 * u + 4) ^ (2 * dt)))));
                const u32x4v vf = {lo[0], lo[1], hi[0], hi[1]};
                oacc[0][dt] = mfma16(vf, pf[0][u], oacc[0][dt]);
                oacc[1][dt] = mfma16(vf, pf[1][u], oacc[1][dt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // (asm variant) the last requests are never consumed: one wait that names both register sets keeps them allocated until they
    // have landed -- otherwise their registers are free for the temporaries below while the loads are still in flight
    PATTN_WAIT4(kst, 0);
    PATTN_WAIT4(vst, 0);

    // ---- finish: the query's sum over its four lanes, O / l, fp16 store (lane: 4 consecutive d per dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float l = lacc[qt][0];  // every row of the ones product is the query's sum, over all four lanes' slots
        const float inv = 1.0f / l;
        const int row = qw0 + 16 * qt + j;
        if (row >= p.S) continue;
        half_t* dst = p.out + ((size_t)(b * p.S + row) * p.Hq + h) * HD + 4 * kb;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const float4_t o = oacc[qt][dt] * inv;
            u32x2 w;
            w[0] = pack_h2(o[0], o[1]);
            w[1] = pack_h2(o[2], o[3]);
            *reinterpret_cast<u32x2*>(dst + 16 * dt) = w;
        }
    }
}

}  // namespace

// q / out [B, S, Hq, 128] fp16, k / v caches [>= B, Tmax, Hkv, 128] fp16 holding rows 0 .. start + S - 1 of every batch entry;
// scale: the score factor (1 / sqrt(128) by default); softcap 0 = none; alibi_slopes [Hq] fp32 on the device or null.
// AWQ_ERR_UNSUPPORTED: head_dim != 128, Hq % Hkv, tensors too large for 32-bit offsets.
int awq_prefill_attention(const uint16_t* q, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* out, int64_t B_, int64_t S_,
                          int64_t Hq_, int64_t Hkv_, int64_t head_dim, int64_t Tmax_, int64_t start_, float scale, float softcap,
                          const float* alibi_slopes, void* stream) {
    if (B_ < 0 || S_ < 0 || Hq_ <= 0 || Hkv_ <= 0 || start_ < 0 || Tmax_ < 0 || B_ > INT32_MAX || S_ > INT32_MAX || Hq_ > 65535 || Hkv_ > 65535 ||
        Tmax_ > INT32_MAX || start_ > INT32_MAX)
        return AWQ_ERR_BAD_SHAPE;
    const int B = (int)B_, S = (int)S_, Hq = (int)Hq_, Hkv = (int)Hkv_, Tmax = (int)Tmax_, start = (int)start_;
    if (!q || !k_cache || !v_cache || !out) return AWQ_ERR_NULL;
    if (B < 0 || S < 0 || Hq <= 0 || Hkv <= 0 || start < 0 || start + S > Tmax) return AWQ_ERR_BAD_SHAPE;
    if (head_dim != HD || Hq % Hkv) return AWQ_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)out) & 15) return AWQ_ERR_BAD_ALIGNMENT;
    if (B == 0 || S == 0) return AWQ_OK;
    if ((uint64_t)S * Hq * HD * 2 >= (1ull << 31) || (uint64_t)(start + S) * Hkv * HD * 2 >= (1ull << 31)) return AWQ_ERR_UNSUPPORTED;
    PrefillAttnParams p;
    p.q = reinterpret_cast<const half_t*>(q);
    p.k = reinterpret_cast<const half_t*>(k_cache);
    p.v = reinterpret_cast<const half_t*>(v_cache);
    p.out = reinterpret_cast<half_t*>(out);
    p.alibi = alibi_slopes;
    p.B = B; p.S = S; p.Hq = Hq; p.Hkv = Hkv; p.Tmax = Tmax; p.start = start;
    p.qblocks = (S + BQ - 1) / BQ;
    p.per_group = p.qblocks * (Hq / Hkv);
    p.groups = B * Hkv;
    p.scale = scale;
    p.softcap = softcap;
    p.scale_log2 = scale * 1.44269504088896f;
    const int rounds = (p.groups + 7) / 8;
    const dim3 grid((unsigned)(8 * rounds * p.per_group));
    hipStream_t st = static_cast<hipStream_t>(stream);
    static std::atomic<unsigned long long> opted0{0}, opted1{0};
    if (alibi_slopes || softcap > 0.f) {
        (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_prefill_attn_kernel<true>), opted1, LDS_BYTES);
        hipLaunchKernelGGL((awq_prefill_attn_kernel<true>), grid, dim3(256), LDS_BYTES, st, p);
    } else {
        (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_prefill_attn_kernel<false>), opted0, LDS_BYTES);
        hipLaunchKernelGGL((awq_prefill_attn_kernel<false>), grid, dim3(256), LDS_BYTES, st, p);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
