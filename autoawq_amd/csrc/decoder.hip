// decoder.hip -- the small kernels around the int4 Linears of a fused decoder block
// (SURVEY.md 8f rank 2): RMSNorm (+ residual add), RoPE + KV-cache append, single-query attention
// over the cache.  All HBM-bound byte movers; fp32 arithmetic, one rounding to fp16.
//
// What they replace in the reference (under /root/reference):
//   awq_rmsnorm         awq_ext.layernorm_forward_cuda(x, weight, out, eps)   awq/modules/fused/norm.py:33-36
//   awq_add_rmsnorm     `h = hidden_states + attn_output` + the next norm     awq/modules/fused/block.py:108-119
//   awq_rope_kv_append  RoPE.forward + WindowedCache.update_kv                awq/modules/fused/attn.py:54-87,265-276, cache.py:40-45
//   awq_decode_attention  flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens, causal=True)  attn.py:291-302
// (awq_ext / flash-attn sources are not in the reference tree: semantics from the call sites and
// from the torch code around them.)
#include "awq_device.h"
#include "awq_internal.h"

namespace {

// Cross-lane moves INSIDE a 16-lane row as DPP operands (a few cycles) instead of __shfl_xor (ds_bpermute: an LDS-crossbar round
// trip of ~100 cycles each, and the attention inner loop made four of them per score, dependent on each other).
template <int CTRL>
AWQ_DEV float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a row, in every lane: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8
#ifndef AWQ_ATTN_NO_DPP
#define AWQ_ATTN_NO_DPP 0  // experiment builds: 1 = the __shfl_xor butterflies of rounds 1-3
#endif
// (dpp: uniform.  A/B on the whole decoder, profiles/r04_attention_split_ab.txt: the DPP form gains 1.5-3 % in single-split launches --
// short contexts, where the kernel is a latency chain on 32 blocks -- and LOSES 6-10 % at 2048 tokens of context, where the
// launch is memory-bound and the faster consumption only puts more requests in flight; so the launch shape picks the form.)
AWQ_DEV float row16_sum(float s, bool dpp) {
    if (AWQ_ATTN_NO_DPP || !dpp) {
#pragma unroll
        for (int x = 8; x > 0; x >>= 1) s += __shfl_xor(s, x, 64);
        return s;
    }
    s += dpp_f<0xB1>(s);
    s += dpp_f<0x4E>(s);
    s += dpp_f<0x124>(s);
    s += dpp_f<0x128>(s);
    return s;
}

AWQ_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- RMSNorm (+ residual add)
// One 256-thread block per row; thread t owns 16-byte chunks t, t+256, ... (H % 8 == 0).
// y = fp16( x * rsqrt(mean(x^2) + eps) * w ), all in fp32, one rounding.
// ADD: r = fp16(residual + x) is written back to `residual` first and is what gets normalised
// (the fp16 rounding of the sum is what torch's `h = hidden_states + attn_output` hands the norm).
template <bool ADD>
__global__ __launch_bounds__(256) void awq_rmsnorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ residual,
                                                         const half_t* __restrict__ w, half_t* __restrict__ out, int H,
                                                         float eps) {
    constexpr int MAXC = 8;  // chunks per thread kept in registers: H <= 16384
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int chunks = H >> 3;
    half8_t v[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i;
        if (c < chunks) {
            v[i] = *reinterpret_cast<const half8_t*>(x + (int64_t)row * H + 8 * c);
            if constexpr (ADD) {
                const half8_t r = *reinterpret_cast<const half8_t*>(residual + (int64_t)row * H + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = (half_t)((float)v[i][e] + (float)r[e]);
                *reinterpret_cast<half8_t*>(residual + (int64_t)row * H + 8 * c) = v[i];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)v[i][e] * (float)v[i][e];
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i;
        if (c < chunks) {
            const half8_t g = *reinterpret_cast<const half8_t*>(w + 8 * c);
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[i][e] * inv * (float)g[e]);
            *reinterpret_cast<half8_t*>(out + (int64_t)row * H + 8 * c) = o;
        }
    }
}

// ---------------------------------------------------------------- RoPE + KV append
// qkv [tokens = B*S, (Hq + 2*Hkv) * D] fp16 as the fused qkv Linear writes it (q heads | k heads |
// v heads).  One block per (token, head slot), D/2 threads; thread i rotates the pair
// (i, i + rot/2) of a q or k head by the angle of position start + s (cos/sin tables
// [max_pos, rot/2] fp32, built like RoPE.precompute_freqs_cis), passes dims >= rot through, and
// writes q to q_out [B*S, Hq, D], k / v to the caches [B, Tmax, Hkv, D] at row start + s.
__global__ __launch_bounds__(128) void awq_rope_kv_append_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ q_out,
                                                                half_t* __restrict__ k_cache, half_t* __restrict__ v_cache,
                                                                const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                const int* __restrict__ pos_dev, int start_pos, int S, int Hq,
                                                                int Hkv, int D, int rot, int Tmax) {
    const int slot = blockIdx.x % (Hq + 2 * Hkv), tok = blockIdx.x / (Hq + 2 * Hkv);
    const int b = tok / S, s = tok % S;
    // A device-side position (hipGraph replay) is not seen by the host checks: clamp it to the cache so that a
    // replay past max_seq_len overwrites the last row instead of memory behind the cache / the cos-sin tables
    // (the caller is expected to roll the window before that: modules/fused/model.py).
    const int pos = min((pos_dev ? *pos_dev : start_pos) + s, Tmax - 1);
    const int i = threadIdx.x;  // 0 .. D/2-1
    const half_t* src = qkv + (int64_t)tok * (Hq + 2 * Hkv) * D + (int64_t)slot * D;
    half_t* dst;
    if (slot < Hq) dst = q_out + ((int64_t)tok * Hq + slot) * D;
    else if (slot < Hq + Hkv) dst = k_cache + (((int64_t)b * Tmax + pos) * Hkv + (slot - Hq)) * D;
    else dst = v_cache + (((int64_t)b * Tmax + pos) * Hkv + (slot - Hq - Hkv)) * D;
    const int half_rot = rot >> 1;
    if (slot < Hq + Hkv && i < half_rot) {
        const float c = cos_t[(int64_t)pos * half_rot + i], sn = sin_t[(int64_t)pos * half_rot + i];
        const float a = (float)src[i], bb = (float)src[i + half_rot];
        dst[i] = (half_t)(a * c - bb * sn);
        dst[i + half_rot] = (half_t)(a * sn + bb * c);
    } else if (slot < Hq + Hkv) {  // pass-through part of a partially rotated head: dims rot .. D-1
        const int d0 = rot + 2 * (i - half_rot);
        if (d0 < D) dst[d0] = src[d0];
        if (d0 + 1 < D) dst[d0 + 1] = src[d0 + 1];
    } else {
        dst[2 * i] = src[2 * i];
        dst[2 * i + 1] = src[2 * i + 1];
    }
}

// ---------------------------------------------------------------- single-query attention over the cache
// grid (splits, Hkv, B), 256 threads.  A block owns one KV head, the G = Hq/Hkv query heads that
// share it and a contiguous chunk of the sequence.  Lane (r = lane >> 4, c = lane & 15) reads the
// 16 bytes [8c, 8c+8) of cache row t0 + 4*(wave + 4*i) + r: a wave instruction covers four whole
// 256-byte rows (D = 128).  (Round 4 tried HEAD-major caches [B, Hkv, Tmax, D] -- four consecutive rows of a head = 1 KiB of contiguous
// memory per instruction, on the theory that the 2.1 TB/s this kernel reaches at 2048 tokens is the rate of scattered 256-byte
// pieces: no gain at 64 / 512 tokens, -1.5 % at 2048 and -6 % at 200 (one block then reads 50 KB from one region); reverted,
// profiles/r04_attention_split_ab.txt.  The kernel is bound by its per-block latency chain, not by the access granularity.)  Scores by 16-lane butterfly, online softmax per lane group, P*V into
// 8 fp32 accumulators per query head; lane groups, waves and finally the splits are merged with
// the usual (max, sum, acc) rule.  Partial results of a split go to `part` [B, Hq, splits, D + 2]
// fp32 and a second tiny kernel finishes; with one split the block writes fp16 directly.
// FUSED (awq_decode_attention_rope): `q` is the raw fused-qkv row [B, (Hq + 2 Hkv) * D]; the kernel
// rotates its query heads itself (partner dims d +- 64 live in lane c ^ 8), takes the NEW token's
// rotated k and its v straight from that row -- cache row `pos` is not read by anybody in this
// launch -- and the one block whose chunk holds `pos` appends them to the caches.  len_dev then
// holds the POSITION (length - 1).  Same roundings as awq_rope_kv_append followed by the plain
// kernel: rotated values pass through fp16.
// EXTRA (plain form only): logit soft-capping s := cap * tanh(s / cap) (attn.py:150,166 `softcap=`) and ALiBi, a per-head linear
// bias slope_h * (t - pos) on the key position (attn.py:89-125,149,165 `alibi_slopes=`), in that order like flash-attn.
template <int G, bool FUSED, bool EXTRA = false>
__global__ __launch_bounds__(256) void awq_decode_attn_kernel(const half_t* __restrict__ q, half_t* __restrict__ kc,
                                                             half_t* __restrict__ vc, half_t* __restrict__ out,
                                                             float* __restrict__ part, const int* __restrict__ len_dev,
                                                             int seq_len, int Hq, int Hkv, int Tmax, float scale,
                                                             int chunk, const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t, float softcap,
                                                             const float* __restrict__ alibi) {
    static_assert(!(FUSED && EXTRA), "soft-capping / ALiBi ride on the plain form");
    constexpr int D = 128;
    __shared__ float sm[4][G][D + 2];
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 4, c = lane & 15;
    // A launch with ONE split starts at row 0 whatever the length is: its first row sets are requested HERE, before the length,
    // the rotation tables and the query have made their own round trips (at 64 tokens of context the kernel is nothing but a chain
    // of dependent round trips on 32 blocks; rows at or past the length are requested too -- they exist, the cache is Tmax rows --
    // and never used: `live` below).  Later row sets and launches with several splits request inside the loop as before.
    constexpr int U = 4;  // row sets requested before any is consumed: 8 loads of 16 bytes in flight per lane
    const int64_t rowstride = (int64_t)Hkv * D;
    const half_t* kb = kc + ((int64_t)b * Tmax * Hkv + hk) * D + 8 * c;
    const half_t* vb = vc + ((int64_t)b * Tmax * Hkv + hk) * D + 8 * c;
#ifndef AWQ_ATTN_NO_EARLY
#define AWQ_ATTN_NO_EARLY 0  // experiment builds: 1 = request every row set inside the loop (round 3's order)
#endif
    const bool early = gridDim.x == 1 && !AWQ_ATTN_NO_EARLY;
    half8_t kv0[U], vv0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int t = 4 * wave + 16 * u + r;
        kv0[u] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        vv0[u] = kv0[u];
        if (early && t < Tmax) {
            kv0[u] = *reinterpret_cast<const half8_t*>(kb + (int64_t)t * rowstride);
            vv0[u] = *reinterpret_cast<const half8_t*>(vb + (int64_t)t * rowstride);
        }
    }
    // device-side lengths are clamped to the cache (see awq_rope_kv_append_kernel): never a row >= Tmax
    const int T = max(1, min((len_dev ? *len_dev : seq_len) + (FUSED ? 1 : 0), Tmax));
    const int pos = T - 1;  // FUSED: the new token's position
    // The rows are dealt to the splits of the launch HERE, from the length the device knows (round 4): a launch sized on the host
    // for a longer context (a hipGraph serves every length up to its bucket's bound) still spreads the rows that exist over
    // all of its splits instead of leaving them to the first few.  16-row granularity (a wave instruction covers 4 rows).
    if (chunk <= 0) chunk = (((T + (int)gridDim.x - 1) / (int)gridDim.x) + 15) & ~15;
    const int t0 = split * chunk, t1 = min(T, t0 + chunk);

    float qf[G][8];
    half8_t k_new = {0, 0, 0, 0, 0, 0, 0, 0}, v_new = k_new;
    if constexpr (FUSED) {
        const half_t* row = q + (int64_t)b * (Hq + 2 * Hkv) * D;
        float cs[8], sn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = (8 * c + e) & 63;
            cs[e] = cos_t[(int64_t)pos * 64 + i];
            sn[e] = sin_t[(int64_t)pos * 64 + i];
        }
        auto rotate = [&](half8_t x) -> half8_t {
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float mine = (float)x[e], other = AWQ_ATTN_NO_DPP ? __shfl_xor(mine, 8, 64) : dpp_f<0x128>(mine);  // row_ror:8 = lane c ^ 8 of the row: dims d and d +- 64
                o[e] = (c < 8) ? (half_t)(mine * cs[e] - other * sn[e]) : (half_t)(other * sn[e] + mine * cs[e]);
            }
            return o;
        };
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const half8_t qr = rotate(*reinterpret_cast<const half8_t*>(row + (int64_t)(hk * G + g) * D + 8 * c));
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[g][e] = (float)qr[e] * scale;
        }
        k_new = rotate(*reinterpret_cast<const half8_t*>(row + (int64_t)(Hq + hk) * D + 8 * c));
        v_new = *reinterpret_cast<const half8_t*>(row + (int64_t)(Hq + Hkv + hk) * D + 8 * c);
        if (pos >= t0 && pos < t1 && wave == 0 && r == 0) {  // append: one block, 16 lanes x 16 bytes per cache
            *reinterpret_cast<half8_t*>(kc + (((int64_t)b * Tmax + pos) * Hkv + hk) * D + 8 * c) = k_new;
            *reinterpret_cast<half8_t*>(vc + (((int64_t)b * Tmax + pos) * Hkv + hk) * D + 8 * c) = v_new;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const half8_t qv = *reinterpret_cast<const half8_t*>(q + ((int64_t)b * Hq + hk * G + g) * D + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[g][e] = (float)qv[e] * scale;
        }
    }
    float slope[G];
#pragma unroll
    for (int g = 0; g < G; ++g) slope[g] = (EXTRA && alibi) ? alibi[hk * G + g] : 0.f;
    const float inv_cap = (EXTRA && softcap > 0.f) ? 1.0f / softcap : 0.f;
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
    }
    for (int tb = t0 + 4 * wave; tb < t1; tb += 16 * U) {  // wave-uniform trip count
        half8_t kv[U], vv[U];
        const bool first = early && tb == 4 * wave;  // (uniform) these row sets were requested at the top of the kernel
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tb + 16 * u + r;
            kv[u] = kv0[u];
            vv[u] = vv0[u];
            if (!first) {
                kv[u] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                vv[u] = kv[u];
                if (t < t1 && !(FUSED && t == pos)) {
                    kv[u] = *reinterpret_cast<const half8_t*>(kb + (int64_t)t * rowstride);
                    vv[u] = *reinterpret_cast<const half8_t*>(vb + (int64_t)t * rowstride);
                }
            }
            if (FUSED && t == pos) {  // the new token: from the qkv row, not from the cache
                kv[u] = k_new;
                vv[u] = v_new;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = tb + 16 * u + r < t1;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += qf[g][e] * (float)kv[u][e];
                s = row16_sum(s, early);  // sum over the 16 lanes of the row
                if constexpr (EXTRA) {
                    if (inv_cap > 0.f) s = softcap * tanhf(s * inv_cap);
                    s += slope[g] * (float)(tb + 16 * u + r - pos);
                }
                if (live) {
                    const float mn = fmaxf(m[g], s);
                    const float corr = __expf(m[g] - mn), p = __expf(s - mn);
                    l[g] = l[g] * corr + p;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[g][e] = o[g][e] * corr + p * (float)vv[u][e];
                    m[g] = mn;
                }
            }
        }
    }
    // merge the four lane groups of the wave (lanes c, c+16, c+32, c+48 hold the same dims)
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int x = 16; x <= 32; x <<= 1) {
            const float mo = __shfl_xor(m[g], x, 64), lo = __shfl_xor(l[g], x, 64);
            const float mn = fmaxf(m[g], mo);
            const float ca = (m[g] == -INFINITY) ? 0.f : __expf(m[g] - mn), cb = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[g][e] = o[g][e] * ca + __shfl_xor(o[g][e], x, 64) * cb;
            l[g] = l[g] * ca + lo * cb;
            m[g] = mn;
        }
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[wave][g][8 * c + e] = o[g][e];
            if (c == 0) {
                sm[wave][g][D] = m[g];
                sm[wave][g][D + 1] = l[g];
            }
        }
    }
    __syncthreads();
    // merge the four waves: thread (g, d) for g < G, d < 128 -> G*128 threads (G <= 2) or a loop
    const int nsplit = gridDim.x;
    for (int idx = tid; idx < G * D; idx += 256) {
        const int g = idx / D, d = idx % D;
        float mn = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sm[w][g][D]);
        float acc = 0.f, ls = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float cw = (sm[w][g][D] == -INFINITY) ? 0.f : __expf(sm[w][g][D] - mn);
            acc += sm[w][g][d] * cw;
            ls += sm[w][g][D + 1] * cw;
        }
        const int64_t head = (int64_t)b * Hq + hk * G + g;
        if (nsplit == 1) {
            out[head * D + d] = (half_t)(ls > 0.f ? acc / ls : 0.f);
        } else {
            float* pp = part + (head * nsplit + split) * (D + 2);
            pp[d] = acc;
            if (d == 0) {
                pp[D] = mn;
                pp[D + 1] = ls;
            }
        }
    }
}

// One block per (sequence, query head): lanes of wave 0 take one split each (max, weights, sum by
// wave reductions), then thread d adds the splits' accumulators with independent loads.  (The first
// version walked the splits serially twice: 12 us for 32 splits, more than the attention itself.)
__global__ __launch_bounds__(128) void awq_decode_attn_combine_kernel(const float* __restrict__ part, half_t* __restrict__ out,
                                                                     int nsplit) {
    constexpr int D = 128;
    __shared__ float wgt[64];
    __shared__ float denom;
    const int64_t head = blockIdx.x;
    const int d = threadIdx.x;
    const float* pp = part + head * nsplit * (D + 2);
    if (d < 64) {  // nsplit <= 64
        const float ms = d < nsplit ? pp[d * (D + 2) + D] : -INFINITY;
        const float ls = d < nsplit ? pp[d * (D + 2) + D + 1] : 0.f;
        float mn = ms;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = fmaxf(mn, __shfl_xor(mn, o, 64));
        const float w = (ms == -INFINITY) ? 0.f : __expf(ms - mn);
        wgt[d] = w;
        const float tot = wave_sum(ls * w);
        if (d == 0) denom = tot;
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < nsplit; ++s) acc += pp[s * (D + 2) + d] * wgt[s];
    out[head * D + d] = (half_t)(denom > 0.f ? acc / denom : 0.f);
}

}  // namespace

int awq_launch_rmsnorm(const uint16_t* x, uint16_t* residual, const uint16_t* w, uint16_t* out, int64_t M, int64_t H,
                       float eps, hipStream_t st) {
    if (M < 0 || H <= 0 || H % 8 || H > 16384) return AWQ_ERR_BAD_SHAPE;
    if (M == 0) return AWQ_OK;
    if (residual)
        hipLaunchKernelGGL(awq_rmsnorm_kernel<true>, dim3((unsigned)M), dim3(256), 0, st, reinterpret_cast<const half_t*>(x),
                           reinterpret_cast<half_t*>(residual), reinterpret_cast<const half_t*>(w),
                           reinterpret_cast<half_t*>(out), (int)H, eps);
    else
        hipLaunchKernelGGL(awq_rmsnorm_kernel<false>, dim3((unsigned)M), dim3(256), 0, st, reinterpret_cast<const half_t*>(x),
                           nullptr, reinterpret_cast<const half_t*>(w), reinterpret_cast<half_t*>(out), (int)H, eps);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_launch_rope_kv_append(const uint16_t* qkv, uint16_t* q_out, uint16_t* k_cache, uint16_t* v_cache, const float* cos_t,
                              const float* sin_t, const int32_t* pos_dev, int start_pos, int B, int S, int Hq, int Hkv, int D,
                              int rot, int Tmax, hipStream_t st) {
    if (B < 0 || S < 0 || Hq < 1 || Hkv < 1 || D < 2 || D % 2 || D > 256 || rot < 0 || rot > D || rot % 2)
        return AWQ_ERR_BAD_SHAPE;
    if (!pos_dev && (start_pos < 0 || start_pos + S > Tmax)) return AWQ_ERR_BAD_SHAPE;
    if (B * S == 0) return AWQ_OK;
    const unsigned blocks = (unsigned)((int64_t)B * S * (Hq + 2 * Hkv));
    hipLaunchKernelGGL(awq_rope_kv_append_kernel, dim3(blocks), dim3(D / 2), 0, st, reinterpret_cast<const half_t*>(qkv),
                       reinterpret_cast<half_t*>(q_out), reinterpret_cast<half_t*>(k_cache),
                       reinterpret_cast<half_t*>(v_cache), cos_t, sin_t, pos_dev, start_pos, S, Hq, Hkv, D, rot, Tmax);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

size_t awq_decode_attention_workspace_bytes_impl(int B, int Hq, int max_splits) {
    return (size_t)B * Hq * max_splits * (128 + 2) * sizeof(float);
}

template <int G, bool FUSED, bool EXTRA = false>
static void launch_attn(dim3 grid, hipStream_t st, const uint16_t* q, uint16_t* kc, uint16_t* vc, uint16_t* out, float* part,
                        const int32_t* len_dev, int seq_len, int Hq, int Hkv, int Tmax, float scale, int chunk,
                        const float* cos_t, const float* sin_t, float softcap = 0.f, const float* alibi = nullptr) {
    hipLaunchKernelGGL((awq_decode_attn_kernel<G, FUSED, EXTRA>), grid, dim3(256), 0, st, reinterpret_cast<const half_t*>(q),
                       reinterpret_cast<half_t*>(kc), reinterpret_cast<half_t*>(vc), reinterpret_cast<half_t*>(out), part,
                       len_dev, seq_len, Hq, Hkv, Tmax, scale, chunk, cos_t, sin_t, softcap, alibi);
}

// cos_t == nullptr: plain form (q [B, Hq, D], rows [0, len) of the caches).  Otherwise the fused form:
// q = raw qkv rows, `len_dev` / `seq_len` carry the POSITION of the new token.
int awq_launch_decode_attention(const uint16_t* q, uint16_t* k_cache, uint16_t* v_cache, uint16_t* out,
                                const int32_t* len_dev, int seq_len, int max_len, int B, int Hq, int Hkv, int D, int Tmax,
                                float scale, void* workspace, size_t workspace_bytes, const float* cos_t,
                                const float* sin_t, hipStream_t st, float softcap, const float* alibi) {
    if (D != 128) return AWQ_ERR_UNSUPPORTED;
    const bool extra = softcap > 0.f || alibi != nullptr;
    if (extra && cos_t) return AWQ_ERR_UNSUPPORTED;  // ALiBi models do not rotate; soft-capping takes the two-launch form
    if (softcap < 0.f) return AWQ_ERR_BAD_SHAPE;
    if (B < 0 || Hq < 1 || Hkv < 1 || Hq % Hkv || Tmax < 1) return AWQ_ERR_BAD_SHAPE;
    const int G = Hq / Hkv;
    if (!(G == 1 || G == 2 || G == 4 || G == 8)) return AWQ_ERR_UNSUPPORTED;
    if (B == 0) return AWQ_OK;
    const bool fused = cos_t != nullptr;
    // the split must not depend on a length that only the device knows: size it for max_len
    const int len_for_split = len_dev ? max_len : seq_len + (fused ? 1 : 0);
    if (len_for_split < 1 || len_for_split > Tmax) return AWQ_ERR_BAD_SHAPE;
    // Splits per KV head: about 128 rows each (AWQ_ATTN_ROWS; A/B on the whole-model decode, profiles/r04_attention_split_ab.txt:
    // 128 rows 769 / 712 / 663 tok/s at 64 / 512 / 2048 tokens of context, 256 rows 772 / 657 / 564, 64 rows 768 / 694 / 589),
    // at most ~1024 blocks (AWQ_ATTN_BLOCKS) and 64 splits (the combine kernel's one-lane-per-split step).  ONE split up to
    // AWQ_ATTN_SINGLE = 256 rows: no partials, no combine launch at all -- what a
    // short context pays for a 32-way split sized for the whole cache was 9.5 + 5 us per layer (profiles/r03_whole_model_gemv_
    // kernel_stats.txt); callers that know the context on the host (modules/fused/decode.py: one hipGraph per length bucket)
    // pass the bucket's bound as max_len.  (Both constants overridable in experiment builds: tools/attn_split_ab.py.)
#ifndef AWQ_ATTN_BLOCKS
#define AWQ_ATTN_BLOCKS 1024
#endif
#ifndef AWQ_ATTN_ROWS
#define AWQ_ATTN_ROWS 128
#endif
#ifndef AWQ_ATTN_SINGLE
#define AWQ_ATTN_SINGLE 256
#endif
    int splits = len_for_split <= AWQ_ATTN_SINGLE ? 1 : (len_for_split + AWQ_ATTN_ROWS - 1) / AWQ_ATTN_ROWS;
    const int max_by_blocks = AWQ_ATTN_BLOCKS / (B * Hkv) > 1 ? AWQ_ATTN_BLOCKS / (B * Hkv) : 1;
    if (splits > max_by_blocks) splits = max_by_blocks;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    if (splits > 1 && (!workspace || workspace_bytes < awq_decode_attention_workspace_bytes_impl(B, Hq, splits))) {
        const size_t per = awq_decode_attention_workspace_bytes_impl(B, Hq, 1);
        splits = workspace ? (int)(workspace_bytes / per) : 1;
        if (splits < 1) splits = 1;
    }
    const int chunk = 0;  // dealt in the kernel from the length it reads
    const dim3 grid((unsigned)splits, (unsigned)Hkv, (unsigned)B);
    float* part = static_cast<float*>(workspace);
#define AWQ_ATTN_CASE(GG)                                                                                                  \
    case GG:                                                                                                               \
        if (extra) launch_attn<GG, false, true>(grid, st, q, k_cache, v_cache, out, part, len_dev, seq_len, Hq, Hkv, Tmax, \
                                                scale, chunk, nullptr, nullptr, softcap, alibi);                            \
        else if (fused) launch_attn<GG, true>(grid, st, q, k_cache, v_cache, out, part, len_dev, seq_len, Hq, Hkv, Tmax, scale, \
                                         chunk, cos_t, sin_t);                                                             \
        else launch_attn<GG, false>(grid, st, q, k_cache, v_cache, out, part, len_dev, seq_len, Hq, Hkv, Tmax, scale,      \
                                    chunk, nullptr, nullptr);                                                              \
        break;
    switch (G) {
        AWQ_ATTN_CASE(1)
        AWQ_ATTN_CASE(2)
        AWQ_ATTN_CASE(4)
        default:
        AWQ_ATTN_CASE(8)
    }
#undef AWQ_ATTN_CASE
    if (hipGetLastError() != hipSuccess) return AWQ_ERR_LAUNCH;
    if (splits > 1) {
        hipLaunchKernelGGL(awq_decode_attn_combine_kernel, dim3((unsigned)(B * Hq)), dim3(128), 0, st, part,
                           reinterpret_cast<half_t*>(out), splits);
        if (hipGetLastError() != hipSuccess) return AWQ_ERR_LAUNCH;
    }
    return AWQ_OK;
}
