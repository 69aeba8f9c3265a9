// gemv_chain.hip -- a CHAIN of dependent decode-sized int4 projections in ONE persistent launch, gfx950.
//
// Decode is a chain of GEMVs in which link l+1 consumes the output of link l
// (awq/modules/fused/block.py:108-119: o_proj -> gate|up -> down -> next block's qkv), but whose
// WEIGHTS depend on nothing.  One launch per Linear (gemv_mfma.hip) pays, per 8-47 MB matrix, a kernel
// boundary, a dispatch ramp, the first-data latency and the split-K exchange tail: 9 us per launch for
// 3-6 us of HBM time (profiles/r01_bench_kernel_trace_stats.txt).  Letting launches overlap recovers most
// of it even with no kernel change (profiles/r02_overlap_probe.txt: 0.36 -> 0.58 of 8 TB/s on three
// streams), but gfx9 ignores hipExtAnyOrderLaunch (profiles/r02_anyorder_probe.txt) and overlapping
// dependent launches on several queues has no dispatch-order guarantee.  So the chain is ONE kernel:
//
//   * grid = 3 blocks of 4 waves per CU, all resident, NO s_barrier after start-up: every wave is an
//     independent worker.  11 of 12 blocks compute, 1 of 12 is a service block (below);
//   * a compute wave owns (column tile, <= 128 rows = one quantisation group at most) of each link: it
//     REQUESTS those packed weights (and the group's zeros / scales) into registers, then waits for its
//     slice of the activations, feeds the MFMAs (the selector-row scheme of gemv_mfma.hip) and at once
//     requests its unit of the NEXT link, before it folds -- the weight stream of link l+1 runs under the
//     exchange latency of link l;
//   * split-K partials are the ONLY thing that travels between links: the waves of a block fold through
//     LDS (arrival counter, last arriver sums) and store one slab of 8-byte {fp32 value, tag} granules
//     with write-through (sc1) 16-byte stores.  A granule validates itself (tag = epoch << 10 | Linear
//     id), so nothing is ever reset, re-armed, fenced or drained (MI355X_MICROARCH.md price list
//     "handoff-1to1", Guideline 16 form R2).  The CONSUMER reduces: while it stages its activations a
//     wave of link l+1 sums the K slices of exactly the 128 columns of link l it needs (fixed order:
//     bitwise reproducible, identical in every consumer), applies link l's bias / residual and rounds to
//     fp16 -- one fabric hop per link instead of reduce -> publish -> poll (the first build of this file
//     did that: 6 us from the last slab to the published vector, profiles/r02_chain_trace_service_hop.txt);
//   * service waves do the same reduction for the links whose fp16 result the caller wants in memory
//     (the end of the chain; any link in tests) -- off the critical path;
//   * the epoch comes from per-XCD arrival counters (a / (G/8) + 1): no host state, no reset kernel,
//     identical under hipGraph replay; every spin is bounded and raises ctrl->err / ctrl->abort.
//
// Roofline: HBM.  Algorithmic bytes per link as for one awq_gemm_forward call.
// Replaces a run of awq_ext.gemv_forward_cuda / gemm_forward_cuda calls (awq/modules/linear/gemm.py:56-58,
// awq/modules/fused/mlp.py:37-62) for M <= 8, g % 128 == 0, N % 32 == 0.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "awq_device.h"
#include "awq_internal.h"
#include "awq_mfma_decode.h"

namespace {

constexpr int NCW = 4;                  // waves per block: four compute waves, or four service waves
constexpr int NTHR = NCW * 64;
constexpr int SVC_EVERY = 12;           // block b is a SERVICE block when b % 12 == 11: 64 of 768 blocks = 256 service waves
constexpr int CW = 256;                 // columns per tile (2 packed words per lane)
constexpr int MAXSETS = 8;              // 16-row sets a wave holds in registers (128 rows = one group at most)
constexpr unsigned SPIN_LIMIT = 1u << 17;
constexpr uint32_t CHAIN_MAGIC = 0x41575143u;  // "AWQC"

enum { XF_SLABS = 1, XF_GATED = 2 };

struct ChainLinkDev {  // one (sub-)link; 128 bytes
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* bias;       // this Linear's epilogue: applied by whoever reduces its slabs
    const half_t* add_res;    // [M, N] or null
    const half_t* x;          // external fp16 rows (link 0), else null
    half_t* y;                // plain fp16 [M, N] (written by service waves) or null
    int K, N;                 // the Linear's full shape
    int tile0, tiles;         // column tiles of this sub-link: [tile0, tile0 + tiles)
    int tiles_full;           // column tiles of the whole Linear (slab layout)
    int S, nsets;             // K slices (blocks per tile), 16-row sets per wave
    int xflags;               // XF_*
    int x_stride;             // external x: halves per row
    int x_col0;               // in-chain x: first column taken of the producer's output
    int prod;                 // in-chain x: index (in this array) of the producer Linear's FIRST sub-link
    int out_id;               // id of this Linear (tag of its slabs)
    int g;                    // group size (multiple of 128)
    uint32_t slab_off;        // byte offset of this LINEAR's slabs in the exchange area: [S][tiles_full][M][64 quads][32 B]
    int first_sub;            // 1 on the first sub-link of a Linear (service jobs are issued there, for the whole Linear)
    int pad_[3];
};
static_assert(sizeof(ChainLinkDev) == 128, "ChainLinkDev layout");

struct ChainHeader {  // 128 bytes, followed by the links
    uint32_t magic, n_links, G, M;
    uint64_t slab_bytes, unused_;
    uint32_t n_linears;
    uint32_t pad0_;
    unsigned long long* trace;  // debug: [n_links][G][NCW][4] wall_clock64 stamps, or null
    uint32_t pad_[20];
};
static_assert(sizeof(ChainHeader) == 128, "ChainHeader layout");

struct ChainCtrl {  // head of the workspace, zeroed once by awq_chain_workspace_init
    unsigned long long arrive[8];  // blocks arrived, per blockIdx % 8: epoch = arrive / (G / 8) + 1
    uint32_t err;                  // OR of give-up codes (sticky)
    uint32_t abort;                // set with err: every spin gives up at once
};
constexpr size_t CTRL_BYTES = 4096;

typedef unsigned long long u64;

// by-value helpers: __builtin_bit_cast applied DIRECTLY to an ext-vector element (`v[2]`) reads element 0
// (the element lvalue is not addressable; hipcc 7.2 takes the vector's address) -- always go through these
AWQ_DEV uint32_t f2u(float v) { return __builtin_bit_cast(uint32_t, v); }
AWQ_DEV float u2f(uint32_t v) { return __builtin_bit_cast(float, v); }

AWQ_DEV uint32_t ld_abort(ChainCtrl* c) { return __hip_atomic_load(&c->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave-uniform: one more turn of a poll loop; true = stop waiting (abort raised by someone, or by us)
AWQ_DEV bool give_up(unsigned& spins, ChainCtrl* c, uint32_t code, int lane) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) != 0) return false;
    if (ld_abort(c)) return true;
    if (spins >= SPIN_LIMIT) {
        if (lane == 0) {
            __hip_atomic_fetch_or(&c->err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return true;
    }
    return false;
}

// The output of Linear P (first sub-link descriptor `P`), columns col .. col + 3 of batch row m, as the four
// fp16 values every reader agrees on: sum of the S slabs in the canonical order (even slices ascending, odd
// slices ascending, even + odd), + bias, rounded to fp16, + residual rounded again.  A wave calls this with
// lanes (qd = lane & 31, sh = lane >> 5): `col` is the lane's quad, `sh` the slice parity it sums; lanes with
// !active request nothing.  The result is valid on every active lane (both parities).
// wave-uniform control flow; `spins` / give-up shared with the caller.
AWQ_DEV half4_t reduce_quad(const ChainLinkDev& P, const unsigned char* slab_base, int M, int m, int col, int sh, bool active,
                            uint32_t stag, unsigned& spins, ChainCtrl* ctrl, uint32_t code, int lane) {
    const int S = P.S;
    const rsrc_t slres = mk_rsrc(slab_base + P.slab_off, (uint32_t)S * (uint32_t)P.tiles_full * (uint32_t)M * 2048u);
    const uint32_t qoff = active ? (uint32_t)(((col >> 8) * M + m) * 2048 + ((col & 255) >> 2) * 32) : OOB;
    const uint32_t sstride = (uint32_t)P.tiles_full * (uint32_t)M * 2048u;
    float4_t part = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 8) {  // this lane: slices s0 + sh, +2, +4, +6
        u32x4 v[4][2];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + sh + 2 * u;
                const uint32_t off = s < S ? qoff : OOB;
                v[u][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(slres, off, (uint32_t)s * sstride, 16 /* sc1 */));
                v[u][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(slres, off == OOB ? OOB : off + 16u, (uint32_t)s * sstride, 16));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (active && s0 + sh + 2 * u < S)
                    ok &= v[u][0][1] == stag && v[u][0][3] == stag && v[u][1][1] == stag && v[u][1][3] == stag;
            if (__all(ok)) break;
            if (give_up(spins, ctrl, code, lane)) break;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)  // slices past S were requested out of range: zeros
            part += float4_t{u2f(v[u][0][0]), u2f(v[u][0][2]), u2f(v[u][1][0]), u2f(v[u][1][2])};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[e] = part[e] + __shfl_xor(part[e], 32, 64);  // even + odd slices: the same on both lanes
    if (P.bias && active) {
        const half4_t b4 = *reinterpret_cast<const half4_t*>(P.bias + col);
        part += float4_t{(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
    }
    half4_t o = {(half_t)part[0], (half_t)part[1], (half_t)part[2], (half_t)part[3]};
    if (P.add_res && active) {  // fp16(fp16(projection) + residual): the two roundings torch makes
        const half4_t r4 = *reinterpret_cast<const half4_t*>(P.add_res + (size_t)m * P.N + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)o[e] + (float)r4[e]);
    }
    return o;
}

// Light wait: until one granule of each of the S slabs holding columns [col, col + ncols) of batch rows < M has
// landed.  Lane i < S probes slice i (its first granule of the range), so a waiting wave costs S x 8 bytes per
// turn instead of re-reading its whole slice of every slab.
AWQ_DEV void probe_slabs(const ChainLinkDev& P, const unsigned char* slab_base, int M, int col, uint32_t stag, unsigned& spins,
                         ChainCtrl* ctrl, uint32_t code, int lane) {
    const int S = P.S;
    const rsrc_t slres = mk_rsrc(slab_base + P.slab_off, (uint32_t)S * (uint32_t)P.tiles_full * (uint32_t)M * 2048u);
    const uint32_t sstride = (uint32_t)P.tiles_full * (uint32_t)M * 2048u;
    const bool probing = lane < S;
    const uint32_t off = probing ? (uint32_t)lane * sstride + (uint32_t)(((col >> 8) * M + (M - 1)) * 2048 + ((col & 255) >> 2) * 32) : OOB;
    for (;;) {
        const u32x2 pv = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(slres, off, 0, 16 /* sc1 */));
        if (__all(!probing || pv[1] == stag)) break;
        __builtin_amdgcn_s_sleep(4);
        if (give_up(spins, ctrl, code, lane)) break;
    }
}

// NREG = live D registers per lane: 2 when M == 1 (rows 0, 1 of the selector MFMA), 4 for M <= 8
template <int NREG>
__global__ __launch_bounds__(NTHR, 3) void awq_chain_kernel(const ChainHeader* __restrict__ plan, unsigned char* __restrict__ ws) {
    constexpr int CWP = CW + 8;  // LDS row pitch of the fold area (floats)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // dynamic LDS: red[NCW][M][CWP] fp32 | xs[NCW][M + 1][128] fp16 | control words
    const int M = (int)plan->M;
    const int G = (int)plan->G;
    const int n_links = (int)plan->n_links;
    const ChainLinkDev* __restrict__ links = reinterpret_cast<const ChainLinkDev*>(plan + 1);
    ChainCtrl* ctrl = reinterpret_cast<ChainCtrl*>(ws);
    const unsigned char* slab_base = ws + CTRL_BYTES;

    float* red = reinterpret_cast<float*>(smem);
    half_t* xs_all = reinterpret_cast<half_t*>(smem + (size_t)NCW * M * CWP * 4);
    uint32_t* lds_ctl = reinterpret_cast<uint32_t*>(smem + (size_t)NCW * M * CWP * 4 + (size_t)NCW * (M + 1) * 128 * 2);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if (tid == 0) {
        const u64 a = __hip_atomic_fetch_add(&ctrl->arrive[b & 7], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_ctl[0] = (uint32_t)(a / (u64)(G >> 3)) + 1u;  // epoch of this launch
        lds_ctl[1] = 0u;                                  // fold arrivals (monotonic)
        lds_ctl[2] = 0u;                                  // folds completed
    }
    for (int i = tid; i < NCW * 128; i += NTHR) xs_all[((i >> 7) * (M + 1) + M) * 128 + (i & 127)] = (half_t)0.f;  // the all-zero row M
    __syncthreads();
    const uint32_t epoch = lds_ctl[0];
    const uint32_t tag_hi = epoch << 10;
    unsigned long long* const trace = plan->trace;
    auto stamp = [&](int l, int slot) {  // phase timeline for tools/chain_probe.py --trace (off unless the plan carries a buffer)
        if (trace && lane == 0) trace[(((size_t)l * G + b) * NCW + wave) * 4 + slot] = wall_clock64();
    };

    // Roles by block: 256-thread blocks spread evenly over the four SIMDs whatever the dispatcher's starting
    // SIMD is (a 5-wave block does not: two such blocks may need 4 waves on one SIMD and then only one fits)
    const bool service = b % SVC_EVERY == SVC_EVERY - 1;
    const int NSW = (G / SVC_EVERY) * NCW;  // service waves
    if (service) {
        // ================================================================ service waves: materialise y where asked
        const int sw = (b / SVC_EVERY) * NCW + wave;  // this wave's index among the service waves
        const int qd = lane & 31, sh = lane >> 5;
        for (int l = 0; l < n_links; ++l) {
            const ChainLinkDev& L = links[l];
            if (!L.y || !L.first_sub) continue;
            const int njobs = L.tiles_full * M * 2;  // (tile, row, half tile): 32 quads x 2 slice parities per wave
            const uint32_t stag = tag_hi | (uint32_t)(L.out_id + 1);
            if (sw < njobs) stamp(l, 0);
            for (int jb = sw; jb < njobs; jb += NSW) {
                const int tl = jb / (2 * M), m = (jb >> 1) % M, hf = jb & 1;
                const int col = tl * CW + hf * 128 + qd * 4;
                const bool active = col < L.N;  // N % 8 == 0: a quad is all in or all out
                unsigned spins = 0;
                probe_slabs(L, slab_base, M, tl * CW + hf * 128, stag, spins, ctrl, 2u, lane);
                stamp(l, 1);
                const half4_t o = reduce_quad(L, slab_base, M, m, col, sh, active, stag, spins, ctrl, 2u, lane);
                if (sh == 0 && active) *reinterpret_cast<u32x2*>(L.y + (size_t)m * L.N + col) = __builtin_bit_cast(u32x2, o);
                stamp(l, 2);
            }
        }
        return;
    }

    // ==================================================================== compute waves
    const int cw = wave;
    const int cb = b - b / SVC_EVERY;  // this block's index among the compute blocks
    const int j = lane & 15, kb = lane >> 4;
    half_t* xs = xs_all + (size_t)cw * (M + 1) * 128;  // this wave's activation rows [M + 1][128]
    float* myred = red + (size_t)cw * M * CWP;

    struct Unit {  // what a wave holds in registers for one link
        u32x2 q[MAXSETS][4];
        u32x2 z;
        u32x4 s[2];
    };
    auto request = [&](int l, Unit& U) {  // P0: issue every load of link l's unit (nothing is waited for)
        const ChainLinkDev& L = links[l];
        const int tiles = L.tiles, nsets = L.nsets;
        const bool has = cb < tiles * L.S;
        const int tile = L.tile0 + cb % tiles, slice = cb / tiles;
        const int row0 = (slice * NCW + cw) * 16 * nsets;
        const int NW = L.N >> 3;
        const uint32_t row_bytes = (uint32_t)NW * 4u;
        const int colw = (tile * 16 + j) * 2;
        const bool act = has && row0 < L.K && colw < NW;
        const rsrc_t wres = mk_rsrc(L.qweight, (uint32_t)L.K * row_bytes);
        const uint32_t voff = act ? (uint32_t)colw * 4u + (uint32_t)(4 * kb) * row_bytes : OOB;
        uint32_t soff = (uint32_t)row0 * row_bytes;
#pragma unroll
        for (int t = 0; t < MAXSETS; ++t) {
            const uint32_t vo = t < nsets ? voff : OOB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                U.q[t][r] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(wres, vo, soff, 2 /* nt */));
                soff += row_bytes;
            }
            soff += 12u * row_bytes;
        }
        const int grp = row0 / L.g;
        const rsrc_t zres = mk_rsrc(L.qzeros, (uint32_t)(L.K / L.g) * row_bytes);
        const rsrc_t sres = mk_rsrc(L.scales, (uint32_t)(L.K / L.g) * (uint32_t)L.N * 2u);
        const uint32_t zo = act ? (uint32_t)grp * row_bytes + (uint32_t)colw * 4u : OOB;
        const uint32_t so = act ? ((uint32_t)grp * (uint32_t)L.N + (uint32_t)colw * 8u) * 2u : OOB;
        U.z = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(zres, zo, 0, 0));
        U.s[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sres, so, 0, 0));
        U.s[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sres, so == OOB ? OOB : so + 16u, 0, 0));
    };

    const uint32_t sel_lo = (j & 1) ? 0x01000C0Cu : 0x0C0C0100u;  // low half of a dword -> slot (j & 1)
    const uint32_t sel_hi = (j & 1) ? 0x03020C0Cu : 0x0C0C0302u;  // high half
    const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    const int arow = min(j >> 1, M);  // A row of this lane = batch row j >> 1 (parity j & 1); rows >= M read zeros

    uint32_t nfold = 0;  // folds this block has taken part in
    Unit U;
    request(0, U);
    for (int l = 0; l < n_links; ++l) {
        const ChainLinkDev& L = links[l];
        const int tiles = L.tiles, nsets = L.nsets, S = L.S;
        const bool has = cb < tiles * S;
        const int tl = L.tile0 + cb % tiles, slice = cb / tiles;  // tile of the whole Linear
        const int row0 = (slice * NCW + cw) * 16 * nsets;
        const int nrows = 16 * nsets;
        float yv[8][NREG];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < NREG; ++r) yv[c][r] = 0.f;

        if (has) stamp(l, 0);
        if (has && row0 < L.K) {
            // ---- P1: this wave's slice of the activations -> LDS rows xs[m][0 .. nrows)
            if (L.xflags & XF_SLABS) {
                // reduce the producer's K slices for exactly the columns this wave needs: quad qd = lane & 31 of the
                // slice, slice parity sh = lane >> 5
                const ChainLinkDev& P = links[L.prod];
                const uint32_t stag = tag_hi | (uint32_t)(P.out_id + 1);
                const bool gated = (L.xflags & XF_GATED) != 0;
                const int qd = lane & 31, sh = lane >> 5;
                const bool active = 4 * qd < nrows;
                const int col = L.x_col0 + row0 + 4 * qd;
                unsigned spins = 0;
                probe_slabs(P, slab_base, M, L.x_col0 + row0, stag, spins, ctrl, 1u, lane);
                if (gated) probe_slabs(P, slab_base, M, L.x_col0 + L.K + row0, stag, spins, ctrl, 1u, lane);
                for (int m = 0; m < M; ++m) {
                    half4_t xv = reduce_quad(P, slab_base, M, m, col, sh, active, stag, spins, ctrl, 1u, lane);
                    if (gated) {  // silu(gate) * up in fp32, one rounding: == awq_silu_and_mul_kernel
                        const half4_t up = reduce_quad(P, slab_base, M, m, col + L.K, sh, active, stag, spins, ctrl, 1u, lane);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gx = (float)xv[e];
                            xv[e] = (half_t)((gx / (1.0f + expf(-gx))) * (float)up[e]);
                        }
                    }
                    if (sh == 0 && active) *reinterpret_cast<u32x2*>(xs + m * 128 + 4 * qd) = __builtin_bit_cast(u32x2, xv);
                }
            } else {
                const bool mine = 2 * lane < nrows;  // lane owns elements (2 lane, 2 lane + 1) of every row
                for (int m = 0; m < M; ++m)
                    if (mine) *reinterpret_cast<uint32_t*>(xs + m * 128 + 2 * lane) =
                                  *reinterpret_cast<const uint32_t*>(L.x + (size_t)m * L.x_stride + row0 + 2 * lane);
            }
            stamp(l, 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // LDS writes of this wave before its own reads
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);

            // ---- P2: MFMAs over the sets of the unit (all inside one group)
            float4_t acc[8];
            float4_t accsx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = float4_t{0.f, 0.f, 0.f, 0.f};
            const half_t* xlane = xs + arow * 128 + 4 * kb;
#pragma unroll
            for (int t = 0; t < MAXSETS; ++t) {
                if (t < nsets) {
                    const u32x2 xq = *reinterpret_cast<const u32x2*>(xlane + 16 * t);
                    const uint32_t x01 = xq[0], x23 = xq[1];
                    const u32x4v a0 = {__builtin_amdgcn_perm(0u, x01, sel_lo), __builtin_amdgcn_perm(0u, x01, sel_hi),
                                       __builtin_amdgcn_perm(0u, x23, sel_lo), __builtin_amdgcn_perm(0u, x23, sel_hi)};
                    accsx = mfma16(a0, ones, accsx);
#pragma unroll
                    for (int wd = 0; wd < 2; ++wd) {
                        const uint32_t q0 = U.q[t][0][wd], q1 = U.q[t][1][wd], q2 = U.q[t][2][wd], q3 = U.q[t][3][wd];
                        const uint32_t h0 = q0 >> 8, h1 = q1 >> 8, h2 = q2 >> 8, h3 = q3 >> 8;
#define AWQ_MMA_J(J)                                                                                     \
    {                                                                                                    \
        const u32x4v bf = {pairb<J>(q0, h0), pairb<J>(q1, h1), pairb<J>(q2, h2), pairb<J>(q3, h3)};      \
        acc[wd * 4 + J] = mfma16(a0, bf, acc[wd * 4 + J]);                                               \
    }
                        AWQ_MMA_J(0)
                        AWQ_MMA_J(1)
                        AWQ_MMA_J(2)
                        AWQ_MMA_J(3)
#undef AWQ_MMA_J
                    }
                }
            }
            // ---- P3: y = s * (acc - (bias_J + z) * sum_x), the unit lies inside one group
#pragma unroll
            for (int wd = 0; wd < 2; ++wd) {
                const uint32_t zw = U.z[wd], zw8 = zw >> 8;
                const uint32_t zp[4] = {pairb<0>(zw, zw8), pairb<1>(zw, zw8), pairb<2>(zw, zw8), pairb<3>(zw, zw8)};
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    const half2_t z2 = u2h2(zp[J]), s2 = u2h2(U.s[wd][J]);
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int e = r & 1;
                        const float raw = __builtin_fmaf(-(float)z2[e], accsx[r], acc[wd * 4 + J][r]);
                        yv[wd * 4 + J][r] = (float)s2[e] * raw;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has) stamp(l, 2);
        // ---- request the NEXT link's unit now: its latency runs under this link's fold and exchange
        if (l + 1 < n_links) request(l + 1, U);
        __builtin_amdgcn_sched_barrier(0);

        if (has) {
            // ---- fold the block's waves through LDS; the last wave to arrive sums and stores the slab.
            // Arrivals and completed folds are counted monotonically.  A wave may be one link ahead of the
            // others (sub-links of one Linear share their input, so nothing else holds it back): it must not
            // overwrite its fold rows before the previous fold has been read.
            {
                unsigned spins = 0;
                while (__hip_atomic_load(&lds_ctl[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != nfold)
                    if (give_up(spins, ctrl, 4u, lane)) break;
            }
            // D row 4 kb + r = (batch row 2 kb + (r >> 1), column parity r & 1); column j*16 + 2c + e
#pragma unroll
            for (int rp = 0; rp < NREG / 2; ++rp) {
                const int m = 2 * kb + rp;
                if (m < M) {
#pragma unroll
                    for (int c = 0; c < 8; c += 2)
                        *reinterpret_cast<float4_t*>(myred + m * CWP + j * 16 + 2 * c) =
                            float4_t{yv[c][2 * rp], yv[c][2 * rp + 1], yv[c + 1][2 * rp], yv[c + 1][2 * rp + 1]};
                }
            }
            uint32_t arrived = 0;
            if (lane == 0)
                arrived = __hip_atomic_fetch_add(&lds_ctl[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            arrived = __builtin_amdgcn_readfirstlane(arrived);
            if (arrived == nfold * NCW + NCW - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t stag = tag_hi | (uint32_t)(L.out_id + 1);
                const rsrc_t slres = mk_rsrc(ws + CTRL_BYTES + L.slab_off, (uint32_t)S * (uint32_t)L.tiles_full * (uint32_t)M * 2048u);
                const uint32_t so = (uint32_t)((slice * L.tiles_full + tl) * M) * 2048u;
                for (int qd = lane; qd < M * 64; qd += 64) {
                    const int m = qd >> 6, c4 = (qd & 63) * 4;
                    float4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < NCW; ++w) {
                        const float* src = red + (size_t)(w * M + m) * CWP + c4;
                        s += float4_t{src[0], src[1], src[2], src[3]};
                    }
                    const u32x4 g0 = {f2u(s[0]), stag, f2u(s[1]), stag};
                    const u32x4 g1 = {f2u(s[2]), stag, f2u(s[3]), stag};
                    __builtin_amdgcn_raw_buffer_store_b128(g0, slres, (uint32_t)qd * 32u + so, 0, 16 /* sc1 */);
                    __builtin_amdgcn_raw_buffer_store_b128(g1, slres, (uint32_t)qd * 32u + 16u + so, 0, 16);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the fold rows have been read
                if (lane == 0) __hip_atomic_store(&lds_ctl[2], nfold + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            ++nfold;
            stamp(l, 3);
        }
    }
}

// ------------------------------------------------------------------------------------------ host side

struct DevInfo {
    int cus = 0;
};
DevInfo dev_info() {
    DevInfo d;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
    hipDeviceProp_t p;
    if (dev >= 0 && hipGetDeviceProperties(&p, dev) == hipSuccess) d.cus = p.multiProcessorCount;
    else (void)hipGetLastError();
    return d;
}

size_t chain_lds_bytes(int M) { return (size_t)NCW * M * (CW + 8) * 4 + (size_t)NCW * (M + 1) * 128 * 2 + 16; }

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

size_t awq_chain_plan_bytes(int64_t n_links) {
    if (n_links <= 0) return 0;
    return sizeof(ChainHeader) + (size_t)n_links * 8 * sizeof(ChainLinkDev);  // a Linear splits into <= 8 sub-links
}

int awq_chain_grid_blocks(void) {
    static const int cached = [] {
        const char* e = getenv("AWQ_CHAIN_GRID");
        if (e && atoi(e) >= 24) return atoi(e) / 24 * 24;
        const DevInfo d = dev_info();
        const int cus = d.cus > 0 ? d.cus : 256;  // no device visible (build box): plan for an MI355X
        return cus * 3 / 24 * 24;                 // 3 blocks per CU; a multiple of 8 (epoch counters) and of 12 (roles)
    }();
    return cached;
}

int awq_chain_build(const AwqChainLink* links, int64_t n, int64_t M, void* workspace, size_t workspace_bytes,
                    void* plan_host, size_t plan_bytes, size_t* workspace_needed) {
    if (!links || !plan_host) return AWQ_ERR_NULL;
    if (n < 1 || n > 1000 || M < 1 || M > 8) return AWQ_ERR_UNSUPPORTED;
    if (plan_bytes < awq_chain_plan_bytes(n)) return AWQ_ERR_WORKSPACE;
    const int Gall = awq_chain_grid_blocks();
    const int G = Gall - Gall / SVC_EVERY;  // compute blocks: what a (sub-)link's tiles x slices must fit
    ChainHeader* H = static_cast<ChainHeader*>(plan_host);
    ChainLinkDev* out = reinterpret_cast<ChainLinkDev*>(H + 1);
    memset(H, 0, sizeof(*H));
    struct Lin { int nsets, S, tiles, nsub, first; size_t slab, off; };
    std::vector<Lin> lin((size_t)n);
    // Slab regions.  A Linear's slabs are read by the next Linear's waves while they stage their input, so
    // Linear i and i + 2 never overlap in time (i + 2 stores only after every slab of i + 1 is complete, i.e.
    // after every wave of i + 1 has finished reading i): two alternating regions serve the whole chain.  A
    // Linear whose result is also materialised (y != NULL) is read by service waves at their own pace: it
    // gets a private region.
    size_t half[2] = {0, 0}, priv = 0;
    int total = 0;
    for (int64_t i = 0; i < n; ++i) {
        const AwqChainLink& a = links[i];
        if (a.K <= 0 || a.N <= 0 || a.group_size <= 0) return AWQ_ERR_BAD_SHAPE;
        if (a.N % 32 || a.K % a.group_size || a.group_size % 128) return AWQ_ERR_UNSUPPORTED;
        if ((int64_t)a.K * a.N / 2 >= ((int64_t)1 << 31)) return AWQ_ERR_UNSUPPORTED;
        if (!a.qweight || !a.scales || !a.qzeros) return AWQ_ERR_NULL;
        const bool gated = (a.flags & AWQ_CHAIN_X_GATED_SILU) != 0;
        if (a.x_from >= 0) {
            if (a.x_from != i - 1) return AWQ_ERR_UNSUPPORTED;  // a pure chain bounds how far blocks run ahead
            const AwqChainLink& p = links[a.x_from];
            if (a.x_col0 < 0 || a.x_col0 % 128 || a.x_col0 + (gated ? 2 : 1) * a.K > p.N) return AWQ_ERR_BAD_SHAPE;
            if (gated && a.K % 128) return AWQ_ERR_BAD_SHAPE;
        } else {
            if (i != 0) return AWQ_ERR_UNSUPPORTED;
            if (!a.x || gated) return a.x ? AWQ_ERR_UNSUPPORTED : AWQ_ERR_NULL;
            if (a.x_stride < a.K || a.x_stride % 2) return AWQ_ERR_BAD_SHAPE;
        }
        if (i == n - 1 && !a.y) return AWQ_ERR_NULL;  // somebody has to want the result
        Lin& L = lin[(size_t)i];
        L.tiles = (int)((a.N + CW - 1) / CW);
        // fat units: a wave holds up to 128 rows, a block 512 -- few K slices, so that a consumer sums few slabs
        L.nsets = a.K >= NCW * 128 ? 8 : (a.K >= NCW * 64 ? 4 : 2);
        const int rpb = NCW * 16 * L.nsets;
        L.S = (int)((a.K + rpb - 1) / rpb);
        if (L.S > 64 || L.S > G) return AWQ_ERR_UNSUPPORTED;
        const int per = G / L.S;  // column ranges of <= G / S tiles, one sub-link after the other
        L.nsub = (L.tiles + per - 1) / per;
        if (L.nsub > 8) return AWQ_ERR_UNSUPPORTED;
        L.slab = align_up((size_t)L.tiles * L.S * (size_t)M * 2048, 4096);
        if (a.y) { L.off = priv; priv += L.slab; }
        else if (L.slab > half[i & 1]) half[i & 1] = L.slab;
        L.first = total;
        total += L.nsub;
    }
    if (total > 1023 || n > 1023) return AWQ_ERR_UNSUPPORTED;  // 10 bits of tag
    const size_t slab_bytes = half[0] + half[1] + priv;
    const size_t need = CTRL_BYTES + slab_bytes;
    if (slab_bytes >= ((size_t)1 << 32)) return AWQ_ERR_UNSUPPORTED;
    if (workspace_needed) *workspace_needed = need;
    if (!workspace) return AWQ_OK;  // size query
    if (workspace_bytes < need) return AWQ_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return AWQ_ERR_BAD_ALIGNMENT;
    int k = 0;
    for (int64_t i = 0; i < n; ++i) {
        const AwqChainLink& a = links[i];
        const Lin& L = lin[(size_t)i];
        const int per = (L.tiles + L.nsub - 1) / L.nsub;
        const size_t off = a.y ? half[0] + half[1] + L.off : ((i & 1) ? half[0] : 0);
        for (int sidx = 0; sidx < L.nsub; ++sidx) {
            ChainLinkDev& d = out[k++];
            memset(&d, 0, sizeof(d));
            d.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
            d.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
            d.scales = reinterpret_cast<const half_t*>(a.scales);
            d.bias = reinterpret_cast<const half_t*>(a.bias);
            d.add_res = reinterpret_cast<const half_t*>(a.add_residual);
            d.y = reinterpret_cast<half_t*>(a.y);
            d.K = (int)a.K; d.N = (int)a.N; d.g = (int)a.group_size;
            d.tile0 = sidx * per;
            d.tiles = (sidx + 1) * per <= L.tiles ? per : L.tiles - sidx * per;
            d.tiles_full = L.tiles;
            d.S = L.S; d.nsets = L.nsets;
            d.out_id = (int)i;
            d.slab_off = (uint32_t)off;
            d.first_sub = sidx == 0;
            if (a.x_from >= 0) {
                d.xflags = XF_SLABS | ((a.flags & AWQ_CHAIN_X_GATED_SILU) ? XF_GATED : 0);
                d.x = nullptr;
                d.prod = lin[(size_t)a.x_from].first;
                d.x_col0 = (int)a.x_col0;
            } else {
                d.xflags = 0;
                d.x = reinterpret_cast<const half_t*>(a.x);
                d.x_stride = (int)a.x_stride;
                d.prod = -1;
            }
        }
    }
    H->magic = CHAIN_MAGIC; H->n_links = (uint32_t)k; H->G = (uint32_t)Gall; H->M = (uint32_t)M;
    H->slab_bytes = slab_bytes; H->n_linears = (uint32_t)n;
    return AWQ_OK;
}

int awq_chain_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) return AWQ_ERR_NULL;
    if (workspace_bytes < CTRL_BYTES) return AWQ_ERR_WORKSPACE;
    // tags of a launch are (epoch << 10 | id) with epoch >= 1, ids >= 1: all-zero memory matches none
    if (hipMemsetAsync(workspace, 0, workspace_bytes, static_cast<hipStream_t>(stream)) != hipSuccess) return AWQ_ERR_LAUNCH;
    return AWQ_OK;
}

int awq_chain_forward(const void* plan_dev, const void* plan_host, void* workspace, size_t workspace_bytes, void* stream) {
    if (!plan_dev || !plan_host || !workspace) return AWQ_ERR_NULL;
    const ChainHeader* H = static_cast<const ChainHeader*>(plan_host);
    if (H->magic != CHAIN_MAGIC || H->n_links < 1) return AWQ_ERR_BAD_SHAPE;
    if (workspace_bytes < CTRL_BYTES + H->slab_bytes) return AWQ_ERR_WORKSPACE;
    const int M = (int)H->M;
    const size_t lds = chain_lds_bytes(M);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const int max_blocks[2] = {
        [] { int nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, awq_chain_kernel<2>, NTHR, chain_lds_bytes(1)) != hipSuccess) { (void)hipGetLastError(); nb = 0; } return nb; }(),
        [] { int nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, awq_chain_kernel<4>, NTHR, chain_lds_bytes(8)) != hipSuccess) { (void)hipGetLastError(); nb = 0; } return nb; }()};
    const DevInfo d = dev_info();
    // every block must be resident at once (they wait for each other): refuse a grid the device cannot hold
    const int nb = max_blocks[M == 1 ? 0 : 1];
    if (d.cus <= 0 || nb <= 0 || (int64_t)nb * d.cus < (int64_t)H->G) return AWQ_ERR_UNSUPPORTED;
    if (M == 1)
        hipLaunchKernelGGL(awq_chain_kernel<2>, dim3(H->G), dim3(NTHR), lds, st, static_cast<const ChainHeader*>(plan_dev),
                           static_cast<unsigned char*>(workspace));
    else
        hipLaunchKernelGGL(awq_chain_kernel<4>, dim3(H->G), dim3(NTHR), lds, st, static_cast<const ChainHeader*>(plan_dev),
                           static_cast<unsigned char*>(workspace));
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_chain_status(const void* workspace, void* stream, uint32_t* err_out) {
    if (!workspace || !err_out) return AWQ_ERR_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t words[2] = {0, 0};
    if (hipMemcpyAsync(words, static_cast<const unsigned char*>(workspace) + offsetof(ChainCtrl, err), sizeof(words),
                       hipMemcpyDeviceToHost, st) != hipSuccess)
        return AWQ_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return AWQ_ERR_LAUNCH;
    *err_out = words[0] | (words[1] ? 0x80000000u : 0u);
    return AWQ_OK;
}

}  // extern "C"
