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
//   * grid = ONE block of 8 waves per CU (its register footprint admits no second one), all resident, NO
//     s_barrier after start-up: six COMPUTE waves and two POLL waves, every wave an independent worker;
//   * unit = (256-column tile, 128 rows = one quantisation group at most) = 16 KB of packed weights.  The
//     units of a link are dealt to the compute waves of the whole grid in order (wave g takes unit g: K
//     groups of one tile are neighbours, so the waves of a block fold together).  A compute wave REQUESTS
//     a unit (and the group's zeros / scales) into registers and holds TWO: while link l waits for its
//     activations the units of links l+1 and l+2 are in flight or landed -- the weight stream runs two
//     links ahead of the dependency chain, HBM never waits for the exchange;
//   * vector-memory loads return in order per wave: a wave that polls behind its own prefetch sees
//     nothing before 16 KB of weights have landed (profiles/r02_chain_v2_reduce_on_read_trace.txt: the
//     stream serialised with the exchange, 10 us per link).  So compute waves never poll memory: the
//     POLL waves, which have nothing else in flight, gather the activations and hand them over in LDS;
//   * split-K partials are the ONLY thing that travels between links: the waves of a block that share a
//     tile fold through LDS (arrival counter, last arriver sums) and store one slab of 8-byte
//     {fp32 value, tag} granules with write-through (sc1) 16-byte stores.  A granule validates itself
//     (tag = epoch << 10 | Linear id): nothing is ever reset, re-armed, fenced or drained
//     (MI355X_MICROARCH.md price list "handoff-1to1", Guideline 16 form R2).  The CONSUMER reduces: a poll
//     wave of link l+1 sums the slabs of exactly the 128 columns of link l each of its compute waves needs
//     (canonical order: bitwise reproducible, identical in every consumer), applies link l's bias /
//     residual and rounds to fp16 -- one fabric hop per link instead of reduce -> publish -> poll;
//   * the same poll waves materialise y for the links whose fp16 result the caller wants in memory (the end
//     of the chain; any link in tests), one link late, when its slabs are known to be complete;
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

constexpr int CW = 256;                 // columns per tile (2 packed words per lane)
constexpr int UPB = 6;                  // units a block holds of one (sub-)link
constexpr int WPU = 1;                  // compute waves per unit (2: 64 rows each -- measured slower: 16 waves, 128 registers, 12-way fold)
constexpr int NCWB = WPU * UPB;         // compute waves per block (waves 1 .. NCWB; wave 0 is the loader)
constexpr int NPW = 3;                  // poll waves per block (waves 13 .. 15)
constexpr int UPP = UPB / NPW;          // units served by one poll wave
constexpr int NWAVES = 1 + NCWB + NPW;
constexpr int NTHR = NWAVES * 64;
constexpr int SLOT_BYTES = 17 * 1024;   // ring slot: 16 KiB of packed weights (128 rows x 128 bytes) + 128 B zeros + 512 B scales
__host__ __device__ constexpr int ring_slots(int M) {  // what the fold / staging areas leave of 160 KiB, at most 8
    const int n = (160 * 1024 - 256 - NCWB * M * (CW + 8) * 4 - 2 * UPB * (M + 1) * 256) / SLOT_BYTES;
    return n > 8 ? 8 : n;
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gl_ptr_t;
constexpr int SETS = 8;                 // 16-row sets per unit: 128 rows
constexpr int MAX_SUB = 16;             // sub-links a Linear may be cut into
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
    half_t* y;                // plain fp16 [M, N] (written by poll waves) or null
    int K, N;                 // the Linear's full shape
    int tile0, tiles;         // column tiles of this sub-link: [tile0, tile0 + tiles)
    int R;                    // units per tile = K / 128
    int units;                // units of this sub-link = tiles * R (<= compute waves of the grid)
    int per;                  // tiles per sub-link of this Linear (the last one may hold fewer)
    int smax;                 // slab slots per tile: blocks that can share a tile
    int xflags;               // XF_*
    int x_stride;             // external x: halves per row
    int x_col0;               // in-chain x: first column taken of the producer's output
    int prod;                 // in-chain x: index (in this array) of the producer Linear's FIRST sub-link
    int out_id;               // id of this Linear (tag of its slabs)
    int g;                    // group size (multiple of 128)
    uint32_t slab_off;        // byte offset of this LINEAR's slabs: [tiles_full][smax][M][64 quads][32 B]
    int last_sub;             // 1 on the last sub-link of a Linear
    int tiles_full;           // column tiles of the whole Linear
    int pad_;
    // (v * magic) >> 32 == v / {R, per, g} for every value the kernel divides (awq_magic_u32); per_magic 0: per == 1
    uint32_t r_magic, per_magic, g_magic;
    uint32_t pad2_[5];
};
static_assert(sizeof(ChainLinkDev) == 160, "ChainLinkDev layout");

struct ChainHeader {  // 128 bytes, followed by the links
    uint32_t magic, n_links, G, M;
    uint64_t slab_bytes, unused_;
    uint32_t n_linears;
    uint32_t pad0_;
    unsigned long long* trace;  // debug: [n_links][G][NWAVES][4] wall_clock64 stamps, or null
    uint32_t inflight;          // fills the loader keeps in flight (1 .. 3)
    uint32_t slack;             // slabs a gather may still miss when it starts (0 | 1)
    uint32_t ns_magic;          // (f * ns_magic) >> 32 == f / ring_slots(M) for every fill number f
    uint32_t pad_[17];
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

// The same for a COMPUTE wave waiting on an LDS word: it must not touch vector memory (a load would return
// behind the wave's own weight prefetch), so it watches the block's LDS abort word, which the poll waves
// set when they give up, and a wall-clock bound (s_memrealtime is a scalar access).
AWQ_DEV bool give_up_lds(unsigned& spins, const uint32_t* lds_abort, unsigned long long& t0, ChainCtrl* c, uint32_t code, int lane) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) != 0) return false;
    if (__hip_atomic_load(lds_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return true;
    const unsigned long long now = wall_clock64();
    if (t0 == 0) t0 = now;
    if (now - t0 > 20000000ull) {  // 0.2 s at 100 MHz
        if (lane == 0) {
            __hip_atomic_fetch_or(&c->err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return true;
    }
    return false;
}

AWQ_DEV uint32_t lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
AWQ_DEV void lds_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Where the slabs of columns [col, col + 128) of Linear P live: which tile, how many blocks contributed
// (slots 0 .. S-1) and the byte offset of slot 0, row 0, quad 0 of the range.
struct SlabRange {
    int S;
    uint32_t base;
};
AWQ_DEV SlabRange slab_range(const ChainLinkDev& P, int M, int col) {
    const int tp = col >> 8;                 // tile of the Linear
    const int tl = P.per_magic ? tp - (int)__umulhi((uint32_t)tp, P.per_magic) * P.per : 0;  // tile within its sub-link: unit numbering restarts there
    const int b_lo = (tl * P.R) / UPB, b_hi = ((tl + 1) * P.R - 1) / UPB;
    SlabRange r;
    r.S = b_hi - b_lo + 1;
    r.base = P.slab_off + (uint32_t)(tp * P.smax * M) * 2048u + (uint32_t)((col & 255) >> 2) * 32u;
    return r;
}

// The output of Linear P, NR ranges of 128 columns (col[r] .. col[r] + 127, each inside one tile) of batch row
// m, as the fp16 values every reader agrees on: sum of the slabs in the canonical order (even slots ascending,
// odd slots ascending, even + odd), + bias, rounded to fp16, + residual rounded again.  Lane (qd = lane & 31,
// sh = lane >> 5) owns quad qd of every range and sums the slots of parity sh; the result is valid on both
// lanes of a quad.  Ranges with !on[r] request nothing.  All ranges are read in the SAME round trips.
template <int NR>
AWQ_DEV void gather_ranges(const ChainLinkDev& P, rsrc_t slres, int M, int m, const int (&col)[NR], const bool (&on)[NR],
                           uint32_t stag, half4_t (&out)[NR], unsigned& spins, ChainCtrl* ctrl, uint32_t code, int lane,
                           unsigned long long* t_loaded = nullptr) {
    const int qd = lane & 31, sh = lane >> 5;
    const uint32_t sstride = (uint32_t)M * 2048u;
    int S[NR];
    uint32_t base[NR];
    int smost = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const SlabRange sr = slab_range(P, M, col[r]);
        S[r] = on[r] ? sr.S : 0;
        base[r] = sr.base + (uint32_t)m * 2048u + (uint32_t)qd * 32u;
        smost = max(smost, S[r]);
    }
    float4_t part[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) part[r] = float4_t{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < smost; s0 += 8) {  // this lane: slots s0 + sh, +2, +4, +6 of every range
        u32x4 v[NR][4][2];
        bool valid[NR][4];  // wave-uniform: the slot (pair: one slot per parity) has been read with good tags on every lane
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) valid[r][u] = s0 + 2 * u >= S[r];  // neither parity's slot exists: nothing to read
        for (;;) {
            // only what is still missing is requested again: the caller starts us when all slabs but one have
            // been seen, so the last slab's DATA arrives with the poll that detects it
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (!valid[r][u]) {
                        const int s = s0 + sh + 2 * u;
                        const uint32_t off = s < S[r] ? base[r] + (uint32_t)s * sstride : OOB;
                        v[r][u][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(slres, off, 0, 16 /* sc1 */));
                        v[r][u][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(slres, off == OOB ? OOB : off + 16u, 0, 16));
                    }
            bool all = true;
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (!valid[r][u]) {
                        const bool mine = s0 + sh + 2 * u < S[r];
                        const bool ok = !mine || (v[r][u][0][1] == stag && v[r][u][0][3] == stag && v[r][u][1][1] == stag && v[r][u][1][3] == stag);
                        valid[r][u] = __all(ok);
                        all &= valid[r][u];
                    }
            if (all) break;
            if (give_up(spins, ctrl, code, lane)) break;
        }
        if (t_loaded) *t_loaded = wall_clock64();
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u)  // slots past S were requested out of range or never: contribute zero
                if (s0 + sh + 2 * u < S[r])
                    part[r] += float4_t{u2f(v[r][u][0][0]), u2f(v[r][u][0][2]), u2f(v[r][u][1][0]), u2f(v[r][u][1][2])};
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float4_t p = part[r];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = p[e] + __shfl_xor(p[e], 32, 64);  // even + odd slots: the same on both lanes
        const int c = col[r] + 4 * qd;
        const bool live = on[r] && c < P.N;  // N % 8 == 0: a quad is all in or all out
        if (P.bias && live) {
            const half4_t b4 = *reinterpret_cast<const half4_t*>(P.bias + c);
            p += float4_t{(float)b4[0], (float)b4[1], (float)b4[2], (float)b4[3]};
        }
        half4_t o = {(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3]};
        if (P.add_res && live) {  // fp16(fp16(projection) + residual): the two roundings torch makes
            const half4_t r4 = *reinterpret_cast<const half4_t*>(P.add_res + (size_t)m * P.N + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)o[e] + (float)r4[e]);
        }
        out[r] = o;
    }
}

// Light wait: until one granule (first quad, last batch row) of all but `slack` slabs of every range has landed.
// Lane (r = lane >> 4, i = lane & 15) probes slot i of range r (ranges of more than 16 slabs: chunk after chunk,
// every slab): a waiting wave costs a few 8-byte loads per turn instead of re-reading every slab.
template <int NR>
AWQ_DEV void probe_ranges(const ChainLinkDev& P, rsrc_t slres, int M, const int (&col)[NR], const bool (&on)[NR], uint32_t stag,
                          int slack, unsigned& spins, ChainCtrl* ctrl, uint32_t code, int lane) {
    static_assert(NR <= 4, "16 lanes per range");
    const int pr = lane >> 4, pi = lane & 15;
    int S = 0, Sr[NR];
    uint32_t base = 0;
    bool small = true;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const SlabRange sr = slab_range(P, M, col[r]);
        Sr[r] = on[r] ? sr.S : 0;
        small &= Sr[r] <= 16;
        if (r == pr && on[r]) {
            S = sr.S;
            base = sr.base + (uint32_t)(M - 1) * 2048u;
        }
    }
    const uint32_t sstride = (uint32_t)M * 2048u;
    if (small) {
        const bool probing = pi < S;
        const uint32_t off = probing ? base + (uint32_t)pi * sstride : OOB;
        for (;;) {
            const u32x2 pv = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(slres, off, 0, 16 /* sc1 */));
            const unsigned long long seen = __ballot(probing && pv[1] == stag);
            bool enough = true;
#pragma unroll
            for (int r = 0; r < NR; ++r) enough &= __builtin_popcountll((seen >> (16 * r)) & 0xFFFFull) >= Sr[r] - slack;
            if (enough) break;
            if (give_up(spins, ctrl, code, lane)) break;
        }
        return;
    }
    for (int s0 = 0; __any(s0 < S); s0 += 16) {
        const bool probing = s0 + pi < S;
        const uint32_t off = probing ? base + (uint32_t)(s0 + pi) * sstride : OOB;
        for (;;) {
            const u32x2 pv = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(slres, off, 0, 16 /* sc1 */));
            if (__all(!probing || pv[1] == stag)) break;
            if (give_up(spins, ctrl, code, lane)) break;
        }
    }
}

// NREG = live D registers per lane: 2 when M == 1 (rows 0, 1 of the selector MFMA), 4 for M <= 8
template <int NREG>
__global__ __launch_bounds__(NTHR, (NWAVES + 3) / 4) void awq_chain_kernel(const ChainHeader* __restrict__ plan, unsigned char* __restrict__ ws) {
    constexpr int CWP = CW + 8;  // LDS row pitch of the fold area (floats)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // dynamic LDS: ring[NS][SLOT_BYTES] | red[NCWB][M][CWP] fp32 | xs[2][UPB][M + 1][128] fp16 (by link parity) | control words
    const int M = (int)plan->M;
    const int G = (int)plan->G;
    const int n_links = (int)plan->n_links;
    const int NS = ring_slots(M);
    const uint32_t ns_magic = plan->ns_magic;
    auto ring_div = [&](uint32_t f) -> uint32_t { return __umulhi(f, ns_magic); };              // f / NS
    auto ring_mod = [&](uint32_t f) -> uint32_t { return f - __umulhi(f, ns_magic) * (uint32_t)NS; };  // f % NS
    const ChainLinkDev* __restrict__ links = reinterpret_cast<const ChainLinkDev*>(plan + 1);
    ChainCtrl* ctrl = reinterpret_cast<ChainCtrl*>(ws);
    const rsrc_t slres = mk_rsrc(ws + CTRL_BYTES, (uint32_t)plan->slab_bytes);

    unsigned char* const ring = smem;
    float* red = reinterpret_cast<float*>(smem + (size_t)NS * SLOT_BYTES);
    half_t* xs_all = reinterpret_cast<half_t*>(reinterpret_cast<unsigned char*>(red) + (size_t)NCWB * M * CWP * 4);
    uint32_t* lds_ctl = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(xs_all) + (size_t)2 * UPB * (M + 1) * 128 * 2);
    uint32_t* const fold_cnt = lds_ctl + 1;   // [2] arrivals per fold group (monotonic)
    uint32_t* const fold_done = lds_ctl + 3;  // [2] folds completed per group (monotonic)
    uint32_t* const lds_abort = lds_ctl + 5;  // set by a wave that gave up: everybody in the block stops waiting
    uint32_t* const xflag = lds_ctl + 8;      // [UPB] link index + 1 whose activations are staged for the unit
    uint32_t* const xdone = lds_ctl + 16;     // [NCWB] link index + 1 the wave has finished reading
    uint32_t* const ready = lds_ctl + 32;     // [NS] fill index + 1 that has landed in the slot
    uint32_t* const freed = lds_ctl + 40;     // [NS] releases of the slot (two per tenant: one per half unit)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if (tid < 64) lds_ctl[tid] = 0u;
    __syncthreads();
    if (tid == 0) {
        const u64 a = __hip_atomic_fetch_add(&ctrl->arrive[b & 7], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_ctl[0] = (uint32_t)(a / (u64)(G >> 3)) + 1u;  // epoch of this launch
    }
    for (int i = tid; i < 2 * UPB * 128; i += NTHR) xs_all[((i >> 7) * (M + 1) + M) * 128 + (i & 127)] = (half_t)0.f;  // the all-zero rows M
    __syncthreads();
    const uint32_t epoch = lds_ctl[0];
    const uint32_t tag_hi = epoch << 10;
    unsigned long long* const trace = plan->trace;
    auto stamp = [&](int l, int slot) {  // phase timeline for tools/chain_probe.py --trace (off unless the plan carries a buffer)
        if (trace && lane == 0) trace[(((size_t)l * G + b) * NWAVES + wave) * 4 + slot] = wall_clock64();
    };
    auto units_here = [&](const ChainLinkDev& L) { return max(0, min(UPB, L.units - b * UPB)); };  // units of a link held by this block

    if (wave == 0) {
        // ================================================================ loader wave: HBM -> LDS ring, paced
        // The ONLY bulk reader of the CU.  It runs ahead of the dependency chain as far as the ring has room
        // (the weights depend on nothing) but keeps at most three fills (51 KB) in flight, so that the CU's
        // memory pipe never holds more than that in front of a poll wave's request.  All of its LDS accesses
        // are inline asm: hipcc would drain the DMA queue (vmcnt(0)) in front of any LDS access it can see.
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;  // LDS byte address of the dynamic segment
        const uint32_t ready_a = lds0 + (uint32_t)(reinterpret_cast<unsigned char*>(ready) - smem);
        const uint32_t freed_a = lds0 + (uint32_t)(reinterpret_cast<unsigned char*>(freed) - smem);
        const uint32_t abort_a = lds0 + (uint32_t)(reinterpret_cast<unsigned char*>(lds_abort) - smem);
        auto lds_read = [](uint32_t addr) -> uint32_t {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            return v;
        };
        auto lds_write = [](uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); };
        const int rr = lane >> 3, cc = lane & 7;
        uint32_t f = 0;  // fill sequence number of this block
        const uint32_t lag = plan->inflight - 1u;  // fills still in flight when the next one is issued
        bool dead = false;
        for (int l = 0; l < n_links && !dead; ++l) {
            const ChainLinkDev& L = links[l];
            const int nblk = units_here(L);
            const uint32_t row_bytes = (uint32_t)(L.N >> 3) * 4u;
            for (int i = 0; i < nblk; ++i, ++f) {
                const int slot = (int)ring_mod(f);
                if (f >= (uint32_t)NS) {  // the slot's previous tenant has been read by both of its waves
                    unsigned spins = 0;
                    unsigned long long t0 = 0;
                    while (lds_read(freed_a + 4u * slot) != (uint32_t)WPU * ring_div(f)) {
                        __builtin_amdgcn_s_sleep(2);
                        if ((++spins & 63u) == 0) {
                            const unsigned long long now = wall_clock64();
                            if (t0 == 0) t0 = now;
                            if (lds_read(abort_a) || now - t0 > 20000000ull) { dead = true; break; }
                        }
                    }
                    if (dead) break;
                }
                if (i == 0) stamp(l, 0);
                const int u = b * UPB + i;
                const int uq = (int)__umulhi((uint32_t)u, L.r_magic);
                const int tile = L.tile0 + uq, row0 = (u - uq * L.R) * 128;
                unsigned char* const dst = ring + (size_t)slot * SLOT_BYTES;
                // 16 x 1 KiB: instruction k moves rows row0 + 8k .. + 7, 128 bytes each (lane = row rr, 16-byte chunk cc)
                const uint32_t coff = min((uint32_t)tile * 128u + (uint32_t)cc * 16u, row_bytes - 16u);  // ragged last tile: stay inside the row
                const unsigned char* src = reinterpret_cast<const unsigned char*>(L.qweight) + (size_t)(row0 + rr) * row_bytes + coff;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    __builtin_amdgcn_global_load_lds((gl_ptr_t)(src + (size_t)k * 8u * row_bytes), (lds_ptr_t)(dst + k * 1024), 16, 0, 2 /* nt */);
                {   // the group's zeros (lanes 0-7: 128 bytes) and scales (lanes 8-39: 512 bytes) behind the weights
                    const int grp = (int)__umulhi((uint32_t)row0, L.g_magic);
                    const unsigned char* zs;
                    if (lane < 8) zs = reinterpret_cast<const unsigned char*>(L.qzeros) + (size_t)grp * row_bytes +
                                       min((uint32_t)tile * 128u + (uint32_t)lane * 16u, row_bytes - 16u);
                    else zs = reinterpret_cast<const unsigned char*>(L.scales) + ((size_t)grp * L.N) * 2u +
                              min((uint32_t)tile * 512u + (uint32_t)(lane - 8) * 16u, (uint32_t)L.N * 2u - 16u);
                    if (lane < 40) __builtin_amdgcn_global_load_lds((gl_ptr_t)zs, (lds_ptr_t)(dst + 16384), 16, 0, 0);
                }
                // fills land in order: everything but the `lag` newest (17 instructions each) is in LDS
                if (lag == 2u) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
                else if (lag == 1u) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (f >= lag) lds_write(ready_a + 4u * ring_mod(f - lag), f - lag + 1u);
            }
        }
        if (lag == 2u) {
            asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            if (f >= 2u) lds_write(ready_a + 4u * ring_mod(f - 2u), f - 1u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lag >= 1u && f >= 1u) lds_write(ready_a + 4u * ring_mod(f - 1u), f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }

    if (wave > NCWB) {
        // ================================================================ poll waves: activations in, results out
        const int pw = wave - NCWB - 1;
        const int qd = lane & 31, sh = lane >> 5;
        const int npoll = G * NPW, pidx = b * NPW + pw;
        auto materialise = [&](int l) {  // y of the Linear whose last sub-link is l (its slabs are complete or about to be)
            const ChainLinkDev& L = links[l];
            const int njobs = L.tiles_full * M * 2;  // (tile, batch row, half tile)
            const uint32_t stag = tag_hi | (uint32_t)(L.out_id + 1);
            for (int jb = pidx; jb < njobs; jb += npoll) {
                const int tl = jb / (2 * M), m = (jb >> 1) % M, hf = jb & 1;
                const int col[1] = {tl * CW + hf * 128};
                const bool on[1] = {true};
                half4_t o[1];
                unsigned spins = 0;
                probe_ranges<1>(L, slres, M, col, on, stag, 0, spins, ctrl, 2u, lane);
                gather_ranges<1>(L, slres, M, m, col, on, stag, o, spins, ctrl, 2u, lane);
                const int c = col[0] + 4 * qd;
                if (sh == 0 && c < L.N) *reinterpret_cast<u32x2*>(L.y + (size_t)m * L.N + c) = __builtin_bit_cast(u32x2, o[0]);
            }
        };
        for (int l = 0; l < n_links; ++l) {
            const ChainLinkDev& L = links[l];
            if ((l & 3) == 0 && ld_abort(ctrl)) lds_st(lds_abort, 1u);  // somebody gave up: release this block's other waves
            const int nblk = units_here(L);
            int kg[UPP];
            bool has[UPP];
            bool any = false;
#pragma unroll
            for (int i = 0; i < UPP; ++i) {
                has[i] = pw * UPP + i < nblk;
                kg[i] = (b * UPB + pw * UPP + i) - (int)__umulhi((uint32_t)(b * UPB + pw * UPP + i), L.r_magic) * L.R;
                any |= has[i];
            }
            if (any) {
                stamp(l, 0);
                unsigned spins = 0;
#pragma unroll
                for (int i = 0; i < UPP; ++i)  // the staging rows alternate by link parity: the waves of the unit are done with link l - 2
#pragma unroll
                    for (int h = 0; h < WPU; ++h)
                        while (has[i] && (int)lds_ld(&xdone[WPU * (pw * UPP + i) + h]) < l - 1)
                            if (give_up(spins, ctrl, 8u, lane)) break;
                half_t* const xs_l = xs_all + (size_t)(l & 1) * UPB * (M + 1) * 128;
                if (L.xflags & XF_SLABS) {
                    const ChainLinkDev& P = links[L.prod];
                    const uint32_t stag = tag_hi | (uint32_t)(P.out_id + 1);
                    const bool gated = (L.xflags & XF_GATED) != 0;
                    int col[UPP], colu[UPP];
#pragma unroll
                    for (int i = 0; i < UPP; ++i) {
                        col[i] = L.x_col0 + kg[i] * 128;
                        colu[i] = col[i] + L.K;
                    }
                    // sub-links of one Linear share their producer: its slabs have been seen complete one link ago
                    const bool seen = l > 0 && links[l - 1].out_id == L.out_id && !gated;
                    if (!seen) probe_ranges<UPP>(P, slres, M, col, has, stag, (int)plan->slack, spins, ctrl, 1u, lane);
                    if (gated) probe_ranges<UPP>(P, slres, M, colu, has, stag, (int)plan->slack, spins, ctrl, 1u, lane);
                    stamp(l, 1);
                    for (int m = 0; m < M; ++m) {
                        half4_t xv[UPP];
                        unsigned long long t_ld = 0;
                        gather_ranges<UPP>(P, slres, M, m, col, has, stag, xv, spins, ctrl, 1u, lane, trace ? &t_ld : nullptr);
                        if (trace && lane == 0 && m == 0) trace[(((size_t)l * G + b) * NWAVES + wave) * 4 + 3] = t_ld;
                        if (gated) {  // silu(gate) * up in fp32, one rounding: == awq_silu_and_mul_kernel
                            half4_t up[UPP];
                            gather_ranges<UPP>(P, slres, M, m, colu, has, stag, up, spins, ctrl, 1u, lane);
#pragma unroll
                            for (int i = 0; i < UPP; ++i)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float gx = (float)xv[i][e];
                                    xv[i][e] = (half_t)((gx / (1.0f + expf(-gx))) * (float)up[i][e]);
                                }
                        }
#pragma unroll
                        for (int i = 0; i < UPP; ++i)
                            if (has[i] && sh == 0)
                                *reinterpret_cast<u32x2*>(xs_l + ((size_t)(pw * UPP + i) * (M + 1) + m) * 128 + 4 * qd) =
                                    __builtin_bit_cast(u32x2, xv[i]);
                    }
                } else {
                    stamp(l, 1);
#pragma unroll
                    for (int i = 0; i < UPP; ++i)
                        for (int m = 0; m < M; ++m)
                            if (has[i])
                                *reinterpret_cast<uint32_t*>(xs_l + ((size_t)(pw * UPP + i) * (M + 1) + m) * 128 + 2 * lane) =
                                    *reinterpret_cast<const uint32_t*>(L.x + (size_t)m * L.x_stride + kg[i] * 128 + 2 * lane);
                }
                if (spins >= 64u && ld_abort(ctrl)) lds_st(lds_abort, 1u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#pragma unroll
                for (int i = 0; i < UPP; ++i)
                    if (has[i] && lane == 0) lds_st(&xflag[pw * UPP + i], (uint32_t)(l + 1));
                stamp(l, 2);
            }
            // the Linear that ended one link ago: every one of its slabs has been consumed by now or is on its way
            if (l > 0 && links[l - 1].last_sub && links[l - 1].y) materialise(l - 1);
        }
        if (links[n_links - 1].y) materialise(n_links - 1);
        return;
    }

    // ==================================================================== compute waves
    const int cw = wave - 1;
    const int ui = cw / WPU, hf = cw % WPU;  // unit of the block, part of the unit (sets hf * SETS / WPU ...)
    const int j = lane & 15, kb = lane >> 4;
    half_t* const xs0 = xs_all + (size_t)ui * (M + 1) * 128;  // the unit's activation rows [M + 1][128], parity 0
    float* myred = red + (size_t)cw * M * CWP;

    const uint32_t sel_lo = (j & 1) ? 0x01000C0Cu : 0x0C0C0100u;  // low half of a dword -> slot (j & 1)
    const uint32_t sel_hi = (j & 1) ? 0x03020C0Cu : 0x0C0C0302u;  // high half
    const u32x4v ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    const int arow = min(j >> 1, M);  // A row of this lane = batch row j >> 1 (parity j & 1); rows >= M read zeros

    // bookkeeping, identical in every wave of the block (pure arithmetic on the link table)
    uint32_t fbase = 0u;            // fills of earlier links
    uint32_t arr0 = 0u, arr1 = 0u;  // arrivals of earlier folds, per group
    uint32_t fin0 = 0u, fin1 = 0u;  // completed folds of earlier links, per group
    int my_pg = -1;                 // group and completion count of the last fold this wave wrote rows for
    uint32_t my_pf = 0u;

    for (int l = 0; l < n_links; ++l) {
        const ChainLinkDev& L = links[l];
        const int R = L.R;
        const int nblk = units_here(L);
        const bool has = ui < nblk;
        const int u0 = b * UPB;
        const int tloc = (int)__umulhi((uint32_t)(u0 + ui), L.r_magic);  // tile within the sub-link
        // the block's units u0 .. u0 + nblk - 1 touch at most two tiles (R >= UPB): group 0 = the first unit's tile
        const int t_first = (int)__umulhi((uint32_t)u0, L.r_magic);
        const int n0 = min(nblk, (t_first + 1) * R - u0);
        const int n1 = nblk - n0;
        const int grp = (has && tloc != t_first) ? 1 : 0;
        float yv[8][NREG];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < NREG; ++r) yv[c][r] = 0.f;

        if (has) {
            stamp(l, 0);
            const uint32_t f = fbase + (uint32_t)ui;
            const int slot = (int)ring_mod(f);
            const unsigned char* const wb = ring + (size_t)slot * SLOT_BYTES;
            {   // the unit's weights have landed in the ring, the poll wave has staged its 128 activations per batch row
                unsigned spins = 0;
                unsigned long long t0 = 0;
                while (lds_ld(&ready[slot]) != f + 1u)
                    if (give_up_lds(spins, lds_abort, t0, ctrl, 32u, lane)) break;
                while (lds_ld(&xflag[ui]) < (uint32_t)(l + 1))  // the poll wave may already be a link ahead
                    if (give_up_lds(spins, lds_abort, t0, ctrl, 16u, lane)) break;
            }
            stamp(l, 1);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);

            // ---- MFMAs over this wave's sets of the unit (all inside one group)
            float4_t acc[8];
            float4_t accsx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = float4_t{0.f, 0.f, 0.f, 0.f};
            const half_t* xlane = xs0 + (size_t)(l & 1) * UPB * (M + 1) * 128 + arow * 128 + 4 * kb + (128 / WPU) * hf;
            const unsigned char* wl = wb + ((128 / WPU) * hf + 4 * kb) * 128 + j * 8;  // this lane's 8 bytes of row 4 kb of the wave's first set
#pragma unroll
            for (int t = 0; t < SETS / WPU; ++t) {
                const u32x2 xq = *reinterpret_cast<const u32x2*>(xlane + 16 * t);
                const uint32_t x01 = xq[0], x23 = xq[1];
                const u32x4v a0 = {__builtin_amdgcn_perm(0u, x01, sel_lo), __builtin_amdgcn_perm(0u, x01, sel_hi),
                                   __builtin_amdgcn_perm(0u, x23, sel_lo), __builtin_amdgcn_perm(0u, x23, sel_hi)};
                accsx = mfma16(a0, ones, accsx);
                const u32x2 w0 = *reinterpret_cast<const u32x2*>(wl + (16 * t + 0) * 128);
                const u32x2 w1 = *reinterpret_cast<const u32x2*>(wl + (16 * t + 1) * 128);
                const u32x2 w2 = *reinterpret_cast<const u32x2*>(wl + (16 * t + 2) * 128);
                const u32x2 w3 = *reinterpret_cast<const u32x2*>(wl + (16 * t + 3) * 128);
#pragma unroll
                for (int wd = 0; wd < 2; ++wd) {
                    const uint32_t q0 = w0[wd], q1 = w1[wd], q2 = w2[wd], q3 = w3[wd];
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
            // ---- y = s * (acc - (bias_J + z) * sum_x), the unit lies inside one group
            const u32x2 zq = *reinterpret_cast<const u32x2*>(wb + 16384 + j * 8);
            const u32x4 sq0 = *reinterpret_cast<const u32x4*>(wb + 16384 + 128 + j * 32);
            const u32x4 sq1 = *reinterpret_cast<const u32x4*>(wb + 16384 + 128 + j * 32 + 16);
#pragma unroll
            for (int wd = 0; wd < 2; ++wd) {
                const uint32_t zw = zq[wd], zw8 = zw >> 8;
                const uint32_t zp[4] = {pairb<0>(zw, zw8), pairb<1>(zw, zw8), pairb<2>(zw, zw8), pairb<3>(zw, zw8)};
                const u32x4 sv = wd ? sq1 : sq0;
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    const half2_t z2 = u2h2(zp[J]), s2 = u2h2(sv[J]);
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int e = r & 1;
                        const float raw = __builtin_fmaf(-(float)z2[e], accsx[r], acc[wd * 4 + J][r]);
                        yv[wd * 4 + J][r] = (float)s2[e] * raw;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&freed[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // this half is done with the slot
            stamp(l, 2);
        }
        if (lane == 0) lds_st(&xdone[cw], (uint32_t)(l + 1));  // xs may be re-staged (also when this link had nothing for us)

        if (has) {
            // ---- fold the waves of the block that share the tile through LDS; the last one to arrive sums and
            // stores the slab.  Arrivals and completed folds are counted monotonically per group; a wave must not
            // overwrite its rows before the fold it last took part in has been read.
            {
                // (a) the fold this wave last wrote rows for has been read; (b) every earlier fold of the group it
                // joins NOW is complete -- a unit changes tile, hence group, from link to link, and a wave that is
                // a link ahead of its new group's stragglers must not be counted among their arrivals
                const uint32_t fing_ = grp ? fin1 : fin0;
                unsigned spins = 0;
                unsigned long long t0 = 0;
                while ((my_pg >= 0 && lds_ld(&fold_done[my_pg]) < my_pf) || lds_ld(&fold_done[grp]) < fing_)
                    if (give_up_lds(spins, lds_abort, t0, ctrl, 4u, lane)) break;
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
            const uint32_t fing = grp ? fin1 : fin0, arrg = grp ? arr1 : arr0;
            my_pg = grp;
            my_pf = fing + 1u;
            uint32_t arrived = 0;
            if (lane == 0) arrived = __hip_atomic_fetch_add(&fold_cnt[grp], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            arrived = __builtin_amdgcn_readfirstlane(arrived);
            const int ng = WPU * (grp ? n1 : n0), w0 = grp ? WPU * n0 : 0;  // waves of the group
            if (arrived == arrg + (uint32_t)ng - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t stag = tag_hi | (uint32_t)(L.out_id + 1);
                const int tile = L.tile0 + tloc;
                const int slot = b - (tloc * R) / UPB;
                const uint32_t so = L.slab_off + (uint32_t)((tile * L.smax + slot) * M) * 2048u;
                for (int qd = lane; qd < M * 64; qd += 64) {
                    const int m = qd >> 6, c4 = (qd & 63) * 4;
                    float4_t s = {0.f, 0.f, 0.f, 0.f};
                    for (int w = w0; w < w0 + ng; ++w) {
                        const float* src = red + (size_t)(w * M + m) * CWP + c4;
                        s += float4_t{src[0], src[1], src[2], src[3]};
                    }
                    const u32x4 g0 = {f2u(s[0]), stag, f2u(s[1]), stag};
                    const u32x4 g1 = {f2u(s[2]), stag, f2u(s[3]), stag};
                    __builtin_amdgcn_raw_buffer_store_b128(g0, slres, (uint32_t)qd * 32u + so, 0, 16 /* sc1 */);
                    __builtin_amdgcn_raw_buffer_store_b128(g1, slres, (uint32_t)qd * 32u + 16u + so, 0, 16);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the fold rows have been read
                if (lane == 0) lds_st(&fold_done[grp], fing + 1u);
            }
            stamp(l, 3);
        }
        fbase += (uint32_t)nblk;
        arr0 += (uint32_t)(WPU * n0);
        arr1 += (uint32_t)(WPU * n1);
        fin0 += n0 > 0 ? 1u : 0u;
        fin1 += n1 > 0 ? 1u : 0u;
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

size_t chain_lds_bytes(int M) {
    return (size_t)ring_slots(M) * SLOT_BYTES + (size_t)NCWB * M * (CW + 8) * 4 + (size_t)2 * UPB * (M + 1) * 128 * 2 + 256;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

size_t awq_chain_plan_bytes(int64_t n_links) {
    if (n_links <= 0) return 0;
    return sizeof(ChainHeader) + (size_t)n_links * MAX_SUB * sizeof(ChainLinkDev);
}

int awq_chain_grid_blocks(void) {
    static const int cached = [] {
        const char* e = getenv("AWQ_CHAIN_GRID");
        if (e && atoi(e) >= 8) return atoi(e) / 8 * 8;
        const DevInfo d = dev_info();
        const int cus = d.cus > 0 ? d.cus : 256;  // no device visible (build box): plan for an MI355X
        return cus / 8 * 8;                       // one block per CU; a multiple of 8 (epoch counters)
    }();
    return cached;
}

int awq_chain_build(const AwqChainLink* links, int64_t n, int64_t M, void* workspace, size_t workspace_bytes,
                    void* plan_host, size_t plan_bytes, size_t* workspace_needed) {
    if (!links || !plan_host) return AWQ_ERR_NULL;
    if (n < 1 || n > 1000 || M < 1 || M > 8) return AWQ_ERR_UNSUPPORTED;
    if (plan_bytes < awq_chain_plan_bytes(n)) return AWQ_ERR_WORKSPACE;
    const int G = awq_chain_grid_blocks();
    const int TCW = G * UPB;  // units a (sub-)link may hold: six per block
    ChainHeader* H = static_cast<ChainHeader*>(plan_host);
    ChainLinkDev* out = reinterpret_cast<ChainLinkDev*>(H + 1);
    memset(H, 0, sizeof(*H));
    struct Lin { int R, tiles, per, nsub, smax, first; size_t slab, off; };
    std::vector<Lin> lin((size_t)n);
    // Slab regions.  A Linear's slabs are read by the next Linear's poll waves while they stage its input, so
    // Linear i and i + 2 never overlap in time (i + 2 stores only after every slab of i + 1 is complete, i.e.
    // after every unit of i + 1 has been given its input, which was read from i): two alternating regions
    // serve the whole chain.  A Linear whose result is also materialised (y != NULL) is read once more, one
    // link later: it gets a private region.
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
        } else {
            if (i != 0) return AWQ_ERR_UNSUPPORTED;
            if (!a.x || gated) return a.x ? AWQ_ERR_UNSUPPORTED : AWQ_ERR_NULL;
            if (a.x_stride < a.K || a.x_stride % 2) return AWQ_ERR_BAD_SHAPE;
        }
        if (i == n - 1 && !a.y) return AWQ_ERR_NULL;  // somebody has to want the result
        Lin& L = lin[(size_t)i];
        L.R = (int)(a.K / 128);
        L.tiles = (int)((a.N + CW - 1) / CW);
        if (L.R > TCW || L.R < UPB) return AWQ_ERR_UNSUPPORTED;  // a block's units touch at most two tiles
        L.per = TCW / L.R;  // whole tiles per sub-link
        if (L.per > L.tiles) L.per = L.tiles;
        L.nsub = (L.tiles + L.per - 1) / L.per;
        if (L.nsub > MAX_SUB) return AWQ_ERR_UNSUPPORTED;
        L.per = (L.tiles + L.nsub - 1) / L.nsub;  // even sub-links
        L.smax = (L.R + UPB - 1) / UPB + 1;
        L.slab = align_up((size_t)L.tiles * L.smax * (size_t)M * 2048, 4096);
        if (a.y) { L.off = priv; priv += L.slab; }
        else if (L.slab > half[i & 1]) half[i & 1] = L.slab;
        L.first = total;
        total += L.nsub;
    }
    if (total > 1023 || n > 1023) return AWQ_ERR_UNSUPPORTED;  // 10 bits of tag
    const size_t slab_bytes = half[0] + half[1] + priv;
    const size_t need = CTRL_BYTES + slab_bytes;
    if (slab_bytes >= ((size_t)1 << 31)) return AWQ_ERR_UNSUPPORTED;
    if (workspace_needed) *workspace_needed = need;
    if (!workspace) return AWQ_OK;  // size query
    if (workspace_bytes < need) return AWQ_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return AWQ_ERR_BAD_ALIGNMENT;
    int k = 0;
    for (int64_t i = 0; i < n; ++i) {
        const AwqChainLink& a = links[i];
        const Lin& L = lin[(size_t)i];
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
            d.tile0 = sidx * L.per;
            d.tiles = (sidx + 1) * L.per <= L.tiles ? L.per : L.tiles - sidx * L.per;
            d.tiles_full = L.tiles;
            d.R = L.R;
            d.per_magic = 0;
            if (!awq_magic_u32((uint32_t)L.R, 1u << 20, &d.r_magic) || !awq_magic_u32((uint32_t)a.group_size, (uint32_t)a.K + 128u, &d.g_magic) ||
                (L.per > 1 && !awq_magic_u32((uint32_t)L.per, 1u << 20, &d.per_magic)))
                return AWQ_ERR_UNSUPPORTED;
            d.units = d.tiles * L.R;
            d.per = L.per;
            d.smax = L.smax;
            d.out_id = (int)i;
            d.slab_off = (uint32_t)off;
            d.last_sub = sidx == L.nsub - 1;
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
    H->magic = CHAIN_MAGIC; H->n_links = (uint32_t)k; H->G = (uint32_t)G; H->M = (uint32_t)M;
    H->slab_bytes = slab_bytes; H->n_linears = (uint32_t)n;
    {
        const char* e = getenv("AWQ_CHAIN_INFLIGHT");
        const int v = e ? atoi(e) : 3;
        if (!awq_magic_u32((uint32_t)ring_slots((int)M), 1u << 20, &H->ns_magic)) return AWQ_ERR_UNSUPPORTED;
        H->inflight = (uint32_t)(v < 1 ? 1 : (v > 3 ? 3 : v));
        const char* e2 = getenv("AWQ_CHAIN_SLACK");
        H->slack = e2 ? (uint32_t)(atoi(e2) != 0) : 1u;
    }
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
