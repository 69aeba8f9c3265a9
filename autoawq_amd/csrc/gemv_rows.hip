// gemv_rows.hip -- decode GEMV (1 <= M <= 4) on the GEMV layout as a ROW-STREAMING kernel, gfx950.
//
// Replaces awq_ext.gemv_forward_cuda(x, qweight, scales, qzeros, group_size) as called by
// awq/modules/linear/gemv.py:178-180 (the layout the reference itself recommends for batch 1,
// README.md:96-97).  Layout (packer gemv.py:94-153):
//   qweight [N, K/8] int32, nibble i of word c = w[n, 8c+i];  qzeros [N, ZW] int32, nibble i of word c =
//   z[n, group 8c+i];  scales [N, 8*ZW] fp16.
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 + (K/g)*N/2 + (K/g)*N*2 + M*K*2 + M*N*2.
//
// Design (no exchange, K-contiguous rows, weights requested first; measured floors in
// profiles/r03_stream_probe3.txt: a wave instruction that reads 1 KiB of ONE row streams as fast as a bare linear
// read, the 16-rows-x-64-bytes pattern a direct MFMA fragment load needs does not):
//  * A UNIT is 1 KiB of one row = one `global_load_dwordx4 ... nt` of a wave (64 lanes x 16 bytes, eight whole
//    128-byte lines).  A wave covers SL units (slots) of a row; at batch 1 that is the WHOLE row for K <= 16384
//    (K = 4096: 2 slots, K = 11008: 6), so a lane's K ranges never change: its 32 SL activations live in 16 SL
//    REGISTERS for the whole launch, already in the (t, t+4) pair order the nibble decode produces.  They reach the
//    block once, by LDS-DMA (`global_load_lds_dwordx4`, no VGPRs), requested just ahead of the first weights; one
//    barrier, and from there a wave shares nothing with anybody: no barrier at the end, no exchange, no workspace.
//    (Larger batches or K: wk waves side by side on a row, and one barrier before y is written.)
//  * Per unit a lane spends 5 VALU ops per packed word on the decode (nibbles stay in place under the fp16
//    exponents 2^10 / 2^6, see gemv_mfma.hip); the dot products are eight v_mfma_f32_4x4x4_16b_f16 (16 independent
//    4 x 4 x 4 products, one per lane quad; the lane's own product is the diagonal element; exact products, fp32
//    accumulation) -- level with 16 v_dot2c_f32_f16 in time, half the VALU issue slots.
//  * Four units form a ROUND.  Group scale and zero point come off once per round and lane: a 4 x 4 transpose-reduce
//    inside each lane quad (a quad = the four chunks of one 128-wide group) leaves lane j of the quad with the group
//    sum of unit j, so ONE scale and ONE zero word per lane serve four units (read from LDS at the START of the round: the
//    scales / zeros of all the rows a wave will touch are contiguous in this layout and arrive in the prologue -- by two
//    registers per lane when they fit 1 KiB / 64 words, else by two or three LDS-DMA instructions -- so the stream itself
//    carries packed weights only):
//        y[n] += s[n,g] * (P - C0 - z[n,g] * SX),   P = sum x*(bias + w),  C0 = sum bias*x,  SX = sum x  over the group,
//    C0 / SX constants of the launch.  One-hot and zero inputs stay exact (tests).
//  * A super-unit (SU) is RPU whole rows = SL * RPU units = R rounds with a compile-time (row, slot) pattern
//    (SL 1: 4 rows, 2: 2 rows, 3: 4 rows / 3 rounds, 4: 1 row, 6: 2 rows / 3 rounds, 8: 1 row / 2 rounds).
//  * Every load of the stream is inline asm with hand-counted `s_waitcnt vmcnt(N)` (vector-memory operations retire
//    in order; hipcc falls back to vmcnt(0) at the loop head for a ring it cannot see through): D super-units
//    stay in flight per wave, a round is re-requested right after it is consumed (main phase); the last D super-units
//    are DRAINED with falling wait counts and no requests (requesting dummies past the end to keep one count made every
//    wave end on a memory round trip).  tools/isa_audit.py audit_inflight_regs reads the ISA back (tests/test_boundary.py).
//  * With ONE wave per SIMD nothing hides instruction latency: the prologue's instruction count is launch time
//    (profiles/r03_gemv_rows_trace.txt).  Hence the zero chunk instead of per-element selects, the group constants by one
//    MFMA chain, batched asm LDS reads, the one-piece-per-wave DMA fast path and the kernel-argument field order.
//  * Cross-lane: two DPP rotations per round, then lanes 12-15 of each 16-lane row park four partial sums per unit in
//    LDS; at the end a lane adds the 4 * SL (* wk) partials of a row and writes y.  Bitwise reproducible.
//  * Static partition: SUs dealt evenly to (block, row group); the grid is a multiple of the CU count.
#include <type_traits>

#include "awq_device.h"
#include "awq_internal.h"

// Debug builds only (tools/rows_experiments.py): -DAWQ_ROWS_DBG=bits switches parts of the kernel off (results are wrong
// by design): 1 = no decode / dot products, 2 = no activation DMA, barrier or LDS reads, 4 = no scale / zero arithmetic and
// no transpose-reduce, 8 = no final fold (no y stores), 16 = no scale / zero DMA or LDS reads, 32 = activations requested AFTER the
// first weights (an ordering experiment: measured slower, every wave then preps x only after its whole ring has landed),
// 64 = dot products by v_dot2c_f32_f16 on the VALU instead of v_mfma_f32_4x4x4_16b_f16.
#ifndef AWQ_ROWS_DBG
#define AWQ_ROWS_DBG 0
#endif

namespace {

struct RowsParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    half_t* y;
    int M, K, N;
    int KW, ZW, SW;             // words per qweight / qzeros row, halfs per scales row
    int C, Cp;                  // 16-byte chunks per row (K / 32); the same rounded up to whole 64-chunk blocks
    int wk, rg;                 // waves side by side on a row, row groups per block (blockDim = (64 * wk, rg))
    int lines_base, lines_rem;  // 128-byte lines per (wave, slot): base (+1 for the first rem of the wk * SL slots)
    int su_total;               // super-units of the matrix: ceil(N / RPU)
    int su_base, su_rem;        // super-units per row group, in units of su_gran: base (+1 for the first rem groups)
    int su_max;                 // most SUs any row group gets
    // (these two sit HERE, inside the run of fields the prologue reads first: at the end of the struct they were fetched by a
    // second, dependent scalar load in the middle of the prologue)
    int su_gran;                // super-units are dealt in multiples of this (2: row PAIRS never straddle two waves)
    int sc_regs;                // a wave's scales (<= 1 KiB) and zero words (<= 64) travel through registers, not LDS-DMA
    int x_bytes;                // LDS: activations [MM][4][Cp] x 16 bytes
    int sc_pitch, z_pitch;      // LDS per wave: its rows' scales (whole KiB) and zero words (whole 256 bytes)
    uint32_t g_magic;           // (k * g_magic) >> 32 == k / g
    unsigned long long* trace;  // debug builds only
    // decoder-block prologue / epilogue (FX template bits; batch 1, whole rows per wave):
    const half_t* norm_w;       // FX_NORM: x := fp16(x * rsqrt(mean(x^2) + eps) * norm_w), == awq_rmsnorm_kernel's arithmetic
    float norm_eps;
    const half_t* res;          // FX_RES: y := fp16(fp16(W x) + res), the two roundings of the unfused add
    // FX_GROUPED (MoE decode, round 6): one virtual batch-1 GEMV per (token, expert) pair over stacked GEMV-layout experts.
    // grid = (8 * pairs, ceil(parts / 8)): blockIdx.x = 8 * pair + xcd, part = 8 * blockIdx.y + xcd -- the blocks that stream
    // part q of ALL pairs are consecutive residents of ONE XCD (ids differing by 8), so pairs that share an expert meet in its L2.
    const int* pair_expert;     // [pairs] expert of pair i (topk_ids flattened); outside [0, E): the pair's rows are not written
    const float* pair_scale;    // FX_SCALE: y := fp16(W x * pair_scale[i]) (fp32 product, one rounding)
    int E, parts;               // parts = blocks one matrix is dealt over
    int e_first;                // expert-parallel shards: pair_expert holds GLOBAL ids, this stack holds experts [e_first, e_first + E)
    uint32_t x_div_magic;       // x row of pair i = i / x_div = (i * magic) >> 32; 0: x_div == 1
    int y_pitch, y_rows;        // halfs between the y rows of two pairs; number of pairs (grid.x = 8 * y_rows)
    long long w_stride, z_stride, s_stride;  // bytes between two experts' qweight / qzeros / scales
};

// FX bits of the kernel template
constexpr int FX_NORM = 1, FX_RES = 2, FX_PAIRS = 4;  // FX_PAIRS: rows (2 i, 2 i + 1) = (gate_i, up_i), y[i] = silu(gate) * up
constexpr int FX_GROUPED = 8, FX_SCALE = 16;          // MoE decode: per-pair expert indirection; routing weight in the epilogue

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int CTRL>
AWQ_DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
AWQ_DEV float dot2(uint32_t a, uint32_t b, float c) { return __builtin_amdgcn_fdot2(u2h2(a), u2h2(b), c, false); }
AWQ_DEV float4_t mfma4(u32x2 a, u32x2 b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, a), __builtin_bit_cast(half4_t, b), c, 0, 0, 0);
}

// FX_NORM: one fp16 pair through the norm, fp16(fp32(fp32(x) * inv) * fp32(w)) -- the arithmetic of awq_rmsnorm_kernel, in four
// mixed-precision FMAs (v_fma_mix reads the fp16 halves directly and rounds like the separate multiplies: the addend is -0.0, so
// signed zeros survive) instead of four converts, four multiplies and a packing convert.
AWQ_DEV uint32_t norm_pair(uint32_t x2, uint32_t w2, float inv) {
    float t0, t1;
    uint32_t r;
    const float nz = -0.0f;  // (the addend: x * y + (-0.0) == x * y for every x * y, signed zeros included)
    asm("v_fma_mix_f32 %0, %2, %3, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %2, %3, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(t0), "=&v"(t1) : "v"(x2), "v"(inv), "v"(nz));
    asm("v_fma_mixlo_f16 %0, %1, %3, %4 op_sel_hi:[0,1,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, %4 op_sel:[0,1,0] op_sel_hi:[0,1,0]"
        : "=&v"(r) : "v"(t0), "v"(t1), "v"(w2), "v"(nz));
    return r;
}

// One round's request: four weight units, non-temporal.  The SGPR base is the tensor pointer itself (a kernel argument,
// never a fresh SALU result: no wait states needed before a vector-memory instruction reads it, cdna_hip_programming.md
// 5.7); row and lane offsets travel in the 32-bit VGPR offset (the launcher rejects tensors of 4 GiB and more).
#define AWQ_ROWS_LPR 4
#define AWQ_ROWS_REQUEST(R, vw0, vw1, vw2, vw3, base)                                                                        \
    asm volatile("global_load_dwordx4 %0, %4, %8 nt\n\tglobal_load_dwordx4 %1, %5, %8 nt\n\t"                                  \
                 "global_load_dwordx4 %2, %6, %8 nt\n\tglobal_load_dwordx4 %3, %7, %8 nt"                                      \
                 : "=&v"(R.q[0]), "=&v"(R.q[1]), "=&v"(R.q[2]), "=&v"(R.q[3])                                                  \
                 : "v"(vw0), "v"(vw1), "v"(vw2), "v"(vw3), "s"(base)                                                           \
                 : "memory")
// The wait names every register of the round (they stay allocated until the data has landed) and prints them as a comment:
// tools/isa_audit.py checks that they ARE the registers the request wrote (no compiler copy of a register whose load is
// still in flight) and that nothing between request and wait touches them.
// the grouped form's request: the same four loads WITHOUT the non-temporal hint -- pairs that share an expert stream the same
// lines from neighbouring CUs of one XCD, and the line a first reader brought in has to stay in that XCD's L2 for the others
#define AWQ_ROWS_REQUEST_KEEP(R, vw0, vw1, vw2, vw3, base)                                                                   \
    asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\t"                                        \
                 "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8"                                            \
                 : "=&v"(R.q[0]), "=&v"(R.q[1]), "=&v"(R.q[2]), "=&v"(R.q[3])                                                  \
                 : "v"(vw0), "v"(vw1), "v"(vw2), "v"(vw3), "s"(base)                                                           \
                 : "memory")
#define AWQ_ROWS_WAIT(R, newer) \
    asm volatile("s_waitcnt vmcnt(%4) ; releases %0 %1 %2 %3" : "+v"(R.q[0]), "+v"(R.q[1]), "+v"(R.q[2]), "+v"(R.q[3]) : "n"(newer))
// 1 KiB (64 x 16 bytes) / 256 bytes (64 x 4) from global memory straight into LDS at M0 + 16 (4) * lane: no VGPRs
#define AWQ_ROWS_DMA16(voff, base, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")
#define AWQ_ROWS_DMA4(voff, base, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")

// Phase stamps for tools/trace_gemv_rows.py (debug build only: -DAWQ_GEMV_TRACE); kept in registers, stored at the very end
// (a store inside the stream would count in vmcnt)
#ifdef AWQ_GEMV_TRACE
#define ROWS_STAMP(slot) ts[slot] = wall_clock64()
#else
#define ROWS_STAMP(slot) do { } while (0)
#endif

struct Round {  // four units in flight
    u32x4 q[4];
};

template <int I, int N, class F>
AWQ_DEV void static_for(F&& f) {  // f(integral_constant<int, I>) ... f(integral_constant<int, N - 1>)
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int rows_per_su(int SL) { return SL == 1 || SL == 3 ? 4 : (SL == 2 || SL == 6 ? 2 : 1); }

// Branch-free select by the lane's position in its quad (hipcc turns a ?: chain on a per-lane condition into exec-masked
// branches): m0 / m1 / m2 are all-ones in the lanes with (lane & 3) == 0 / 1 / 2; (m & a) | (~m & b) is one v_bfi_b32.
struct QuadSel {
    uint32_t m0, m1, m2;
    AWQ_DEV uint32_t bits(uint32_t a, uint32_t b, uint32_t c, uint32_t d) const {
        const uint32_t cd = (m2 & c) | (~m2 & d), bcd = (m1 & b) | (~m1 & cd);
        return (m0 & a) | (~m0 & bcd);
    }
    AWQ_DEV int operator()(int a, int b, int c, int d) const { return (int)bits((uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)d); }
    AWQ_DEV float operator()(float a, float b, float c, float d) const {
        return __builtin_bit_cast(float, bits(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b),
                                              __builtin_bit_cast(uint32_t, c), __builtin_bit_cast(uint32_t, d)));
    }
};

// SL: 1-KiB slots of a row per wave (1, 2, 3, 4, 6, 8); D: super-units in flight per wave (1 | 2); MM: batch rows
template <int SL, int D, int MM, int FX = 0>
__global__ __launch_bounds__(512) void awq_gemv_rows_kernel(RowsParams p) {
    static_assert(FX == 0 || MM == 1, "the block prologue / epilogue is built for batch 1");
    static_assert(!(FX & FX_SCALE) || (FX & FX_GROUPED), "the routing weight belongs to the grouped form");
    static_assert(!(FX & FX_GROUPED) || !(FX & (FX_NORM | FX_RES)), "grouped form: silu pairs or the routing weight only");
    constexpr int RPU = rows_per_su(SL);  // rows per super-unit
    constexpr int R = SL * RPU / 4;       // rounds per super-unit; unit u of an SU = (row u / SL, slot u % SL)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: xs[MM][4 pieces][Cp + 1] x 16 bytes (piece j of chunk c at (j * (Cp + 1) + c) * 16; chunk Cp = zeros) | scales / zeros | red[MM][rows of the block][wk * SL][4]
    const int lane = threadIdx.x & 63;
    const int wki = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rgi = __builtin_amdgcn_readfirstlane(threadIdx.y);
#ifdef AWQ_GEMV_TRACE
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ROWS_STAMP(0);
    // the five tensor bases: the kernel arguments themselves, or (FX_GROUPED) this pair's expert / activation row / output row
    const uint32_t* qw_base = p.qweight;
    const uint32_t* qz_base = p.qzeros;
    const half_t* sc_base = p.scales;
    const half_t* x_base = p.x;
    half_t* y_base = p.y;
    float pscale = 1.f;
    int part = blockIdx.x;
    if constexpr (FX & FX_GROUPED) {
        const int pr = blockIdx.x >> 3;
        part = 8 * blockIdx.y + (blockIdx.x & 7);
        if (part >= p.parts) return;
        const int e = p.pair_expert[pr] - p.e_first;  // (scalar load: uniform index)
        if ((unsigned)e >= (unsigned)p.E) return;
        qw_base = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.qweight) + (long long)e * p.w_stride);
        qz_base = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.qzeros) + (long long)e * p.z_stride);
        sc_base = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(p.scales) + (long long)e * p.s_stride);
        x_base = p.x + (long long)(p.x_div_magic ? (int)__umulhi((uint32_t)pr, p.x_div_magic) : pr) * p.K;
        y_base = p.y + (long long)pr * p.y_pitch;
        if constexpr (FX & FX_SCALE) pscale = p.pair_scale[pr];
        // opaque from here (never rematerialised next to a load that reads them), and five wait states between whatever wrote
        // these SGPRs and the first vector-memory instruction that takes one as its base (cdna_hip_programming.md 5.7 item 2)
        asm volatile("s_nop 4" : "+s"(qw_base), "+s"(qz_base), "+s"(sc_base), "+s"(x_base));
    }
    const int gi = part * p.rg + rgi;
    const int t0 = (gi * p.su_base + min(gi, p.su_rem)) * p.su_gran;  // first SU of this wave's row group
    const int nt = (p.su_base + (gi < p.su_rem ? 1 : 0)) * p.su_gran;
    const int last_row = p.N - 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;  // LDS byte address of the dynamic segment
    const int sc_off = p.x_bytes + (rgi * p.wk + wki) * (p.sc_pitch + p.z_pitch), z_off = sc_off + p.sc_pitch;
    // activation planes have a pitch of Cp + 1 chunks: chunk Cp of every plane is ZERO, and a lane without a chunk of its own
    // (ragged rows) reads that one -- no per-element select in the prologue
    const int Cq = p.Cp + 1;
    constexpr int PLANES = 4 * (MM + ((FX & FX_NORM) ? 1 : 0));
    if (threadIdx.y == 0 && threadIdx.x < PLANES) *reinterpret_cast<u32x4*>(smem + (size_t)((threadIdx.x * Cq + p.Cp) * 16)) = u32x4{0u, 0u, 0u, 0u};

    // ---- 1. activations first (they must be in registers when the first weights land): this wave's share of the block's
    //         LDS-DMA copy, piece j of 64 chunks per instruction
    if constexpr (!(AWQ_ROWS_DBG & 2)) {
        if (p.rg == 4 * MM) {  // the usual shape (4 waves, batch 1): every wave moves exactly ONE piece per slot -- no loop, no
#pragma unroll                 // scalar bookkeeping ahead of the first requests
            for (int s = 0; s < SL; ++s) {
                const int cb = s * p.wk + wki;
                const int c = min(cb * 64 + lane, p.C - 1);
                const int j = rgi & 3, m = rgi >> 2;
                AWQ_ROWS_DMA16((uint32_t)((m * p.K + 32 * c + 8 * j) * 2), x_base, lds0 + (uint32_t)(((m * 4 + j) * Cq + cb * 64) * 16));
            }
        } else
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            const int cb = s * p.wk + wki;
            const int c = min(cb * 64 + lane, p.C - 1);
            for (int jm = rgi; jm < 4 * MM; jm += p.rg) {
                const int j = jm & 3, m = jm >> 2;
                const uint32_t src = (uint32_t)((m * p.K + 32 * c + 8 * j) * 2);
                const uint32_t dst = lds0 + (uint32_t)(((m * 4 + j) * Cq + cb * 64) * 16);
                AWQ_ROWS_DMA16(src, x_base, dst);
            }
        }
    }
    if constexpr (FX & FX_NORM) {  // the norm weights travel like one more batch row (plane MM of xs)
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            const int cb = s * p.wk + wki;
            const int c = min(cb * 64 + lane, p.C - 1);
            for (int j = rgi; j < 4; j += p.rg)
                AWQ_ROWS_DMA16((uint32_t)((32 * c + 8 * j) * 2), p.norm_w, lds0 + (uint32_t)(((MM * 4 + j) * Cq + cb * 64) * 16));
        }
    }
    // ---- 2. scales and zero words of every row this wave will touch: contiguous in this layout, copied as they are
    // (When a wave's share is at most 1 KiB of scales and 64 zero words -- every 7B / 70B decode shape -- it travels through two
    // REGISTERS instead: one 16-byte and one 4-byte load per lane, written to the wave's LDS region once they have landed.)
    u32x4 scv;    // (deliberately not initialised: a zeroing move on the other path would sit between request and wait)
    uint32_t zv;
    auto issue_scales = [&]() __attribute__((always_inline)) {
        if constexpr (!(AWQ_ROWS_DBG & 16)) {
            const int rows_w = max(min((t0 + nt) * RPU, p.N) - t0 * RPU, 0);
            const int sc_bytes = rows_w * p.SW * 2, zd = rows_w * p.ZW;
            const uint32_t sc_src = (uint32_t)(t0 * RPU * p.SW * 2), z_src = (uint32_t)(t0 * RPU * p.ZW * 4);  // byte offsets
            if (p.sc_regs) {
                const uint32_t so = sc_bytes > 0 ? sc_src + (uint32_t)min(16 * lane, sc_bytes - 16) : 0u;
                const uint32_t zo = zd > 0 ? z_src + 4u * (uint32_t)min(lane, zd - 1) : 0u;
                asm volatile("global_load_dwordx4 %0, %2, %4\n\tglobal_load_dword %1, %3, %5"
                             : "=&v"(scv), "=&v"(zv) : "v"(so), "v"(zo), "s"(sc_base), "s"(qz_base) : "memory");
            } else {
                for (int o = 0; o < sc_bytes; o += 1024)
                    AWQ_ROWS_DMA16(sc_src + (uint32_t)min(o + 16 * lane, sc_bytes - 16), sc_base, lds0 + (uint32_t)(sc_off + o));
                for (int o = 0; o < zd; o += 64) AWQ_ROWS_DMA4(z_src + 4u * (uint32_t)min(o + lane, zd - 1), qz_base, lds0 + (uint32_t)(z_off + 4 * o));
            }
        }
    };
    if constexpr (!(AWQ_ROWS_DBG & 128)) issue_scales();

    // FX_RES: lane e adds the residual of row e of this wave (at most 64 rows, the launcher checks); requested here, ahead of
    // the ring, so that it has landed long before the fold (a load issued at the fold would expose its whole round trip)
    uint32_t resv = 0;
    if constexpr (FX & FX_RES) {
        const int nrows = max(min((t0 + nt) * RPU, p.N) - t0 * RPU, 0);
        const int row = min(t0 * RPU + min(lane, max(nrows - 1, 0)), last_row);
        asm volatile("global_load_ushort %0, %1, %2" : "=&v"(resv) : "v"((uint32_t)(row * 2)), "s"(p.res) : "memory");
    }
    // ---- 3. this lane's chunk of a row per slot, then the ring: D super-units of R rounds each
    uint32_t woff[SL];
    int cidx[SL];
    bool act[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int ws = s * p.wk + wki;
        const int l0 = ws * p.lines_base + min(ws, p.lines_rem);
        const int nl = p.lines_base + (ws < p.lines_rem ? 1 : 0);
        const int c = 8 * l0 + lane;
        act[s] = lane < 8 * nl && c < p.C;
        cidx[s] = act[s] ? c : p.Cp;  // the zero chunk
        woff[s] = nl > 0 && act[s] ? 16u * (uint32_t)c : 0u;  // bytes; a lane or slot without weights (padding) reads the row's first 16 bytes
    }
    Round ring[D][R];
    auto request = [&](Round& Rd, int t, int r) {  // round r of SU t of this row group
        const bool live = t < nt;  // past the last SU the request is kept (the counted waits need it) but reads 16 bytes
        const int row0 = min((t0 + t) * RPU, last_row);
        const int ra = min(row0 + (4 * r + 0) / SL, last_row), rb = min(row0 + (4 * r + 1) / SL, last_row);
        const int rc = min(row0 + (4 * r + 2) / SL, last_row), rd = min(row0 + (4 * r + 3) / SL, last_row);
        const int rowb = p.KW * 4;  // bytes per row
        const uint32_t wa = (uint32_t)(ra * rowb) + (live ? woff[(4 * r + 0) % SL] : 0u), wb = (uint32_t)(rb * rowb) + (live ? woff[(4 * r + 1) % SL] : 0u);
        const uint32_t wc = (uint32_t)(rc * rowb) + (live ? woff[(4 * r + 2) % SL] : 0u), wd = (uint32_t)(rd * rowb) + (live ? woff[(4 * r + 3) % SL] : 0u);
        if constexpr (FX & FX_GROUPED) {
            AWQ_ROWS_REQUEST_KEEP(Rd, wa, wb, wc, wd, qw_base);
        } else {
            AWQ_ROWS_REQUEST(Rd, wa, wb, wc, wd, qw_base);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            request(ring[d][r], d, r);
            // (experiment 128: scales / zeros right behind the FIRST round -- still older than the R D - 1 rounds every wait
            // leaves in flight, so they have landed when the first round is consumed)
            if constexpr (AWQ_ROWS_DBG & 128)
                if (d == 0 && r == 0) issue_scales();
        }
    ROWS_STAMP(1);

    // ---- 4. off the critical path: lane j of a quad finishes unit 4 r + j of every SU in round r -- that unit's row within
    //         the SU, slot and group, and where its scale / zero word sit in LDS
    const int uj = lane & 3;
    const QuadSel sel4{uj == 0 ? ~0u : 0u, uj == 1 ? ~0u : 0u, uj == 2 ? ~0u : 0u};
    int grp[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) grp[s] = (int)__umulhi((uint32_t)(32 * min(cidx[s], p.C - 1)), p.g_magic);
    int myslot[R], myrow[R], sc_at[R], z_at[R];
    uint32_t zsh[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        myslot[r] = sel4((4 * r + 0) % SL, (4 * r + 1) % SL, (4 * r + 2) % SL, (4 * r + 3) % SL);
        myrow[r] = sel4((4 * r + 0) / SL, (4 * r + 1) / SL, (4 * r + 2) / SL, (4 * r + 3) / SL);
        const int g = sel4(grp[(4 * r + 0) % SL], grp[(4 * r + 1) % SL], grp[(4 * r + 2) % SL], grp[(4 * r + 3) % SL]);
        zsh[r] = 4u * (uint32_t)(g & 7);
        sc_at[r] = sc_off + (myrow[r] * p.SW + g) * 2;          // + t * RPU * SW * 2 per SU
        z_at[r] = z_off + (myrow[r] * p.ZW + (g >> 3)) * 4;     // + t * RPU * ZW * 4 per SU
    }
    const int sc_step = RPU * p.SW * 2, z_step = RPU * p.ZW * 4;

    // ---- 5. the DMA pieces have landed (they are older than the ring); the activations are shared by the block
    asm volatile("s_waitcnt vmcnt(%3) ; releases %0 %1 %2" : "+v"(resv), "+v"(scv), "+v"(zv) : "n"(AWQ_ROWS_LPR * R * D) : "memory");
    if constexpr (!(AWQ_ROWS_DBG & 16)) {
        if (p.sc_regs) {  // the wave's own LDS region: its later reads follow these writes in order, no barrier needed
            *reinterpret_cast<u32x4*>(smem + sc_off + 16 * lane) = scv;
            *reinterpret_cast<uint32_t*>(smem + z_off + 4 * lane) = zv;
        }
    }
    if constexpr (!(AWQ_ROWS_DBG & 2)) __builtin_amdgcn_s_barrier();
    ROWS_STAMP(2);

    // ---- activations LDS -> registers: (t, t+4) pairs; group constants C0 = sum bias * x, SX = sum x
    const int NC = p.wk * SL;
    const int rows_blk = p.su_max * p.rg * RPU;
    float* red = reinterpret_cast<float*>(smem + (size_t)p.x_bytes + (size_t)p.rg * p.wk * (p.sc_pitch + p.z_pitch));
    uint32_t xp[MM][SL][16];
    float c0g[MM][R], sxg[MM][R];
    float inv = 1.f;
    const u32x2 sum_rows = {(lane & 3) == 0 ? 0x3C003C00u : ((lane & 3) == 1 ? 0x64006400u : 0u),
                            (lane & 3) == 0 ? 0x3C003C00u : ((lane & 3) == 1 ? 0x54005400u : 0u)};
    constexpr int SB = SL <= 3 ? SL : (SL == 8 ? 1 : 2);  // slots per LDS -> register batch: 16 SB registers in flight (SL = 8 has none to spare)
    // FX_NORM with the whole row in ONE batch (SL <= 3: K <= 6144): the row statistic is taken from the batch's own registers, and
    // the norm weights are read in the same batch -- no second pass over x in LDS (round 6: 7.38 -> 7.05 us on qkv with the DPP sum
    // and the mixed-precision FMAs, profiles/r06_fx_overhead.txt)
    constexpr bool NORM_INLINE = (FX & FX_NORM) && SB == SL;
    auto wave_sum = [&](float ss) __attribute__((always_inline)) {
        // without the LDS crossbar (six dependent ds_bpermute round trips on the launch's critical path): DPP inside the 16-lane
        // rows, then the four row sums through scalar registers
        ss += dpp_mov<0xB1>(ss);   // quad_perm [1,0,3,2]
        ss += dpp_mov<0x4E>(ss);   // quad_perm [2,3,0,1]
        ss += dpp_mov<0x124>(ss);  // row_ror:4
        ss += dpp_mov<0x128>(ss);  // row_ror:8
        const int si = __builtin_bit_cast(int, ss);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(si, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(si, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(si, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(si, 48));
        return (r0 + r1) + (r2 + r3);
    };
    if constexpr ((FX & FX_NORM) && !NORM_INLINE) {  // the row statistic: a wave covers the whole row (wk == 1), inactive lanes add nothing
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < SL; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 d = *reinterpret_cast<const u32x4*>(smem + (size_t)((j * Cq + cidx[s]) * 16));
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) q = dot2(d[i], d[i], q);
                ss += q;  // (a lane without a chunk reads the zero chunk)
            }
        ss = wave_sum(ss);
        inv = rsqrtf(ss / (float)p.K + p.norm_eps);
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        float c0s[SL], sxs[SL];
        // The LDS reads are inline asm, a BATCH of slots at a time behind ONE wait (left to the compiler, the reads of a slot
        // went out two at a time with a wait after each pair: 0.48 us for 8 reads at K = 4096, 1.28 us at K = 11008, on the
        // critical path of the launch -- profiles/r03_gemv_rows_trace.txt); the C0 / SX sums run as four independent chains.
        static_assert(SL % SB == 0, "batches cover the slots");
#pragma unroll
        for (int s0 = 0; s0 < SL; s0 += SB) {
            u32x4 dj[SB][4];
#pragma unroll
            for (int sb = 0; sb < SB; ++sb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dj[sb][j] = u32x4{0x3C003C00u + lane, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
                    if constexpr (!(AWQ_ROWS_DBG & 2))
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(dj[sb][j]) : "v"(lds0 + (uint32_t)(((m * 4 + j) * Cq + cidx[s0 + sb]) * 16)));
                }
            u32x4 wj[NORM_INLINE ? SB : 1][4];
            if constexpr (NORM_INLINE) {
#pragma unroll
                for (int sb = 0; sb < SB; ++sb)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(wj[sb][j]) : "v"(lds0 + (uint32_t)(((MM * 4 + j) * Cq + cidx[s0 + sb]) * 16)));
            }
            if constexpr (!(AWQ_ROWS_DBG & 2)) {
#pragma unroll
                for (int sb = 0; sb < SB; ++sb)  // (LDS operations return in order: the first wait covers every read of the batch)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dj[sb][0]), "+v"(dj[sb][1]), "+v"(dj[sb][2]), "+v"(dj[sb][3]));
            }
            if constexpr (NORM_INLINE) {
#pragma unroll
                for (int sb = 0; sb < SB; ++sb)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wj[sb][0]), "+v"(wj[sb][1]), "+v"(wj[sb][2]), "+v"(wj[sb][3]));
                float ss = 0.f;  // the row statistic, summed in the order of the separate pass (slot, piece, pair)
#pragma unroll
                for (int sb = 0; sb < SB; ++sb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float q = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) q = dot2(dj[sb][j][i], dj[sb][j][i], q);
                        ss += q;  // (a lane without a chunk read the zero chunk)
                    }
                inv = rsqrtf(wave_sum(ss) / (float)p.K + p.norm_eps);
            }
#pragma unroll
            for (int sb = 0; sb < SB; ++sb) {
                const int s = s0 + sb;
                float4_t sums = {0.f, 0.f, 0.f, 0.f};  // [0] = sum x, [1] = sum bias * x of this lane's 32 activations
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 d = dj[sb][j];
                    if constexpr (NORM_INLINE) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) d[i] = norm_pair(d[i], wj[sb][j][i], inv);  // == awq_rmsnorm_kernel
                    } else if constexpr (FX & FX_NORM) {
                        const u32x4 wv = *reinterpret_cast<const u32x4*>(smem + (size_t)(((MM * 4 + j) * Cq + cidx[s]) * 16));
#pragma unroll
                        for (int i = 0; i < 4; ++i) d[i] = norm_pair(d[i], wv[i], inv);  // == awq_rmsnorm_kernel
                    }
                    xp[m][s][4 * j + 0] = __builtin_amdgcn_perm(d[2], d[0], 0x05040100u);  // (x0, x4)  bias 1024
                    xp[m][s][4 * j + 1] = __builtin_amdgcn_perm(d[2], d[0], 0x07060302u);  // (x1, x5)  bias 64
                    xp[m][s][4 * j + 2] = __builtin_amdgcn_perm(d[3], d[1], 0x05040100u);  // (x2, x6)  bias 1024
                    xp[m][s][4 * j + 3] = __builtin_amdgcn_perm(d[3], d[1], 0x07060302u);  // (x3, x7)  bias 64
                    // the group constants off the VALU: a 4x4x4 MFMA whose A rows are (1, 1, 1, 1) [lane 0 of the quad] and
                    // (1024, 1024, 64, 64) [lane 1] leaves sum x in register 0 and sum bias * x in register 1 of every lane
                    sums = mfma4(sum_rows, u32x2{xp[m][s][4 * j + 0], xp[m][s][4 * j + 1]}, sums);
                    sums = mfma4(sum_rows, u32x2{xp[m][s][4 * j + 2], xp[m][s][4 * j + 3]}, sums);
                }
                float c0 = sums[1], sx = sums[0];
                c0 += dpp_mov<0xB1>(c0);  // quad_perm [1,0,3,2]
                sx += dpp_mov<0xB1>(sx);
                c0 += dpp_mov<0x4E>(c0);  // quad_perm [2,3,0,1]
                sx += dpp_mov<0x4E>(sx);
                c0s[s] = c0;
                sxs[s] = sx;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c0g[m][r] = sel4(c0s[(4 * r + 0) % SL], c0s[(4 * r + 1) % SL], c0s[(4 * r + 2) % SL], c0s[(4 * r + 3) % SL]);
            sxg[m][r] = sel4(sxs[(4 * r + 0) % SL], sxs[(4 * r + 1) % SL], sxs[(4 * r + 2) % SL], sxs[(4 * r + 3) % SL]);
        }
    }
    const bool odd = lane & 1, hi = lane & 2, writer = (lane & 12) == 12;
    // LDS index of this lane's unit in round r of SU 0 of its row group (advances by RPU rows per SU)
    int red_lane[R];
#pragma unroll
    for (int r = 0; r < R; ++r) red_lane[r] = (((rgi * p.su_max) * RPU + myrow[r]) * NC + wki * SL + myslot[r]) * 4 + (lane >> 4);
    const int red_step = RPU * NC * 4;
    ROWS_STAMP(3);

    // ---- one round: wait until at most NEWER later operations are outstanding (= this round has landed), consume it, and --
    // in the main phase -- request the same round of super-unit t + D into the registers just freed
    auto do_round = [&](auto newer_c, auto rerequest_c, Round& Rd, const int t, const int r) __attribute__((always_inline)) {
        constexpr int NEWER = decltype(newer_c)::value;
        constexpr bool REREQUEST = decltype(rerequest_c)::value;
        {
            {
                AWQ_ROWS_WAIT(Rd, NEWER);
#ifdef AWQ_GEMV_TRACE
                if (t == 0 && r == 0) ROWS_STAMP(4);
#endif
                // this round's scale and zero word are requested from LDS NOW (left to the compiler the two reads sat behind the
                // last MFMA with a wait right after them: a full LDS round trip per round, 0.4-0.8 us per launch)
                uint32_t scl_raw = 0x3C00u, z_raw = 0u;
                if constexpr (!(AWQ_ROWS_DBG & 16))
                    asm volatile("ds_read_u16 %0, %2\n\tds_read_b32 %1, %3"
                                 : "=&v"(scl_raw), "=&v"(z_raw)
                                 : "v"(lds0 + (uint32_t)(sc_at[r] + t * sc_step)), "v"(lds0 + (uint32_t)(z_at[r] + t * z_step))
                                 : "memory");
                float pu[MM][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = (4 * r + u) % SL;
                    float pa[MM], pb[MM];
#pragma unroll
                    for (int m = 0; m < MM; ++m) pa[m] = pb[m] = 0.f;
                    if constexpr (AWQ_ROWS_DBG & 1) {
#pragma unroll
                        for (int m = 0; m < MM; ++m) pa[m] = __builtin_bit_cast(float, Rd.q[u][0] ^ Rd.q[u][1] ^ Rd.q[u][2] ^ Rd.q[u][3]);
                    } else if constexpr (!(AWQ_ROWS_DBG & 64)) {
                        // v_mfma_f32_4x4x4_16b_f16: 16 independent 4 x 4 x 4 products, one per lane quad.  A row i = lane i of the
                        // quad (4 weights), B column j = lane j (4 activations): D[i][j] = w(lane i) . x(lane j), register i of
                        // lane j.  The lane's own dot product is the diagonal element, register (lane & 3); the other twelve
                        // products of the quad are discarded.  Eight MFMAs per unit replace sixteen VALU dot products.
                        float4_t acc[MM];
#pragma unroll
                        for (int m = 0; m < MM; ++m) acc[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t w = Rd.q[u][j], w8 = w >> 8;
                            const u32x2 a0 = {and_or(w, 0x000F000Fu, 0x64006400u), and_or(w, 0x00F000F0u, 0x54005400u)};
                            const u32x2 a1 = {and_or(w8, 0x000F000Fu, 0x64006400u), and_or(w8, 0x00F000F0u, 0x54005400u)};
#pragma unroll
                            for (int m = 0; m < MM; ++m) {
                                acc[m] = mfma4(a0, u32x2{xp[m][s][4 * j + 0], xp[m][s][4 * j + 1]}, acc[m]);
                                acc[m] = mfma4(a1, u32x2{xp[m][s][4 * j + 2], xp[m][s][4 * j + 3]}, acc[m]);
                            }
                        }
#pragma unroll
                        for (int m = 0; m < MM; ++m) pa[m] = sel4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t w = Rd.q[u][j], w8 = w >> 8;
                            const uint32_t b0 = and_or(w, 0x000F000Fu, 0x64006400u), b1 = and_or(w, 0x00F000F0u, 0x54005400u);
                            const uint32_t b2 = and_or(w8, 0x000F000Fu, 0x64006400u), b3 = and_or(w8, 0x00F000F0u, 0x54005400u);
#pragma unroll
                            for (int m = 0; m < MM; ++m) {
                                pa[m] = dot2(xp[m][s][4 * j + 0], b0, pa[m]);
                                pb[m] = dot2(xp[m][s][4 * j + 1], b1, pb[m]);
                                pa[m] = dot2(xp[m][s][4 * j + 2], b2, pa[m]);
                                pb[m] = dot2(xp[m][s][4 * j + 3], b3, pb[m]);
                            }
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MM; ++m) pu[m][u] = pa[m] + pb[m];
                }
                if constexpr (!(AWQ_ROWS_DBG & 16)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(scl_raw), "+v"(z_raw));
                const float scl = (float)__builtin_bit_cast(half_t, (unsigned short)scl_raw);
                const float zf = (float)((z_raw >> zsh[r]) & 15u);
                // every value derived from the round exists before the round is requested again: the old contents are dead
                // at the request, so the round keeps its registers (no copy of a register with a load in flight)
#pragma unroll
                for (int m = 0; m < MM; ++m) asm volatile("" ::"v"(pu[m][0]), "v"(pu[m][1]), "v"(pu[m][2]), "v"(pu[m][3]));
                if constexpr (REREQUEST) request(Rd, t + D, r);
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    float v;
                    if constexpr (AWQ_ROWS_DBG & 4) {
                        v = (pu[m][0] + pu[m][1]) + (pu[m][2] + pu[m][3]) + scl + zf;
                    } else {
                        // 4 x 4 transpose-reduce inside the quad: lane j ends with the quad (= group) sum of unit j
                        const float s01 = odd ? pu[m][0] : pu[m][1], k01 = odd ? pu[m][1] : pu[m][0];
                        const float s23 = odd ? pu[m][2] : pu[m][3], k23 = odd ? pu[m][3] : pu[m][2];
                        const float r01 = k01 + dpp_mov<0xB1>(s01), r23 = k23 + dpp_mov<0xB1>(s23);
                        const float snd = hi ? r01 : r23, kp = hi ? r23 : r01;
                        const float rq = kp + dpp_mov<0x4E>(snd);
                        v = scl * __builtin_fmaf(-zf, sxg[m][r], rq - c0g[m][r]);
                        v += dpp_mov<0x124>(v);  // row_ror:4
                        v += dpp_mov<0x128>(v);  // row_ror:8  -> sum over the four quads of this 16-lane row
                    }
                    if (writer) red[(m * rows_blk * NC) * 4 + red_lane[r] + t * red_step] = v;
                }
            }
        }
    };
    // ---- stream.  Main phase: super-units [0, nt - D), every consumed round re-requested D super-units ahead (always a live
    // one), so exactly the R D - 1 other rounds of the ring are in flight behind the one being waited for.  Drain phase: the
    // last D super-units, oldest first, no requests, the allowed count falling by one round each time.  (The first version kept
    // requesting 16-byte dummies past the end to keep ONE wait count: every wave then ended on a full memory round trip.)
    // The launcher gives every row group a multiple of D super-units (D = 2 only when they divide evenly), so the ring position
    // of every super-unit is static: ONE straight-line drain, no alternative register mappings for the compiler to reconcile
    // with copies of registers whose loads are still in flight (it did exactly that for a two-variant drain; tools/isa_audit.py
    // audit_inflight_regs, run by tests/test_boundary.py, reads the ISA back).
    const int n_main = nt - D;  // nt == 0 (a row group without work) skips both phases
    for (int tbase = 0; tbase < n_main; tbase += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                do_round(std::integral_constant<int, AWQ_ROWS_LPR * (R * D - 1)>{}, std::true_type{}, ring[d][r], tbase + d, r);
        }
    }
    if (nt > 0) {
        static_for<0, R * D>([&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value, dpos = i / R, r = i % R;
            do_round(std::integral_constant<int, AWQ_ROWS_LPR * (R * D - 1 - i)>{}, std::false_type{}, ring[dpos][r], n_main + dpos, r);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the 16-byte dummies of a row group without work; nothing is outstanding otherwise
    ROWS_STAMP(5);
    if constexpr (!(AWQ_ROWS_DBG & 8)) {
        // ---- add the 4 * wk * SL partials of every row, write y.  wk == 1: a wave folds the rows it produced itself
        // (its own LDS writes, in order: no barrier); otherwise the block folds all of its rows behind one barrier.
        if (p.wk == 1) {
            ROWS_STAMP(6);
            const int nrows = min((t0 + nt) * RPU, p.N) - t0 * RPU;
            auto row_sum = [&](int m, int j) {
                const float4_t* rp = reinterpret_cast<const float4_t*>(red + ((size_t)(m * rows_blk + rgi * p.su_max * RPU + j) * NC) * 4);
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < SL; ++w) {
                    const float4_t v = rp[w];
                    sum += (v[0] + v[1]) + (v[2] + v[3]);
                }
                return sum;
            };
            if constexpr (FX & FX_PAIRS) {  // rows (2 e, 2 e + 1) = (gate, up) of output e: the unfused path's two roundings
                for (int e = lane; e < (nrows >> 1); e += 64) {
                    const float gt = (float)(half_t)row_sum(0, 2 * e), up = (float)(half_t)row_sum(0, 2 * e + 1);
                    y_base[((t0 * RPU) >> 1) + e] = (half_t)awq_silu_mul_f32(gt, up);  // == awq_silu_and_mul_kernel
                }
            } else if constexpr (FX & FX_RES) {
                if (lane < nrows) {
                    const half_t r = __builtin_bit_cast(half_t, (unsigned short)resv);
                    y_base[t0 * RPU + lane] = (half_t)((float)(half_t)row_sum(0, lane) + (float)r);
                }
            } else {
                for (int e = lane; e < nrows * MM; e += 64) {
                    const int m = MM == 1 ? 0 : e / nrows, j = MM == 1 ? e : e - m * nrows;
                    y_base[(int64_t)m * p.N + t0 * RPU + j] = (FX & FX_SCALE) ? (half_t)(row_sum(m, j) * pscale) : (half_t)row_sum(m, j);
                }
            }
        } else {
            __syncthreads();
            ROWS_STAMP(6);
            const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthr = blockDim.x * blockDim.y;
            for (int rgj = 0; rgj < p.rg; ++rgj) {
                const int gj = part * p.rg + rgj;
                const int tj0 = (gj * p.su_base + min(gj, p.su_rem)) * p.su_gran, ntj = (p.su_base + (gj < p.su_rem ? 1 : 0)) * p.su_gran;
                const int nr = min((tj0 + ntj) * RPU, p.N) - tj0 * RPU;
                for (int e = tid; e < nr * MM; e += nthr) {
                    const int m = MM == 1 ? 0 : e / nr, j = MM == 1 ? e : e - m * nr;
                    const float4_t* rp = reinterpret_cast<const float4_t*>(red + ((size_t)(m * rows_blk + rgj * p.su_max * RPU + j) * NC) * 4);
                    float sum = 0.f;
                    for (int w = 0; w < NC; ++w) {
                        const float4_t v = rp[w];
                        sum += (v[0] + v[1]) + (v[2] + v[3]);
                    }
                    y_base[(int64_t)m * p.N + tj0 * RPU + j] = (half_t)sum;
                }
            }
        }
    }
#ifdef AWQ_GEMV_TRACE
    ROWS_STAMP(7);
    if (p.trace && lane == 0) {
        unsigned long long* o = p.trace + ((size_t)blockIdx.x * 8 + (threadIdx.y * (blockDim.x >> 6) + wki)) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
#endif
}

template <int SL, int D, int MM, int FX = 0>
int launch_rows(const RowsParams& p, int blocks, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> opted{0};
        (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_rows_kernel<SL, D, MM, FX>), opted);
    }
    const dim3 grid = (FX & FX_GROUPED) ? dim3(8u * (unsigned)p.y_rows, (unsigned)(blocks + 7) / 8u) : dim3((unsigned)blocks);
    hipLaunchKernelGGL((awq_gemv_rows_kernel<SL, D, MM, FX>), grid, dim3(64 * p.wk, p.rg), lds, st, p);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// the decoder-block variants (batch 1): every (SL, D) of the plain kernel x {norm, residual, norm + residual, silu pairs, norm + silu pairs}
template <int SL, int D>
int launch_rows_fx(int fx, const RowsParams& p, int blocks, size_t lds, hipStream_t st) {
    if constexpr (SL == 8) {  // K > 12288: 128 activation registers leave no room for the norm's temporaries (it would spill)
        if (fx & FX_NORM) return AWQ_ERR_UNSUPPORTED;
    }
    switch (fx) {
        case FX_NORM:
            if constexpr (SL != 8) return launch_rows<SL, D, 1, FX_NORM>(p, blocks, lds, st);
            return AWQ_ERR_UNSUPPORTED;
        case FX_RES: return launch_rows<SL, D, 1, FX_RES>(p, blocks, lds, st);
        case FX_NORM | FX_RES:
            if constexpr (SL != 8) return launch_rows<SL, D, 1, FX_NORM | FX_RES>(p, blocks, lds, st);
            return AWQ_ERR_UNSUPPORTED;
        case FX_PAIRS: return launch_rows<SL, D, 1, FX_PAIRS>(p, blocks, lds, st);
        case FX_NORM | FX_PAIRS:
            if constexpr (SL != 8) return launch_rows<SL, D, 1, FX_NORM | FX_PAIRS>(p, blocks, lds, st);
            return AWQ_ERR_UNSUPPORTED;
        case FX_GROUPED: return launch_rows<SL, D, 1, FX_GROUPED>(p, blocks, lds, st);
        case FX_GROUPED | FX_PAIRS: return launch_rows<SL, D, 1, FX_GROUPED | FX_PAIRS>(p, blocks, lds, st);
        case FX_GROUPED | FX_SCALE: return launch_rows<SL, D, 1, FX_GROUPED | FX_SCALE>(p, blocks, lds, st);
        default: return AWQ_ERR_UNSUPPORTED;
    }
}

// Slots per wave for a row of `slots` 1-KiB units at batch M: the smallest legal SL (1, 2, 3, 4, 6, 8) that covers the whole
// row if the activations fit the register budget (16 SL M registers: SL <= 8 at M = 1, 3 at 2, 2 at 3, 1 at 4), else the
// largest that fits (wk waves then share a row).
int pick_sl(int slots, int M, int forced) {
    static const int legal[] = {1, 2, 3, 4, 6, 8};
    if (forced > 0) {
        for (int s : legal)
            if (s == forced) return s;
        return 0;
    }
    const int cap = M == 1 ? 8 : (6 / M < 1 ? 1 : 6 / M);
    int best = 1;
    for (int s : legal) {
        if (s > cap) break;
        best = s;
        if (s >= slots) break;
    }
    return best;
}

}  // namespace

bool awq_gemv_rows_supports(int M, int K, int N, int g) {
    if (M < 1 || M > 4 || N < 1 || K < 128) return false;
    if (g < 128 || g % 128 || K % g) return false;  // a lane quad (4 x 32 weights) never straddles a group
    const int slots = (K / 32 + 63) / 64;
    const int SL = pick_sl(slots, M, 0);
    return (slots + SL - 1) / SL <= 8;  // wk waves side by side, at most 8
}

#ifdef AWQ_GEMV_TRACE
static unsigned long long* g_rows_trace = nullptr;
extern "C" __attribute__((visibility("default"))) void awq_debug_set_trace_rows(void* dev_buf) {
    g_rows_trace = static_cast<unsigned long long*>(dev_buf);
}
#endif

// waves: waves per block wanted (0 = auto, <= 8); depth: super-units in flight per wave (1 | 2, 0 = auto);
// bpc: blocks per CU (0 = auto); sl: slots per wave (0 = auto).
// fx (decoder-block prologue / epilogue, batch 1 with whole rows per wave, else AWQ_ERR_UNSUPPORTED and the caller runs the
// separate launches): norm_w != null -> x is RMS-normalised while it is brought into registers; res != null -> y = fp16(fp16(W x)
// + res); pairs -> rows (2 i, 2 i + 1) are (gate_i, up_i) and y [N / 2] = silu(gate) * up.
int awq_launch_gemv_rows(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                         uint16_t* y, int M, int K, int N, int g, int ZW, int waves, int depth, int bpc, int sl, hipStream_t st,
                         const AwqRowsFx* fxa) {
    if (!awq_gemv_rows_supports(M, K, N, g)) return AWQ_ERR_UNSUPPORTED;
    const int fx = fxa ? (fxa->norm_w ? FX_NORM : 0) | (fxa->res ? FX_RES : 0) | (fxa->pairs ? FX_PAIRS : 0) |
                             (fxa->pair_expert ? FX_GROUPED : 0) | (fxa->pair_scale ? FX_SCALE : 0) : 0;
    if (fx && (M != 1 || ((fx & FX_PAIRS) && ((fx & FX_RES) || N % 2)))) return AWQ_ERR_UNSUPPORTED;
    if ((fx & FX_SCALE) && !(fx & FX_GROUPED)) return AWQ_ERR_UNSUPPORTED;
    if ((fx & FX_GROUPED) && ((fx & (FX_NORM | FX_RES)) || ((fx & FX_PAIRS) && (fx & FX_SCALE)) || fxa->num_pairs < 1 || fxa->num_pairs > 8191 ||
                              fxa->num_experts < 1 || fxa->x_div < 1))
        return AWQ_ERR_UNSUPPORTED;
    if ((int64_t)N * K / 2 >= ((int64_t)1 << 32) || (int64_t)N * ZW * 16 >= ((int64_t)1 << 32) || (int64_t)M * K * 2 >= ((int64_t)1 << 31))
        return AWQ_ERR_UNSUPPORTED;  // 32-bit byte offsets
    RowsParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N;
    p.KW = K / 8; p.ZW = ZW; p.SW = 8 * ZW;
    p.C = K / 32;
    const int slots = (p.C + 63) / 64;
    const int SL = pick_sl(slots, M, sl);
    if (SL == 0 || 16 * SL * M > 128) return AWQ_ERR_UNSUPPORTED;
    p.wk = (slots + SL - 1) / SL;
    if (p.wk > 8 || (fx && p.wk != 1)) return AWQ_ERR_UNSUPPORTED;
    p.Cp = p.wk * SL * 64;
    // Defaults from the sweeps (profiles/r03_gemv_rows_sweep.txt): four waves per block and one block per CU start fastest
    // (1024 waves: the dispatch ramp and the per-block copy of x are what a short launch pays for); long streams take two.
    if (waves <= 0 || waves > 8) waves = p.wk > 4 ? 8 : 4;
    p.rg = 1;
    while (p.rg * 2 * p.wk <= waves) p.rg *= 2;
    const int lines = (p.C + 7) / 8, tslots = p.wk * SL;
    p.lines_base = lines / tslots;
    p.lines_rem = lines % tslots;
    if (!awq_magic_u32((uint32_t)g, (uint32_t)K + 32u, &p.g_magic)) return AWQ_ERR_UNSUPPORTED;
    const int RPU = rows_per_su(SL);
    p.su_total = (N + RPU - 1) / RPU;
    p.su_gran = (fx & FX_PAIRS) && RPU == 1 ? 2 : 1;  // a (gate, up) row pair stays with one wave
    const int su_units = p.su_total / p.su_gran;      // N is even with pairs
    const bool bpc_auto = bpc <= 0;
    if (bpc_auto) bpc = p.su_total > 8 * 256 * p.rg ? 2 : 1;
    if ((fx & FX_GROUPED) && bpc_auto) bpc = 1;  // the pairs multiply the blocks in flight: parts = 256 per matrix, or `parts` asked for
    const bool one_round = SL * RPU == 4;
    int blocks = 0;
    auto partition = [&](int per_cu) {
        blocks = 256 * per_cu;
        const int max_blocks = (su_units + p.rg - 1) / p.rg;  // at least one SU per row group
        if (blocks > max_blocks) blocks = max_blocks;
        const int groups = blocks * p.rg;
        p.su_base = su_units / groups;
        p.su_rem = su_units % groups;
        p.su_max = (p.su_base + (p.su_rem ? 1 : 0)) * p.su_gran;
    };
    partition(bpc);
    if (fx & FX_GROUPED) {
        // parts per matrix: the pairs multiply the blocks in flight -- as many blocks as the chip holds at once (the kernel's
        // register count leaves room for 4 / 3 / 2 blocks per CU at SL <= 2 / <= 4 / above), between 64 and 512 parts, a multiple
        // of 8 (dense XCD map); fewer, longer blocks amortise the prologue (x into registers: 28 KB per block at K = 14336).
        // Mixtral bs = 4 (8 pairs), us per MoE block by (parts w1|w3, parts w2): (64, 64) 125.8, (128, 64) 108.3, (256, 64) 109.6,
        // (512, 64) 110.1, (256, 32) 115.6, (256, 96) 116.3 (profiles/r06_moe_rows.txt).  fxa->parts > 0 forces a count.
        const int occ = SL <= 2 ? 4 : (SL <= 4 ? 3 : 2);
        int want = fxa->parts > 0 ? fxa->parts : (256 * occ / fxa->num_pairs + 7) / 8 * 8;
        if (fxa->parts <= 0) want = want < 64 ? 64 : (want > 512 ? 512 : want);
        blocks = want;
        const int max_blocks = (su_units + p.rg - 1) / p.rg;
        if (blocks > max_blocks) blocks = max_blocks;
        const int groups = blocks * p.rg;
        p.su_base = su_units / groups;
        p.su_rem = su_units % groups;
        p.su_max = (p.su_base + (p.su_rem ? 1 : 0)) * p.su_gran;
    }
    // Two super-units in flight per wave need an even count in EVERY row group (one straight-line drain).  Where the rows do
    // not divide that way (4096 x 11008: 5504 super-units over 1024 row groups) the same bytes stay in flight through two
    // blocks per CU with one super-unit each instead (7.0 us; one block per CU and one in flight: 8.0).
    if (!(fx & FX_GROUPED) && bpc_auto && bpc == 1 && (depth < 1 || depth > 2) && one_round && p.su_max >= 2 && p.su_max <= 8 &&
        (p.su_rem != 0 || (p.su_base * p.su_gran) % 2)) {
        bpc = 2;
        partition(bpc);
    }
    if ((fx & FX_RES) && p.su_max * RPU > 64) return AWQ_ERR_UNSUPPORTED;  // one residual element per lane
    p.norm_w = fxa ? reinterpret_cast<const half_t*>(fxa->norm_w) : nullptr;
    p.norm_eps = fxa ? fxa->norm_eps : 0.f;
    p.res = fxa ? reinterpret_cast<const half_t*>(fxa->res) : nullptr;
    p.pair_expert = nullptr; p.pair_scale = nullptr;
    p.E = 0; p.parts = blocks; p.x_div_magic = 0; p.y_pitch = 0; p.y_rows = 0; p.e_first = 0;
    p.w_stride = p.z_stride = p.s_stride = 0;
    if (fx & FX_GROUPED) {
        p.pair_expert = fxa->pair_expert;
        p.pair_scale = fxa->pair_scale;
        p.E = fxa->num_experts;
        p.e_first = fxa->first_expert;
        p.y_rows = fxa->num_pairs;
        p.y_pitch = (fx & FX_PAIRS) ? N / 2 : N;
        if (fxa->x_div > 1 && !awq_magic_u32((uint32_t)fxa->x_div, (uint32_t)fxa->num_pairs + 1u, &p.x_div_magic)) return AWQ_ERR_UNSUPPORTED;
        p.w_stride = (long long)N * p.KW * 4;
        p.z_stride = (long long)N * p.ZW * 4;
        p.s_stride = (long long)N * p.SW * 2;
        if ((int64_t)(fxa->num_pairs / fxa->x_div + 1) * K * 2 >= ((int64_t)1 << 31)) return AWQ_ERR_UNSUPPORTED;
    }
    if (depth < 1 || depth > 2) depth = p.su_max >= 2 && p.su_max <= 8 && bpc == 1 && one_round ? 2 : 1;
    if (!one_round || (SL == 4 && M > 1) || (SL == 2 && M > 3)) depth = 1;  // instantiated combinations (register budget)
    if (p.su_rem != 0 || (p.su_base * p.su_gran) % 2) depth = 1;  // two in flight only when every row group gets an even count
    p.trace = nullptr;
#ifdef AWQ_GEMV_TRACE
    p.trace = g_rows_trace;
#endif
    p.x_bytes = (M + ((fx & FX_NORM) ? 1 : 0)) * 4 * (p.Cp + 1) * 16;  // chunk Cp of every plane is the zero chunk
    p.sc_regs = (AWQ_ROWS_DBG & 256) ? 0 : (p.su_max * RPU * p.SW * 2 <= 1024 && p.su_max * RPU * p.ZW <= 64);
    p.sc_pitch = (p.su_max * RPU * p.SW * 2 + 1023) / 1024 * 1024;
    p.z_pitch = (p.su_max * RPU * p.ZW * 4 + 255) / 256 * 256;
    const size_t lds = (size_t)p.x_bytes + (size_t)p.rg * p.wk * (p.sc_pitch + p.z_pitch) +
                       (size_t)M * p.su_max * p.rg * RPU * p.wk * SL * 4 * sizeof(float);
    if (lds > 160 * 1024) return AWQ_ERR_UNSUPPORTED;
#define AWQ_ROWS_CASE(SLV, DV, MV)                                                        \
    if (SL == SLV && depth == DV && M == MV) {                                            \
        if constexpr (MV == 1) {                                                          \
            if (fx) return launch_rows_fx<SLV, DV>(fx, p, blocks, lds, st);               \
        }                                                                                 \
        return launch_rows<SLV, DV, MV>(p, blocks, lds, st);                              \
    }
    AWQ_ROWS_CASE(1, 1, 1) AWQ_ROWS_CASE(1, 1, 2) AWQ_ROWS_CASE(1, 1, 3) AWQ_ROWS_CASE(1, 1, 4)
    AWQ_ROWS_CASE(1, 2, 1) AWQ_ROWS_CASE(1, 2, 2) AWQ_ROWS_CASE(1, 2, 3) AWQ_ROWS_CASE(1, 2, 4)
    AWQ_ROWS_CASE(2, 1, 1) AWQ_ROWS_CASE(2, 1, 2) AWQ_ROWS_CASE(2, 1, 3) AWQ_ROWS_CASE(2, 1, 4)
    AWQ_ROWS_CASE(2, 2, 1) AWQ_ROWS_CASE(2, 2, 2) AWQ_ROWS_CASE(2, 2, 3)
    AWQ_ROWS_CASE(3, 1, 1) AWQ_ROWS_CASE(3, 1, 2)
    AWQ_ROWS_CASE(4, 1, 1) AWQ_ROWS_CASE(4, 1, 2) AWQ_ROWS_CASE(4, 2, 1)
    AWQ_ROWS_CASE(6, 1, 1)
    AWQ_ROWS_CASE(8, 1, 1)
#undef AWQ_ROWS_CASE
    return AWQ_ERR_UNSUPPORTED;
}
