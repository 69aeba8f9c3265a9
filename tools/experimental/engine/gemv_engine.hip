// gemv_engine.hip -- a whole CHAIN of dependent batch-1 GEMV-layout Linears in ONE persistent launch, gfx950 (round 6).
//
// EXPERIMENT, NOT IN THE PRODUCT LIBRARY ("one decision-grade attempt ... or close it with counters").  Correct
// (every op checked against the dequantised-weights product, bitwise reproducible, hipGraph-replayable, no hang in any run), and
// SLOWER than one launch per Linear: 39.5 us per 7B layer against 27.0 (profiles/r06_engine_probe.txt, which also holds the per-edge
// costs and the switch-off experiments that close the item: with arithmetic AND hand-off waits switched off the skeleton alone takes
// 26.1 us per layer).  Built and run by probe_engine.py in this directory.
//
// The reference runs five dependent GEMVs per decoder layer (awq/modules/linear/gemv.py:177-180 called from
// awq/modules/fused/block.py:108-119 and fused/mlp.py:46-62), each `awq_ext.gemv_forward_cuda` its own launch.  With one launch
// per Linear (gemv_rows.hip) the 7B decode step sits at 0.50 of the HBM peak for three rounds although HBM traffic is 1.008 x
// the algorithmic bytes: 35 % of a layer is per-launch fixed cost (dispatch ramp, first-data latency, tail).  This kernel is the
// weight-streaming ENGINE of /opt/skills/guides/MI355X_MICROARCH.md (price list rows ldsdma-fill, prefetch-credit, allgather,
// engine-vs-launches): one block per CU for the whole chain,
//   * wave 0 = LOADER: streams this CU's rows of op 0, op 1, ... through a ring of NS LDS slots by LDS-DMA
//     (`global_load_lds_dwordx4 ... nt`, 1 KiB per instruction, no VGPRs), never waiting for a dependency -- the weights depend
//     on nothing, so the ring runs AHEAD across every op -> op edge (up to NS x 18 KiB per CU = ~29 MB chip-wide);
//   * waves 1..NC = CONSUMERS: the arithmetic of gemv_rows.hip (nibbles decoded in place under the fp16 exponents 2^10 / 2^6,
//     v_mfma_f32_4x4x4_16b_f16 dot products, y += s (P - C0 - z SX)) on `ds_read_b128` units of the slot; a lane's activations
//     live in registers for the whole op;
//   * an op's output vector reaches every CU as 8-byte {2 x fp16, tag} GRANULES (one agent-scope store each; the consumers sweep
//     the granules of the rows they need with agent-scope loads until every tag carries the launch's epoch): no flag, no fence,
//     no grid barrier.
// Every spin is bounded (sticky error word + NaN-free early exit); all state is re-armed by the launch itself (epoch on the
// device), so the launch replays inside a hipGraph.
//
// Roofline: HBM.  Algorithmic bytes per op as in gemv_rows.hip (K N / 2 + scales + zeros + 2 K + 2 N).
#include <string.h>
#include <type_traits>

#include "../../../autoawq_amd/csrc/awq_device.h"
#include "../../../autoawq_amd/csrc/awq_internal.h"
#include "awq_engine.h"

namespace {

constexpr int ENG_SLMAX = 6;            // 1-KiB units of a row: K <= 12288
constexpr int ENG_WSLOT = 18 * 1024;    // weight bytes of a ring slot
constexpr int ENG_NWI = 18;             // weight DMA instructions per slot (always all of them: fixed wait counts)
constexpr int ENG_SC_OFF = ENG_WSLOT;   // scales of the slot's rows (one 1-KiB DMA)
constexpr int ENG_Z_OFF = ENG_WSLOT + 1024;  // zero words of the slot's rows (one 256-byte DMA)
constexpr int ENG_SLOT = ENG_WSLOT + 1024 + 256;  // 19712 bytes
constexpr int ENG_IPS = ENG_NWI + 2;    // vector-memory instructions per slot
constexpr uint32_t ENG_SPIN = 1u << 19; // bound of every spin (x ~0.2 us)

struct EngOp {  // 64 bytes, built on the host by awq_engine_describe
    const uint32_t* qw;
    const uint32_t* qz;
    const half_t* sc;
    half_t* y;           // plain fp16 output [N] or null
    uint32_t gran_off;   // byte offset of this op's OUTPUT granules in the granule buffer
    int K, N, ZW;
    int C, SL, R, rowb;  // 16-byte chunks per row, 1-KiB units per row, rows per slot, bytes per row
};
static_assert(sizeof(EngOp) == AWQ_ENGINE_OP_BYTES, "descriptor size is part of the ABI");

AWQ_DEV EngOp load_op(const EngOp* ops, int i) {  // scalar loads (see lds_ld's note on uniformity)
    struct Raw { uint32_t w[sizeof(EngOp) / 4]; } r;
    const __attribute__((address_space(4))) uint32_t* src = reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(ops + i));
#pragma unroll
    for (int j = 0; j < (int)(sizeof(EngOp) / 4); ++j) r.w[j] = src[j];
    return __builtin_bit_cast(EngOp, r);
}

struct EngParams {
    const EngOp* ops;
    int n_ops;
    const half_t* x;            // plain fp16 input of op 0
    unsigned long long* gran;   // granule buffer
    uint32_t* ctrl;             // [0] epoch (>= 1), [1] blocks done, [2] sticky error
    int ring_slots;             // NS
    int xs_off, y_off, ctl_off; // LDS byte offsets
    int inflight;               // slots the loader keeps in flight (1..3)
    int thin;                   // the loader keeps ONE slot in flight while a consumer of its CU sweeps granules
    int dbg;                    // measurement only (results are wrong by design): 1 = no arithmetic, 2 = the gather accepts any tag, 4 = the loader issues no DMA
    unsigned long long* trace;  // debug: [block][op][8] wall-clock stamps (null = off)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

AWQ_DEV float4_t eng_mfma4(u32x2 a, u32x2 b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, a), __builtin_bit_cast(half4_t, b), c, 0, 0, 0);
}
// Control words live in LDS and are accessed by inline asm: a `volatile` access makes hipcc drain vmcnt AND lgkmcnt around it, which
// would empty the loader's DMA queue at every look at the ring.  Reads go through v_readfirstlane: the value is wave-uniform, and the
// compiler must KNOW it is (the loops around these reads decide which SGPR-based DMA instructions are issued: with a "divergent"
// condition every descriptor field lands in VGPRs).  `a` is an LDS byte address.
AWQ_DEV uint32_t lds_ld(uint32_t a) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
template <class T>
AWQ_DEV const T* uniform_ptr(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}
AWQ_DEV void lds_st(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
// the descriptor table through the CONSTANT address space: invariant memory, so uniform indices give scalar loads (a vector load in
// the loader would sit in the same in-order queue as its DMA instructions)


// 1 KiB (64 x 16 bytes) / 256 bytes from global memory into LDS at M0 + 16 (4) * lane; s_nop 4: the SGPR base may be fresh from a
// v_readfirstlane (5 wait states before a VMEM instruction reads it; hipcc pads nothing inside an asm statement)
#define ENG_DMA16_NT(voff, base, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")
#define ENG_DMA16(voff, base, ldsaddr) \
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")
#define ENG_DMA4(voff, base, ldsaddr) \
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")

// LDS control words (byte offsets from ctl_off): single writer each, monotonic
constexpr int CW_READY = 0;    // loader: slots landed
constexpr int CW_ABORT = 4;    // anybody: give up
constexpr int CW_CDONE = 32;   // [8] consumer: slots consumed
constexpr int CW_GDONE = 64;   // [8] consumer: gathers finished (op index + 1)
constexpr int CW_XDONE = 96;   // [8] consumer: activations of op in registers (op index + 1)
constexpr int CW_ODONE = 128;  // [8] consumer: rows of op computed (op index + 1)
constexpr int CW_GATHER = 8;   // consumers: how many of them sweep granules right now (ds_add / ds_sub; the loader thins itself: MI355X_MICROARCH.md gather-pass)
constexpr int CW_BYTES = 160;

// A wave-uniform value the VALU computed (integer division, LDS read ...) back in an SGPR.  Without it hipcc keeps the value -- and every
// loop counter, comparison and branch that depends on it -- in VGPRs with exec-masked control flow (measured: ~1000 cycles per row).
AWQ_DEV int U(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct CuShare {  // this CU's share of an op: row pairs dealt evenly over the grid
    int row0, rows, slots;
};
AWQ_DEV CuShare cu_share(const EngOp& o, int b, int nb) {
    const int P = o.N >> 1, pb = P / nb, pr = P - pb * nb;
    CuShare s;
    s.row0 = U(2 * (b * pb + min(b, pr)));
    s.rows = U(2 * (pb + (b < pr ? 1 : 0)));
    s.slots = U((s.rows + o.R - 1) / o.R);
    return s;
}

// min over the four words of a per-consumer array in ONE LDS round trip (a round trip costs 0.1 - 0.2 us next to the DMA stream: three
// serial ones per look at the ring made the LOADER the bottleneck, profiles/r06_engine_probe.txt).  Words of consumers that do not
// exist are initialised to all-ones.
AWQ_DEV uint32_t lds_min4(uint32_t a) {  // (the eight words of an array: two 16-byte reads, one wait)
    u32x4 v, v2;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(v2) : "v"(a) : "memory");
    const uint32_t m = min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(v2[0], v2[1]), min(v2[2], v2[3])));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
}
// the same for the consumed-slot counters, together with the gather word, still one round trip
AWQ_DEV uint32_t lds_min4_and(uint32_t a4, uint32_t a1, uint32_t& word) {
    u32x4 v, v2;
    uint32_t g;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b32 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v), "=&v"(v2), "=&v"(g) : "v"(a4), "v"(a1) : "memory");
    word = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
    const uint32_t m = min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(v2[0], v2[1]), min(v2[2], v2[3])));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
}
template <int NC>
AWQ_DEV uint32_t min_cdone(uint32_t lds0, int ctl) { return lds_min4(lds0 + (uint32_t)(ctl + CW_CDONE)); }

// ---------------------------------------------------------------------------------------------------------------- loader
template <int NC>
AWQ_DEV void eng_loader(const EngParams& p, unsigned char* smem, uint32_t lds0, int lane) {
    const int b = blockIdx.x, nb = gridDim.x, NS = p.ring_slots, ctl = p.ctl_off;
    uint32_t issued = 0, published = 0;  // slots, counted over the whole chain
    int op = 0;
    uint32_t kbase = 0;  // first slot of `op`
    bool have = false;
    EngOp o;
    CuShare sh;
    uint32_t total = 0;
    for (int i = 0; i < p.n_ops; ++i) total += (uint32_t)cu_share(load_op(p.ops, i), b, nb).slots;
    uint32_t spins = 0, free_upto = (uint32_t)NS;  // slots [0, free_upto) may be filled: their ring positions have been consumed
    int ring_pos = 0;
    while (published < total) {
        int lim = p.inflight;
        if (p.thin || issued >= free_upto) {  // one LDS round trip: the consumers' slot counters and the gather word
            uint32_t gth;
            free_upto = lds_min4_and(lds0 + (uint32_t)(ctl + CW_CDONE), lds0 + (uint32_t)(ctl + CW_GATHER), gth) + (uint32_t)NS;
            if (p.thin && gth) lim = 1;
        }
        const bool can_issue = issued < total && (int)(issued - published) < lim && issued < free_upto;
        if (can_issue) {
            if (!have) {
                o = load_op(p.ops, op);
                sh = cu_share(o, b, nb);
                have = true;
            }
            while (issued - kbase >= (uint32_t)sh.slots) {  // next op (an op may have no slot on this CU)
                kbase += (uint32_t)sh.slots;
                ++op;
                o = load_op(p.ops, op);
                sh = cu_share(o, b, nb);
            }
            const int ks = (int)(issued - kbase);
            if (p.trace && ks == 0) p.trace[((size_t)blockIdx.x * p.n_ops + op) * 8 + 7] = wall_clock64();
            const int row0 = sh.row0 + ks * o.R, nr = min(o.R, sh.rows - ks * o.R);
            const uint32_t slot = (uint32_t)U((int)(lds0 + (uint32_t)(ring_pos * ENG_SLOT)));
            ring_pos = ring_pos + 1 == NS ? 0 : ring_pos + 1;
            const uint32_t wsrc = (uint32_t)row0 * (uint32_t)o.rowb, wlast = (uint32_t)(nr * o.rowb - 16);
            const uint32_t* qw = uniform_ptr(o.qw);
            const half_t* scp = uniform_ptr(o.sc);
            const uint32_t* qzp = uniform_ptr(o.qz);
            asm volatile("s_nop 4" ::: "memory");  // (qw / slot may be fresh from v_readfirstlane: five wait states before a VMEM instruction reads them)
            if (!(p.dbg & 4)) {
#pragma unroll
                for (int i = 0; i < ENG_NWI; ++i) {
                    const uint32_t off = min((uint32_t)(i * 1024 + lane * 16), wlast);
                    ENG_DMA16_NT(wsrc + off, qw, slot + (uint32_t)(i * 1024));
                }
            }
            const int SW = 8 * o.ZW;
            const uint32_t soff = min((uint32_t)(lane * 16), (uint32_t)(nr * SW * 2 - 16));
            ENG_DMA16((uint32_t)(row0 * SW * 2) + soff, scp, slot + (uint32_t)ENG_SC_OFF);
            const uint32_t zoff = 4u * min((uint32_t)lane, (uint32_t)(nr * o.ZW - 1));
            ENG_DMA4((uint32_t)(row0 * o.ZW * 4) + zoff, qzp, slot + (uint32_t)ENG_Z_OFF);
            ++issued;
            spins = 0;
            if ((int)(issued - published) < lim && issued < total) continue;
        }
        // publish the oldest outstanding slot (ENG_IPS instructions each, retired in order)
        const int out = (int)(issued - published);
        if (out >= 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ENG_IPS) : "memory");
        } else if (out == 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ENG_IPS) : "memory");
        } else if (out == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {  // ring full and nothing in flight: the consumers are behind
            __builtin_amdgcn_s_sleep(2);
            if (++spins > ENG_SPIN || lds_ld(lds0 + (uint32_t)(ctl + CW_ABORT))) {
                if (spins > ENG_SPIN) {
                    lds_st(lds0 + (uint32_t)(ctl + CW_ABORT), 1u);
                    __hip_atomic_store(p.ctrl + 2, 0x100u + (uint32_t)op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            continue;
        }
        ++published;
        lds_st(lds0 + (uint32_t)(ctl + CW_READY), published);
    }
}

// -------------------------------------------------------------------------------------------------------------- consumer
// Bounded wait on LDS control words; false = give up (abort raised by somebody, or by this wave with `code`)
#define ENG_WAIT_RET(cond, code, ret)                                                                                         \
    {                                                                                                                    \
        uint32_t spins_ = 0;                                                                                             \
        while (!(cond)) {                                                                                                \
            __builtin_amdgcn_s_sleep(1);                                                                                 \
            if (++spins_ > ENG_SPIN || lds_ld(lds0 + (uint32_t)(ctl + CW_ABORT))) {                                                   \
                if (spins_ > ENG_SPIN) {                                                                                 \
                    lds_st(lds0 + (uint32_t)(ctl + CW_ABORT), 1u);                                                                    \
                    __hip_atomic_store(p.ctrl + 2, (uint32_t)(code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        \
                }                                                                                                        \
                return ret;                                                                                              \
            }                                                                                                            \
        }                                                                                                                \
        asm volatile("" ::: "memory");                                                                                   \
    }
#define ENG_WAIT(cond, code) ENG_WAIT_RET(cond, code, )

template <int NC>
AWQ_DEV bool all_ge(uint32_t lds0, int off, uint32_t v) {
    return lds_min4(lds0 + (uint32_t)off) >= v;
}
template <int NC>
AWQ_DEV bool all_ge_serial(uint32_t lds0, int off, uint32_t v) {
    bool ok = true;
#pragma unroll
    for (int w = 0; w < NC; ++w) ok = ok && lds_ld(lds0 + (uint32_t)(off + 4 * w)) >= v;
    return ok;
}

template <int CTRL>
AWQ_DEV float eng_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

#define ENG_STAMP(slot)                                                                                          \
    if (p.trace && w == 0) p.trace[((size_t)blockIdx.x * p.n_ops + op) * 8 + (slot)] = wall_clock64() /* all lanes, one word: no divergent branch */

// One row of a slot as it comes out of LDS: SL 16-byte weight chunks, the scale and the zero word of the lane's group per unit
template <int SL>
struct RowRegs {
    u32x4 q[SL];
    uint32_t sc[SL], zw[SL];
};

// Steps 2 + 3 of an op for one consumer wave.  Consumer w = (row group w / KS, unit part kw = w % KS): it works on the 1-KiB units
// s = kw, kw + KS, ... of rows rg, rg + NC / KS, ... of every slot -- SL of them per row (a compile-time constant: straight-line code
// per row; 0 = this wave has no unit of these rows).  Two waves per SIMD hide each other's LDS latency, and a wave holds 16 SL
// activation registers instead of 16 per unit of the whole row.  false = gave up (abort raised).
template <int NC, int KS, int SL>
AWQ_DEV bool eng_consume_op(const EngParams& p, unsigned char* smem, uint32_t lds0, int w, int lane, const EngOp& o, const CuShare& sh,
                            uint32_t kbase, int op) {
    const int NS = p.ring_slots, ctl = p.ctl_off;
    const int kw = w % KS, rg = w / KS;
    constexpr int NR = NC / KS;  // row groups
    const u32x2 sum_rows = {(lane & 3) == 0 ? 0x3C003C00u : ((lane & 3) == 1 ? 0x64006400u : 0u),
                            (lane & 3) == 0 ? 0x3C003C00u : ((lane & 3) == 1 ? 0x54005400u : 0u)};
    const int q = lane & 3;
    // ---- 2. activations LDS -> registers in the (t, t + 4) pair order of the nibble decode; per-lane constants
    //         SX = sum x, C0 = sum bias x over the lane's 32 activations of a unit (gemv_rows.hip)
    uint32_t xp[SL][16];
    float c0[SL], sx[SL];
    // This wave's unit u is unit s = kw + KS u of a row: chunk 64 s + lane, group (64 s + lane) / 4 -> byte offsets = a per-lane base +
    // a compile-time multiple of u (the instruction's offset field), zero-nibble shift independent of u; the wave's LAST unit may
    // be the row's last and run past it (K % 2048 != 0): its lanes beyond the row read the row's last chunk against zero activations.
    const int sl_ = kw + KS * (SL - 1);  // the wave's last unit
    const int cl = min(sl_ * 64 + lane, o.C - 1), gl = cl >> 2;
    const int woff0 = kw * 1024 + lane * 16, soff0 = kw * 32 + (lane >> 2) * 2, zoff0 = kw * 8 + (lane >> 5) * 4;
    const uint32_t zsh0 = 4u * (uint32_t)((lane >> 2) & 7);
    const int woffl = cl * 16, soffl = gl * 2, zoffl = (gl >> 3) * 4;
    const uint32_t zshl = 4u * (uint32_t)(gl & 7);
#pragma unroll
    for (int u = 0; u < SL; ++u) {
        const int c = (kw + KS * u) * 64 + lane;
        const bool act = c < o.C;
        const int cc = act ? c : o.C - 1;
        float4_t sums = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 d = *reinterpret_cast<const u32x4*>(smem + p.xs_off + cc * 64 + j * 16);
            if (!act) d = u32x4{0u, 0u, 0u, 0u};
            xp[u][4 * j + 0] = __builtin_amdgcn_perm(d[2], d[0], 0x05040100u);  // (x0, x4)  bias 1024
            xp[u][4 * j + 1] = __builtin_amdgcn_perm(d[2], d[0], 0x07060302u);  // (x1, x5)  bias 64
            xp[u][4 * j + 2] = __builtin_amdgcn_perm(d[3], d[1], 0x05040100u);  // (x2, x6)  bias 1024
            xp[u][4 * j + 3] = __builtin_amdgcn_perm(d[3], d[1], 0x07060302u);  // (x3, x7)  bias 64
            sums = eng_mfma4(sum_rows, u32x2{xp[u][4 * j + 0], xp[u][4 * j + 1]}, sums);
            sums = eng_mfma4(sum_rows, u32x2{xp[u][4 * j + 2], xp[u][4 * j + 3]}, sums);
        }
        sx[u] = sums[0];
        c0[u] = sums[1];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(lds0 + (uint32_t)(ctl + CW_XDONE + 4 * w), (uint32_t)op + 1u);
    ENG_STAMP(2);

    // ---- 3. this CU's slots of the op: rows rg, rg + NR, ... of every slot.  The loop is kept SMALL on purpose: a wave issues one
    //         instruction every ~5 cycles, so the instruction count per row IS the consumer's rate (a first version with a generic
    //         next-item / prefetch state machine spent ~150 instructions of control per row: 0.4 us per 1-KiB unit,
    //         profiles/r06_engine_probe.txt).  LDS latency is hidden by the second wave of the SIMD, not by software pipelining.
    const int SW = 8 * o.ZW;
    const int R = o.R, rows = sh.rows, slots = sh.slots;
    float* ypart = reinterpret_cast<float*>(smem + p.y_off) + kw * 4 + (lane >> 4);
    const int dbg = p.dbg;
    const int rstep = NR * o.rowb, sstep = NR * SW * 2, zstep = NR * o.ZW * 4;  // bytes from a row of this wave to its next one
    int rp = U((int)(kbase % (uint32_t)NS));
    uint32_t ready_seen = 0;
    for (int ks = 0; ks < slots; ++ks) {
        const uint32_t k = kbase + (uint32_t)ks;
        if (ready_seen <= k) {
            ENG_WAIT_RET((ready_seen = lds_ld(lds0 + (uint32_t)(ctl + CW_READY))) > k, 0x500u + (uint32_t)op, false);
        }
        const int nr = min(R, rows - ks * R);
        const int slot = rp * ENG_SLOT;
        const unsigned char* wrow = smem + slot + rg * o.rowb;
        const unsigned char* srow = smem + slot + ENG_SC_OFF + rg * SW * 2;
        const unsigned char* zrow = smem + slot + ENG_Z_OFF + rg * o.ZW * 4;
        float* yrow = ypart + (ks * R + rg) * KS * 4;
        for (int j = rg; j < nr; j += NR) {
            RowRegs<SL> r;
#pragma unroll
            for (int u = 0; u < SL; ++u) {
                const bool last = u == SL - 1;
                r.q[u] = *reinterpret_cast<const u32x4*>(wrow + (last ? woffl : woff0 + u * KS * 1024));
                r.sc[u] = *reinterpret_cast<const unsigned short*>(srow + (last ? soffl : soff0 + u * KS * 32));
                r.zw[u] = *reinterpret_cast<const uint32_t*>(zrow + (last ? zoffl : zoff0 + u * KS * 8));
            }
            float acc_row = 0.f;
#pragma unroll
            for (int u = 0; u < SL; ++u) {
                float4_t a = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
                if (!(dbg & 1)) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const uint32_t wv = r.q[u][jj], w8 = wv >> 8;
                        const u32x2 a0 = {and_or(wv, 0x000F000Fu, 0x64006400u), and_or(wv, 0x00F000F0u, 0x54005400u)};
                        const u32x2 a1 = {and_or(w8, 0x000F000Fu, 0x64006400u), and_or(w8, 0x00F000F0u, 0x54005400u)};
                        a = eng_mfma4(a0, u32x2{xp[u][4 * jj + 0], xp[u][4 * jj + 1]}, a);
                        a2 = eng_mfma4(a1, u32x2{xp[u][4 * jj + 2], xp[u][4 * jj + 3]}, a2);
                    }
                } else {
                    a[0] = __builtin_bit_cast(float, r.q[u][0] ^ r.q[u][1] ^ r.q[u][2] ^ r.q[u][3]);
                }
                // the lane's own dot product is the diagonal element: register (lane & 3)
                const float pa = q == 0 ? a[0] + a2[0] : (q == 1 ? a[1] + a2[1] : (q == 2 ? a[2] + a2[2] : a[3] + a2[3]));
                const float zf = (float)((r.zw[u] >> (u == SL - 1 ? zshl : zsh0)) & 15u);
                const float scl = (float)__builtin_bit_cast(half_t, (unsigned short)r.sc[u]);
                acc_row += scl * __builtin_fmaf(-zf, sx[u], pa - c0[u]);
            }
            // 16-lane sums by DPP; every lane of a DPP row stores the row's partial sum (same value, same address: no branch);
            // the publisher adds the 4 KS partial sums of a row
            acc_row += eng_dpp<0xB1>(acc_row);   // quad_perm [1,0,3,2]
            acc_row += eng_dpp<0x4E>(acc_row);   // quad_perm [2,3,0,1]
            acc_row += eng_dpp<0x124>(acc_row);  // row_ror:4
            acc_row += eng_dpp<0x128>(acc_row);  // row_ror:8
            *yrow = acc_row;
            wrow += rstep;
            srow += sstep;
            zrow += zstep;
            yrow += NR * KS * 4;
        }
        rp = rp + 1 == NS ? 0 : rp + 1;
        // (the rows' data are in registers / consumed: LDS operations of a wave complete in order, so this store follows every read)
        lds_st(lds0 + (uint32_t)(ctl + CW_CDONE + 4 * w), k + 1u);
    }
    ENG_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(lds0 + (uint32_t)(ctl + CW_CDONE + 4 * w), kbase + (uint32_t)sh.slots);
    lds_st(lds0 + (uint32_t)(ctl + CW_ODONE + 4 * w), (uint32_t)op + 1u);
    ENG_STAMP(4);
    return true;
}

// A consumer without a unit of this op's rows (K <= 1024 (KS - 1)): zero partial sums, and the op's bookkeeping
template <int NC, int KS>
AWQ_DEV void eng_consume_none(const EngParams& p, unsigned char* smem, uint32_t lds0, int w, int lane, const CuShare& sh, uint32_t kbase, int op) {
    const int ctl = p.ctl_off, kw = w % KS, rg = w / KS;
    constexpr int NR = NC / KS;
    lds_st(lds0 + (uint32_t)(ctl + CW_XDONE + 4 * w), (uint32_t)op + 1u);
    float* ypart = reinterpret_cast<float*>(smem + p.y_off);
    for (int r = rg; r < sh.rows; r += NR) ypart[(r * KS + kw) * 4 + (lane & 3)] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(lds0 + (uint32_t)(ctl + CW_CDONE + 4 * w), kbase + (uint32_t)sh.slots);
    lds_st(lds0 + (uint32_t)(ctl + CW_ODONE + 4 * w), (uint32_t)op + 1u);
}

template <int NC, int KS>
AWQ_DEV void eng_consumer(const EngParams& p, unsigned char* smem, uint32_t lds0, int w, int lane, uint32_t epoch) {
    const int b = blockIdx.x, nb = gridDim.x, ctl = p.ctl_off;
    uint32_t kbase = 0;
    for (int op = 0; op < p.n_ops; ++op) {
        const EngOp o = load_op(p.ops, op);
        const CuShare sh = cu_share(o, b, nb);
        const int G = o.K >> 1;  // granules (pairs of activations) this op consumes
        // ---- 1. gather the activations into the LDS staging vector xs [K] fp16 (this wave: 64-granule batches w, w + NC, ...)
        ENG_WAIT(all_ge<NC>(lds0, ctl + CW_XDONE, (uint32_t)op), 0x200u + (uint32_t)op);  // everybody is done with the previous xs
        if (w == 0) ENG_STAMP(0);
        asm volatile("ds_add_u32 %0, %1" ::"v"(lds0 + (uint32_t)(ctl + CW_GATHER)), "v"(1u) : "memory");
        const int nbat = (G + 63) >> 6;
        uint32_t* xs32 = reinterpret_cast<uint32_t*>(smem + p.xs_off);
        if (op == 0) {
            const uint32_t* x32 = reinterpret_cast<const uint32_t*>(p.x);
            for (int bt = w; bt < nbat; bt += NC) {
                const int g = min(bt * 64 + lane, G - 1);  // (clamped, not masked: a per-lane branch makes hipcc treat the loop state as divergent)
                xs32[g] = x32[g];
            }
        } else {
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned char*>(p.gran) + load_op(p.ops, op - 1).gran_off);
            constexpr int GB = 16;  // loads in flight per lane
            for (int bt0 = w; bt0 < nbat; bt0 += NC * GB) {
                uint32_t spins = 0;
                for (;;) {
                    unsigned long long v[GB];
#pragma unroll
                    for (int u = 0; u < GB; ++u) {
                        const int g = min((bt0 + u * NC) * 64 + lane, G - 1);
                        v[u] = __hip_atomic_load(src + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bool okg = true;
#pragma unroll
                    for (int u = 0; u < GB; ++u) okg = okg && (uint32_t)(v[u] >> 32) == epoch;
                    if (__builtin_amdgcn_ballot_w64(!okg) == 0ull || (p.dbg & 2)) {
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            const int g = min((bt0 + u * NC) * 64 + lane, G - 1);
                            xs32[g] = (uint32_t)v[u];
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                    spins = (uint32_t)U((int)spins + 1);
                    if (spins > ENG_SPIN || lds_ld(lds0 + (uint32_t)(ctl + CW_ABORT))) {
                        if (spins > ENG_SPIN) {
                            lds_st(lds0 + (uint32_t)(ctl + CW_ABORT), 1u);
                            __hip_atomic_store(p.ctrl + 2, 0x300u + (uint32_t)op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        return;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("ds_sub_u32 %0, %1" ::"v"(lds0 + (uint32_t)(ctl + CW_GATHER)), "v"(1u) : "memory");
        lds_st(lds0 + (uint32_t)(ctl + CW_GDONE + 4 * w), (uint32_t)op + 1u);
        ENG_WAIT(all_ge<NC>(lds0, ctl + CW_GDONE, (uint32_t)op + 1u), 0x400u + (uint32_t)op);
        if (w == 0) ENG_STAMP(1);

        bool ok = true;
        const int slw = U((o.SL - (w % KS) + KS - 1) / KS);  // this wave's units per row
        switch (slw) {
            case 0: eng_consume_none<NC, KS>(p, smem, lds0, w, lane, sh, kbase, op); break;
            case 1: ok = eng_consume_op<NC, KS, 1>(p, smem, lds0, w, lane, o, sh, kbase, op); break;
            case 2: ok = eng_consume_op<NC, KS, 2>(p, smem, lds0, w, lane, o, sh, kbase, op); break;
            case 3: ok = eng_consume_op<NC, KS, 3>(p, smem, lds0, w, lane, o, sh, kbase, op); break;
            default:
                if constexpr (ENG_SLMAX / KS > 3) {
                    switch (slw) {
                        case 4: ok = eng_consume_op<NC, KS, 4>(p, smem, lds0, w, lane, o, sh, kbase, op); break;
                        case 5: ok = eng_consume_op<NC, KS, 5>(p, smem, lds0, w, lane, o, sh, kbase, op); break;
                        default: ok = eng_consume_op<NC, KS, 6>(p, smem, lds0, w, lane, o, sh, kbase, op); break;
                    }
                }
                break;
        }
        if (!ok) return;
        kbase += (uint32_t)sh.slots;

        // ---- 4. consumer 0 publishes the CU's rows: granules {y[2 g], y[2 g + 1], epoch} for every CU, plain y for the caller
        if (w == 0) {
            ENG_WAIT(all_ge<NC>(lds0, ctl + CW_ODONE, (uint32_t)op + 1u), 0x600u + (uint32_t)op);
            ENG_STAMP(5);
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(p.gran) + o.gran_off);
            const float4_t* yp = reinterpret_cast<const float4_t*>(smem + p.y_off);  // [row][KS] x four partial sums
            uint32_t* yo = reinterpret_cast<uint32_t*>(o.y);
            const int pairs = sh.rows >> 1, pair0 = sh.row0 >> 1;
            for (int e0 = 0; e0 < pairs; e0 += 64) {
                const int e = min(e0 + lane, pairs - 1);
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const float4_t r0 = yp[2 * e * KS + k], r1 = yp[(2 * e + 1) * KS + k];
                    s0 += (r0[0] + r0[1]) + (r0[2] + r0[3]);
                    s1 += (r1[0] + r1[1]) + (r1[2] + r1[3]);
                }
                const half2_t h = {(half_t)s0, (half_t)s1};
                const uint32_t v = h22u(h);
                __hip_atomic_store(dst + pair0 + e, (unsigned long long)v | ((unsigned long long)epoch << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (yo) yo[pair0 + e] = v;
            }
            ENG_STAMP(6);
        }
    }
}

template <int NC, int KS>
__global__ __launch_bounds__(64 * (NC + 1)) void awq_gemv_engine_kernel(EngParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (threadIdx.x < CW_BYTES / 4)  // (consumer w's words sit at index 8 + 8 a + w of the four per-consumer arrays)
        lds_st(lds0 + (uint32_t)(p.ctl_off + 4 * (int)threadIdx.x), threadIdx.x >= 8 && (int)(threadIdx.x & 7) >= NC ? 0xFFFFFFFFu : 0u);
    EngParams pp = p;
    {  // debug: ctrl words 8-9 hold a trace buffer pointer (0 = off)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.ctrl[8]), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.ctrl[9]);
        pp.trace = reinterpret_cast<unsigned long long*>(((unsigned long long)hi << 32) | lo);
    }
    __syncthreads();
    if (wave == 0)
        eng_loader<NC>(pp, smem, lds0, lane);
    else
        eng_consumer<NC, KS>(pp, smem, lds0, wave - 1, lane, epoch);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {  // the last block of the launch closes the epoch (every block has read it by then)
        const uint32_t done = __hip_atomic_fetch_add(p.ctrl + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done + 1u == gridDim.x) {
            __hip_atomic_store(p.ctrl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.ctrl, epoch + 1u == 0u ? 1u : epoch + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// rows per ring slot: the weight bytes fit the slot, the rows' scales one 1-KiB DMA instruction, their zero words one 256-byte one
int eng_rows_per_slot(int K, int ZW) {
    int R = ENG_WSLOT / (K / 2);
    if (R > 1024 / (16 * ZW)) R = 1024 / (16 * ZW);
    if (R > 64 / ZW) R = 64 / ZW;
    return R;
}

}  // namespace

bool awq_engine_supports(int64_t K, int64_t N, int64_t g, int64_t ZW) {
    if (g != 128 || K < 128 || K % 128 || K > ENG_SLMAX * 2048 || N < 2 || N % 2) return false;
    if (ZW < (K / 128 + 7) / 8) return false;
    if (ZW > 64 || eng_rows_per_slot((int)K, (int)ZW) < 1) return false;
    if (N * K / 2 >= ((int64_t)1 << 32) || N * ZW * 16 >= ((int64_t)1 << 32)) return false;  // 32-bit byte offsets
    return true;
}

int awq_engine_describe(void* desc, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* y, int64_t K,
                        int64_t N, int64_t group_size, int64_t zeros_width, uint64_t granule_offset) {
    if (!desc || !qweight || !scales || !qzeros) return AWQ_ERR_NULL;
    if (!awq_engine_supports(K, N, group_size, zeros_width)) return AWQ_ERR_UNSUPPORTED;
    if (granule_offset % 8 || granule_offset + (uint64_t)N * 4 >= ((uint64_t)1 << 32)) return AWQ_ERR_BAD_SHAPE;
    if ((reinterpret_cast<uintptr_t>(qweight) & 15) || (reinterpret_cast<uintptr_t>(scales) & 15) || (reinterpret_cast<uintptr_t>(qzeros) & 3) ||
        (reinterpret_cast<uintptr_t>(y) & 3))
        return AWQ_ERR_BAD_ALIGNMENT;
    EngOp o;
    memset(&o, 0, sizeof(o));
    o.qw = reinterpret_cast<const uint32_t*>(qweight);
    o.qz = reinterpret_cast<const uint32_t*>(qzeros);
    o.sc = reinterpret_cast<const half_t*>(scales);
    o.y = reinterpret_cast<half_t*>(y);
    o.gran_off = (uint32_t)granule_offset;
    o.K = (int)K; o.N = (int)N; o.ZW = (int)zeros_width;
    o.C = (int)(K / 32);
    o.SL = (o.C + 63) / 64;
    o.R = eng_rows_per_slot((int)K, (int)zeros_width);
    o.rowb = (int)(K / 2);
    memcpy(desc, &o, sizeof(o));
    return AWQ_OK;
}

size_t awq_engine_granule_bytes(int64_t N) { return N > 0 ? (size_t)(N / 2) * 8 : 0; }
size_t awq_engine_ctrl_bytes(void) { return 64; }

int awq_engine_forward(const uint16_t* x, const void* ops_dev, int64_t n_ops, int64_t max_N, int64_t max_K, void* granules,
                       void* ctrl, uint32_t flags, void* stream) {
    if (!x || !ops_dev || !granules || !ctrl) return AWQ_ERR_NULL;
    if (n_ops < 1 || n_ops > 4096 || max_K < 128 || max_K > ENG_SLMAX * 2048 || max_N < 2) return AWQ_ERR_BAD_SHAPE;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return AWQ_ERR_LAUNCH;
    const int64_t max_rows_per_cu = 2 * ((max_N / 2 + cus - 1) / cus);
    EngParams p;
    p.ops = static_cast<const EngOp*>(ops_dev);
    p.n_ops = (int)n_ops;
    p.x = reinterpret_cast<const half_t*>(x);
    p.gran = static_cast<unsigned long long*>(granules);
    p.ctrl = static_cast<uint32_t*>(ctrl);
    int nc = (int)((flags >> 4) & 0xFu), ns = (int)((flags >> 8) & 0xFu), inflight = (int)(flags & 0xFu);
    if (nc == 0) nc = 6;
    if (nc != 3 && nc != 6) return AWQ_ERR_UNSUPPORTED;  // 3: one wave per SIMD on whole rows; 6: two per SIMD, each half of a row's units
    const int ks = nc == 6 ? 2 : 1;
    if (inflight < 1 || inflight > 3) inflight = 2;
    p.inflight = inflight;
    p.dbg = (int)((flags >> 12) & 0xFu);
    p.thin = (flags >> 16) & 1u ? 0 : 1;  // bit 16: do NOT thin the loader during gathers
    p.trace = nullptr;  // the kernel takes it from ctrl words 8-9
    const int xs_bytes = (int)((max_K * 2 + 63) / 64 * 64);
    const int y_bytes = (int)(max_rows_per_cu * 16 * ks);  // four fp32 partial sums per (row, unit part)
    int max_ns = (160 * 1024 - xs_bytes - y_bytes - CW_BYTES) / ENG_SLOT;
    if (max_ns > 15) max_ns = 15;
    if (ns == 0 || ns > max_ns) ns = max_ns;
    if (ns < 2) return AWQ_ERR_UNSUPPORTED;
    p.ring_slots = ns;
    p.xs_off = ns * ENG_SLOT;
    p.y_off = p.xs_off + xs_bytes;
    p.ctl_off = p.y_off + y_bytes;
    const size_t lds = (size_t)p.ctl_off + CW_BYTES;
    // One block per CU, every block resident for the whole launch: the LDS footprint (> 80 KiB) admits one block per CU and the grid
    // is the CU count.  (The row shares are a function of gridDim, the descriptors are not.)
    if (lds <= 80 * 1024) return AWQ_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static std::atomic<unsigned long long> opted3{0}, opted6{0};
    if (nc == 3) {
        (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_engine_kernel<3, 1>), opted3);
        hipLaunchKernelGGL((awq_gemv_engine_kernel<3, 1>), dim3((unsigned)cus), dim3(256), lds, st, p);
    } else {
        (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_engine_kernel<6, 2>), opted6);
        hipLaunchKernelGGL((awq_gemv_engine_kernel<6, 2>), dim3((unsigned)cus), dim3(448), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
