// gemv_batch.hip -- batched decode (1 <= M <= 128 per launch) on the GEMV / GEMVFast layouts: weights stream through LDS by DMA into MFMA,
// the activations live in REGISTERS as MFMA A fragments, the K range of a tile is split over the waves of ONE block, gfx950.
//
// Replaces awq_ext.gemmv2_forward_cuda (awq/modules/linear/gemv.py:168-176: the reference's kernel for more than 8 rows on
// this layout) and awq_ext.gemv_forward_cuda (:178-180) for the batches above the row-streaming kernel (gemv_rows.hip, M <= 4).
// Layout: qweight [N, K/8] int32 (nibble i of word c = w[n, 8c+i]), qzeros [N, ZW] int32 (nibble i of word c =
// z[n, group 8c+i]), scales [N, 8 ZW] fp16.  group_size == 128.
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 + (K/g)*N/2 + (K/g)*N*2 + M*K*2 + M*N*2.
//
// Why this shape (round 5; what the predecessors measured):
//  * gemv_lds.hip (M <= 16, M K <= 32768) stages ALL of x in LDS as A fragments per block -- 64 KB through registers and two
//    barriers before the first MFMA: start-up bound (11.5 us at 4096 x 11008, M = 8, against 5.2 us at M = 1), and it cannot
//    take M = 16 at K = 4096 or any batch at K = 11008.  The register-fetch N-major forms (gemm_skinny MODE 1 / 2, round 4's
//    staged experiment, profiles/r05_first_call/skinny_nk.txt) read 16 rows x 64 bytes per wave instruction: 21-25 us at every M.
//  * Here a block's eight waves split the K range of a 16-row tile: wave wk owns GW = 4 consecutive 128-k groups (512 k), so its
//    activations are MI x 16 MFMA A fragments that stay in (at most 128) REGISTERS for the whole pass -- no block barrier before
//    the first MFMA, no LDS read on the A side of any MFMA.
//  * Weights: a PIECE = 16 rows x 256 bytes of ONE wave's K range, by four LDS-DMA instructions (`global_load_lds_dwordx4`, 1 KiB
//    each = 4 rows x 256 B; the eight waves of the block together read the tile's rows end to end) + one for the rows' scales
//    (16 bytes per row) + one for the zero words: LDM = 6 identical vector-memory instructions per piece, so
//    `s_waitcnt vmcnt(6 (RD - 1))` names exactly one ring slot; no VGPR is a DMA destination.  A 16-byte chunk XOR swizzle applied
//    on the GLOBAL side (lane i of a row fetches chunk i ^ row) makes the ds_read_b128 fragment reads (lane (n, kq): chunk
//    4 u + kq of row n = the B operands of four MFMAs) bank-conflict free without padding.
//  * Activations (first version, profiles/r05_gemv_batch_trace_first.txt: 16 bytes per lane straight into fragment position =
//    16 rows x 4 x 16 B per instruction; ISSUING those took 1.3 us per wave at M = 8 and 5.4 us at M = 32 -- the CU's one address
//    path walks 32 cache lines per instruction): XS form -- each batch row's 1 KiB of the wave's K range arrives by ONE coalesced
//    LDS-DMA instruction into a wave-private staging area (swizzled on the global side), then 16 MI ds_read_b128 put the fragments
//    into registers; no barrier (the wave reads only what it requested itself).  The staging area is dead afterwards and holds the
//    partial-tile buffers.  The area holds eight rows per wave (64 KB per block beside a two-deep ring); more rows arrive in CHUNKS of
//    eight, each requested once the previous one has been read (lanes of the other half keep what they hold).  The direct form stays
//    selectable (flag) for A/B runs.
//  * Decode: nibbles stay in place under the exponents 2^10 / 2^6 (five VALU per word), four packed subtracts of (bias + z): the
//    EXACT integers w - z in fp16; products and the fp32 accumulation in v_mfma_f32_16x16x32_f16 are exact per term; the group's
//    scale multiplies the group's fp32 sum (4 MI fused multiply-adds per group instead of four packed multiplies per word: the first
//    version spent 13 VALU per word and 0.85 us per piece, VALU-bound).  One-hot and zero inputs stay exact; f(2x) = 2 f(x).
//  * The wk partial tiles of a block meet in LDS behind ONE raw s_barrier per tile (the DMA ring stays in flight across it; LDS
//    writes are drained by hand: `__syncthreads()` would wait for vmcnt(0)), summed in wave order: bitwise reproducible.  Nothing
//    crosses a CU: no workspace, no exchange.  y is parked in LDS (fp32) and written after the stream has drained (a store
//    inside the stream would break the counted waits: stores count in vmcnt but do not retire in order with loads).
//  * K beyond one pass (8 waves x 512 k) is walked in PASSES: the A fragments of the next K range are re-requested (a drain: they
//    queue behind the ring), the ring keeps running across the pass edge, partial sums add up in the parked tiles.
//  * M > 32 (round 6): still ONE launch up to 128 rows -- ROW PARTS of <= 32 rows, each part in its own block, the blocks that walk the
//    same tiles residents of one XCD (see the kernel head; the first form -- the parts as wave groups of one block -- stays
//    selectable: AWQ_GEMM_FLAG_WAVES = 1); beyond 128 rows the C API walks balanced chunks of <= 128.
//  * What bounds a unit (tile x pass) is not memory: the per-wave phase times of the -DAWQ_GEMV_TRACE build show ~0 us spent waiting
//    for a piece; the two waves of a SIMD issue ~0.75 us each per unit (requests, decode VALU, 32 MFMAs at MI 2, exchange) and do not
//    overlap much (MFMA-only + decode-only switch-off runs add up to the full cost): profiles/r06_gemv_batch_trace.txt.  Ring depth
//    therefore hardly matters (one slot and two lazy ones within 1 - 3 %), and the successor unit's request is division- and branch-free.
#include <type_traits>

#include "awq_device.h"
#include "awq_internal.h"

namespace {

constexpr int GW = 4;                          // 128-k groups per wave and pass
constexpr int PIECE_W = 16 * GW * 64;          // bytes of weights per piece: 16 rows x GW groups x 64 bytes
constexpr int PIECE_B = PIECE_W + 1024 + 256;  // + the rows' scales (64 x 16-byte slots) + zero words (64 x 4)
constexpr int LDM = GW + 2;                    // vector-memory instructions per piece request

struct BatchParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    half_t* y;
    int M, K, N;
    int KW, ZW, SW;  // words per qweight / qzeros row, halfs per scales row
    int G;           // groups: K / 128
    int wk, wt;      // waves side by side on a tile's K range, tile owners per block (wk * wt == 8)
    int passes;      // ceil(G / (wk * GW))
    int tiles_base, tiles_rem, tiles_max;  // tiles per owner: base (+1 for the first rem owners)
    int ring_off, pbuf_off, pbuf_pitch, ystage_off;  // LDS byte offsets; pbuf_pitch: bytes per wave (XS: the wave's staging area)
    int xs_rows;     // XS: batch rows the staging area holds at a time (min(M, 8): more rows arrive in chunks of eight)
    int rs, rows_part;  // round 6: ROW PARTS -- the wt wave groups of a block are (wt / rs tile owners) x (rs row parts of rows_part <= 32 batch rows)
    int brs;            // ... or brs row parts ACROSS blocks (blocks of one XCD that walk the same tiles for different batch rows; rs == 1 then)
    int GP;          // FAST (GEMVFast layout): rows of scales / qzeros [GP, N]
    unsigned long long* trace;             // debug builds only (tools/trace_gemv_batch.py)
};

// Debug builds only (tools/trace_gemv_batch.py): -DAWQ_GEMV_TRACE stamps the phases of every wave (kept in registers, stored after the
// stream has drained); -DAWQ_BT_DBG=bits switches parts off (results are wrong by design): 1 = every lane requests the SAME 16 bytes of
// x (direct form: the request count stays, the 16-rows-x-64-bytes scatter goes), 2 = no decode / MFMA, 4 = no partial-tile exchange.
#ifndef AWQ_BT_DBG
#define AWQ_BT_DBG 0
#endif
#ifdef AWQ_GEMV_TRACE
#define BT_STAMP(slot) ts[slot] = wall_clock64()
#else
#define BT_STAMP(slot) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define AWQ_BT_DMA16(voff, base, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")
#define AWQ_BT_DMA4(voff, base, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")
#define AWQ_BT_LOAD16(dst, voff, base) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(base) : "memory")
// 16 register quads per statement (asm operand lists are bounded)
#define AWQ_BT_WAIT16(X, o, cnt)                                                                                                   \
    asm volatile("s_waitcnt vmcnt(%16)"                                                                                            \
                 : "+v"(X[o + 0]), "+v"(X[o + 1]), "+v"(X[o + 2]), "+v"(X[o + 3]), "+v"(X[o + 4]), "+v"(X[o + 5]), "+v"(X[o + 6]), \
                   "+v"(X[o + 7]), "+v"(X[o + 8]), "+v"(X[o + 9]), "+v"(X[o + 10]), "+v"(X[o + 11]), "+v"(X[o + 12]),              \
                   "+v"(X[o + 13]), "+v"(X[o + 14]), "+v"(X[o + 15])                                                               \
                 : "n"(cnt))

// ds_read_b128 into the lanes of `mask` only (the other lanes keep what the register holds): the second half-chunk of the staged
// activations merges into the fragment registers without a temporary (a select needs both copies live: register spills at MI 2).
// The reads are invisible to hipcc's lgkmcnt bookkeeping: AWQ_BT_LGKM16 waits and names the registers.
#define AWQ_BT_LDS_READ16_MASKED(dst, addr, OFF, mask, save)                                                                        \
    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tds_read_b128 %0, %2 offset:" #OFF "\n\ts_mov_b64 exec, %1"      \
                 : "+v"(dst), "=&s"(save)                                                                                           \
                 : "v"(addr), "s"(mask)                                                                                             \
                 : "scc")  /* s_and_b64 writes SCC: undeclared, a compare hipcc had made BEFORE the statement decided a branch after it */
#define AWQ_BT_LGKM16(X, o)                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                             \
                 : "+v"(X[o + 0]), "+v"(X[o + 1]), "+v"(X[o + 2]), "+v"(X[o + 3]), "+v"(X[o + 4]), "+v"(X[o + 5]), "+v"(X[o + 6]),  \
                   "+v"(X[o + 7]), "+v"(X[o + 8]), "+v"(X[o + 9]), "+v"(X[o + 10]), "+v"(X[o + 11]), "+v"(X[o + 12]),               \
                   "+v"(X[o + 13]), "+v"(X[o + 14]), "+v"(X[o + 15]))

AWQ_DEV float4_t mfma16(u32x4 a, u32x4 b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}

// staging swizzle of the activations (XS form): chunk j (16 bytes = 8 k) of batch row m sits at chunk slot
// (j & 48) | ((j & 15) ^ f(m)), f(m) = the two 2-bit halves of m & 15 swapped: the sixteen lanes a ds_read_b128 serves together
// (rows {0-3, 12-15} of one kq with rows {4-11} of the next) then hit sixteen different 16-byte columns
AWQ_DEV int xs_f(int m) { return ((m & 3) << 2) | ((m >> 2) & 3); }

// MI: 16-row batch tiles (1 | 2); RD: ring slots per wave -- 1 | 3: that many pieces in flight, re-requested after a piece has been
// consumed; 2 (LAZY, the default): the NEXT piece is requested the moment a piece has landed, i.e. its flight overlaps the consumption
// of the current one while no wave ever holds more than one request in the memory system's queues (measured: with two or three
// pieces per wave requested up front every request takes 2.5 us instead of 1.2 -- the queues, not the latency, set the pace -- and the
// waves start later: profiles/r05_gemv_batch_trace_v3.txt); XS: the activations reach the registers through a wave-private LDS
// staging area (coalesced LDS-DMA) instead of 16-byte fragment loads
// FAST: the GEMVFast layout's buffers (awq/modules/linear/gemv_fast.py:26-65): qweight int16 [N/4, K] -- element [r, 64 b + 16 i + 8 h + t],
// nibble j = w[4 r + i, 64 b + 32 h + 8 j + t] -- scales / qzeros fp16 [GP, N] with qzeros = -(s z); effective weight w s + qzeros.  A
// tile's 16 rows are four int16 rows whose 512 k of the wave are 1 KiB contiguous each (one DMA instruction per int16 row); a 16-byte
// chunk (i, h) of a 64-k block holds, dword d, the nibble pairs (8 j + 2 d, 8 j + 2 d + 1): lane kq takes the chunk of (block kq / 2,
// half kq % 2) of its row, step c its dword c -- so the A fragment of step c is dword c of each of the lane's four activation chunks:
// one 4 x 4 dword transpose per pass, no pair permute.  Nibbles decode to 16 + w (exponent 2^4), the group folds y += s (acc - 16 sx) + qzeros sx with sx = sum of x over the
// group (one ones-MFMA chain per pass, parked in LDS), the arithmetic of csrc/gemv_fast.hip.
template <int MI, int RD, bool XS, bool FAST = false>
__global__ __launch_bounds__(512) void awq_gemv_batch_kernel(BatchParams p) {
    static_assert(!FAST || XS, "the GEMVFast form keeps its group sums in the staging area");
    constexpr int NA = MI * GW * 4;  // A fragments (16 bytes each) per lane
    constexpr bool LAZY = RD == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int wki = wave % p.wk, twi = wave / p.wk;
    // Round 6 (one launch for 33 .. 128 rows, the matrix streamed ONCE), first form (rs = 2 | 4, kept selectable):
    // the wave groups of a block that used to own different tiles own the SAME tiles for different 32-row PARTS of the batch; K is
    // then walked in rs x as many passes (wk = 8 / wt waves side by side on a tile) and every block stages ALL of x: 21 / 35 / 37 us at
    // 64 / 96 / 128 rows of 4096 x 11008, against 19 / 25 / 29 for the parts across blocks below.
    const int rsi = twi % p.rs, toi = twi / p.rs;  // row part, tile owner of the block
    // Row parts ACROSS blocks (brs > 1, the default above 32 rows): the activations a CU pulls through its L2 port are M K 2 / brs bytes
    // instead of all M K 2 (at 64 rows x 4096 k: 512 KB per CU against 88 KB of weights -- the activations, not the matrix, were the
    // launch's L2 -> CU traffic, staged twice because a part's K range then needed two passes); the brs blocks that walk the same tiles
    // are brs consecutive residents of ONE XCD (block ids b, b + 8, ..: the dispatcher deals blocks to the eight XCDs round robin), so
    // their weight requests meet in that XCD's L2 (FETCH_SIZE 1.10 x algorithmic at 64 rows; with neighbouring block ids, i.e.
    // different XCDs, it was 1.94 x: profiles/r06_batch_parts.txt).
    const int bq = (int)blockIdx.x >> 3;
    const int bpart = p.brs > 1 ? bq % p.brs : 0;
    const int oblock = p.brs > 1 ? (bq / p.brs) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    const int nob = (int)gridDim.x / p.brs;                     // blocks with different tiles
    const int row_base = (bpart * p.rs + rsi) * p.rows_part;    // first batch row of this wave group
    const int M = max(0, min(p.rows_part, p.M - row_base));
    // owner = the wk waves that share tiles; owner ids interleave the blocks (consecutive owners sit on different CUs)
    const int owner = toi * nob + oblock;
    const int t0 = owner * p.tiles_base + min(owner, p.tiles_rem);
    const int ntile = p.tiles_base + (owner < p.tiles_rem ? 1 : 0);
    const int nunit = ntile * p.passes;  // live units of this wave, flat: u = pass * ntile + tile
    const int ring = p.ring_off + wave * RD * PIECE_B;
    const int rowbytes = p.KW * 4;
    const int xs_w = p.pbuf_off + wave * p.pbuf_pitch;  // XS: this wave's staging area [M][1 KiB]; later its partial-tile buffers
#ifdef AWQ_GEMV_TRACE
    unsigned long long ts[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 12 .. 15: time spent waiting / requesting / consuming / exchanging
    unsigned long long tph = 0;
#define BT_PHASE(slot) do { const unsigned long long now_ = wall_clock64(); ts[slot] += now_ - tph; tph = now_; } while (0)
#else
#define BT_PHASE(slot) do { } while (0)
#endif
    BT_STAMP(0);

    // ---- request flat unit u into ring slot u % RD (past the end: clamped addresses -- the counted waits need the requests)
    //      request_at: unit u = (pass ps, tile tl) given by the caller (the stream knows its successor without a division); a dead unit
    //      re-reads the owner's first piece (valid memory, nobody consumes it)
    auto request_at = [&](int u, int ps_u, int tl_u, bool live) {
        const int ps = live ? ps_u : 0, tl = live ? tl_u : 0;
        const int g0 = (ps * p.wk + wki) * GW;
        const int row0 = (ntile > 0 ? t0 + tl : 0) * 16;
        const uint32_t slot = lds0 + (uint32_t)(ring + (u % RD) * PIECE_B);
        if constexpr (FAST) {
            const int rb2 = p.K * 2;  // bytes of an int16 row
#pragma unroll
            for (int i = 0; i < GW; ++i) {  // int16 row i of the tile: its 1 KiB of this wave's K range, chunk slots permuted for the reads
                constexpr int T[4] = {0, 2, 3, 1};
                const int kqq = ((lane >> 2) & 3) ^ T[i];
                const int off = 256 * (lane >> 4) + 128 * (kqq >> 1) + 32 * (lane & 3) + 16 * (kqq & 1);
                const int byte = min(256 * g0 + off, rb2 - 16);
                const uint32_t voff = (uint32_t)(min((row0 >> 2) + i, (p.N >> 2) - 1) * rb2 + byte);
                AWQ_BT_DMA16(voff, p.qweight, slot + 1024u * i);
            }
            // scales / qzeros of groups g0 .. g0 + 3 for the tile's 16 rows: 32 bytes per group, lane l: group (l >> 3) & 3, dword l & 7
            const int gr = min(g0 + ((lane >> 3) & 3), p.GP - 1);
            const uint32_t vsz = (uint32_t)(((gr * p.N + min(row0, p.N - 16)) * 2) + 4 * (lane & 7));
            AWQ_BT_DMA4(vsz, p.scales, slot + (uint32_t)PIECE_W);
            AWQ_BT_DMA4(vsz, p.qzeros, slot + (uint32_t)(PIECE_W + 1024));
            return;
        }
#pragma unroll
        for (int i = 0; i < GW; ++i) {
            const int r = i * 4 + (lane >> 4);  // row of the tile (four rows of 256 bytes per instruction)
            const int c = ((lane & 15) ^ r) & 15;
            const int byte = min(g0 * 64 + 16 * c, rowbytes - 16);  // past the row end (a ragged or dead piece): its last chunk (A is 0 there)
            const uint32_t voff = (uint32_t)(min(row0 + r, p.N - 1) * rowbytes + byte);
            AWQ_BT_DMA16(voff, p.qweight, slot + 1024u * i);
        }
        {
            const int r = min(row0 + n, p.N - 1);
            const int sbyte = min((2 * g0) & ~15, p.SW * 2 - 16);
            AWQ_BT_DMA16((uint32_t)(r * p.SW * 2 + sbyte), p.scales, slot + (uint32_t)PIECE_W);
            const int zword = min(g0 >> 3, p.ZW - 1);
            AWQ_BT_DMA4((uint32_t)((r * p.ZW + zword) * 4), p.qzeros, slot + (uint32_t)(PIECE_W + 1024));
        }
    };

    auto request = [&](int u) {
        const bool live = u < nunit;
        const int uu = live ? u : 0;
        const int nt1 = max(ntile, 1);
        const int ps = uu / nt1;
        request_at(u, ps, uu - ps * nt1, live);
    };

    // ---- activations of pass ps -> A fragments in registers (pair-permuted), zero for batch rows >= M and groups >= G.
    //      `first_piece`: the ring's first request goes out right behind the first chunk of activations (pass 0 only).
    u32x4 afr[NA];
    auto permute_a = [&](int g0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int u = 0; u < GW; ++u) {
                const bool valid = (16 * mi + n < M) && (g0 + u < p.G);
                const uint32_t slo = valid ? 0x05040100u : 0x0C0C0C0Cu, shi = valid ? 0x07060302u : 0x0C0C0C0Cu;  // 0x0C: the constant 0
                if constexpr (FAST) {
                    // natural pair order, but step c wants dword c of each of the lane's four chunks: a 4 x 4 dword transpose, ONCE per pass
                    // (assembled per MFMA it cost four copies each and spilled at MI 2); invalid rows / groups: zeros
                    const uint32_t sid = valid ? 0x03020100u : 0x0C0C0C0Cu;
                    const u32x4 d0 = afr[(mi * GW + u) * 4 + 0], d1 = afr[(mi * GW + u) * 4 + 1], d2 = afr[(mi * GW + u) * 4 + 2], d3 = afr[(mi * GW + u) * 4 + 3];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        u32x4 v;
                        v[0] = __builtin_amdgcn_perm(d0[c], d0[c], sid);
                        v[1] = __builtin_amdgcn_perm(d1[c], d1[c], sid);
                        v[2] = __builtin_amdgcn_perm(d2[c], d2[c], sid);
                        v[3] = __builtin_amdgcn_perm(d3[c], d3[c], sid);
                        afr[(mi * GW + u) * 4 + c] = v;
                    }
                    continue;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32x4 d = afr[(mi * GW + u) * 4 + c];
                    u32x4 v;
                    v[0] = __builtin_amdgcn_perm(d[2], d[0], slo);  // (x0, x4)  bias 1024
                    v[1] = __builtin_amdgcn_perm(d[2], d[0], shi);  // (x1, x5)  bias 64
                    v[2] = __builtin_amdgcn_perm(d[3], d[1], slo);  // (x2, x6)  bias 1024
                    v[3] = __builtin_amdgcn_perm(d[3], d[1], shi);  // (x3, x7)  bias 64
                    afr[(mi * GW + u) * 4 + c] = v;
                }
            }
    };
    auto load_a = [&](int ps, auto first_piece_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_piece_c)::value;  // pass 0: nothing older is in flight, the ring starts here
        const int g0 = (ps * p.wk + wki) * GW;
        if constexpr (XS) {
            // chunks of eight batch rows through the staging area (rows 16 mi + 8 half ..): one coalesced 1-KiB instruction per row (the
            // wave's 512 k of it), swizzled on the global side; a chunk is requested once the previous one has been read
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    // (straight-line on purpose: a wave-uniform skip of an absent chunk made hipcc keep two copies of the fragment registers;
                    //  an absent chunk requests nothing, reads stale bytes and merges nothing)
                    const int r0 = 16 * mi + 8 * half;
                    const int r1 = max(min(r0 + 8, M), r0 + 1);
                    const bool present = r0 < M;
                    for (int m = r0; m < r1 && present; ++m) {
                        const int j = (lane & 48) | ((lane & 15) ^ xs_f(m));
                        const int byte = min(256 * g0 + 16 * j, p.K * 2 - 16);
                        AWQ_BT_DMA16((uint32_t)((row_base + m) * p.K * 2 + byte), p.x, lds0 + (uint32_t)(xs_w + (m - r0) * 1024));
                    }
                    if (FIRST && mi == 0 && half == 0) {
                        request(0);
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LDM) : "memory");  // the chunk is older than the piece
                    } else if (present) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    const int m = min(max(16 * mi + n, r0), r1 - 1);  // (lanes of the other half / of rows >= M read some row of the chunk)
                    const int f = xs_f(m);
                    {
                        if (half == 0) {  // every lane takes what it reads (lanes 8-15: overwritten by the second half, or zeroed below)
                            const unsigned char* row = smem + xs_w + (m - r0) * 1024;
#pragma unroll
                            for (int u = 0; u < GW; ++u)
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    afr[(mi * GW + u) * 4 + c] = *reinterpret_cast<const u32x4*>(row + 16 * (16 * u + ((4 * kq + c) ^ f)));
                        } else {  // lanes 8-15 of every 16 only (none if the chunk is absent)
                            const unsigned long long mask = present ? 0xFF00FF00FF00FF00ull : 0ull;
                            unsigned long long save;  // (an SGPR pair each statement uses as scratch)
                            const uint32_t rb = lds0 + (uint32_t)(xs_w + (m - r0) * 1024);
                            uint32_t ad[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) ad[c] = rb + 16u * (uint32_t)((4 * kq + c) ^ f);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                AWQ_BT_LDS_READ16_MASKED(afr[(mi * GW + 0) * 4 + c], ad[c], 0, mask, save);
                                AWQ_BT_LDS_READ16_MASKED(afr[(mi * GW + 1) * 4 + c], ad[c], 256, mask, save);
                                AWQ_BT_LDS_READ16_MASKED(afr[(mi * GW + 2) * 4 + c], ad[c], 512, mask, save);
                                AWQ_BT_LDS_READ16_MASKED(afr[(mi * GW + 3) * 4 + c], ad[c], 768, mask, save);
                            }
                            AWQ_BT_LGKM16(afr, mi * 16);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the chunk has been read: the next one may land on it
                }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int u = 0; u < GW; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {  // (lanes of batch rows >= M repeat row M - 1's addresses: no more lines per instruction)
                        const int m = min(16 * mi + n, M - 1);
                        const int kk = min(128 * (g0 + u) + 32 * kq + 8 * c, p.K - 8);
                        AWQ_BT_LOAD16(afr[(mi * GW + u) * 4 + c], (AWQ_BT_DBG & 1) ? 0u : (uint32_t)(((row_base + m) * p.K + kk) * 2), p.x);
                    }
            if constexpr (FIRST) {
                request(0);
                AWQ_BT_WAIT16(afr, 0, LDM);
                if constexpr (NA > 16) AWQ_BT_WAIT16(afr, 16, LDM);
            } else {
                AWQ_BT_WAIT16(afr, 0, 0);
                if constexpr (NA > 16) AWQ_BT_WAIT16(afr, 16, 0);
            }
        }
        permute_a(g0);
        if constexpr (FAST) {
            // sx[mi][u][m = 4 kq + r] = sum of x over group g0 + u: a ones-MFMA chain (every column of D is the row sum); lane n == 0 of
            // each kq parks it in the (now dead) staging area behind the partial-tile buffers
            const u32x4 ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int u = 0; u < GW; ++u) {
                    float4_t sx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sx = mfma16(afr[(mi * GW + u) * 4 + c], ones, sx);
                    }
                    if (n == 0) *reinterpret_cast<float4_t*>(smem + xs_w + 4096 + ((mi * GW + u) * 4 + kq) * 16) = sx;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // the activations, the ring's first piece right behind them; the other RD - 1 pieces once the fragments are in registers
    // (their requests then issue in the shadow of the first piece's flight: profiles/r05_gemv_batch_trace_v2.txt -- issued up
    // front they delayed every wave's start by ~1 us at RD 2)
    load_a(0, std::true_type{});
    BT_STAMP(3);
    if constexpr (!LAZY) {
#pragma unroll
        for (int d = 1; d < RD; ++d) request(d);
    }
    BT_STAMP(1);

    // ---- stream
    float4_t* ystage = reinterpret_cast<float4_t*>(smem + p.ystage_off);  // [wt][tiles_max][MI][64 lanes]
    auto pbuf = [&](int w, int parity, int mi) { return reinterpret_cast<float4_t*>(smem + p.pbuf_off + w * p.pbuf_pitch + (parity * MI + mi) * 1024); };
    int u = 0, it = 0;  // live units requested so far; iterations (the parity of the partial-tile buffer)
    for (int ps = 0; ps < p.passes; ++ps) {
        if (ps > 0) {  // the next K range of the activations (they queue behind the ring: a drain)
            if constexpr (XS) {
                if (p.wk > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // nobody still reads the partial tiles the staging overwrites
            }
            load_a(ps, std::false_type{});
        }
        const int g0 = (ps * p.wk + wki) * GW;
        for (int tl = 0; tl < p.tiles_max; ++tl, ++it) {
            const bool live = tl < ntile;  // wave-uniform (and the same for the wk waves of an owner)
            float4_t acc[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = float4_t{0.f, 0.f, 0.f, 0.f};
#ifdef AWQ_GEMV_TRACE
            tph = wall_clock64();
#endif
            if (live) {
                if constexpr (LAZY) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // only this piece was in flight
                    BT_PHASE(12);
                    request_at(u + 1, tl + 1 < ntile ? ps : ps + 1, tl + 1 < ntile ? tl + 1 : 0, u + 1 < nunit);  // (its slot was read out before the previous iteration ended)
                    BT_PHASE(13);
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LDM * (RD - 1)) : "memory");
                    BT_PHASE(12);
                }
#ifdef AWQ_GEMV_TRACE
                if (u < 3) ts[4 + 2 * u] = wall_clock64();
#endif
                const unsigned char* slot = smem + ring + (u % RD) * PIECE_B;
                const uint32_t zw = *reinterpret_cast<const uint32_t*>(slot + PIECE_W + 1024 + 4 * lane);
                const u32x4 sq = *reinterpret_cast<const u32x4*>(slot + PIECE_W + 16 * lane);  // 8 scales: groups (g0 & ~7) ..
                u32x4 wq[GW];
#pragma unroll
                for (int uu = 0; uu < GW; ++uu) wq[uu] = *reinterpret_cast<const u32x4*>(slot + n * 256 + (((4 * uu + kq) ^ n) & 15) * 16);
                if constexpr (FAST) {
                    const int r4 = n >> 2, i4 = n & 3;
                    constexpr int T[4] = {0, 2, 3, 1};
                    const int tq = (r4 == 0 ? T[0] : r4 == 1 ? T[1] : r4 == 2 ? T[2] : T[3]) ^ kq;
                    u32x4 wf[GW];
#pragma unroll
                    for (int uu = 0; uu < GW; ++uu) wf[uu] = *reinterpret_cast<const u32x4*>(slot + 1024 * r4 + 16 * (16 * uu + 4 * tq + i4));
#pragma unroll
                    for (int uu = 0; uu < GW; ++uu) {
                        const uint32_t sw2 = *reinterpret_cast<const uint32_t*>(slot + PIECE_W + 32 * uu + 4 * (n >> 1));
                        const uint32_t zw2 = *reinterpret_cast<const uint32_t*>(slot + PIECE_W + 1024 + 32 * uu + 4 * (n >> 1));
                        const float sc = (float)u2h2(sw2)[n & 1];
                        const float zc = __builtin_fmaf(-16.f, sc, (float)u2h2(zw2)[n & 1]);
                        float4_t gacc[MI];
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) gacc[mi] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t q = wf[uu][c];
                            // nibble j of both int16 -> the pair (16 + w[8 j + 2 c], 16 + w[8 j + 2 c + 1]) under the exponent 2^4
                            const u32x4 b = {and_or(q << 6, 0x03C003C0u, 0x4C004C00u), and_or(q << 2, 0x03C003C0u, 0x4C004C00u),
                                             and_or(q >> 2, 0x03C003C0u, 0x4C004C00u), and_or(q >> 6, 0x03C003C0u, 0x4C004C00u)};
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) gacc[mi] = mfma16(afr[(mi * GW + uu) * 4 + c], b, gacc[mi]);
                        }
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            const float4_t sx = *reinterpret_cast<const float4_t*>(smem + xs_w + 4096 + ((mi * GW + uu) * 4 + kq) * 16);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[mi][r] = __builtin_fmaf(sc, gacc[mi][r], __builtin_fmaf(zc, sx[r], acc[mi][r]));
                        }
                    }
                } else if constexpr (AWQ_BT_DBG & 2) {
#pragma unroll
                    for (int uu = 0; uu < GW; ++uu) acc[0] += __builtin_bit_cast(float4_t, wq[uu]) + __builtin_bit_cast(float4_t, sq) + (float)zw;
                    if constexpr (AWQ_BT_DBG & 8) __builtin_amdgcn_s_sleep(21);  // (the consumption's duration without its instructions)
                } else if constexpr (AWQ_BT_DBG & 16) {  // the MFMAs on undecoded words: no decode VALU
#pragma unroll
                    for (int uu = 0; uu < GW; ++uu)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const u32x4 b = {wq[uu][c], wq[uu][(c + 1) & 3], sq[c], zw};
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma16(afr[(mi * GW + uu) * 4 + c], b, acc[mi]);
                        }
                } else
#pragma unroll
                for (int uu = 0; uu < GW; ++uu) {
                    const int gi = (g0 & 7) + uu;  // index of the group in the zero word and in the 8-scale chunk
                    const uint32_t z = (zw >> (4 * gi)) & 15u;
                    const half2_t zlo = u2h2(0x64006400u | z | (z << 16));         // (1024 + z, 1024 + z)
                    const half2_t zhi = u2h2(0x54005400u | (z << 4) | (z << 20));  // (64 + z, 64 + z)
                    // (g0 & 7 is 0 or 4: the piece's four scales are the low or the high 8 bytes of the chunk)
                    const uint32_t sw = (g0 & 4) ? sq[2 + (uu >> 1)] : sq[uu >> 1];
                    const float sc = (float)u2h2(sw)[uu & 1];
                    float4_t gacc[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) gacc[mi] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t ww = wq[uu][c], w8 = ww >> 8;
                        u32x4 b;  // the exact integers w - z
                        b[0] = h22u(u2h2(and_or(ww, 0x000F000Fu, 0x64006400u)) - zlo);
                        b[1] = h22u(u2h2(and_or(ww, 0x00F000F0u, 0x54005400u)) - zhi);
                        b[2] = h22u(u2h2(and_or(w8, 0x000F000Fu, 0x64006400u)) - zlo);
                        b[3] = h22u(u2h2(and_or(w8, 0x00F000F0u, 0x54005400u)) - zhi);
                        if constexpr (AWQ_BT_DBG & 32) {  // the decode without the MFMAs
                            gacc[0] += __builtin_bit_cast(float4_t, b);
                        } else
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) gacc[mi] = mfma16(afr[(mi * GW + uu) * 4 + c], b, gacc[mi]);
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mi][r] = __builtin_fmaf(sc, gacc[mi][r], acc[mi][r]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of the slot has returned before it is overwritten
                BT_PHASE(14);
                if constexpr (!LAZY) {
                    if constexpr (RD == 1) request_at(u + 1, tl + 1 < ntile ? ps : ps + 1, tl + 1 < ntile ? tl + 1 : 0, u + 1 < nunit);
                    else request(u + RD);
                    BT_PHASE(13);
                }
#ifdef AWQ_GEMV_TRACE
                if (u < 3) ts[5 + 2 * u] = wall_clock64();
#endif
                ++u;
            }
            // ---- the wk partial tiles meet in LDS; lane (n, kq) holds D[m = 4 kq + r][n] in acc[mi][r]
            if (p.wk > 1 && !(AWQ_BT_DBG & 4)) {
                if (live) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) pbuf(wave, it & 1, mi)[lane] = acc[mi];
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (live) {
                    const int f = wki + p.wk * lane;  // this wave's share of the tile's MI x 64 float4 slots
                    if (f < 64 * MI) {
                        const int mi = f >> 6, sl = f & 63;
                        float4_t s = pbuf(twi * p.wk, it & 1, mi)[sl];
                        for (int j = 1; j < p.wk; ++j) s += pbuf(twi * p.wk + j, it & 1, mi)[sl];
                        float4_t* dst = ystage + ((twi * p.tiles_max + tl) * MI + mi) * 64 + sl;
                        *dst = ps > 0 ? *dst + s : s;
                    }
                }
            } else if (live) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    float4_t* dst = ystage + ((twi * p.tiles_max + tl) * MI + mi) * 64 + lane;
                    *dst = ps > 0 ? *dst + acc[mi] : acc[mi];
                }
            }
#ifdef AWQ_GEMV_TRACE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            BT_PHASE(15);
#endif
        }
    }
    BT_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy requests past the end
    if (p.wk > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- y: the wk waves of an owner share its tiles; item (m, n) of a tile from slot (m / 4) * 16 + n, element m % 4
    for (int tl = wki; tl < ntile; tl += p.wk) {
        const int row0 = (t0 + tl) * 16;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const float* src = reinterpret_cast<const float*>(ystage + ((twi * p.tiles_max + tl) * MI + mi) * 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int item = lane + 64 * q, ml = item >> 4, nn = item & 15;
                const int m = 16 * mi + ml;
                if (m < M && row0 + nn < p.N) p.y[(int64_t)(row_base + m) * p.N + row0 + nn] = (half_t)src[((ml >> 2) * 16 + nn) * 4 + (ml & 3)];
            }
        }
    }
#ifdef AWQ_GEMV_TRACE
    BT_STAMP(11);
    if (p.trace && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) p.trace[((size_t)blockIdx.x * 8 + wave) * 16 + i] = ts[i];
    }
#endif
}

struct BatchPlan {
    int MI, RD, XS, xs_rows, wk, wt, passes, blocks, tiles_base, tiles_rem, tiles_max, rs, rows_part, brs;
    size_t ring, pbuf_pitch, ystage;
};

// form: 0 = auto, 1 = activations through the LDS staging area (XS), 2 = direct fragment loads
bool plan_batch(int M, int K, int N, int g, int form, int rd_req, BatchPlan* out, bool fast = false) {
    if (M < 1 || M > 128 || N < 1 || K < 128 || K % 128 || g != 128) return false;
    const int parts_req = (form >> 4) & 15;         // 0 = auto, 1 = every part inside the block, 2 .. 4 = that many parts across blocks
    form &= 15;
    if (fast && (N % 16 || form == 2)) return false;  // GEMVFast: whole 4-row bundles, staged form only
    if ((int64_t)N * K / 2 >= ((int64_t)1 << 31) || (int64_t)M * K * 2 >= ((int64_t)1 << 31)) return false;  // 32-bit lane offsets
    BatchPlan b;
    const int nparts = M > 64 ? 4 : (M > 32 ? 2 : 1);  // row parts of at most 32 rows (balanced: 33 rows = 17 + 16)
    // AUTO (profiles/r06_batch_parts.txt): the parts go ACROSS blocks -- two up to 64 rows, three up to 96 where the tile lists are long
    // (N > 4096: 240 blocks with 27 - 32 rows each beat 256 blocks with 20 - 24), else four; 17 .. 32 rows also split in two where a
    // block would own a single tile (N <= 4096: 8.3 vs 9.6 us at 4096 x 4096, 19.8 vs 21.6 at 11008 x 4096, M = 32)
    const int tiles_all = (N + 15) / 16;
    int auto_brs = nparts;
    if (M > 64 && M <= 96 && tiles_all > 256) auto_brs = 3;
    if (M > 16 && M <= 32 && tiles_all <= 256) auto_brs = 2;
    b.brs = parts_req >= 2 ? parts_req : (parts_req == 1 ? 1 : auto_brs);
    b.rs = b.brs > 1 ? 1 : nparts;
    b.rows_part = (M + b.rs * b.brs - 1) / (b.rs * b.brs);
    if (b.rows_part > 32) return false;
    const int MP = b.rows_part;                     // rows a wave group holds
    b.MI = MP > 16 ? 2 : 1;
    const int G = K / 128;
    int wk = 1;
    while (wk < 8 / b.rs && wk * GW < G) wk *= 2;  // (row parts take wave groups: at most 8 / rs waves side by side on a tile)
    b.wk = wk;
    b.wt = 8 / wk;
    b.passes = (G + wk * GW - 1) / (wk * GW);
    const int tiles = (N + 15) / 16;
    const int towners = b.wt / b.rs;                // tile owners per block
    const int want = (tiles + towners - 1) / towners;
    int nob = want < 256 ? want : 256;              // blocks with different tiles
    if (b.brs > 1) {                                // whole rounds of the eight XCDs, brs residents of an XCD per tile list
        const int q = (want + 7) / 8 < 32 / b.brs ? (want + 7) / 8 : 32 / b.brs;
        nob = 8 * q;
    }
    b.blocks = nob * b.brs;
    const int owners = nob * towners;
    b.tiles_base = tiles / owners;
    b.tiles_rem = tiles % owners;
    b.tiles_max = b.tiles_base + (b.tiles_rem ? 1 : 0);
    b.ystage = (size_t)b.wt * b.tiles_max * b.MI * 1024;
    const size_t plain = wk > 1 ? (size_t)2 * b.MI * 1024 : 0;
    const size_t budget = 160 * 1024;
    // XS: the staging area holds min(M, 8) batch rows per wave (more rows arrive in chunks of eight); the partial-tile buffers live
    // in it afterwards
    const int rows = MP < 8 ? MP : 8;
    size_t staged = (size_t)rows * 1024 > plain ? (size_t)rows * 1024 : plain;
    if (fast && staged < (size_t)4096 + 256 * b.MI) staged = (size_t)4096 + 256 * b.MI;  // (the group sums sit behind the partial-tile buffers)
    const bool xs = form != 2 && b.ystage + 8 * staged + (size_t)8 * PIECE_B <= budget;
    if ((form == 1 || fast) && !xs) return false;
    b.XS = xs ? 1 : 0;
    b.xs_rows = rows;
    b.pbuf_pitch = xs ? staged : plain;
    const size_t fixed = b.ystage + 8 * b.pbuf_pitch;
    // ring slots, AUTO: two (lazy) wherever they fit, else one -- the per-wave phase times (profiles/r06_gemv_batch_trace.txt) show no
    // wave waiting for a piece in either form: a unit is bound by what the two waves of a SIMD issue (requests, decode, MFMA, exchange);
    // one slot and two measure within 1 - 3 % of each other (profiles/r06_batch_parts.txt)
    int rd = (rd_req >= 1 && rd_req <= 3) ? rd_req : 2;
    while (rd > 1 && fixed + (size_t)8 * rd * PIECE_B > budget) --rd;
    if (fixed + (size_t)8 * rd * PIECE_B > budget) return false;
    b.RD = rd;
    b.ring = (size_t)8 * rd * PIECE_B;
    *out = b;
    return true;
}

}  // namespace

#ifdef AWQ_GEMV_TRACE
static unsigned long long* g_batch_trace = nullptr;
extern "C" __attribute__((visibility("default"))) void awq_debug_set_trace_batch(void* dev_buf) {
    g_batch_trace = static_cast<unsigned long long*>(dev_buf);
}
#endif

bool awq_gemv_batch_supports(int M, int K, int N, int g) {
    BatchPlan b;
    return plan_batch(M, K, N, g, 0, 0, &b);
}

namespace {
int launch_batch(const void* x, const void* qweight, const void* scales, const void* qzeros, void* y, int M, int K, int N, int g, int ZW, int GP,
                 int form, int depth, bool fast, hipStream_t st) {
    BatchPlan b;
    if (!plan_batch(M, K, N, g, form, depth, &b, fast)) return AWQ_ERR_UNSUPPORTED;
    BatchParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N;
    p.KW = K / 8; p.ZW = ZW; p.SW = 8 * ZW; p.GP = GP;
    p.G = K / 128;
    p.wk = b.wk; p.wt = b.wt; p.passes = b.passes;
    p.tiles_base = b.tiles_base; p.tiles_rem = b.tiles_rem; p.tiles_max = b.tiles_max;
    p.ring_off = 0;
    p.pbuf_off = (int)b.ring;
    p.pbuf_pitch = (int)b.pbuf_pitch;
    p.xs_rows = b.xs_rows;
    p.rs = b.rs; p.rows_part = b.rows_part; p.brs = b.brs;
    p.ystage_off = (int)(b.ring + 8 * b.pbuf_pitch);
#ifdef AWQ_GEMV_TRACE
    p.trace = g_batch_trace;
#else
    p.trace = nullptr;
#endif
    const size_t lds = b.ring + 8 * b.pbuf_pitch + b.ystage;
#define AWQ_BT_CASE(MIV, RDV, XSV, FASTV)                                                                                               \
    if (b.MI == MIV && b.RD == RDV && b.XS == XSV && fast == FASTV) {                                                                   \
        static std::atomic<unsigned long long> opted{0};                                                                                \
        if (!awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_batch_kernel<MIV, RDV, (XSV != 0), FASTV>), opted)) return AWQ_ERR_LAUNCH; \
        hipLaunchKernelGGL((awq_gemv_batch_kernel<MIV, RDV, (XSV != 0), FASTV>), dim3((unsigned)b.blocks), dim3(512), lds, st, p);      \
        return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;                                                               \
    }
    AWQ_BT_CASE(1, 1, 0, false) AWQ_BT_CASE(1, 2, 0, false) AWQ_BT_CASE(1, 3, 0, false) AWQ_BT_CASE(2, 1, 0, false) AWQ_BT_CASE(2, 2, 0, false)
    AWQ_BT_CASE(2, 3, 0, false) AWQ_BT_CASE(1, 1, 1, false) AWQ_BT_CASE(1, 2, 1, false) AWQ_BT_CASE(1, 3, 1, false) AWQ_BT_CASE(2, 1, 1, false)
    AWQ_BT_CASE(2, 2, 1, false) AWQ_BT_CASE(2, 3, 1, false)
    AWQ_BT_CASE(1, 1, 1, true) AWQ_BT_CASE(1, 2, 1, true) AWQ_BT_CASE(2, 1, 1, true) AWQ_BT_CASE(2, 2, 1, true)
#undef AWQ_BT_CASE
    return AWQ_ERR_UNSUPPORTED;
}
}  // namespace

// form: how the activations reach the registers (0 = auto, 1 = LDS staging area, 2 = direct fragment loads); depth: ring slots per wave
// (1 | 3: that many pieces in flight; 2: the next piece is requested when a piece lands; 0 = auto)
int awq_launch_gemv_batch(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* y,
                          int M, int K, int N, int g, int ZW, int form, int depth, hipStream_t st) {
    if (ZW * 8 < K / 128) return AWQ_ERR_BAD_SHAPE;
    return launch_batch(x, qweight, scales, qzeros, y, M, K, N, g, ZW, 0, form, depth, false, st);
}

bool awq_gemv_batch_fast_supports(int M, int K, int N, int g) {
    BatchPlan b;
    return plan_batch(M, K, N, g, 0, 0, &b, true);
}

// the same kernel on the GEMVFast layout's buffers (qweight int16 [N/4, K], scales / qzeros fp16 [group_rows, N]); depth 1 | 2
int awq_launch_gemv_batch_fast(const uint16_t* x, const int16_t* qweight, const uint16_t* scales, const uint16_t* qzeros, uint16_t* y,
                               int M, int K, int N, int g, int group_rows, int depth, hipStream_t st) {
    if (group_rows < K / 128) return AWQ_ERR_BAD_SHAPE;
    const int parts = (depth >> 4) & 15;  // (row parts across blocks: see plan_batch)
    depth &= 15;
    if (depth > 2) depth = 2;
    return launch_batch(x, qweight, scales, qzeros, y, M, K, N, g, 0, group_rows, parts << 4, depth, true, st);
}
