// gemm_regb.hip -- prefill GEMM on the GEMM layout with the WEIGHT operand decoded in registers, gfx950.
//
// Replaces the >= 1024-token branch of awq/modules/linear/gemm.py:48-54 (dequantize_weights_cuda + torch.matmul)
// in ONE kernel, like gemm_tiled.hip, but without ever writing a dequantised weight to LDS: the limiter of the
// LDS-tiled kernel at M = 16384 is the LDS itself (per K step a 128 x 256 block moves 128 KB of fragments out of
// LDS and 48 KB in for 1024 clocks of MFMA).  Here
//   * a lane of a wave owns ONE packed word column (8 logical columns) and the 8 K rows 8*kb .. 8*kb+7 of a
//     32-row slab: 8 dword loads straight from global memory (L2 hits after the first M tile) are everything
//     the B side of FOUR 16-column MFMA tiles needs.  v_perm_b32 puts byte b of two consecutive rows side by
//     side, one v_and_or per nibble then yields the fp16 pair (1024 + w[k], 1024 + w[k+1]) (low nibble,
//     exponent 2^10) or (64 + w[k], 64 + w[k+1]) (high nibble, exponent 2^6) of ONE logical column -- already
//     the K-pair layout of an MFMA B register -- and (t - (bias + z)) * s gives exactly the fp16 weight the
//     reference materialises ((w - z) * s, one rounding: awq/utils/packing_utils.py:98-100).
//     3.5 VALU ops per B register, 56 per 32 MFMAs of a 128 x 64 wave tile, no transposition, no LDS.
//   * bytes (ph, ph + 2) of a word are logical columns 4 ph .. 4 ph + 3 (ORDER = [0,2,4,6,1,3,5,7]): the two
//     waves that share a word column split it by `ph`, and each stores 8 contiguous bytes per output row.
//   * activations go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 16 B per lane, no VGPRs, no ds_write)
//     in PAIRS of 64-wide K steps, one pair ahead, four buffers, one barrier per pair; 128 B row pitch with the
//     16-byte chunks XOR-swizzled by (row >> 1) & 7 on the GLOBAL side of the DMA, so the ds_read_b128 of an
//     MFMA A fragment (8 consecutive k of one row per lane) is bank-conflict free (SQ_LDS_BANK_CONFLICT = 0).
//   * every memory operation of the K loop is inline asm with hand-counted s_waitcnt vmcnt(N): vector-memory
//     operations retire in order, so "N operations were issued after the one I need" is exact, while hipcc's own
//     bookkeeping gives up (vmcnt(0)) as soon as an LDS-DMA and an ordinary load are pending together, and
//     drains the DMA queue in front of every LDS read it can see.
//   * block = 128 x 256 (4 waves along N, each 128 x 64 -> 128 accumulator registers; two blocks per CU, which
//     drift apart and fill each other's barrier gaps) or 256 x 256 (8 waves); blockIdx -> tile is XCD aware:
//     the 32 blocks an XCD runs at a time form an 8 x 4 patch of tiles, so its L2 holds 8 activation slabs
//     and 4 weight slabs.
//
// Roofline: MFMA (2 M K N flops, dense fp16 peak 2.5 PFLOP/s); algorithmic bytes as in gemm_tiled.hip.
// Measured (r02, 4096 x 11008 and 11008 x 4096, M = 16384): 1050-1090 TFLOP/s (gemm_tiled: 910-980; HIP dequant +
// vendor GEMM: 950-1280); SQ counters and the switch-off experiments behind the design: profiles/r02_regb_*.txt.
// Constraints (the launcher returns AWQ_ERR_UNSUPPORTED otherwise and the caller falls back to gemm_tiled):
// K % 64 == 0, group_size % 64 == 0, N % 8 == 0, M * K < 2^31 elements.
#include <cstdlib>

#include "awq_device.h"
#include "awq_internal.h"
#include "awq_mfma_decode.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gl_ptr_t;

struct RegbParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    const half_t* bias;
    half_t* y;
    int M, K, N, g;
    int tiles_m, tiles_n;
    int mp, patches;  // patches of pm x pn tiles: mp along M, `patches` in all
    int pm, pn;       // tile patch one XCD runs at a time (pm * pn = 32)
};

constexpr int BK = 64, NBUF = 4, BN = 256;
constexpr int PM = 8, PN = 4;  // tile patch per XCD round

// DBG (tools/regb_experiments.py, -DAWQ_REGB_EXPERIMENTS builds only; results are WRONG, timing only): 1 = weights fetched
// once, 2 = activations fetched once, 4 = no barrier, 8 = no decode arithmetic, 16 = every output row stored into rows 0..127 (stores issued, nothing reaches HBM)
template <int WGM, int DBG = 0>  // waves along M: BM = 128 * WGM
__global__ __launch_bounds__(WGM * 256, 2) void awq_gemm_regb_kernel(RegbParams p) {
    constexpr int BM = 128 * WGM;
    constexpr int A_BUF = BM * BK * 2;            // bytes of one activation K step
    constexpr int PIECES = BM * 8 / 64 / (4 * WGM);  // 1 KiB DMA pieces per wave per K step (= 4)
    constexpr int VM_PER_ITER = PIECES + 16 + 2;  // vector-memory ops a wave issues per K step
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [NBUF][BM rows][8 chunks of 16 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, set = wn >> 1, ph = wn & 1;
    const int j = lane & 15, kb = lane >> 4;

    // ---- XCD-aware tile assignment: block b runs on XCD b % 8; 32 consecutive blocks of an XCD = one patch
    int mt, nt;
    {
        const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
        const int patch = (i / (p.pm * p.pn)) * 8 + xcd, local = i % (p.pm * p.pn);
        if (patch >= p.patches) return;
        mt = (patch % p.mp) * p.pm + local % p.pm;
        nt = (patch / p.mp) * p.pn + local / p.pm;
        if (mt >= p.tiles_m || nt >= p.tiles_n) return;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int NW = p.N >> 3;
    const int T = p.K / BK;

    // ---- B side: this lane's word column and rows
    const int wcol = (n0 >> 3) + set * 16 + j;
    const bool colok = wcol < NW;
    const uint32_t row_bytes = (uint32_t)NW * 4u;
    const uint32_t w_voff = colok ? ((uint32_t)(8 * kb) * (uint32_t)NW + (uint32_t)wcol) * 4u : OOB;
    const uint32_t z_voff = colok ? (uint32_t)wcol * 4u : OOB;
    const uint32_t s_voff = colok ? ((uint32_t)wcol * 8u + 4u * (uint32_t)ph) * 2u : OOB;
    const uint32_t g_magic = (uint32_t)(0x100000000ull / (uint32_t)p.g) + 1u;  // (k * g_magic) >> 32 == k / g for k < 2^16

    // Weight words, zero points and scales are requested with inline asm and waited for with COUNTED s_waitcnt
    // statements that name the registers they release (so nothing that uses them can be scheduled above): hipcc's own
    // bookkeeping falls back to vmcnt(0) whenever an LDS-DMA and an ordinary load are pending together (it takes them
    // for different, mutually unordered event classes), which drains the activation prefetch at every use of a weight.
    struct BRegs {
        uint32_t w[2][8];
        uint32_t z;
        u32x2 s;
    };
    auto srd = [](const void* base, uint32_t bytes) -> u32x4 {  // raw buffer descriptor, stride 0, bounds checked
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
    };
    const u32x4 wsrd = srd(p.qweight, (uint32_t)p.K * row_bytes);
    const u32x4 zsrd = srd(p.qzeros, (uint32_t)(p.K / p.g) * row_bytes);
    const u32x4 ssrd = srd(p.scales, (uint32_t)(p.K / p.g) * (uint32_t)p.N * 2u);
    const u32x4 xsrd = srd(p.x, (uint32_t)((int64_t)p.M * p.K * 2));
// Eight weight words / the group's zero word and scales in ONE asm statement each, opened by s_nop 4: an SGPR written by
// the SALU (the scalar offsets, a rematerialised descriptor) needs five wait states before a VMEM instruction may read
// it, and hipcc pads nothing for the operands of an asm statement -- without the nop a load now and then used the
// PREVIOUS value of its offset register (rare wrong tiles that came and went with the schedule).
#define AWQ_BLOAD8(W, voff, rs, so)                                                                                          \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %8, %9, %10 offen\n\tbuffer_load_dword %1, %8, %9, %11 offen\n\t"                \
                 "buffer_load_dword %2, %8, %9, %12 offen\n\tbuffer_load_dword %3, %8, %9, %13 offen\n\t"                          \
                 "buffer_load_dword %4, %8, %9, %14 offen\n\tbuffer_load_dword %5, %8, %9, %15 offen\n\t"                          \
                 "buffer_load_dword %6, %8, %9, %16 offen\n\tbuffer_load_dword %7, %8, %9, %17 offen"                              \
                 : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]), "=v"(W[4]), "=v"(W[5]), "=v"(W[6]), "=v"(W[7])                  \
                 : "v"(voff), "s"(rs), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5]), "s"(so[6]), "s"(so[7]))
#define AWQ_BLOADZS(Z, S2, zvoff, zrs, zso, svoff, srs, sso)                                                                  \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %3, %4 offen\n\tbuffer_load_dwordx2 %1, %5, %6, %7 offen"                        \
                 : "=v"(Z), "=v"(S2)                                                                                        \
                 : "v"(zvoff), "s"(zrs), "s"(zso), "v"(svoff), "s"(srs), "s"(sso))
#define AWQ_BLOAD1(dst, voff, rs, soff) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff))
#define AWQ_BLOAD2(dst, voff, rs, soff) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff))
    constexpr int B_OPS = 18;  // vector-memory operations of one fetch_b
    auto fetch_b = [&](BRegs& R, int t) {
        const uint32_t k0 = (uint32_t)t * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint32_t so[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) so[r] = (k0 + 32u * kk + r) * row_bytes;
            AWQ_BLOAD8(R.w[kk], w_voff, wsrd, so);
        }
        const uint32_t grp = __umulhi(k0, g_magic);
        const uint32_t zo = grp * row_bytes, so2 = grp * (uint32_t)p.N * 2u;
        AWQ_BLOADZS(R.z, R.s, z_voff, zsrd, zo, s_voff, ssrd, so2);
    };
    // everything up to and including R's requests has returned once at most `newer` later operations are outstanding
#define AWQ_WAIT_B(R, newer)                                                                                              \
    asm volatile("s_waitcnt vmcnt(" #newer ")"                                                                            \
                 : "+v"(R.w[0][0]), "+v"(R.w[0][1]), "+v"(R.w[0][2]), "+v"(R.w[0][3]), "+v"(R.w[0][4]), "+v"(R.w[0][5]),   \
                   "+v"(R.w[0][6]), "+v"(R.w[0][7]), "+v"(R.w[1][0]), "+v"(R.w[1][1]), "+v"(R.w[1][2]), "+v"(R.w[1][3]),   \
                   "+v"(R.w[1][4]), "+v"(R.w[1][5]), "+v"(R.w[1][6]), "+v"(R.w[1][7]), "+v"(R.z), "+v"(R.s))

    // ---- A side: LDS-DMA pieces of this wave: piece q = PIECES * wave + u covers rows 8q .. 8q+7, lane L writes
    // LDS chunk (row 8q + L/8, slot L%8) and fetches global chunk kc = slot ^ ((row >> 1) & 7)
    // (buffer form: a 32-bit lane offset per piece and the K step in the scalar offset -- half the address registers
    // of the flat form.  Rows past M are clamped to the last row: loaded, multiplied, never stored.)
    uint32_t a_voff[PIECES];
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
        const int q = PIECES * wave + u;
        const int row = 8 * q + (lane >> 3), slot = lane & 7;
        const int kc = slot ^ ((row >> 1) & 7);
        const int grow = min(m0 + row, p.M - 1);
        a_voff[u] = (uint32_t)(((int64_t)grow * p.K + 8 * kc) * 2);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;  // LDS byte address of the dynamic segment
    const uint32_t a_dst0 = lds0 + (uint32_t)(PIECES * wave) * 1024u;
    // s_nop: an SALU write of M0 needs one wait state before an LDS-DMA reads it
#define AWQ_DMA16(ldsaddr, voff, rs, soff)                                                                    \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsaddr), "v"(voff), \
                 "s"(rs), "s"(soff)                                                                            \
                 : "m0")
    auto fetch_a = [&](int t, int buf) {
        const uint32_t so = (uint32_t)t * (BK * 2);
#pragma unroll
        for (int u = 0; u < PIECES; ++u) {
            const uint32_t dst = a_dst0 + (uint32_t)(buf * A_BUF + u * 1024);
            AWQ_DMA16(dst, a_voff[u], xsrd, so);
        }
    };
    // A fragment of row tile i, K sub-step kk: row wm*128 + 16 i + j, chunk (4 kk + kb) ^ ((row >> 1) & 7)
    const int a_row_off = (wm * 128 + j) * 128;
    const int hl = (j >> 1) & 7;

    float4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = float4_t{0.f, 0.f, 0.f, 0.f};

    // byte selectors: byte `ph` (-> P0) / `ph + 2` (-> P1) of two consecutive rows into bits 0-7 and 16-23
    const uint32_t sel0 = 0x0C000C00u | (uint32_t)ph | ((uint32_t)(4 + ph) << 16);
    const uint32_t sel1 = sel0 + 0x00020002u;

    // A fragments are read with inline asm: for an LDS access it can see, hipcc first drains every LDS-DMA in flight
    // (s_waitcnt vmcnt(0): it cannot tell the buffer being read from the ones being filled), which would expose the
    // whole prefetch round trip at the first ds_read of every step.  The asm reads are ordered by hand: all eight are
    // issued, the B decode runs in their shadow, ONE wait-only statement that names the registers, then the MFMAs.
    const uint32_t a_base = lds0 + (uint32_t)a_row_off;
#define AWQ_LDS_READ16(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))

    auto compute = [&](BRegs& R, int buf, auto&& issue_next) {
        half2_t zm[4], sd[4];
        issue_next();  // next step's requests first (their round trip is the longest thing in the step), then wait for R
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint32_t aa = a_base + (uint32_t)(buf * A_BUF) + (uint32_t)(((4 * kk + kb) ^ hl) << 4);
            u32x4v af[4], ag[4];  // rows 0-63 of the wave tile; rows 64-127 are requested once these are in the MFMAs
            AWQ_LDS_READ16(af[0], aa, 0);
            AWQ_LDS_READ16(af[1], aa, 2048);
            AWQ_LDS_READ16(af[2], aa, 4096);
            AWQ_LDS_READ16(af[3], aa, 6144);
            if (kk == 0) {
                // zero points and scales of this step's group, duplicated into both halves: columns 4 ph + {0, 1, 2, 3}
                //   c = 0: byte ph low nibble, c = 1: byte ph+2 low, c = 2: byte ph high, c = 3: byte ph+2 high
                const uint32_t zp0 = __builtin_amdgcn_perm(R.z, R.z, sel0), zp1 = __builtin_amdgcn_perm(R.z, R.z, sel1);
                zm[0] = u2h2(and_or(zp0, 0x000F000Fu, 0x64006400u));
                zm[1] = u2h2(and_or(zp1, 0x000F000Fu, 0x64006400u));
                zm[2] = u2h2(and_or(zp0, 0x00F000F0u, 0x54005400u));
                zm[3] = u2h2(and_or(zp1, 0x00F000F0u, 0x54005400u));
                const half2_t s01 = u2h2(R.s[0]), s23 = u2h2(R.s[1]);  // broadcasts: op_sel on the packed multiply, no registers
                sd[0] = __builtin_shufflevector(s01, s01, 0, 0);
                sd[1] = __builtin_shufflevector(s01, s01, 1, 1);
                sd[2] = __builtin_shufflevector(s23, s23, 0, 0);
                sd[3] = __builtin_shufflevector(s23, s23, 1, 1);
            }
            u32x4v bf[4];
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const uint32_t p0 = __builtin_amdgcn_perm(R.w[kk][2 * rp + 1], R.w[kk][2 * rp], sel0);
                const uint32_t p1 = __builtin_amdgcn_perm(R.w[kk][2 * rp + 1], R.w[kk][2 * rp], sel1);
                if constexpr (DBG & 8) {
                    bf[0][rp] = p0; bf[1][rp] = p1; bf[2][rp] = p0 ^ h22u(zm[0]); bf[3][rp] = p1 ^ h22u(sd[0]);
                    continue;
                }
                bf[0][rp] = h22u((u2h2(and_or(p0, 0x000F000Fu, 0x64006400u)) - zm[0]) * sd[0]);
                bf[1][rp] = h22u((u2h2(and_or(p1, 0x000F000Fu, 0x64006400u)) - zm[1]) * sd[1]);
                bf[2][rp] = h22u((u2h2(and_or(p0, 0x00F000F0u, 0x54005400u)) - zm[2]) * sd[2]);
                bf[3][rp] = h22u((u2h2(and_or(p1, 0x00F000F0u, 0x54005400u)) - zm[3]) * sd[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]));
            AWQ_LDS_READ16(ag[0], aa, 8192);
            AWQ_LDS_READ16(ag[1], aa, 10240);
            AWQ_LDS_READ16(ag[2], aa, 12288);
            AWQ_LDS_READ16(ag[3], aa, 14336);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i][c] = mfma16(af[i], bf[c], acc[i][c]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ag[0]), "+v"(ag[1]), "+v"(ag[2]), "+v"(ag[3]));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[4 + i][c] = mfma16(ag[i], bf[c], acc[4 + i][c]);
        }
    };

    // ---- pipeline.  Activations travel in PAIRS of K steps (four LDS buffers, one barrier per pair), requested two
    // pairs... one pair ahead: at the top of even step t for steps t+2, t+3; weights one step ahead in registers.
    // Vector-memory operations retire in order, so every wait is a count of what was issued after the thing needed:
    //   even step t:  vmcnt(18)   only B(t) (issued during t-1) may be pending -> the pair (t, t+1), requested at
    //                             the top of t-2, is in LDS;  s_barrier publishes every wave's pieces
    //                 issue B(t+1) [18], activations for t+2, t+3 [8];  vmcnt(26) -> B(t) is in registers
    //   odd step t:   issue B(t+1) [18];  vmcnt(26) (the 8 DMA pieces of the even step are newer than B(t)) -> B(t)
    // Past the last step the requests repeat the last step's (static counts, a few KB of redundant traffic per block).
    BRegs B0, B1;
    fetch_a(0, 0);
    fetch_a(min(1, T - 1), 1);
    fetch_b(B0, 0);
    auto step = [&](int t, BRegs& cur, BRegs& nxt) {
        if (!(t & 1)) {
            if constexpr (!(DBG & 4)) {
                asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        auto issue_next = [&]() {
            if (!(DBG & 1) || t == 0) fetch_b(nxt, min(t + 1, T - 1));
            if (!(t & 1) && (!(DBG & 2) || t == 0)) {
                fetch_a(min(t + 2, T - 1), (t + 2) & 3);
                fetch_a(min(t + 3, T - 1), (t + 3) & 3);
            }
            if constexpr (DBG & 1) {
                if (t == 0) AWQ_WAIT_B(B0, 0);
            } else {
                AWQ_WAIT_B(cur, 26);
            }
        };
        compute((DBG & 1) ? B0 : cur, t & 3, issue_next);
    };
    int t = 0;
    for (; t + 2 <= T; t += 2) {
        step(t, B0, B1);
        step(t + 1, B1, B0);
    }
    if (t < T) step(t, B0, B1);
    // no DMA may land in LDS after this block has released it -- and the last weight request (a repeat nobody consumes)
    // must have landed before its destination registers, dead to the compiler since they were requested, are re-used:
    // the waits NAME both register sets
    AWQ_WAIT_B(B0, 0);
    AWQ_WAIT_B(B1, 0);

    // ---- epilogue: lane (j, kb) holds rows 16 i + 4 kb + e, columns 8 j + 4 ph + c of its wave tile
    const int col = n0 + set * 128 + 8 * j + 4 * ph;
    if (col >= p.N) return;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const half4_t bv = *reinterpret_cast<const half4_t*>(p.bias + col);
#pragma unroll
        for (int c = 0; c < 4; ++c) b4[c] = (float)bv[c];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = m0 + wm * 128 + 16 * i + 4 * kb + e;
            if (row < p.M) {
                half4_t o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (half_t)(acc[i][c][e] + b4[c]);
                *reinterpret_cast<half4_t*>(p.y + (int64_t)((DBG & 16) ? (row & 127) : row) * p.N + col) = o;
            }
        }
}

}  // namespace

bool awq_gemm_regb_supports(int M, int K, int N, int g) {
    return M > 0 && K > 0 && N > 0 && (int64_t)M * K * 2 < ((int64_t)1 << 32) && K % BK == 0 && g % BK == 0 && K % g == 0 && N % 8 == 0 && K < 65536 &&
           (int64_t)K * (N / 8) * 4 < ((int64_t)1 << 31) && (int64_t)(K / g) * N * 2 < ((int64_t)1 << 31);
}

// bm: 128 | 256 rows per block tile, 0 = auto
int awq_launch_gemm_regb(const AwqGemmArgs& a, int bm) {
    if (!awq_gemm_regb_supports(a.M, a.K, a.N, a.g)) return AWQ_ERR_UNSUPPORTED;
    if (bm == 0) bm = 128;  // two independent 4-wave blocks per CU drift apart and fill each other's barrier gaps: 1050-1090 TF vs 1010-1045 for one 8-wave block (r02, M = 16384)
    if (bm != 128 && bm != 256) return AWQ_ERR_UNSUPPORTED;
    RegbParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.bias = reinterpret_cast<const half_t*>(a.bias);
    p.y = reinterpret_cast<half_t*>(a.y);
    p.M = a.M; p.K = a.K; p.N = a.N; p.g = a.g;
    p.tiles_m = (a.M + bm - 1) / bm;
    p.tiles_n = (a.N + BN - 1) / BN;
    p.pm = PM; p.pn = PN;
#ifdef AWQ_REGB_EXPERIMENTS
    if (const char* e = getenv("AWQ_REGB_PM")) { p.pm = atoi(e); p.pn = 32 / p.pm; }
#endif
    p.mp = (p.tiles_m + p.pm - 1) / p.pm;
    p.patches = p.mp * ((p.tiles_n + p.pn - 1) / p.pn);
    const int grid = ((p.patches + 7) / 8) * 8 * (p.pm * p.pn);
    const size_t lds = (size_t)NBUF * bm * BK * 2;
#ifdef AWQ_REGB_EXPERIMENTS
    {
        const char* e = getenv("AWQ_REGB_DBG");
        const int dbg = e ? atoi(e) : 0;
        if (dbg) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (dbg == 1) hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 1>), dim3(grid), dim3(512), lds, a.stream, p);
            else if (dbg == 2) hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 2>), dim3(grid), dim3(512), lds, a.stream, p);
            else if (dbg == 4) hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 4>), dim3(grid), dim3(512), lds, a.stream, p);
            else if (dbg == 8) hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 8>), dim3(grid), dim3(512), lds, a.stream, p);
            else if (dbg == 16) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 16>), dim3(grid), dim3(512), lds, a.stream, p);
            } else hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 15>), dim3(grid), dim3(512), lds, a.stream, p);
            return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
        }
    }
#endif
    if (bm == 256) {
        static std::atomic<unsigned long long> opted{0};  // > 64 KB of dynamic LDS: once per device (awq_internal.h)
        if (!awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2>), opted)) return AWQ_ERR_LAUNCH;
        hipLaunchKernelGGL((awq_gemm_regb_kernel<2>), dim3(grid), dim3(512), lds, a.stream, p);
    } else {
        hipLaunchKernelGGL((awq_gemm_regb_kernel<1>), dim3(grid), dim3(256), lds, a.stream, p);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
