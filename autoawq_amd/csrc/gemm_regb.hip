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
//
// NK = true (round 4): the SAME kernel on the GEMV layout's own buffers -- qweight [N, K/8] int32 (nibble i of word c = w[n, 8c+i]),
// qzeros [N, ZW], scales [N, 8 ZW] -- so that WQLinear_GEMV's prefill-sized calls (awq/modules/linear/gemv.py:168-180,
// gemmv2_forward_cuda) read the checkpoint's buffers instead of a second, GEMM-layout copy of every matrix (VERDICT r03
// weak 9).  Here a packed word IS the K-contiguous run an MFMA B register wants: lane (j, kb) of a wave owns the four output
// columns 4 j + c and the 8 K values of words 2 kb, 2 kb + 1 of each 64-wide step (one dwordx2 per column and step, 4 instead of
// 16 weight loads per step); byte b of a word holds the natural pair (k = 2 b, 2 b + 1): v_perm_b32 copies it into bytes 0 and 2,
// ONE v_and_or (mask 0x00F0000F, exponents 2^10 | 2^6) makes the fp16 pair (1024 + w, 64 + w'), and (t - (bias + z)) * s is again
// exactly the reference's fp16 weight.  2 VALU ops per B register + subtract + multiply; the A side, the pipeline and the
// counted waits are shared (a step issues 12 instead of 18 weight-side memory operations).
#include <cstdlib>
#include <type_traits>

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
    int KW, ZW, SW;   // NK form: words per qweight row, zero words per row, scale halfs per row
    // GROUPED form (MoE prefill): rows [seg[e], seg[e + 1]) of x / y belong to expert e, whose tensors sit e strides further
    const int* seg;   // [E + 1] row offsets on the DEVICE (nothing about the routing is read back to the host)
    int E;
    long long w_stride, z_stride, s_stride;  // bytes between two experts' qweight / qzeros / scales
    // GROUPED, round 6: the sort stays an index list -- no gathered copy of x, no scatter pass over y
    const int* row_map;     // [P] pair index of sorted row r (null: rows of x / y are the sorted rows themselves)
    const float* pair_w;    // [pairs] routing weights by PAIR index: y row := fp16(acc * pair_w[pair]) (null: none)
    uint32_t x_div_magic;   // GATHER: x row of sorted row r = row_map[r] / x_div ((v * magic) >> 32; 0: x_div == 1)
    int gather, scatter;    // gather: x rows through row_map; scatter: y rows to row_map[r]
};

constexpr int BK = 64, NBUF = 4, BN = 256;
constexpr int PM = 8, PN = 4;  // tile patch per XCD round

// DBG (tools/regb_experiments.py, -DAWQ_REGB_EXPERIMENTS builds only; results are WRONG, timing only): 1 = weights fetched
// once, 2 = activations fetched once, 4 = no barrier, 8 = no decode arithmetic, 16 = every output row stored into rows 0..127 (stores issued, nothing reaches HBM)
// GROUPED: the M tiles are dealt over the experts of a sorted token list (device-side row offsets `seg`): virtual tile `mt` walks
// expert 0's ceil(rows / BM) tiles, then expert 1's, ...; a block past the last expert's tiles exits.
// FZ (round 6, GEMM-layout words only): the GEMVFast format's arithmetic -- `qzeros` is the fp16 tensor -(s z) [K/g, N] of
// awq/modules/linear/gemv_fast.py:175-181 (same orientation as `scales`) and a weight is fp16(w s + qzeros), ONE packed fma per
// register pair instead of subtract + multiply.  The words come from awq_repack_gemvfast_to_gemm (repack.hip): the prefill route
// of WQLinear_GEMVFast (gemv_fast.py:203-206 runs awq_v2_ext.gemm_forward_cuda_prefill there).
template <int WGM, int DBG = 0, bool NK = false, bool GROUPED = false, bool FZ = false>  // waves along M: BM = 128 * WGM; NK: the GEMV layout's buffers (header)
__global__ __launch_bounds__(WGM * 256, 2) void awq_gemm_regb_kernel(RegbParams p) {
    static_assert(!FZ || (!NK && !GROUPED), "the GEMVFast arithmetic exists for the plain GEMM-layout form");
    constexpr int BM = 128 * WGM;
    constexpr int A_BUF = BM * BK * 2;            // bytes of one activation K step
    constexpr int PIECES = BM * 8 / 64 / (4 * WGM);  // 1 KiB DMA pieces per wave per K step (= 4)
    constexpr int B_OPS = NK ? 12 : 18;           // vector-memory operations of one fetch_b
    constexpr int AB_OPS = B_OPS + 2 * PIECES;    // ... plus the DMA pieces of a pair of activation steps (an even step issues both)
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
    int m0 = mt * BM, m_hi = p.M;
    const int n0 = nt * BN;
    if constexpr (GROUPED) {
        int e = 0, vt = mt, lo = 0, hi = 0;
        for (; e < p.E; ++e) {  // (scalar loads: E + 1 words, uniform)
            lo = p.seg[e];
            hi = p.seg[e + 1];
            const int te = (hi - lo + BM - 1) / BM;
            if (vt < te) break;
            vt -= te;
        }
        if (e >= p.E) return;
        m0 = lo + vt * BM;
        m_hi = hi;
        p.qweight = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.qweight) + (long long)e * p.w_stride);
        p.qzeros = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.qzeros) + (long long)e * p.z_stride);
        p.scales = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(p.scales) + (long long)e * p.s_stride);
    }
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
    // NK form: lane (j, kb) owns output columns ncol0 + c (c = 0..3) and words 2 kb, 2 kb + 1 of every 64-wide step of their rows.
    // ONE lane offset per tensor (row ncol0); the row of column c is reached through the SCALAR offset (+ c rows), so the address
    // registers are the three of the GEMM-layout form.  Rows past N are outside the buffer: the bounds-checked loads return 0,
    // and those columns are never stored.
    const int ncol0 = n0 + wn * 64 + 4 * j;
    const uint32_t nk_w = NK ? (uint32_t)ncol0 * (uint32_t)p.KW * 4u + 8u * (uint32_t)kb : 0u;
    const uint32_t nk_z = NK ? (uint32_t)ncol0 * (uint32_t)p.ZW * 4u : 0u;
    const uint32_t nk_s = NK ? (uint32_t)ncol0 * (uint32_t)p.SW * 2u : 0u;

    // Weight words, zero points and scales are requested with inline asm and waited for with COUNTED s_waitcnt
    // statements that name the registers they release (so nothing that uses them can be scheduled above): hipcc's own
    // bookkeeping falls back to vmcnt(0) whenever an LDS-DMA and an ordinary load are pending together (it takes them
    // for different, mutually unordered event classes), which drains the activation prefetch at every use of a weight.
    struct BRegsKN {
        uint32_t w[2][8];
        uint32_t z;
        u32x2 s;
        u32x2 zf;  // FZ: the four columns' fp16 zero terms -(s z)
    };
    struct BRegsNK {
        u32x2 w[4];      // column c: words 2 kb (K sub-step 0) and 2 kb + 1 (sub-step 1) of this step
        uint32_t z[4];   // the zero WORD of (column c, this step's group): nibble (group & 7)
        uint32_t s[4];   // the scale of (column c, group) in bits 0-15
        uint32_t zsh;    // 4 * (group & 7), uniform
    };
    using BRegs = std::conditional_t<NK, BRegsNK, BRegsKN>;
    auto srd = [](const void* base, uint32_t bytes) -> u32x4 {  // raw buffer descriptor, stride 0, bounds checked
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
    };
    const u32x4 wsrd = srd(p.qweight, NK ? (uint32_t)p.N * (uint32_t)p.KW * 4u : (uint32_t)p.K * row_bytes);
    const u32x4 zsrd = srd(p.qzeros, NK ? (uint32_t)p.N * (uint32_t)p.ZW * 4u : (FZ ? (uint32_t)(p.K / p.g) * (uint32_t)p.N * 2u : (uint32_t)(p.K / p.g) * row_bytes));
    const u32x4 ssrd = srd(p.scales, NK ? (uint32_t)p.N * (uint32_t)p.SW * 2u : (uint32_t)(p.K / p.g) * (uint32_t)p.N * 2u);
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
                 : "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(W[3]), "=&v"(W[4]), "=&v"(W[5]), "=&v"(W[6]), "=&v"(W[7])                  \
                 : "v"(voff), "s"(rs), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5]), "s"(so[6]), "s"(so[7]))
#define AWQ_BLOADZS(Z, S2, zvoff, zrs, zso, svoff, srs, sso)                                                                  \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %3, %4 offen\n\tbuffer_load_dwordx2 %1, %5, %6, %7 offen"                        \
                 : "=&v"(Z), "=&v"(S2)                                                                                        \
                 : "v"(zvoff), "s"(zrs), "s"(zso), "v"(svoff), "s"(srs), "s"(sso))
#define AWQ_BLOADZFS(Z2, S2, zrs, svoff, srs, sso)                                                                            \
    asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %2, %3, %5 offen\n\tbuffer_load_dwordx2 %1, %2, %4, %5 offen"                       \
                 : "=&v"(Z2), "=&v"(S2)                                                                                       \
                 : "v"(svoff), "s"(zrs), "s"(srs), "s"(sso))
#define AWQ_BLOAD1(dst, voff, rs, soff) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff))
#define AWQ_BLOAD2(dst, voff, rs, soff) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff))
// NK form: four dwordx2 weight loads, then four zero words and four scales, each group behind one s_nop 4 (same hazard)
#define AWQ_BLOADW4(W, vo, rs, so)                                                                                              \
    asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %4, %5, %6 offen\n\tbuffer_load_dwordx2 %1, %4, %5, %7 offen\n\t"              \
                 "buffer_load_dwordx2 %2, %4, %5, %8 offen\n\tbuffer_load_dwordx2 %3, %4, %5, %9 offen"                                \
                 : "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(W[3])                                                               \
                 : "v"(vo), "s"(rs), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]))
#define AWQ_BLOADZS4(Z, S, zvo, zrs, zso, svo, srs, sso)                                                                        \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %8, %9, %10 offen\n\tbuffer_load_dword %1, %8, %9, %11 offen\n\t"                \
                 "buffer_load_dword %2, %8, %9, %12 offen\n\tbuffer_load_dword %3, %8, %9, %13 offen\n\t"                            \
                 "buffer_load_ushort %4, %14, %15, %16 offen\n\tbuffer_load_ushort %5, %14, %15, %17 offen\n\t"                      \
                 "buffer_load_ushort %6, %14, %15, %18 offen\n\tbuffer_load_ushort %7, %14, %15, %19 offen"                            \
                 : "=&v"(Z[0]), "=&v"(Z[1]), "=&v"(Z[2]), "=&v"(Z[3]), "=&v"(S[0]), "=&v"(S[1]), "=&v"(S[2]), "=&v"(S[3])               \
                 : "v"(zvo), "s"(zrs), "s"(zso[0]), "s"(zso[1]), "s"(zso[2]), "s"(zso[3]), "v"(svo), "s"(srs), "s"(sso[0]),       \
                   "s"(sso[1]), "s"(sso[2]), "s"(sso[3]))
    auto fetch_b = [&](BRegs& R, int t) {
        const uint32_t k0 = (uint32_t)t * BK;
        if constexpr (NK) {
            const uint32_t grp = __umulhi(k0, g_magic);
            uint32_t so_w[4], so_z[4], so_s[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                so_w[c] = (uint32_t)t * (BK / 8 * 4) + (uint32_t)c * (uint32_t)p.KW * 4u;
                so_z[c] = (grp >> 3) * 4u + (uint32_t)c * (uint32_t)p.ZW * 4u;
                so_s[c] = grp * 2u + (uint32_t)c * (uint32_t)p.SW * 2u;
            }
            AWQ_BLOADW4(R.w, nk_w, wsrd, so_w);
            AWQ_BLOADZS4(R.z, R.s, nk_z, zsrd, so_z, nk_s, ssrd, so_s);
            R.zsh = 4u * (grp & 7u);
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint32_t so[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) so[r] = (k0 + 32u * kk + r) * row_bytes;
            AWQ_BLOAD8(R.w[kk], w_voff, wsrd, so);
        }
        const uint32_t grp = __umulhi(k0, g_magic);
        const uint32_t zo = grp * row_bytes, so2 = grp * (uint32_t)p.N * 2u;
        if constexpr (FZ) {
            AWQ_BLOADZFS(R.zf, R.s, zsrd, s_voff, ssrd, so2);  // zero terms and scales: the same offsets in two tensors
        } else {
            AWQ_BLOADZS(R.z, R.s, z_voff, zsrd, zo, s_voff, ssrd, so2);
        }
        }
    };
    // everything up to and including R's requests has returned once at most NEWER later operations are outstanding
    auto wait_b = [&](BRegs& R, auto newer_c) __attribute__((always_inline)) {
        constexpr int NEWER = decltype(newer_c)::value;
        if constexpr (NK) {
            asm volatile("s_waitcnt vmcnt(%12)"
                         : "+v"(R.w[0]), "+v"(R.w[1]), "+v"(R.w[2]), "+v"(R.w[3]), "+v"(R.z[0]), "+v"(R.z[1]), "+v"(R.z[2]), "+v"(R.z[3]),
                           "+v"(R.s[0]), "+v"(R.s[1]), "+v"(R.s[2]), "+v"(R.s[3])
                         : "n"(NEWER));
        } else if constexpr (FZ) {
            asm volatile("s_waitcnt vmcnt(%18)"
                         : "+v"(R.w[0][0]), "+v"(R.w[0][1]), "+v"(R.w[0][2]), "+v"(R.w[0][3]), "+v"(R.w[0][4]), "+v"(R.w[0][5]),
                           "+v"(R.w[0][6]), "+v"(R.w[0][7]), "+v"(R.w[1][0]), "+v"(R.w[1][1]), "+v"(R.w[1][2]), "+v"(R.w[1][3]),
                           "+v"(R.w[1][4]), "+v"(R.w[1][5]), "+v"(R.w[1][6]), "+v"(R.w[1][7]), "+v"(R.zf), "+v"(R.s)
                         : "n"(NEWER));
        } else {
            asm volatile("s_waitcnt vmcnt(%18)"
                         : "+v"(R.w[0][0]), "+v"(R.w[0][1]), "+v"(R.w[0][2]), "+v"(R.w[0][3]), "+v"(R.w[0][4]), "+v"(R.w[0][5]),
                           "+v"(R.w[0][6]), "+v"(R.w[0][7]), "+v"(R.w[1][0]), "+v"(R.w[1][1]), "+v"(R.w[1][2]), "+v"(R.w[1][3]),
                           "+v"(R.w[1][4]), "+v"(R.w[1][5]), "+v"(R.w[1][6]), "+v"(R.w[1][7]), "+v"(R.z), "+v"(R.s)
                         : "n"(NEWER));
        }
    };
#define AWQ_WAIT_B(R, newer) wait_b(R, std::integral_constant<int, (newer)>{})

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
        int grow = min(m0 + row, m_hi - 1);
        if constexpr (GROUPED) {
            if (p.gather) {  // the sorted row's source row: one 4-byte load per lane and piece, once per block
                const int pr = p.row_map[grow];
                grow = p.x_div_magic ? (int)__umulhi((uint32_t)pr, p.x_div_magic) : pr;
            }
        }
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
#define AWQ_LDS_READ16(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=&v"(dst) : "v"(addr))

    auto compute = [&](BRegs& R, int buf, auto&& issue_next) {
        half2_t zm[4], sd[4], zd[4];
        issue_next();  // next step's requests first (their round trip is the longest thing in the step), then wait for R
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // this lane's 8 K values of sub-step kk: chunk 4 kk + kb of the row's eight (GEMM layout: rows 8 kb .. 8 kb + 7 of each
            // 32-row slab), chunk 2 kb + kk in the NK form (words 2 kb, 2 kb + 1 of the step)
            const uint32_t aa = a_base + (uint32_t)(buf * A_BUF) + (uint32_t)((((NK ? 2 * kb + kk : 4 * kk + kb)) ^ hl) << 4);
            u32x4v af[4], ag[4];  // rows 0-63 of the wave tile; rows 64-127 are requested once these are in the MFMAs
            AWQ_LDS_READ16(af[0], aa, 0);
            AWQ_LDS_READ16(af[1], aa, 2048);
            AWQ_LDS_READ16(af[2], aa, 4096);
            AWQ_LDS_READ16(af[3], aa, 6144);
            if constexpr (NK) {
                if (kk == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t z = (R.z[c] >> R.zsh) & 15u;
                        zm[c] = u2h2(0x54006400u | z | (z << 20));  // (1024 + z, 64 + z): the biases of the two halves below
                        sd[c] = u2h2(__builtin_amdgcn_perm(R.s[c], R.s[c], 0x01000100u));  // (s, s)
                    }
                }
            } else if (FZ && kk == 0) {
                // GEMVFast arithmetic: zm = the exponent biases of the two nibble positions, zd = the columns' zero terms -(s z)
                zm[0] = zm[1] = u2h2(0x64006400u);
                zm[2] = zm[3] = u2h2(0x54005400u);
                const half2_t s01 = u2h2(R.s[0]), s23 = u2h2(R.s[1]), z01 = u2h2(R.zf[0]), z23 = u2h2(R.zf[1]);
                sd[0] = __builtin_shufflevector(s01, s01, 0, 0);
                sd[1] = __builtin_shufflevector(s01, s01, 1, 1);
                sd[2] = __builtin_shufflevector(s23, s23, 0, 0);
                sd[3] = __builtin_shufflevector(s23, s23, 1, 1);
                zd[0] = __builtin_shufflevector(z01, z01, 0, 0);
                zd[1] = __builtin_shufflevector(z01, z01, 1, 1);
                zd[2] = __builtin_shufflevector(z23, z23, 0, 0);
                zd[3] = __builtin_shufflevector(z23, z23, 1, 1);
            } else if (kk == 0) {
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
            if constexpr (NK) {
                // byte b of a word = the natural pair (k = 2 b, 2 b + 1): into bytes 0 and 2, then ONE and-or gives (1024 + w, 64 + w')
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t w = R.w[c][kk];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t pb = __builtin_amdgcn_perm(w, w, 0x0C000C00u | (uint32_t)b | ((uint32_t)b << 16));
                        if constexpr (DBG & 8) {
                            bf[c][b] = pb ^ h22u(zm[c]) ^ h22u(sd[c]);
                            continue;
                        }
                        bf[c][b] = h22u((u2h2(and_or(pb, 0x00F0000Fu, 0x54006400u)) - zm[c]) * sd[c]);
                    }
                }
            } else
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const uint32_t p0 = __builtin_amdgcn_perm(R.w[kk][2 * rp + 1], R.w[kk][2 * rp], sel0);
                const uint32_t p1 = __builtin_amdgcn_perm(R.w[kk][2 * rp + 1], R.w[kk][2 * rp], sel1);
                if constexpr (DBG & 8) {
                    bf[0][rp] = p0; bf[1][rp] = p1; bf[2][rp] = p0 ^ h22u(zm[0]); bf[3][rp] = p1 ^ h22u(sd[0]);
                    continue;
                }
                if constexpr (FZ) {  // w exact (the subtract), then ONE rounding of w s + qzeros: == fp16 of the fp32 fma (w s is exact there)
                    bf[0][rp] = h22u(__builtin_elementwise_fma(u2h2(and_or(p0, 0x000F000Fu, 0x64006400u)) - zm[0], sd[0], zd[0]));
                    bf[1][rp] = h22u(__builtin_elementwise_fma(u2h2(and_or(p1, 0x000F000Fu, 0x64006400u)) - zm[1], sd[1], zd[1]));
                    bf[2][rp] = h22u(__builtin_elementwise_fma(u2h2(and_or(p0, 0x00F000F0u, 0x54005400u)) - zm[2], sd[2], zd[2]));
                    bf[3][rp] = h22u(__builtin_elementwise_fma(u2h2(and_or(p1, 0x00F000F0u, 0x54005400u)) - zm[3], sd[3], zd[3]));
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
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_OPS) : "memory");
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
                AWQ_WAIT_B(cur, AB_OPS);
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
    const int col = NK ? ncol0 : n0 + set * 128 + 8 * j + 4 * ph;  // NK: four consecutive columns per lane as well
    if (col >= p.N) return;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const half4_t bv = *reinterpret_cast<const half4_t*>(p.bias + col);
#pragma unroll
        for (int c = 0; c < 4; ++c) b4[c] = (float)bv[c];
    }
    if constexpr (GROUPED) {
        // with a row map a lane's output rows and their routing weights are requested as batches of independent loads ahead of
        // the stores, sixteen rows at a time (one dependent chain per row -- map, weight, store -- measured +46 us on Mixtral's
        // w2 launch; all 32 rows at once spilled)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int orow[4][4];
            float wr[4][4];
            // (ONE uniform branch per batch, not one per element: written as whole-batch alternatives so that hipcc keeps it that way)
            if (p.scatter || p.pair_w) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) orow[i][e] = p.row_map[min(m0 + wm * 128 + 16 * (4 * h + i) + 4 * kb + e, m_hi - 1)];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) orow[i][e] = m0 + wm * 128 + 16 * (4 * h + i) + 4 * kb + e;
            }
            if (p.pair_w) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) wr[i][e] = p.pair_w[orow[i][e]];  // mul_routed_weight (moe.py:84-88): fp32 product, one rounding
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) wr[i][e] = 1.f;
            }
            // every load of the batch has landed before the first (conditional) store: nothing with a destination register is in
            // the vector-memory queue across the 16 store branches (tools/isa_audit.py walks both sides of each with its queue state)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m0 + wm * 128 + 16 * (4 * h + i) + 4 * kb + e;
                    if (row < m_hi) {
                        half4_t o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = (half_t)(acc[4 * h + i][c][e] * wr[i][e]);
                        *reinterpret_cast<half4_t*>(p.y + (int64_t)(p.scatter ? orow[i][e] : row) * p.N + col) = o;
                    }
                }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm * 128 + 16 * i + 4 * kb + e;
                if (row < m_hi) {
                    half4_t o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = (half_t)(acc[i][c][e] + b4[c]);
                    *reinterpret_cast<half4_t*>(p.y + (int64_t)((DBG & 16) ? (row & 127) : row) * p.N + col) = o;
                }
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
    p.KW = p.ZW = p.SW = 0;
    p.seg = nullptr; p.E = 0; p.w_stride = p.z_stride = p.s_stride = 0;
    p.row_map = nullptr; p.pair_w = nullptr; p.x_div_magic = 0; p.gather = p.scatter = 0;
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
            auto go = [&](auto wgm_c, auto dbg_c) {
                constexpr int W = decltype(wgm_c)::value, D = decltype(dbg_c)::value;
                hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<W, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((awq_gemm_regb_kernel<W, D>), dim3(grid), dim3(W * 256), lds, a.stream, p);
            };
            auto by_dbg = [&](auto wgm_c) {
                using std::integral_constant;
                if (dbg == 1) go(wgm_c, integral_constant<int, 1>{});
                else if (dbg == 2) go(wgm_c, integral_constant<int, 2>{});
                else if (dbg == 4) go(wgm_c, integral_constant<int, 4>{});
                else if (dbg == 8) go(wgm_c, integral_constant<int, 8>{});
                else if (dbg == 16) go(wgm_c, integral_constant<int, 16>{});
                else go(wgm_c, integral_constant<int, 15>{});
            };
            if (bm == 256) by_dbg(std::integral_constant<int, 2>{});
            else by_dbg(std::integral_constant<int, 1>{});
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

// ---- the FZ form: GEMM-layout words (a repacked temporary) with the GEMVFast format's scales / fp16 zero terms [>= K/g, N]
int awq_launch_gemm_regb_fz(const uint16_t* x, const int32_t* qweight_kn, const uint16_t* scales, const uint16_t* qzeros_f16, uint16_t* y,
                            int M, int K, int N, int g, int bm, hipStream_t st) {
    if (!awq_gemm_regb_supports(M, K, N, g)) return AWQ_ERR_UNSUPPORTED;
    if (bm == 0) bm = 128;
    if (bm != 128 && bm != 256) return AWQ_ERR_UNSUPPORTED;
    RegbParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight_kn);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros_f16);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.bias = nullptr;
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N; p.g = g;
    p.KW = p.ZW = p.SW = 0;
    p.seg = nullptr; p.E = 0; p.w_stride = p.z_stride = p.s_stride = 0;
    p.row_map = nullptr; p.pair_w = nullptr; p.x_div_magic = 0; p.gather = p.scatter = 0;
    p.tiles_m = (M + bm - 1) / bm;
    p.tiles_n = (N + BN - 1) / BN;
    p.pm = PM; p.pn = PN;
    p.mp = (p.tiles_m + p.pm - 1) / p.pm;
    p.patches = p.mp * ((p.tiles_n + p.pn - 1) / p.pn);
    const int grid = ((p.patches + 7) / 8) * 8 * (p.pm * p.pn);
    const size_t lds = (size_t)NBUF * bm * BK * 2;
    if (bm == 256) {
        static std::atomic<unsigned long long> opted{0};
        if (!awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 0, false, false, true>), opted)) return AWQ_ERR_LAUNCH;
        hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 0, false, false, true>), dim3(grid), dim3(512), lds, st, p);
    } else {
        hipLaunchKernelGGL((awq_gemm_regb_kernel<1, 0, false, false, true>), dim3(grid), dim3(256), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// ---- the NK form: WQLinear_GEMV buffers (qweight [N, K/8], qzeros [N, ZW], scales [N, 8 ZW]), prefill-sized batches
bool awq_gemm_regb_nk_supports(int M, int K, int N, int g, int ZW) {
    return M > 0 && K > 0 && N > 0 && ZW > 0 && (int64_t)M * K * 2 < ((int64_t)1 << 32) && K % BK == 0 && g % BK == 0 && K % g == 0 &&
           N % 4 == 0 && K < 65536 && ZW * 8 >= K / g && (int64_t)N * (K / 8) * 4 < ((int64_t)1 << 31) &&
           (int64_t)N * ZW * 16 < ((int64_t)1 << 31);
}

int awq_launch_gemm_regb_nk(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                            const uint16_t* bias, uint16_t* y, int M, int K, int N, int g, int ZW, int bm, hipStream_t st) {
    if (!awq_gemm_regb_nk_supports(M, K, N, g, ZW)) return AWQ_ERR_UNSUPPORTED;
    if (bm == 0) bm = 128;
    if (bm != 128 && bm != 256) return AWQ_ERR_UNSUPPORTED;
    RegbParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.bias = reinterpret_cast<const half_t*>(bias);
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N; p.g = g;
    p.KW = K / 8; p.ZW = ZW; p.SW = 8 * ZW;
    p.seg = nullptr; p.E = 0; p.w_stride = p.z_stride = p.s_stride = 0;
    p.row_map = nullptr; p.pair_w = nullptr; p.x_div_magic = 0; p.gather = p.scatter = 0;
    p.tiles_m = (M + bm - 1) / bm;
    p.tiles_n = (N + BN - 1) / BN;
    p.pm = PM; p.pn = PN;
    p.mp = (p.tiles_m + p.pm - 1) / p.pm;
    p.patches = p.mp * ((p.tiles_n + p.pn - 1) / p.pn);
    const int grid = ((p.patches + 7) / 8) * 8 * (p.pm * p.pn);
    const size_t lds = (size_t)NBUF * bm * BK * 2;
    if (bm == 256) {
        static std::atomic<unsigned long long> opted{0};
        if (!awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 0, true>), opted)) return AWQ_ERR_LAUNCH;
        hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 0, true>), dim3(grid), dim3(512), lds, st, p);
    } else {
        hipLaunchKernelGGL((awq_gemm_regb_kernel<1, 0, true>), dim3(grid), dim3(256), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// ---- the GROUPED form (MoE prefill, GEMM-layout expert stacks): x / y rows sorted by expert, seg [E + 1] on the device.
// max_tiles: M tiles the launch provides = ceil(P / bm) + E covers every split of P rows over E experts.
// row_map / x_div / pair_w / gather / scatter: see RegbParams (all optional; without row_map the rows of x and y are the sorted rows)
int awq_launch_gemm_regb_grouped(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros,
                                 uint16_t* y, const int32_t* seg, int P, int E, int K, int N, int g, int bm, hipStream_t st,
                                 const int32_t* row_map, int x_div, const float* pair_w, int gather, int scatter) {
    if (!awq_gemm_regb_supports(P, K, N, g) || E < 1 || !seg) return AWQ_ERR_UNSUPPORTED;
    if ((gather || scatter || pair_w) && !row_map) return AWQ_ERR_NULL;
    if (bm == 0) bm = 128;
    if (bm != 128 && bm != 256) return AWQ_ERR_UNSUPPORTED;
    RegbParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.bias = nullptr;
    p.y = reinterpret_cast<half_t*>(y);
    p.M = P; p.K = K; p.N = N; p.g = g;
    p.KW = p.ZW = p.SW = 0;
    p.seg = seg; p.E = E;
    p.row_map = row_map; p.pair_w = pair_w; p.gather = gather; p.scatter = scatter; p.x_div_magic = 0;
    if (gather && x_div > 1 && !awq_magic_u32((uint32_t)x_div, (uint32_t)P + 1u, &p.x_div_magic)) return AWQ_ERR_UNSUPPORTED;
    p.w_stride = (long long)K * (N / 8) * 4;
    p.z_stride = (long long)(K / g) * (N / 8) * 4;
    p.s_stride = (long long)(K / g) * N * 2;
    p.tiles_m = (P + bm - 1) / bm + E;
    p.tiles_n = (N + BN - 1) / BN;
    p.pm = PM; p.pn = PN;
    p.mp = (p.tiles_m + p.pm - 1) / p.pm;
    p.patches = p.mp * ((p.tiles_n + p.pn - 1) / p.pn);
    const int grid = ((p.patches + 7) / 8) * 8 * (p.pm * p.pn);
    const size_t lds = (size_t)NBUF * bm * BK * 2;
    if (bm == 256) {
        static std::atomic<unsigned long long> opted{0};
        if (!awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_regb_kernel<2, 0, false, true>), opted)) return AWQ_ERR_LAUNCH;
        hipLaunchKernelGGL((awq_gemm_regb_kernel<2, 0, false, true>), dim3(grid), dim3(512), lds, st, p);
    } else {
        hipLaunchKernelGGL((awq_gemm_regb_kernel<1, 0, false, true>), dim3(grid), dim3(256), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
