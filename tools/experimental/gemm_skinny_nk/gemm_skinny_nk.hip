// gemm_skinny_nk.hip -- EXPERIMENTAL (round 4, NOT in the product build, NOT yet run on a GPU): csrc/gemm_skinny.hip with an N-major
// form (template flag NK) for the GEMV layout's own buffers -- qweight [N, K/8] int32 (nibble i of word c = w[n, 8c+i]),
// qzeros [N, ZW], scales [N, 8 ZW] -- the same transformation csrc/gemm_regb.hip got in round 4: lane (j, kb)
// owns output columns 4 j + c and words 2 kb, 2 kb + 1 of each 64-wide step (one dwordx2 per column and step), byte b of a word
// is the natural K pair: v_perm_b32 + ONE v_and_or (mask 0x00F0000F, exponents 2^10 | 2^6), (t - (bias + z)) * s exact; the A
// side reads chunk 2 kb + kk.  Purpose: batches of 17 ... 64 rows on WQLinear_GEMV without a GEMM-layout copy of the weights
// (today: 16-row chunks of the decode kernels, 47 / 93 us at 4096 x 11008 for M = 32 / 64; the GEMM-layout form: 14 / 18 us).
// MODE 2 ("wide", K % 256 == 0): the same N-major form with 256-wide steps -- lane (j, kb) reads 32 contiguous bytes (words 8 kb ..
// 8 kb + 7 of the step: two dwordx4) of each of its four weight rows, so a row's 128 bytes are one full line per wave
// instruction pair instead of four 32-byte visits (what held csrc/gemm_regb.hip's N-major form at 0.29 of the MFMA peak:
// profiles/r04_pmc_mfma_prefill.txt); MFMA sub-step w of the step uses word w against x chunk w of 64-wide step 4 ws + kb
// (a lane-dependent LDS step offset); 16 instead of 48 vector-memory operations per 256 k; two steps in flight.
// Next round: run tools/experimental/gemm_skinny_nk/probe.py on the GPU; if it is green, merge the flag into csrc/gemm_skinny.hip
// and route awq_gemv_forward (17 <= M <= 64) to it.
//
// ---- original header of csrc/gemm_skinny.hip:
// gemm_skinny.hip -- batched decode GEMM (9 <= M <= 64) on the GEMM layout, weights decoded in registers, gfx950.
//
// Replaces awq_ext.gemm_forward_cuda (awq/modules/linear/gemm.py:56-58) for the batch sizes between the decode kernel
// (gemv_mfma.hip, M <= 16) and the prefill kernel (gemm_regb.hip).  HBM-bound like the decode kernel (the matrix is
// read once; 23 MB at 4096 x 11008), but with 2-4 MFMA row tiles per decoded weight fragment.  The LDS-tiled kernel
// spends 137 instructions per 8 MFMAs here (dequantise to LDS, barrier per 64 rows): 21.7 us at M = 32, 34.8 at M = 64.
//   * the B side is gemm_regb.hip's: a lane owns one packed word column and 8 K rows of a 32-row slab (8 dword loads),
//     v_perm + v_and_or give K-pair fp16 registers, (t - (bias + z)) * s is the reference's fp16 weight exactly; the
//     two waves that share a word column split its bytes (ph), each 64 logical columns wide.  Three 64-row steps of
//     weight words are in flight per wave (inline asm, counted s_waitcnt);
//   * the activations of the block's whole K slice are brought into LDS ONCE by LDS-DMA (<= 128 KB: 64 rows x 1024
//     columns), in 64-wide steps of 128 B rows with gemm_regb's XOR swizzle -- after that single barrier the K loop
//     has no barrier at all: eight waves per block run free;
//   * block = 256 columns x (K / S) rows on 8 waves: 4 column waves x 2 halves of the K slice, folded through LDS;
//     the S K-slices of a tile are combined in-launch through the sentinel slabs of gemv_mfma.hip / gemm_tiled.hip,
//     with one reducer block per group of row tiles (the last R slices), so the poll is one round trip.
// Algorithmic bytes as in gemm_tiled.hip; roofline: HBM.  Constraints (else AWQ_ERR_UNSUPPORTED -> gemm_tiled):
// 9 <= M <= 64, K % 64 == 0, group_size % 64 == 0, N % 8 == 0.  (From 9 rows, where the decode kernel needs a second MFMA per
// fragment and a 16-row exchange, this kernel is ahead on the 4096-row matrices -- 4096 x 11008, M = 16: 13.8 vs 15.6 us,
// 4096 x 4096: 11.8 vs 12.9 -- and level or slightly behind on 11008 x 4096 / 4096 x 22016, which the dispatch leaves alone.)
#include <cstdlib>
#include <type_traits>

#include "awq_device.h"
#include "awq_internal.h"
#include "awq_mfma_decode.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct SkinnyParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    const half_t* bias;
    half_t* y;
    int M, K, N, g;
    int S, sps;        // K slices per tile, 64-row steps per slice
    uint32_t g_magic;  // (k * g_magic) >> 32 == k / g
    float* slabs;      // exchange region [S][tiles][4 waves][MI*4][64 lanes] float4, sentinel-filled
    int* err;
    int KW, ZW, SW;    // NK form: words per qweight row, zero words per row, scale halfs per row
};

// Eight weight words / the group's zero word and scales in ONE asm statement each, opened by s_nop 4: an SGPR written by
// the SALU (the scalar offsets, a rematerialised descriptor) needs five wait states before a VMEM instruction may read
// it, and hipcc pads nothing for the operands of an asm statement -- without the nop a load now and then used the
// PREVIOUS value of its offset register (rare wrong tiles that came and went with the schedule).
#define AWQ_SK_BLOAD8(W, voff, rs, so)                                                                                          \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %8, %9, %10 offen\n\tbuffer_load_dword %1, %8, %9, %11 offen\n\t"                \
                 "buffer_load_dword %2, %8, %9, %12 offen\n\tbuffer_load_dword %3, %8, %9, %13 offen\n\t"                          \
                 "buffer_load_dword %4, %8, %9, %14 offen\n\tbuffer_load_dword %5, %8, %9, %15 offen\n\t"                          \
                 "buffer_load_dword %6, %8, %9, %16 offen\n\tbuffer_load_dword %7, %8, %9, %17 offen"                              \
                 : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]), "=v"(W[4]), "=v"(W[5]), "=v"(W[6]), "=v"(W[7])                  \
                 : "v"(voff), "s"(rs), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5]), "s"(so[6]), "s"(so[7]))
#define AWQ_SK_BLOADZS(Z, S2, zvoff, zrs, zso, svoff, srs, sso)                                                                  \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %3, %4 offen\n\tbuffer_load_dwordx2 %1, %5, %6, %7 offen"                        \
                 : "=v"(Z), "=v"(S2)                                                                                        \
                 : "v"(zvoff), "s"(zrs), "s"(zso), "v"(svoff), "s"(srs), "s"(sso))
#define AWQ_SK_BLOAD1(dst, voff, rs, soff) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff))
#define AWQ_SK_BLOAD2(dst, voff, rs, soff) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff))
#define AWQ_SK_BLOADW4(W, vo, rs, so)                                                                                           \
    asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %4, %5, %6 offen\n\tbuffer_load_dwordx2 %1, %4, %5, %7 offen\n\t"              \
                 "buffer_load_dwordx2 %2, %4, %5, %8 offen\n\tbuffer_load_dwordx2 %3, %4, %5, %9 offen"                                \
                 : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3])                                                               \
                 : "v"(vo), "s"(rs), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]))
#define AWQ_SK_BLOADZS4(Z, S, zvo, zrs, zso, svo, srs, sso)                                                                     \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %8, %9, %10 offen\n\tbuffer_load_dword %1, %8, %9, %11 offen\n\t"                \
                 "buffer_load_dword %2, %8, %9, %12 offen\n\tbuffer_load_dword %3, %8, %9, %13 offen\n\t"                            \
                 "buffer_load_ushort %4, %14, %15, %16 offen\n\tbuffer_load_ushort %5, %14, %15, %17 offen\n\t"                      \
                 "buffer_load_ushort %6, %14, %15, %18 offen\n\tbuffer_load_ushort %7, %14, %15, %19 offen"                            \
                 : "=v"(Z[0]), "=v"(Z[1]), "=v"(Z[2]), "=v"(Z[3]), "=v"(S[0]), "=v"(S[1]), "=v"(S[2]), "=v"(S[3])               \
                 : "v"(zvo), "s"(zrs), "s"(zso[0]), "s"(zso[1]), "s"(zso[2]), "s"(zso[3]), "v"(svo), "s"(srs), "s"(sso[0]),       \
                   "s"(sso[1]), "s"(sso[2]), "s"(sso[3]))
#define AWQ_SK_BLOADW8(W, vo, rs, so)                                                                                           \
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %8, %9, %10 offen\n\tbuffer_load_dwordx4 %1, %8, %9, %10 offen offset:16\n\t"   \
                 "buffer_load_dwordx4 %2, %8, %9, %11 offen\n\tbuffer_load_dwordx4 %3, %8, %9, %11 offen offset:16\n\t"             \
                 "buffer_load_dwordx4 %4, %8, %9, %12 offen\n\tbuffer_load_dwordx4 %5, %8, %9, %12 offen offset:16\n\t"             \
                 "buffer_load_dwordx4 %6, %8, %9, %13 offen\n\tbuffer_load_dwordx4 %7, %8, %9, %13 offen offset:16"                  \
                 : "=v"(W[0][0]), "=v"(W[0][1]), "=v"(W[1][0]), "=v"(W[1][1]), "=v"(W[2][0]), "=v"(W[2][1]), "=v"(W[3][0]),     \
                   "=v"(W[3][1])                                                                                                \
                 : "v"(vo), "s"(rs), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]))
#define AWQ_SK_DMA16(ldsaddr, voff, rs, soff)                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsaddr), "v"(voff), \
                 "s"(rs), "s"(soff)                                                                             \
                 : "m0")
#define AWQ_SK_LDS_READ16(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
template <int MI, int MODE>  // 16-row tiles per block: BM = 16 * MI (32 | 64); MODE 0: GEMM layout, 1 / 2: the GEMV layout's buffers (header)
__global__ __launch_bounds__(512, 2) void awq_gemm_skinny_kernel(SkinnyParams p) {
    constexpr bool WIDE = MODE == 2, NK = MODE == 1, NMAJOR = MODE != 0;
    constexpr int BM = 16 * MI;
    constexpr int A_STEP = BM * 128;  // bytes of one 64-wide activation step in LDS
    constexpr int PER = MI * 4;       // 16-byte accumulator chunks per lane
    constexpr int B_OPS = WIDE ? 16 : NK ? 12 : 18;  // vector-memory operations of one weight fetch
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [sps][BM][8 chunks]; later the K-half fold area

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, kh = wave >> 2, set = cw >> 1, ph = cw & 1;
    const int j = lane & 15, kb = lane >> 4;
    const int tile = blockIdx.x, slice = blockIdx.y, tiles = gridDim.x;
    const int n0 = tile * 256;
    const int NW = p.N >> 3;
    const int T = p.K >> 6;
    const int st0 = slice * p.sps, nst = min(p.sps, T - st0);  // this block's 64-row steps: [st0, st0 + nst)
    const int nunits = WIDE ? nst >> 2 : nst;                  // WIDE: units of four steps (the launcher keeps st0 and nst multiples of 4)
    const int half0 = (nunits + 1) >> 1;                       // the first K half takes the odd unit
    const int w0 = kh ? half0 : 0, w1 = kh ? nunits : half0;    // this wave's units, relative to st0

    auto srd = [](const void* base, uint32_t bytes) -> u32x4 {
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
    };
    const uint32_t row_bytes = (uint32_t)NW * 4u;
    const u32x4 wsrd = srd(p.qweight, NMAJOR ? (uint32_t)p.N * (uint32_t)p.KW * 4u : (uint32_t)p.K * row_bytes);
    const u32x4 zsrd = srd(p.qzeros, NMAJOR ? (uint32_t)p.N * (uint32_t)p.ZW * 4u : (uint32_t)(p.K / p.g) * row_bytes);
    const u32x4 ssrd = srd(p.scales, NMAJOR ? (uint32_t)p.N * (uint32_t)p.SW * 2u : (uint32_t)(p.K / p.g) * (uint32_t)p.N * 2u);
    const u32x4 xsrd = srd(p.x, (uint32_t)((int64_t)p.M * p.K * 2));

    // ---- activations of the whole K slice -> LDS, once (piece q: step q / (BM/8), rows 8 (q % (BM/8)) .. + 7)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    {
        const int npieces = nst * (BM / 8);
        const int slot = lane & 7;
        for (int q = wave; q < npieces; q += 8) {
            const int st = q / (BM / 8), row = 8 * (q % (BM / 8)) + (lane >> 3);
            const int kc = slot ^ ((row >> 1) & 7);
            const uint32_t voff = (uint32_t)(((int64_t)min(row, p.M - 1) * p.K + 8 * kc) * 2);  // rows past M: a valid row, never stored
            const uint32_t so = (uint32_t)(st0 + st) * 128u;
            const uint32_t dst = lds0 + (uint32_t)(st * A_STEP + (q % (BM / 8)) * 1024);
            AWQ_SK_DMA16(dst, voff, xsrd, so);
        }
    }

    // ---- weights: this lane's word column, three steps in flight
    const int wcol = (n0 >> 3) + set * 16 + j;
    const bool colok = wcol < NW;
    const uint32_t w_voff = colok ? ((uint32_t)(8 * kb) * (uint32_t)NW + (uint32_t)wcol) * 4u : OOB;
    const uint32_t z_voff = colok ? (uint32_t)wcol * 4u : OOB;
    const uint32_t s_voff = colok ? ((uint32_t)wcol * 8u + 4u * (uint32_t)ph) * 2u : OOB;
    // NK form: ONE lane offset per tensor (the row of column ncol0), the row of column c through the scalar offset; rows past N
    // lie outside the descriptors (bounds-checked loads return 0) and are never stored
    const int ncol0 = n0 + cw * 64 + 4 * j;
    const uint32_t nk_w = NMAJOR ? (uint32_t)ncol0 * (uint32_t)p.KW * 4u + (WIDE ? 32u : 8u) * (uint32_t)kb : 0u;
    const uint32_t nk_z = NMAJOR ? (uint32_t)ncol0 * (uint32_t)p.ZW * 4u : 0u;
    const uint32_t nk_s = NMAJOR ? (uint32_t)ncol0 * (uint32_t)p.SW * 2u : 0u;
    struct BRegsKN {
        uint32_t w[2][8];
        uint32_t z;
        u32x2 s;
    };
    struct BRegsNK {
        u32x2 w[4];      // column c: words 2 kb (K sub-step 0) and 2 kb + 1 (sub-step 1) of this step
        uint32_t z[4];   // the zero WORD of (column c, this step's group): nibble (group & 7)
        uint32_t s[4];   // the scale of (column c, group) in bits 0-15
        uint32_t zsh;    // 4 * (group & 7), uniform
    };
    struct BRegsWide {
        u32x4 w[4][2];   // column c: words 8 kb .. 8 kb + 7 of this 256-wide step (k = 64 kb + 8 word + i)
        uint32_t z[4];   // the zero word of (column c, this LANE's group)
        uint32_t s[4];   // the scale of (column c, this lane's group) in bits 0-15
        uint32_t zsh;    // 4 * (group & 7), per lane
    };
    using BRegs = std::conditional_t<WIDE, BRegsWide, std::conditional_t<NK, BRegsNK, BRegsKN>>;
    auto wait_b = [&](BRegs& R, auto newer_c) __attribute__((always_inline)) {
        constexpr int NEWER = decltype(newer_c)::value;
        if constexpr (WIDE) {
            asm volatile("s_waitcnt vmcnt(%16)"
                         : "+v"(R.w[0][0]), "+v"(R.w[0][1]), "+v"(R.w[1][0]), "+v"(R.w[1][1]), "+v"(R.w[2][0]), "+v"(R.w[2][1]),
                           "+v"(R.w[3][0]), "+v"(R.w[3][1]), "+v"(R.z[0]), "+v"(R.z[1]), "+v"(R.z[2]), "+v"(R.z[3]), "+v"(R.s[0]),
                           "+v"(R.s[1]), "+v"(R.s[2]), "+v"(R.s[3])
                         : "n"(NEWER));
        } else if constexpr (NK) {
            asm volatile("s_waitcnt vmcnt(%12)"
                         : "+v"(R.w[0]), "+v"(R.w[1]), "+v"(R.w[2]), "+v"(R.w[3]), "+v"(R.z[0]), "+v"(R.z[1]), "+v"(R.z[2]), "+v"(R.z[3]),
                           "+v"(R.s[0]), "+v"(R.s[1]), "+v"(R.s[2]), "+v"(R.s[3])
                         : "n"(NEWER));
        } else {
            asm volatile("s_waitcnt vmcnt(%18)"
                         : "+v"(R.w[0][0]), "+v"(R.w[0][1]), "+v"(R.w[0][2]), "+v"(R.w[0][3]), "+v"(R.w[0][4]), "+v"(R.w[0][5]),
                           "+v"(R.w[0][6]), "+v"(R.w[0][7]), "+v"(R.w[1][0]), "+v"(R.w[1][1]), "+v"(R.w[1][2]), "+v"(R.w[1][3]),
                           "+v"(R.w[1][4]), "+v"(R.w[1][5]), "+v"(R.w[1][6]), "+v"(R.w[1][7]), "+v"(R.z), "+v"(R.s)
                         : "n"(NEWER));
        }
    };
#define AWQ_SK_WAIT_B(R, newer) wait_b(R, std::integral_constant<int, (newer)>{})
    auto fetch_b = [&](BRegs& R, int st) {  // st relative to st0; past the wave's range: the last step again (static counts)
        if constexpr (WIDE) {  // st = unit index: k of the unit, this lane's 64-wide part of it and its group
            const uint32_t K0 = (uint32_t)st0 * 64u + (uint32_t)min(st, max(w1 - 1, w0)) * 256u;
            const uint32_t grp = __umulhi(K0 + 64u * (uint32_t)kb, p.g_magic);
            uint32_t so_w[4], so_z[4], so_s[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                so_w[c] = (K0 >> 1) + (uint32_t)c * (uint32_t)p.KW * 4u;
                so_z[c] = (uint32_t)c * (uint32_t)p.ZW * 4u;
                so_s[c] = (uint32_t)c * (uint32_t)p.SW * 2u;
            }
            AWQ_SK_BLOADW8(R.w, nk_w, wsrd, so_w);
            const uint32_t zvo = nk_z + (grp >> 3) * 4u, svo = nk_s + grp * 2u;
            AWQ_SK_BLOADZS4(R.z, R.s, zvo, zsrd, so_z, svo, ssrd, so_s);
            R.zsh = 4u * (grp & 7u);
        } else {
        const uint32_t k0 = (uint32_t)(st0 + min(st, max(w1 - 1, w0))) * 64u;
        if constexpr (NK) {
            const uint32_t grp = __umulhi(k0, p.g_magic);
            uint32_t so_w[4], so_z[4], so_s[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                so_w[c] = (k0 >> 3) * 4u + (uint32_t)c * (uint32_t)p.KW * 4u;
                so_z[c] = (grp >> 3) * 4u + (uint32_t)c * (uint32_t)p.ZW * 4u;
                so_s[c] = grp * 2u + (uint32_t)c * (uint32_t)p.SW * 2u;
            }
            AWQ_SK_BLOADW4(R.w, nk_w, wsrd, so_w);
            AWQ_SK_BLOADZS4(R.z, R.s, nk_z, zsrd, so_z, nk_s, ssrd, so_s);
            R.zsh = 4u * (grp & 7u);
            return;
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint32_t so[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) so[r] = (k0 + 32u * kk + r) * row_bytes;
            AWQ_SK_BLOAD8(R.w[kk], w_voff, wsrd, so);
        }
        const uint32_t grp = __umulhi(k0, p.g_magic);
        const uint32_t zo = grp * row_bytes, so2 = grp * (uint32_t)p.N * 2u;
        AWQ_SK_BLOADZS(R.z, R.s, z_voff, zsrd, zo, s_voff, ssrd, so2);
        }
        }
    };
    BRegs B0, B1, B2;
    fetch_b(B0, w0);
    fetch_b(B1, w0 + 1);
    if constexpr (!WIDE) fetch_b(B2, w0 + 2);  // (WIDE: two units of 256 in flight)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WIDE ? 2 : 3) * B_OPS) : "memory");  // everything older than the three weight fetches: the DMA pieces
    __builtin_amdgcn_s_barrier();

    float4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = float4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t sel0 = 0x0C000C00u | (uint32_t)ph | ((uint32_t)(4 + ph) << 16);
    const uint32_t sel1 = sel0 + 0x00020002u;
    const uint32_t a_base = lds0 + (uint32_t)(j * 128);
    const int hl = (j >> 1) & 7;

    auto compute = [&](BRegs& R, int st) {
        half2_t zm[4], sd[4];
        if constexpr (WIDE) {  // st = unit index; sub-step w: word w of every column against x chunk w of 64-wide step 4 st + kb
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t z = (R.z[c] >> R.zsh) & 15u;
                zm[c] = u2h2(0x54006400u | z | (z << 20));
                sd[c] = u2h2(__builtin_amdgcn_perm(R.s[c], R.s[c], 0x01000100u));
            }
            const uint32_t a_unit = a_base + (uint32_t)((4 * st + kb) * A_STEP);
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const uint32_t aa = a_unit + (uint32_t)((w ^ hl) << 4);
                u32x4v af[MI];
                AWQ_SK_LDS_READ16(af[0], aa, 0);
                if constexpr (MI > 1) AWQ_SK_LDS_READ16(af[1], aa, 2048);
                if constexpr (MI > 2) {
                    AWQ_SK_LDS_READ16(af[2], aa, 4096);
                    AWQ_SK_LDS_READ16(af[3], aa, 6144);
                }
                u32x4v bf[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t word = R.w[c][w >> 2][w & 3];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t pb = __builtin_amdgcn_perm(word, word, 0x0C000C00u | (uint32_t)b | ((uint32_t)b << 16));
                        bf[c][b] = h22u((u2h2(and_or(pb, 0x00F0000Fu, 0x54006400u)) - zm[c]) * sd[c]);
                    }
                }
                if constexpr (MI > 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]));
                else if constexpr (MI > 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][c] = mfma16(af[i], bf[c], acc[i][c]);
            }
        } else {
        if constexpr (NK) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t z = (R.z[c] >> R.zsh) & 15u;
                zm[c] = u2h2(0x54006400u | z | (z << 20));  // (1024 + z, 64 + z): the biases of the two halves below
                sd[c] = u2h2(__builtin_amdgcn_perm(R.s[c], R.s[c], 0x01000100u));  // (s, s)
            }
        } else {
            const uint32_t zp0 = __builtin_amdgcn_perm(R.z, R.z, sel0), zp1 = __builtin_amdgcn_perm(R.z, R.z, sel1);
            zm[0] = u2h2(and_or(zp0, 0x000F000Fu, 0x64006400u));
            zm[1] = u2h2(and_or(zp1, 0x000F000Fu, 0x64006400u));
            zm[2] = u2h2(and_or(zp0, 0x00F000F0u, 0x54005400u));
            zm[3] = u2h2(and_or(zp1, 0x00F000F0u, 0x54005400u));
            const half2_t s01 = u2h2(R.s[0]), s23 = u2h2(R.s[1]);
            sd[0] = __builtin_shufflevector(s01, s01, 0, 0);
            sd[1] = __builtin_shufflevector(s01, s01, 1, 1);
            sd[2] = __builtin_shufflevector(s23, s23, 0, 0);
            sd[3] = __builtin_shufflevector(s23, s23, 1, 1);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint32_t aa = a_base + (uint32_t)(st * A_STEP) + (uint32_t)((((NK ? 2 * kb + kk : 4 * kk + kb)) ^ hl) << 4);
            u32x4v af[MI];
            AWQ_SK_LDS_READ16(af[0], aa, 0);
            if constexpr (MI > 1) AWQ_SK_LDS_READ16(af[1], aa, 2048);
            if constexpr (MI > 2) {
                AWQ_SK_LDS_READ16(af[2], aa, 4096);
                AWQ_SK_LDS_READ16(af[3], aa, 6144);
            }
            u32x4v bf[4];
            if constexpr (NK) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t w = R.w[c][kk];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t pb = __builtin_amdgcn_perm(w, w, 0x0C000C00u | (uint32_t)b | ((uint32_t)b << 16));
                        bf[c][b] = h22u((u2h2(and_or(pb, 0x00F0000Fu, 0x54006400u)) - zm[c]) * sd[c]);
                    }
                }
            } else
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const uint32_t p0 = __builtin_amdgcn_perm(R.w[kk][2 * rp + 1], R.w[kk][2 * rp], sel0);
                const uint32_t p1 = __builtin_amdgcn_perm(R.w[kk][2 * rp + 1], R.w[kk][2 * rp], sel1);
                bf[0][rp] = h22u((u2h2(and_or(p0, 0x000F000Fu, 0x64006400u)) - zm[0]) * sd[0]);
                bf[1][rp] = h22u((u2h2(and_or(p1, 0x000F000Fu, 0x64006400u)) - zm[1]) * sd[1]);
                bf[2][rp] = h22u((u2h2(and_or(p0, 0x00F000F0u, 0x54005400u)) - zm[2]) * sd[2]);
                bf[3][rp] = h22u((u2h2(and_or(p1, 0x00F000F0u, 0x54005400u)) - zm[3]) * sd[3]);
            }
            if constexpr (MI > 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]));
            else if constexpr (MI > 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i][c] = mfma16(af[i], bf[c], acc[i][c]);
        }
        }
    };

    // ---- K loop, no barrier: wait for a step's words (the two younger fetches stay in flight), multiply, refill
    int st = w0;
    if constexpr (WIDE) {
        for (; st + 2 <= w1; st += 2) {
            AWQ_SK_WAIT_B(B0, B_OPS); compute(B0, st);     fetch_b(B0, st + 2);
            AWQ_SK_WAIT_B(B1, B_OPS); compute(B1, st + 1); fetch_b(B1, st + 3);
        }
        AWQ_SK_WAIT_B(B0, 0);  // (names both register sets: see the comment below)
        AWQ_SK_WAIT_B(B1, 0);
        if (st < w1) compute(B0, st);
        st = w1;
    }
    for (; !WIDE && st + 3 <= w1; st += 3) {
        AWQ_SK_WAIT_B(B0, 2 * B_OPS); compute(B0, st);     fetch_b(B0, st + 3);
        AWQ_SK_WAIT_B(B1, 2 * B_OPS); compute(B1, st + 1); fetch_b(B1, st + 4);
        AWQ_SK_WAIT_B(B2, 2 * B_OPS); compute(B2, st + 2); fetch_b(B2, st + 5);
    }
    // The last refills are never consumed: to the compiler their destination registers are dead the moment they are
    // requested, so it would hand them to the temporaries of the steps below while the loads are still in flight -- and a
    // late load then overwrites a live value.  One wait that NAMES all three register sets keeps them allocated until
    // everything has landed; the (at most two) remaining steps then need no wait of their own.
    if constexpr (!WIDE) {
        AWQ_SK_WAIT_B(B0, 0);
        AWQ_SK_WAIT_B(B1, 0);
        AWQ_SK_WAIT_B(B2, 0);
        if (st < w1) compute(B0, st);
        if (st + 1 < w1) compute(B1, st + 1);
    }

    // ---- fold the two K halves through LDS (the activation area is dead once every wave is past its K loop)
    __syncthreads();
    float4_t* fold = reinterpret_cast<float4_t*>(smem) + (size_t)cw * PER * 64 + lane;
    if (kh == 1) {
#pragma unroll
        for (int c = 0; c < PER; ++c) fold[c * 64] = acc[c >> 2][c & 3];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int c = 0; c < PER; ++c) acc[c >> 2][c & 3] += fold[c * 64];

    // ---- split-K combine: sentinel slabs, R reducers (the last R slices), reducer r owns row tiles i = r, r + R, ...
    const int S = p.S;
    if (S > 1) {
        constexpr uint32_t SENT = 0xFFFFFFFFu, QNAN = 0x7FC00000u;
        constexpr uint32_t TILE_BYTES = 4u * PER * 1024u;
        constexpr int R = MI >= 4 ? 4 : MI >= 2 ? 2 : 1;  // launcher guarantees S >= R (MI = 1, experimental: one reducer)
        const rsrc_t slres = mk_rsrc(p.slabs, (uint32_t)S * (uint32_t)tiles * TILE_BYTES);
        const uint32_t lane_off = (uint32_t)cw * (PER * 1024u) + (uint32_t)lane * 16u;
        const int ridx = slice - (S - R);  // >= 0: a reducer
        const uint32_t mine = ((uint32_t)slice * (uint32_t)tiles + (uint32_t)tile) * TILE_BYTES + lane_off;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (ridx >= 0 && (i % R) == ridx) continue;  // reduced here
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32x4 b = __builtin_bit_cast(u32x4, acc[i][c]);
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (b[e] == SENT) ? QNAN : b[e];
                // soffset stays the constant 0 on 16-byte buffer stores (gemm_tiled.hip: store-data hazard)
                __builtin_amdgcn_raw_buffer_store_b128(b, slres, mine + (uint32_t)(i * 4 + c) * 1024u, 0, 16 /* sc1 */);
            }
        }
        if (ridx < 0) return;
        const u32x4 sent4 = {SENT, SENT, SENT, SENT};
        // (the row tile is a COMPILE-TIME index inside the unrolled loop: a run-time index into acc[] sends the whole
        // accumulator array to scratch memory -- 10 x slower, and scratch traffic counts in vmcnt)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if ((i % R) != ridx) continue;
            float4_t sum[4] = {float4_t{0.f, 0.f, 0.f, 0.f}, float4_t{0.f, 0.f, 0.f, 0.f}, float4_t{0.f, 0.f, 0.f, 0.f}, float4_t{0.f, 0.f, 0.f, 0.f}};
            for (int k0 = 0; k0 < S - 1; k0 += 4) {  // four other slices x four chunks = 16 loads in flight per poll
                u32x4 v[4][4];
                uint32_t soff[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    soff[u] = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)(((uint32_t)(k0 + u + (k0 + u >= slice ? 1 : 0)) * (uint32_t)tiles + (uint32_t)tile) * TILE_BYTES));
                for (unsigned spins = 0;; ++spins) {
                    uint32_t pending = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            v[u][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                    slres, (k0 + u < S - 1) ? lane_off + (uint32_t)(i * 4 + c) * 1024u : OOB, soff[u], 16));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            pending |= (v[u][c][0] == SENT) | (v[u][c][1] == SENT) | (v[u][c][2] == SENT) | (v[u][c][3] == SENT);
                    if (!pending) break;
                    if (spins > (1u << 18)) {
                        *p.err = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {  // slice order, this block's own partial at its place: bitwise reproducible
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (k0 + u == slice && k0 + u < S - 1) sum[c] += acc[i][c];
                        sum[c] += __builtin_bit_cast(float4_t, v[u][c]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + u < S - 1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            __builtin_amdgcn_raw_buffer_store_b128(sent4, slres, lane_off + soff[u] + (uint32_t)(i * 4 + c) * 1024u, 0, 16);
                    }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][c] = (slice == S - 1) ? sum[c] + acc[i][c] : sum[c];
        }
    }

    // ---- epilogue: lane (j, kb) holds rows 16 i + 4 kb + e, columns 8 j + 4 ph + c of its wave's 128 columns
    const int col = NMAJOR ? ncol0 : n0 + set * 128 + 8 * j + 4 * ph;  // NK: four consecutive columns per lane as well
    if (col >= p.N) return;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const half4_t bv = *reinterpret_cast<const half4_t*>(p.bias + col);
#pragma unroll
        for (int c = 0; c < 4; ++c) b4[c] = (float)bv[c];
    }
    const int ridx = S > 1 ? slice - (S - (MI >= 4 ? 4 : MI >= 2 ? 2 : 1)) : -1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (S > 1 && (i % (MI >= 4 ? 4 : MI >= 2 ? 2 : 1)) != ridx) continue;  // another reducer writes this row tile
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * i + 4 * kb + e;
            if (row < p.M) {
                half4_t o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (half_t)(acc[i][c][e] + b4[c]);
                *reinterpret_cast<half4_t*>(p.y + (int64_t)row * p.N + col) = o;
            }
        }
    }
}

template <int MI, int MODE>
int launch_skinny(const SkinnyParams& p, dim3 grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> opted{0};
    (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_skinny_kernel<MI, MODE>), opted);
    hipLaunchKernelGGL((awq_gemm_skinny_kernel<MI, MODE>), grid, dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace

// The experimental entry point (tools/experimental/gemm_skinny_nk/probe.py): x [M, K] fp16, GEMV-layout buffers, y [M, N];
// workspace = a split-K workspace of libawq_hip.so (awq_gemm_workspace_bytes / _init: control words + sentinel-filled exchange).
// nk = 0 runs the ORIGINAL form on GEMM-layout buffers (a check that the copy of the kernel still is the kernel), 1 the N-major
// form with 64-wide steps, 2 the wide form (K % 256 == 0).
extern "C" __attribute__((visibility("default"))) int awq_exp_gemm_skinny(const uint16_t* x, const int32_t* qweight, const uint16_t* scales,
                                                                          const int32_t* qzeros, uint16_t* y, int M, int K, int N, int g, int ZW,
                                                                          int nk, int splitk, void* workspace, size_t workspace_bytes,
                                                                          void* stream) {
    uint32_t magic;
    if (!(M >= 1 && M <= 64 && K >= 128 && K % 64 == 0 && g % 64 == 0 && K % g == 0 && N % 8 == 0 && K < 65536 &&
          awq_magic_u32((uint32_t)g, (uint32_t)K + 64u, &magic)))
        return AWQ_ERR_UNSUPPORTED;
    if ((int64_t)(M > 32 ? 4 : 2) * ((N + 255) / 256) > 256) return AWQ_ERR_UNSUPPORTED;
    if (nk < 0 || nk > 2 || (nk == 2 && K % 256)) return AWQ_ERR_UNSUPPORTED;
    // (MI = 1 -- one 16-row tile, batches up to 16 -- exists in this experimental copy only, for the N-major forms)
    const int MI = (M <= 16 && nk) ? 1 : M <= 32 ? 2 : 4, BM = 16 * MI, R = MI >= 4 ? 4 : MI >= 2 ? 2 : 1;
    const int tiles = (N + 255) / 256, T = K / 64;
    const size_t tile_bytes = (size_t)4 * MI * 4 * 1024;
    int S = splitk > 0 ? splitk : (tiles >= 32 ? 4 : K / 512);
    if (splitk <= 0) {
        if (MI == 1) S = (512 + tiles - 1) / tiles;  // a 16-row block is light: at least ~512 blocks
        if (S < 4) S = 4;
        if (S > 16) S = 16;
    }
    if (S > T / 2) S = T / 2;
    if (S < 1) S = 1;
    int sps = (T + S - 1) / S;
    const int sps_max = (128 * 1024) / (BM * 128);
    if (sps > sps_max) sps = sps_max;
    if (nk == 2) {  // whole units of four steps per slice (sps_max is a multiple of 4)
        sps = (sps + 3) & ~3;
        if (sps > sps_max) sps = sps_max;
    }
    S = (T + sps - 1) / sps;
    if (S > 1 && S < R) {
        if (T / 2 >= R) {
            sps = (T + R - 1) / R;
            if (nk == 2) sps = (sps + 3) & ~3;
            S = (T + sps - 1) / sps;
        }
        if (S < R) {
            S = 1;
            sps = T;
            if (sps > (128 * 1024) / (BM * 128)) return AWQ_ERR_UNSUPPORTED;
        }
    }
    char* ws = static_cast<char*>(workspace);
    const size_t half = workspace_bytes > AWQ_WS_COUNTER_BYTES ? ((workspace_bytes - AWQ_WS_COUNTER_BYTES) / 2) & ~(size_t)255 : 0;
    if (S > 1 && (!ws || (size_t)S * tiles * tile_bytes > half)) return AWQ_ERR_WORKSPACE;
    SkinnyParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.bias = nullptr;
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N; p.g = g;
    p.S = S; p.sps = sps;
    p.g_magic = magic;
    p.slabs = ws ? reinterpret_cast<float*>(ws + AWQ_WS_COUNTER_BYTES) : nullptr;
    p.err = reinterpret_cast<int*>(ws);
    p.KW = K / 8; p.ZW = ZW; p.SW = 8 * ZW;
    const size_t a_bytes = (size_t)sps * BM * 128, fold_bytes = (size_t)4 * MI * 4 * 1024;
    const size_t lds = a_bytes > fold_bytes ? a_bytes : fold_bytes;
    if (S > 1 && R * tiles > 256 * (lds <= 80 * 1024 ? 2 : 1)) return AWQ_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)tiles, (unsigned)S, 1u);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nk == 2) return MI == 1 ? launch_skinny<1, 2>(p, grid, lds, st) : MI == 2 ? launch_skinny<2, 2>(p, grid, lds, st) : launch_skinny<4, 2>(p, grid, lds, st);
    if (nk == 1) return MI == 1 ? launch_skinny<1, 1>(p, grid, lds, st) : MI == 2 ? launch_skinny<2, 1>(p, grid, lds, st) : launch_skinny<4, 1>(p, grid, lds, st);
    return MI == 2 ? launch_skinny<2, 0>(p, grid, lds, st) : launch_skinny<4, 0>(p, grid, lds, st);
}
