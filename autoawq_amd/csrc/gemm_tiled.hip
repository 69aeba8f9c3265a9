// gemm_tiled.hip -- fused int4-dequant + MFMA GEMM on the GEMM layout for M > 16 (prefill and
// batched decode), gfx950.
//
// Replaces the large-batch branch of awq/modules/linear/gemm.py:48-54 (dequantize_weights_cuda
// followed by torch.matmul) and awq_ext.gemm_forward_cuda (:56-58) for M > 16 in ONE kernel: the
// fp16 weight matrix is never materialised in HBM.
//
// Roofline: MFMA for M >= ~512 (AI = 2*M*K*N / bytes >> ridge 312 flop/B at M = 16384), HBM below.
// Algorithmic bytes per call: K*N/2 + (K/g)*(N/8)*4 + (K/g)*N*2 + M*K*2 + M*N*2;
// flops 2*M*K*N.
//
// Numerics: weights are dequantised to exactly the fp16 values the reference materialises
// ((w - z) * s with one rounding, awq/utils/packing_utils.py:98-100) and multiplied with fp32
// accumulation by v_mfma_f32_16x16x32_f16 -- the same arithmetic as dequant + fp16 GEMM.
//
// Structure (block = BM x BN output tile, BM in {32, 64, 128}, 4 waves, K step 64, LDS double buffered)
//   * A (activations): global 16-byte loads -> registers -> LDS rows of 72 halfs (144 B pitch:
//     conflict-free ds_read_b64 of MFMA A fragments, 16-byte aligned ds_write_b128).
//   * B (weights): each thread owns ONE packed word column and 4 rows of the K step; the word is
//     decoded with the (nib | 0x6400) - (1024 + z) trick (awq_device.h), scaled, and written as 16
//     bytes (8 N-adjacent fp16) into 16-column sub-tiles  [n/16][k][16].  The N-major layout is
//     turned into K-major MFMA B fragments by ds_read_b64_tr_b16 (CDNA4 transpose read): lane
//     (n, kb) receives k = 4*kb + {0..3} from one read and k = 16 + 4*kb + {0..3} from a second;
//     the A fragment uses the same K-slot order, so no data is ever transposed by the VALU.
//   * one barrier per K step: tile t+1 is fetched into registers before tile t is multiplied and
//     stored into the other LDS buffer afterwards.
#include <cstdlib>

#include "awq_device.h"
#include "awq_internal.h"

namespace {

struct TiledParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    const half_t* bias;
    half_t* y;
    int M, K, N, g;
    uint32_t g_magic;  // 2^32 / g + 1: (k * g_magic) >> 32 == k / g for k < K (K * g < 2^32, checked by the launcher)
    int tiles_m, tiles_n;
    int S, steps_per_slice;  // split-K: K steps [slice*steps_per_slice, ...) per block
    float* slabs;            // exchange region [S-1][tiles][4 waves][4*NT][64 lanes] float4, sentinel-filled
    int* err;
};

typedef short short4_t __attribute__((__vector_size__(4 * sizeof(short))));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int BK_DEFAULT = 64;  // K step of every tile except the four-wave 128 x 256 one (32: two blocks fit a CU)
constexpr uint32_t OOB = 0x80000000u;

AWQ_DEV rsrc_t mk_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

AWQ_DEV float4_t mfma16(half8_t a, half8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// BM x BN output tile per block; waves are laid out 2 x 2 (128 x 128), 2 x 4 (128 x 256: 8 waves,
// each 64 x 64 like the 128 x 128 tile, half the activation traffic per flop) or 1 x 4 (BM < 128).
// Phase timestamps for tools/trace_tiled.py (debug build only: -DAWQ_GEMV_TRACE)
#ifdef AWQ_GEMV_TRACE
__device__ unsigned long long* g_awq_trace_tiled = nullptr;
#define AWQ_TSTAMP(slot)                                                                                 \
    do {                                                                                                 \
        if (g_awq_trace_tiled && lane == 0)                                                              \
            g_awq_trace_tiled[((size_t)blockIdx.x * 8 + wave) * 16 + (slot)] = wall_clock64();            \
    } while (0)
#else
#define AWQ_TSTAMP(slot) do { } while (0)
#endif

// FAT: the 128 x 256 tile on FOUR waves of 64 x 128 (instead of eight of 64 x 64): a quarter fewer LDS
// fragment bytes per MFMA (12 fragments feed 32 MFMAs instead of 8 feeding 16), 128 accumulator registers.
template <int BM, int BN, bool SPLITK, int BK = BK_DEFAULT, bool FAT = false>
__global__ __launch_bounds__((FAT ? BM / 64 : (BM >= 128 ? 2 : 1)) * (BN >= 256 ? (FAT ? 2 : 4) : (BM >= 128 ? 2 : 4)) * 64, (FAT && BM == 128) ? 2 : 1)
void awq_gemm_tiled_kernel(TiledParams p) {
    constexpr int APITCH = BK + 8;  // halfs per A row in LDS (144 / 80 bytes)
    constexpr int CPR = BK / 8;     // 16-byte activation chunks per row of a K step
    constexpr int WGM = FAT ? BM / 64 : (BM >= 128 ? 2 : 1);   // waves along M
    constexpr int WGN = BN >= 256 ? (FAT ? 2 : 4) : (BM >= 128 ? 2 : 4);   // waves along N
    constexpr int NTHR = WGM * WGN * 64;
    constexpr int WM = BM / WGM;            // rows per wave
    constexpr int MI = WM / 16;             // 16-row MFMA tiles per wave
    constexpr int WN = BN / WGN;            // columns per wave
    constexpr int NT = WN / 16;             // 16-column MFMA tiles per wave
    constexpr int ACH = BM * CPR / NTHR;    // 16-byte activation chunks per thread per K step
    constexpr int BASSIGN = (BN / 8) * (BK / 4);  // (word column, 4-row group) assignments per K step
    constexpr int WPT = BASSIGN >= NTHR ? BASSIGN / NTHR : 1;  // per thread (threads past BASSIGN idle in the B staging)
    static_assert(ACH >= 1 && BM * CPR % NTHR == 0, "activation chunks must divide evenly");
    constexpr int A_BYTES = BM * APITCH * 2;
    constexpr int B_BYTES = BK * BN * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][A | B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l15 = lane & 15, kb = lane >> 4;
    // consecutive blocks walk the N tiles of one M tile (they share the activation rows in L2)
    const int tile_id = blockIdx.x % (p.tiles_m * p.tiles_n), slice = blockIdx.x / (p.tiles_m * p.tiles_n);
    const int mt = tile_id / p.tiles_n, nt = tile_id % p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int NW = p.N >> 3;
    AWQ_TSTAMP(0);

    const rsrc_t xres = mk_rsrc(p.x, (uint32_t)((int64_t)p.M * p.K * 2));
    const rsrc_t wres = mk_rsrc(p.qweight, (uint32_t)((int64_t)p.K * NW * 4));
    const rsrc_t zres = mk_rsrc(p.qzeros, (uint32_t)((int64_t)(p.K / p.g) * NW * 4));
    const rsrc_t sres = mk_rsrc(p.scales, (uint32_t)((int64_t)(p.K / p.g) * p.N * 2));

    // ---- per-thread staging assignments
    // A: ACH chunks of 16 bytes: chunk c = tid + NTHR*i -> row c/8, 8 halfs at k = 8*(c%8)
    uint32_t a_voff[ACH];
    int a_lds[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + NTHR * i, row = c / CPR, kc = c % CPR;
        a_voff[i] = (m0 + row < p.M) ? (uint32_t)(((int64_t)(m0 + row) * p.K + 8 * kc) * 2) : OOB;
        a_lds[i] = (row * APITCH + 8 * kc) * 2;
    }
    // B: word column wc (of BN/8), rows 4*rg .. 4*rg+3 of the K step; WPT such assignments
    int b_wc[WPT], b_rg[WPT];
    uint32_t b_voff[WPT], z_voff[WPT], s_voff[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int idx = tid + NTHR * i;
        b_wc[i] = idx % (BN / 8);
        b_rg[i] = idx / (BN / 8);  // 0 .. BK/4 - 1
        const int w = (n0 >> 3) + b_wc[i];
        const bool ok = w < NW && idx < BASSIGN;
        b_voff[i] = ok ? (uint32_t)(((int64_t)(4 * b_rg[i]) * NW + w) * 4) : OOB;
        z_voff[i] = ok ? (uint32_t)w * 4u : OOB;
        s_voff[i] = ok ? (uint32_t)w * 16u : OOB;
    }

    struct Regs {
        u32x4 a[ACH];
        uint32_t w[WPT][4], z[WPT];
        u32x4 s[WPT];
    };

    // global -> registers for K step t.  `valid == false` keeps the instruction count (the
    // outstanding-load bookkeeping of the pipelined loop stays static) but moves every lane out
    // of range: zeros, no memory traffic.
    auto fetch = [&](Regs& R, int t, bool valid = true) {
        const uint32_t k0 = (uint32_t)t * BK;
        const uint32_t kill = valid ? 0u : OOB;
#pragma unroll
        for (int i = 0; i < ACH; ++i)
            R.a[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, a_voff[i] | kill, valid ? k0 * 2u : 0u, 0));
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                R.w[i][r] = __builtin_amdgcn_raw_buffer_load_b32(wres, b_voff[i] | kill, valid ? (k0 + (uint32_t)r) * (uint32_t)NW * 4u : 0u, 0);
            const uint32_t grp = __umulhi(k0 + 4u * (uint32_t)b_rg[i], p.g_magic);  // row / g without a divide
            R.z[i] = __builtin_amdgcn_raw_buffer_load_b32(zres, (z_voff[i] + grp * (uint32_t)NW * 4u) | kill, 0, 0);
            R.s[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sres, (s_voff[i] + grp * (uint32_t)p.N * 2u) | kill, 0, 0));
        }
    };

    auto stage = [&](const Regs& R, int buf) {  // registers -> LDS (dequantising B)
        unsigned char* A = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* B = A + A_BYTES;
#pragma unroll
        for (int i = 0; i < ACH; ++i) *reinterpret_cast<u32x4*>(A + a_lds[i]) = R.a[i];
        if (BASSIGN < NTHR && tid >= BASSIGN) return;  // wave-uniform: this thread has no word of the B tile
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const uint32_t qz = R.z[i];
            const half2_t z0 = u2h2(awq_pair_magic<0>(qz)), z1 = u2h2(awq_pair_magic<1>(qz));
            const half2_t z2 = u2h2(awq_pair_magic<2>(qz)), z3 = u2h2(awq_pair_magic<3>(qz));
            const u32x4 sv = R.s[i];
            // sub-tile (wc/2) of 16 columns: [k][16] halfs, this word is the (wc&1) half of a row
            unsigned char* dst = B + ((b_wc[i] >> 1) * BK * 16 + (b_wc[i] & 1) * 8) * 2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t q = R.w[i][r];
                u32x4 o;
                o[0] = h22u(awq_dq_pair<0>(q, z0, u2h2(sv[0])));
                o[1] = h22u(awq_dq_pair<1>(q, z1, u2h2(sv[1])));
                o[2] = h22u(awq_dq_pair<2>(q, z2, u2h2(sv[2])));
                o[3] = h22u(awq_dq_pair<3>(q, z3, u2h2(sv[3])));
                *reinterpret_cast<u32x4*>(dst + (4 * b_rg[i] + r) * 32) = o;
            }
        }
    };

    float4_t acc[MI][NT];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = float4_t{0.f, 0.f, 0.f, 0.f};

    // fragment addresses (bytes) inside a buffer
    //   A: row wm*WM + 16*i + l15, k = kk*32 + 4*kb (+16 for the second half)
    const int a_frag = ((wm * WM + l15) * APITCH + 4 * kb) * 2;
    //   B (tr read): sub-tile (wn*WN/16 + jn); lane t=l15 of group kb reads 4 halfs of row
    //   kk*32 + 4*kb + (t>>2) at columns 4*(t&3); the hardware hands lane l15 column l15
    const int b_frag = ((wn * NT) * BK * 16 + (4 * kb + (l15 >> 2)) * 16 + 4 * (l15 & 3)) * 2;

    auto compute = [&](int buf) {
        const unsigned char* A = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* B = A + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            half8_t bf[NT];
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
                const unsigned char* bp = B + b_frag + (jn * BK * 16 + kk * 32 * 16) * 2;
                const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4_t*)(bp));
                const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4_t*)(bp + 16 * 16 * 2));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                bf[jn] = __builtin_bit_cast(half8_t, u32x4{l2[0], l2[1], h2[0], h2[1]});
            }
            half8_t af[MI];  // every fragment of the K slab is requested before the first MFMA: with one shared
                             // A register set each of the MI groups waited out an LDS round trip of its own
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const unsigned char* ap = A + a_frag + (16 * i * APITCH + kk * 32) * 2;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(ap);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(ap + 32);
                af[i] = __builtin_bit_cast(half8_t, u32x4{lo[0], lo[1], hi[0], hi[1]});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) acc[i][jn] = mfma16(af[i], bf[jn], acc[i][jn]);
        }
    };

    // One barrier per K step; global loads run DEPTH tiles ahead in registers: tile t+DEPTH is
    // requested before tile t is multiplied, tile t+1 (requested DEPTH-1 steps ago) is decoded into
    // the other LDS buffer afterwards.  The small-batch tiles (BM <= 64) are latency-bound -- a
    // block's tile is 8 KB per step -- so they keep 4 tiles in flight (17 registers each); the
    // 128-row tiles keep one (a second register set there costs the co-resident block:
    // 675 -> 413 TF at M = 16384).
    constexpr int DEPTH = BM >= 128 ? 1 : 4;
    const int T = p.K / BK;
    const int t0 = slice * p.steps_per_slice, t1 = min(T, t0 + p.steps_per_slice);
    Regs R[DEPTH];
    if constexpr (DEPTH == 1) {
        fetch(R[0], t0);
        stage(R[0], 0);
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            if (t + 1 < t1) fetch(R[0], t + 1);
            compute((t - t0) & 1);
            if (t + 1 < t1) stage(R[0], (t - t0 + 1) & 1);
            __syncthreads();
        }
    } else {
        // Steady state without a branch: every step issues exactly one tile's loads (out of
        // range past the slice) and decodes exactly one, so the compiler's s_waitcnt vmcnt(N)
        // lets DEPTH-1 tiles stay in flight (with conditional fetches it fell back to vmcnt(0)
        // every step and the whole round trip was exposed: 1.1 us per K step).
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(R[d], t0 + d, t0 + d < t1);
        AWQ_TSTAMP(1);
        stage(R[0], 0);
        __syncthreads();
        AWQ_TSTAMP(2);
        int t = t0;
        for (; t + DEPTH <= t1; t += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {  // (t - t0) is a multiple of the even DEPTH: LDS buffer = d & 1
                fetch(R[d], t + d + DEPTH, t + d + DEPTH < t1);
                compute(d & 1);
                stage(R[(d + 1) % DEPTH], (d + 1) & 1);  // past the slice: decodes zeros into the idle buffer
                __syncthreads();
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d)  // tail: fewer than DEPTH steps, all of them already requested
            if (t + d < t1) {
                compute(d & 1);
                if (t + d + 1 < t1) stage(R[d + 1], (d + 1) & 1);
                __syncthreads();
            }
    }

    AWQ_TSTAMP(3);
    // ---- split-K combine in one fabric hop through self-validating slabs (same protocol as
    // gemv_mfma.hip): a wave's accumulators travel as [i][jn][lane] float4, fully coalesced; the
    // block of the last K slice polls, adds in slice order, then runs the epilogue.
    if constexpr (SPLITK) {
        constexpr uint32_t SENT = 0xFFFFFFFFu, QNAN = 0x7FC00000u;
        constexpr uint32_t WAVE_BYTES = MI * NT * 64 * 16, TILE_BYTES = WGM * WGN * WAVE_BYTES;
        const uint32_t ntiles = (uint32_t)(p.tiles_m * p.tiles_n);
        const rsrc_t slres = mk_rsrc(p.slabs, (uint32_t)(p.S - 1) * ntiles * TILE_BYTES);
        const uint32_t lane_off = (uint32_t)wave * WAVE_BYTES + (uint32_t)lane * 16u;
        // NOTE on every 16-byte buffer store below: the slab offset goes into the VGPR offset and
        // soffset stays the constant 0.  With an SGPR soffset the compiler's hazard recogniser
        // assumes "VALU may overwrite the data VGPRs of a >8-byte MUBUF store right away" is safe;
        // on gfx950 it is not (the next v_add clobbered dword 0 of the last chunk in some lanes,
        // profiles/r01_store_hazard.txt) -- with soffset = 0 it inserts the wait states.
        if (slice != p.S - 1) {
            const uint32_t pbase = lane_off + ((uint32_t)slice * ntiles + (uint32_t)tile_id) * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    u32x4 b = __builtin_bit_cast(u32x4, acc[i][jn]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = (b[e] == SENT) ? QNAN : b[e];
                    __builtin_amdgcn_raw_buffer_store_b128(b, slres, pbase + (uint32_t)(i * NT + jn) * 1024u, 0, 16);
                }
            AWQ_TSTAMP(4);
            return;
        }
        const u32x4 sent4 = {SENT, SENT, SENT, SENT};
        // Every poll is a fabric round trip, so a round requests up to 16 chunks per lane -- all
        // (i, jn) accumulators of GSL slices -- before anything is checked: S = 6 at BM = 32 takes
        // 2 round trips instead of 10 (a 32-row batch at 4096 x 11008: 24 -> see profiles/).
        constexpr int PER = MI * NT;                    // 16-byte chunks per lane per slice
        constexpr int GSL = PER >= 16 ? 1 : 16 / PER;   // slices per round
        for (int sl0 = 0; sl0 < p.S - 1; sl0 += GSL) {  // slice order: bitwise reproducible
            u32x4 v[GSL][PER];
            uint32_t soff[GSL];
#pragma unroll
            for (int g = 0; g < GSL; ++g) soff[g] = ((uint32_t)(sl0 + g) * ntiles + (uint32_t)tile_id) * TILE_BYTES;
            for (unsigned spins = 0;; ++spins) {
                uint32_t pending = 0;
#pragma unroll
                for (int g = 0; g < GSL; ++g)
#pragma unroll
                    for (int c = 0; c < PER; ++c)  // slices past S-1: out of range, zeros, no traffic
                        v[g][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                slres, (sl0 + g < p.S - 1) ? lane_off + (uint32_t)c * 1024u : OOB,
                                                                soff[g], 16));
#pragma unroll
                for (int g = 0; g < GSL; ++g)
#pragma unroll
                    for (int c = 0; c < PER; ++c)
                        pending |= (v[g][c][0] == SENT) | (v[g][c][1] == SENT) | (v[g][c][2] == SENT) | (v[g][c][3] == SENT);
                if (!pending) break;
                if (spins > (1u << 18)) {
                    *p.err = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int g = 0; g < GSL; ++g)
#pragma unroll
                for (int c = 0; c < PER; ++c) acc[c / NT][c % NT] += __builtin_bit_cast(float4_t, v[g][c]);
#pragma unroll
            for (int g = 0; g < GSL; ++g)
                if (sl0 + g < p.S - 1) {
#pragma unroll
                    for (int c = 0; c < PER; ++c)
                        __builtin_amdgcn_raw_buffer_store_b128(sent4, slres, lane_off + soff[g] + (uint32_t)c * 1024u, 0, 16);
                }
        }
    }

    AWQ_TSTAMP(5);
    // ---- epilogue: D register r of lane (col l15, quad kb) is row 4*kb + r of its 16x16 tile
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const int col = n0 + wn * WN + 16 * jn + l15;
            if (col >= p.N) continue;
            const float b = p.bias ? (float)p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + 16 * i + 4 * kb + r;
                if (row < p.M) p.y[(int64_t)row * p.N + col] = (half_t)(acc[i][jn][r] + b);
            }
        }
    }
    AWQ_TSTAMP(6);
}

template <int BM, int BN, int BK = BK_DEFAULT, bool FAT = false>
void launch_tiled(const TiledParams& p, unsigned grid, hipStream_t st) {
    constexpr size_t lds = 2 * (BM * (BK + 8) * 2 + BK * BN * 2);
    static std::atomic<unsigned long long> opted{0}, opted_split{0};
    (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_tiled_kernel<BM, BN, false, BK, FAT>), opted);
    if constexpr (!FAT) (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_tiled_kernel<BM, BN, true, BK, FAT>), opted_split);
    constexpr int NTHR = (FAT ? BM / 64 : (BM >= 128 ? 2 : 1)) * (BN >= 256 ? (FAT ? 2 : 4) : (BM >= 128 ? 2 : 4)) * 64;
    if constexpr (FAT) {  // chip-filling grids only: never split (its split-K form would spill)
        hipLaunchKernelGGL((awq_gemm_tiled_kernel<BM, BN, false, BK, FAT>), dim3(grid), dim3(NTHR), lds, st, p);
    } else {
        if (p.S > 1) hipLaunchKernelGGL((awq_gemm_tiled_kernel<BM, BN, true, BK, FAT>), dim3(grid), dim3(NTHR), lds, st, p);
        else hipLaunchKernelGGL((awq_gemm_tiled_kernel<BM, BN, false, BK, FAT>), dim3(grid), dim3(NTHR), lds, st, p);
    }
}

}  // namespace

#ifdef AWQ_GEMV_TRACE
extern "C" __attribute__((visibility("default"))) void awq_debug_set_trace_tiled(void* dev_buf) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_awq_trace_tiled), &dev_buf, sizeof(void*));
}
#endif

// AWQ_TILED_FAT=0 falls back to the 8-wave 64 x 64-per-wave tiles everywhere (A/B measurements)
static const bool g_fat = true;  // 64 x 128 "fat" waves for the chip-filling tiles (the r01 A/B switch read an environment variable here)

bool awq_gemm_tiled_supports(int M, int K, int N, int g) {
    if (M < 1) return false;
    if (K % 64 || N % 8) return false;
    if (g % 4) return false;  // a thread's 4 rows of a K step share one group
    if ((int64_t)M * K * 2 >= ((int64_t)1 << 31) || (int64_t)K * N / 2 >= ((int64_t)1 << 31)) return false;
    if ((int64_t)(K + 64) * g >= ((int64_t)1 << 32)) return false;  // exactness of the multiply-high group index
    return true;
}

int awq_launch_gemm_tiled(const AwqGemmArgs& a, int bn, int splitk) {
    if (!awq_gemm_tiled_supports(a.M, a.K, a.N, a.g)) return AWQ_ERR_UNSUPPORTED;
    if (bn == 0) bn = 128;
    if (!(bn == 128 || bn == 256)) return AWQ_ERR_UNSUPPORTED;
    TiledParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.bias = reinterpret_cast<const half_t*>(a.bias);
    p.y = reinterpret_cast<half_t*>(a.y);
    p.M = a.M; p.K = a.K; p.N = a.N; p.g = a.g;
    p.g_magic = (uint32_t)((((uint64_t)1 << 32) / (uint64_t)a.g) + 1);
    int BM = a.M <= 32 ? 32 : (a.M <= 64 ? 64 : 128);  // smallest tile that holds the batch: less split-K exchange
    // Chip-filling grids (>= 512 blocks, no K split) run fat-wave tiles -- each wave owns 64 x 128, 12 LDS
    // fragments feed 32 MFMAs: 256 x 256 / K step 64 on eight waves if 256-row tiles still fill the chip (the B
    // decode amortised over twice the rows: 1 block of 139 KB per CU), else 128 x 256 / K step 32 on four waves
    // (53 KB, two blocks per CU).  profiles/r01_gemm_tiled_vs_two_pass.txt: 739 -> 981-1020 TF at M = 16384.
    int fat = 0;  // 0: regular tiles, 1: 128 x 256 x 32, 2: 256 x 256 x 64
    if (g_fat && bn == 256 && splitk <= 1 && BM == 128) {
        if ((int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256) >= 512) { fat = 2; BM = 256; }
        else if ((int64_t)((a.M + 127) / 128) * ((a.N + 255) / 256) >= 512) fat = 1;
    }
    if (BM < 128) bn = 128;  // (32|64) x 256 with four tiles in flight drops to one wave per SIMD: 22 -> 27 us at M = 32
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = (a.N + bn - 1) / bn;
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    const int T = a.K / (fat == 1 ? 32 : BK_DEFAULT);
    // split K until ~2 blocks per CU are in the grid (small M: few output tiles, long K loops)
    // r69 / r92 sweeps: every slice of a 128-row tile ships 64 KB through the exchange, so those split
    // less (~320 blocks); nothing gains past 8 slices
    int S = splitk > 0 ? splitk : (BM == 128 ? (int)((320 + tiles / 2) / tiles) : (int)((512 + tiles - 1) / tiles));
    if (S > (splitk > 0 ? 16 : 8)) S = splitk > 0 ? 16 : 8;
    if (S < 1) S = 1;
    if (S > T / 4) S = T / 4 > 0 ? T / 4 : 1;  // at least 4 K steps per block
    const size_t tile_bytes = (size_t)BM * bn * sizeof(float);
    if (S > 1) {
        const size_t fit = a.exchange ? a.exchange_bytes / ((size_t)tiles * tile_bytes) + 1 : 1;
        if ((size_t)S > fit) S = (int)fit;
    }
    if (S < 1) S = 1;
    const int sps = (T + S - 1) / S;
    S = (T + sps - 1) / sps;
    p.S = S; p.steps_per_slice = sps;
    p.slabs = a.exchange;
    p.err = a.counters;
    if (S > 1 && (!a.exchange || !a.counters)) return AWQ_ERR_WORKSPACE;
    if (tiles * S > 0x7FFFFFFF) return AWQ_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)(tiles * S);
    if (BM == 32) launch_tiled<32, 128>(p, grid, a.stream);
    else if (BM == 64) launch_tiled<64, 128>(p, grid, a.stream);
    else if (bn == 128) launch_tiled<128, 128>(p, grid, a.stream);
    else if (fat == 2) launch_tiled<256, 256, 64, true>(p, grid, a.stream);
    else if (fat == 1) launch_tiled<128, 256, 32, true>(p, grid, a.stream);
    else launch_tiled<128, 256>(p, grid, a.stream);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
