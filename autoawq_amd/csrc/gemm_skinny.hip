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
                 : "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(W[3]), "=&v"(W[4]), "=&v"(W[5]), "=&v"(W[6]), "=&v"(W[7])                  \
                 : "v"(voff), "s"(rs), "s"(so[0]), "s"(so[1]), "s"(so[2]), "s"(so[3]), "s"(so[4]), "s"(so[5]), "s"(so[6]), "s"(so[7]))
#define AWQ_SK_BLOADZS(Z, S2, zvoff, zrs, zso, svoff, srs, sso)                                                                  \
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %3, %4 offen\n\tbuffer_load_dwordx2 %1, %5, %6, %7 offen"                        \
                 : "=&v"(Z), "=&v"(S2)                                                                                        \
                 : "v"(zvoff), "s"(zrs), "s"(zso), "v"(svoff), "s"(srs), "s"(sso))
#define AWQ_SK_BLOAD1(dst, voff, rs, soff) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff))
#define AWQ_SK_BLOAD2(dst, voff, rs, soff) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff))
#define AWQ_SK_DMA16(ldsaddr, voff, rs, soff)                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsaddr), "v"(voff), \
                 "s"(rs), "s"(soff)                                                                             \
                 : "m0")
#define AWQ_SK_LDS_READ16(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=&v"(dst) : "v"(addr))
#define AWQ_SK_WAIT_B(R, newer)                                                                                            \
    asm volatile("s_waitcnt vmcnt(" #newer ")"                                                                             \
                 : "+v"(R.w[0][0]), "+v"(R.w[0][1]), "+v"(R.w[0][2]), "+v"(R.w[0][3]), "+v"(R.w[0][4]), "+v"(R.w[0][5]),    \
                   "+v"(R.w[0][6]), "+v"(R.w[0][7]), "+v"(R.w[1][0]), "+v"(R.w[1][1]), "+v"(R.w[1][2]), "+v"(R.w[1][3]),    \
                   "+v"(R.w[1][4]), "+v"(R.w[1][5]), "+v"(R.w[1][6]), "+v"(R.w[1][7]), "+v"(R.z), "+v"(R.s))

template <int MI>  // 16-row tiles per block: BM = 16 * MI (32 | 64)
__global__ __launch_bounds__(512, 2) void awq_gemm_skinny_kernel(SkinnyParams p) {
    constexpr int BM = 16 * MI;
    constexpr int A_STEP = BM * 128;  // bytes of one 64-wide activation step in LDS
    constexpr int PER = MI * 4;       // 16-byte accumulator chunks per lane
    constexpr int B_OPS = 18;         // vector-memory operations of one weight fetch
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
    const int half0 = (nst + 1) >> 1;                          // the first K half takes the odd step
    const int w0 = kh ? half0 : 0, w1 = kh ? nst : half0;       // this wave's steps, relative to st0

    auto srd = [](const void* base, uint32_t bytes) -> u32x4 {
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
    };
    const uint32_t row_bytes = (uint32_t)NW * 4u;
    const u32x4 wsrd = srd(p.qweight, (uint32_t)p.K * row_bytes);
    const u32x4 zsrd = srd(p.qzeros, (uint32_t)(p.K / p.g) * row_bytes);
    const u32x4 ssrd = srd(p.scales, (uint32_t)(p.K / p.g) * (uint32_t)p.N * 2u);
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
    struct BRegs {
        uint32_t w[2][8];
        uint32_t z;
        u32x2 s;
    };
    auto fetch_b = [&](BRegs& R, int st) {  // st relative to st0; past the wave's range: the last step again (static counts)
        const uint32_t k0 = (uint32_t)(st0 + min(st, max(w1 - 1, w0))) * 64u;
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
    };
    BRegs B0, B1, B2;
    fetch_b(B0, w0);
    fetch_b(B1, w0 + 1);
    fetch_b(B2, w0 + 2);
    asm volatile("s_waitcnt vmcnt(54)" ::: "memory");  // everything older than the three weight fetches: the DMA pieces
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
        const uint32_t zp0 = __builtin_amdgcn_perm(R.z, R.z, sel0), zp1 = __builtin_amdgcn_perm(R.z, R.z, sel1);
        half2_t zm[4], sd[4];
        zm[0] = u2h2(and_or(zp0, 0x000F000Fu, 0x64006400u));
        zm[1] = u2h2(and_or(zp1, 0x000F000Fu, 0x64006400u));
        zm[2] = u2h2(and_or(zp0, 0x00F000F0u, 0x54005400u));
        zm[3] = u2h2(and_or(zp1, 0x00F000F0u, 0x54005400u));
        const half2_t s01 = u2h2(R.s[0]), s23 = u2h2(R.s[1]);
        sd[0] = __builtin_shufflevector(s01, s01, 0, 0);
        sd[1] = __builtin_shufflevector(s01, s01, 1, 1);
        sd[2] = __builtin_shufflevector(s23, s23, 0, 0);
        sd[3] = __builtin_shufflevector(s23, s23, 1, 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint32_t aa = a_base + (uint32_t)(st * A_STEP) + (uint32_t)(((4 * kk + kb) ^ hl) << 4);
            u32x4v af[MI];
            AWQ_SK_LDS_READ16(af[0], aa, 0);
            AWQ_SK_LDS_READ16(af[1], aa, 2048);
            if constexpr (MI > 2) {
                AWQ_SK_LDS_READ16(af[2], aa, 4096);
                AWQ_SK_LDS_READ16(af[3], aa, 6144);
            }
            u32x4v bf[4];
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
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i][c] = mfma16(af[i], bf[c], acc[i][c]);
        }
    };

    // ---- K loop, no barrier: wait for a step's words (the two younger fetches stay in flight), multiply, refill
    int st = w0;
    for (; st + 3 <= w1; st += 3) {
        AWQ_SK_WAIT_B(B0, 36); compute(B0, st);     fetch_b(B0, st + 3);
        AWQ_SK_WAIT_B(B1, 36); compute(B1, st + 1); fetch_b(B1, st + 4);
        AWQ_SK_WAIT_B(B2, 36); compute(B2, st + 2); fetch_b(B2, st + 5);
    }
    // The last refills are never consumed: to the compiler their destination registers are dead the moment they are
    // requested, so it would hand them to the temporaries of the steps below while the loads are still in flight -- and a
    // late load then overwrites a live value.  One wait that NAMES all three register sets keeps them allocated until
    // everything has landed; the (at most two) remaining steps then need no wait of their own.
    AWQ_SK_WAIT_B(B0, 0);
    AWQ_SK_WAIT_B(B1, 0);
    AWQ_SK_WAIT_B(B2, 0);
    if (st < w1) compute(B0, st);
    if (st + 1 < w1) compute(B1, st + 1);

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
        constexpr int R = MI >= 4 ? 4 : 2;  // launcher guarantees S >= R
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
    const int col = n0 + set * 128 + 8 * j + 4 * ph;
    if (col >= p.N) return;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const half4_t bv = *reinterpret_cast<const half4_t*>(p.bias + col);
#pragma unroll
        for (int c = 0; c < 4; ++c) b4[c] = (float)bv[c];
    }
    const int ridx = S > 1 ? slice - (S - (MI >= 4 ? 4 : 2)) : -1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (S > 1 && (i % (MI >= 4 ? 4 : 2)) != ridx) continue;  // another reducer writes this row tile
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

template <int MI>
int launch_skinny(const SkinnyParams& p, dim3 grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> opted{0};
    (void)awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemm_skinny_kernel<MI>), opted);
    hipLaunchKernelGGL((awq_gemm_skinny_kernel<MI>), grid, dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace

bool awq_gemm_skinny_supports(int M, int K, int N, int g) {
    uint32_t magic;
    // all reducer blocks of a launch must be co-resident (2 per tile up to 32 rows, 4 above; one 128 KB block per CU in the
    // worst case): N <= 32768 / 16384.  Wider matrices (7B gate|up at M > 32) take the LDS-tiled kernel.
    if ((int64_t)(M > 32 ? 4 : 2) * ((N + 255) / 256) > 256) return false;
    return M >= 9 && M <= 64 && K >= 128 && K % 64 == 0 && g % 64 == 0 && K % g == 0 && N % 8 == 0 && K < 65536 &&
           (int64_t)K * (N / 8) * 4 < ((int64_t)1 << 31) && (int64_t)(K / g) * N * 2 < ((int64_t)1 << 31) &&
           awq_magic_u32((uint32_t)g, (uint32_t)K + 64u, &magic);
}

// splitk: K slices per tile, 0 = auto
int awq_launch_gemm_skinny(const AwqGemmArgs& a, int splitk) {
    if (!awq_gemm_skinny_supports(a.M, a.K, a.N, a.g)) return AWQ_ERR_UNSUPPORTED;
    const int MI = a.M <= 32 ? 2 : 4, BM = 16 * MI, R = MI >= 4 ? 4 : 2;
    const int tiles = (a.N + 255) / 256, T = a.K / 64;
    const size_t tile_bytes = (size_t)4 * MI * 4 * 1024;
    // K split (r02 sweep, profiles/r02_skinny_sweep.txt): wide matrices (>= 32 column tiles) want FOUR long slices (4096 x
    // 11008, M = 32: 14.1 us at S = 4, 18.8-20.7 at 6-8; the exchange grows with S M N), narrow ones a slice of ~512 rows
    // (4096 x 4096: S = 8; 11008 x 4096: S = 16); the activation slice has to fit 128 KB of LDS (below)
    int S = splitk > 0 ? splitk : (tiles >= 32 ? 4 : a.K / 512);
    if (splitk <= 0) {
        if (S < 4) S = 4;
        if (S > 16) S = 16;
    }
    if (S > T / 2) S = T / 2;
    if (S < 1) S = 1;
    int sps = (T + S - 1) / S;
    // 128 KB of LDS for the activation slice (one block per CU at M > 32): with 96 KB a 64-row batch had to take six
    // slices instead of four: 4096 x 11008, M = 64: 25.2 -> 17.9 us
    const int sps_max = (128 * 1024) / (BM * 128);
    if (sps > sps_max) sps = sps_max;  // more slices than asked for: the slice has to fit
    S = (T + sps - 1) / sps;
    if (S > 1 && S < R) {  // the reducers are the last R slices: fewer slices than that -> exactly R, or no split at all
        if (T / 2 >= R) {
            sps = (T + R - 1) / R;
            S = (T + sps - 1) / sps;
        }
        if (S < R) {
            S = 1;
            sps = T;
            if (sps > (128 * 1024) / (BM * 128)) return AWQ_ERR_UNSUPPORTED;
        }
    }
    if (S > 1) {
        if (!a.exchange || !a.counters) return AWQ_ERR_WORKSPACE;
        if ((size_t)S * tiles * tile_bytes > a.exchange_bytes) return AWQ_ERR_UNSUPPORTED;
    }
    SkinnyParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(a.qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(a.qzeros);
    p.scales = reinterpret_cast<const half_t*>(a.scales);
    p.x = reinterpret_cast<const half_t*>(a.x);
    p.bias = reinterpret_cast<const half_t*>(a.bias);
    p.y = reinterpret_cast<half_t*>(a.y);
    p.M = a.M; p.K = a.K; p.N = a.N; p.g = a.g;
    p.S = S; p.sps = sps;
    if (!awq_magic_u32((uint32_t)a.g, (uint32_t)a.K + 64u, &p.g_magic)) return AWQ_ERR_UNSUPPORTED;
    p.slabs = a.exchange;
    p.err = a.counters;
    const size_t a_bytes = (size_t)sps * BM * 128, fold_bytes = (size_t)4 * MI * 4 * 1024;
    const size_t lds = a_bytes > fold_bytes ? a_bytes : fold_bytes;
    // Every reducer block (the last R slices of every tile) polls the other slices of its tile, so all R * tiles of them have
    // to be RESIDENT at once, or the resident ones spin on blocks that cannot be scheduled.  Two 512-thread
    // blocks fit a CU up to 80 KB of LDS each.  awq_gemm_skinny_supports() already keeps R * tiles <= 256; this is the net
    // under it.
    if (S > 1 && R * tiles > 256 * (lds <= 80 * 1024 ? 2 : 1)) return AWQ_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)tiles, (unsigned)S, 1u);
    return MI == 2 ? launch_skinny<2>(p, grid, lds, a.stream) : launch_skinny<4>(p, grid, lds, a.stream);
}
