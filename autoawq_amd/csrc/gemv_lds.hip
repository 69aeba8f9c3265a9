// gemv_lds.hip -- batched decode (2 <= M <= 16) on the GEMV layout: weights stream through LDS into MFMA, gfx950.
//
// Replaces awq_ext.gemv_forward_cuda / gemmv2_forward_cuda for small batches (awq/modules/linear/gemv.py:168-180) where the
// row-streaming kernel (gemv_rows.hip, VALU work proportional to M) stops paying.  Layout:
//   qweight [N, K/8] int32 (nibble i of word c = w[n, 8c+i]), qzeros [N, ZW] int32, scales [N, 8 ZW] fp16.
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 + (K/g)*N/2 + (K/g)*N*2 + M*K*2 + M*N*2.
//
// Why LDS: v_mfma_f32_16x16x32_f16 wants 16 DIFFERENT output rows in the 16 lanes l % 16 of an operand register, the memory
// system wants a wave instruction to read ONE row's contiguous bytes (profiles/r03_stream_probe3.txt: 16 rows x 64 bytes per
// instruction streams at a third of the rate of 1 KiB of one row).  LDS-DMA (`global_load_lds_dwordx4`) reconciles the two: a
// DMA instruction reads 512 contiguous bytes of each of two rows, and the LDS read that feeds the MFMA is a 16-byte
// ds_read_b128 per lane (n = l % 16, kb = l / 16) -- four packed words = the B fragments of four MFMAs.  A 16-byte chunk
// XOR-swizzle applied on the GLOBAL side of the DMA (lane i fetches chunk (i & 31) ^ (n & 15) of its row piece) makes
// those reads bank-conflict free without padding.
//  * PIECE = 16 rows x 512 bytes (1024 K) = 8 DMA instructions + one for the 8 group scales of each row (16 bytes per row)
//    + one for the zero word: 10 vector-memory instructions per piece, identical for every piece, so the in-order counter
//    `s_waitcnt vmcnt(10 (RD - 1))` names exactly one ring slot.  No VGPR is a DMA destination: nothing to audit.
//  * Activations: the block stages x ONCE as MFMA A fragments in LDS (pair-permuted to the (t, t+4) order the nibble
//    decode produces, one zero row for the unused batch rows) plus C0[m][g] = sum bias*x and SX[m][g] = sum x per group.
//  * Per 128-K group: 4 MFMAs (bias-coded weights), then y[m][n] += s[n,g] * (acc - C0[m][g] - z[n,g] * SX[m][g]): the
//    group factorisation of gemv_mfma.hip / gemv_rows.hip.  One-hot and zero inputs stay exact.
//  * A wave owns whole 16-row tiles (KS = 1) -- no cross-wave step at all -- or, for matrices with few tiles, the KS waves of a
//    block share a tile's pieces round-robin and their partial sums meet in LDS behind one barrier.  Nothing crosses a CU.
//  * y is parked in LDS and written after the stream has drained (a store inside the stream would make the counted
//    waits unreliable: stores count in vmcnt but do not retire in order with loads).
#include "awq_device.h"
#include "awq_internal.h"

namespace {

constexpr int PIECE_W = 8192;             // bytes of weights per piece (16 rows x 512 bytes)
constexpr int PIECE_B = PIECE_W + 1024 + 256;  // + scales (64 x 16-byte slots, 16 used) + zero words (64 x 4, 16 used)
constexpr int LDM = 10;                   // vector-memory instructions per piece request

struct LdsParams {
    const uint32_t* qweight;
    const uint32_t* qzeros;
    const half_t* scales;
    const half_t* x;
    half_t* y;
    int M, K, N;
    int KW, ZW, SW;
    int npiece;          // pieces per row: ceil(K / 1024)
    int G;               // groups: K / 128 (group_size == 128 in this kernel)
    int tiles;           // ceil(N / 16)
    int ks;              // waves sharing a tile (1 | 2 | 4)
    int tiles_base, tiles_rem, tiles_max;  // tiles per tile-owner (wave group): base (+1 for the first rem)
    int xf_bytes;        // LDS: A fragments [K/32][M+1][4] x 16 bytes
    int cg_bytes;        // LDS: C0 and SX, each [G][16] floats
    int ring_off, ybuf_off;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define AWQ_LDS_DMA16(voff, base, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")
#define AWQ_LDS_DMA4(voff, base, ldsaddr) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(ldsaddr) : "memory", "m0")

AWQ_DEV float4_t mfma16(u32x4 a, u32x4 b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}
AWQ_DEV float dot2(uint32_t a, uint32_t b, float c) { return __builtin_amdgcn_fdot2(u2h2(a), u2h2(b), c, false); }

// NW: waves per block (4 | 8); RD: pieces in flight per wave
template <int NW, int RD>
__global__ __launch_bounds__(NW * 64) void awq_gemv_lds_kernel(LdsParams p) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kb = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const int M = p.M;
    float* c0s = reinterpret_cast<float*>(smem + p.xf_bytes);
    float* sxs = reinterpret_cast<float*>(smem + p.xf_bytes + p.cg_bytes);
    const int ring = p.ring_off + wave * RD * PIECE_B;

    // ---- which tiles / pieces this wave streams: owner = a group of ks waves; owners are dealt the tiles evenly
    // (owner ids interleave the blocks: consecutive owners sit on different CUs, so a partial last round is spread over the chip)
    const int owner = (wave / p.ks) * gridDim.x + blockIdx.x, kq = wave % p.ks;  // NW / ks owners per block
    const int t0 = owner * p.tiles_base + min(owner, p.tiles_rem);
    const int ntile = p.tiles_base + (owner < p.tiles_rem ? 1 : 0);
    const int ppt = (p.npiece - kq + p.ks - 1) / p.ks;  // pieces of a tile this wave takes: kq, kq + ks, ...
    const int nunit = ntile * ppt;                 // flat units (tile-major)

    // request flat unit u into ring slot u % RD (past the end: the same addresses again, clamped -- the counted waits need it)
    const int rowl = lane >> 5, ch = lane & 31;    // DMA lane -> (row of the pair, chunk slot)
    auto request = [&](int u) {
        const bool live = u < nunit;  // past the end: one 16-byte line per instruction (the counted waits need the requests)
        const int uu = live ? u : 0;
        const int tl = uu / max(ppt, 1), pi = kq + (uu - tl * max(ppt, 1)) * p.ks;
        const int row0 = live ? (t0 + tl) * 16 : 0;
        const uint32_t slot = lds0 + (uint32_t)(ring + (u % RD) * PIECE_B);
        const int rowbytes = p.KW * 4, pbase = pi * 512;
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // rows 2 i, 2 i + 1: 512 bytes each, chunk-swizzled by the row
            const int r = 2 * i + rowl;
            const int q = ch ^ (r & 15);
            const int byte = min(pbase + 16 * q, rowbytes - 16);  // last piece of a ragged row: repeat its last chunk (x is 0 there)
            const uint32_t voff = live ? (uint32_t)(min(row0 + r, p.N - 1) * rowbytes + byte) : 0u;
            AWQ_LDS_DMA16(voff, p.qweight, slot + 1024u * i);
        }
        {   // the 8 group scales (16 bytes) / the zero word of this piece for row `lane & 15`
            const int r = min(row0 + n, p.N - 1);
            const uint32_t vs = (uint32_t)(r * p.SW * 2 + min(pi * 16, p.SW * 2 - 16));
            AWQ_LDS_DMA16(vs, p.scales, slot + (uint32_t)PIECE_W);
            const uint32_t vz = (uint32_t)((r * p.ZW + min(pi, p.ZW - 1)) * 4);
            AWQ_LDS_DMA4(vz, p.qzeros, slot + (uint32_t)(PIECE_W + 1024));
        }
    };
    // ---- activations: every thread requests its chunks of x FIRST (asm loads: a compiler-managed load would be waited for
    //      with vmcnt(0), i.e. behind the whole ring), then the ring, then `vmcnt(LDM RD)` = "x has landed"
    constexpr int XL = 4096 / NT;  // chunks of 8 activations per thread: M K <= 32768 (the launcher checks)
    const int xchunks = p.K >> 3, xtotal = M * xchunks;
    u32x4 xr[XL];
#pragma unroll
    for (int t = 0; t < XL; ++t) {
        const int e = min(tid + NT * t, xtotal - 1);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(xr[t]) : "v"((uint32_t)(e * 16)), "s"(p.x) : "memory");
    }
#pragma unroll
    for (int d = 0; d < RD; ++d) request(d);
    if constexpr (XL == 16)
        asm volatile("s_waitcnt vmcnt(%16) ; releases %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14 %15"
                     : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]), "+v"(xr[8 % XL]),
                       "+v"(xr[9 % XL]), "+v"(xr[10 % XL]), "+v"(xr[11 % XL]), "+v"(xr[12 % XL]), "+v"(xr[13 % XL]), "+v"(xr[14 % XL]), "+v"(xr[15 % XL])
                     : "n"(LDM * RD));
    else
        asm volatile("s_waitcnt vmcnt(%8) ; releases %0 %1 %2 %3 %4 %5 %6 %7"
                     : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7])
                     : "n"(LDM * RD));
    // -> MFMA A fragments in LDS, pair-permuted; fragment (c = k / 32 chunk, i = word of the chunk) for batch row m and K-lane
    //    kb sits at (((c4 * 4 + i) * (M + 1) + m) * 4 + kb) * 16 with c = 4 c4 + kb: one ds_read_b128 per MFMA
    {
#pragma unroll
        for (int t = 0; t < XL; ++t) {
            const int e = tid + NT * t;
            if (e < xtotal) {
                const int m = e / xchunks, c8 = e - m * xchunks;  // c8: 8 activations k = 8 c8 ..
                const u32x4 d = xr[t];
                u32x4 v;
                v[0] = __builtin_amdgcn_perm(d[2], d[0], 0x05040100u);  // (x0, x4)  bias 1024
                v[1] = __builtin_amdgcn_perm(d[2], d[0], 0x07060302u);  // (x1, x5)  bias 64
                v[2] = __builtin_amdgcn_perm(d[3], d[1], 0x05040100u);  // (x2, x6)  bias 1024
                v[3] = __builtin_amdgcn_perm(d[3], d[1], 0x07060302u);  // (x3, x7)  bias 64
                const int c = c8 >> 2, i = c8 & 3, c4 = c >> 2, kbb = c & 3;  // k = 32 c + 8 i
                *reinterpret_cast<u32x4*>(smem + ((((c4 * 4 + i) * (M + 1) + m) * 4 + kbb) * 16)) = v;
            }
        }
        for (int e = tid; e < p.npiece * 8 * 4 * 4; e += NT) {  // the zero row of every (c4, i), padded groups included
            const int ci = e >> 2, kbb = e & 3;
            *reinterpret_cast<u32x4*>(smem + (((ci * (M + 1) + M) * 4 + kbb) * 16)) = u32x4{0u, 0u, 0u, 0u};
        }
        // fragments of the padded groups (K .. 1024 npiece): zero
        for (int e = tid; e < (p.npiece * 8 - p.G) * 4 * M * 4; e += NT) {
            const int kbb = e & 3, m = (e >> 2) % M, ci = p.G * 4 + (e >> 2) / M;
            *reinterpret_cast<u32x4*>(smem + (((ci * (M + 1) + m) * 4 + kbb) * 16)) = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        // C0[g][m] = sum over the group of bias * x, SX[g][m] = sum x  (rows m >= M and padded groups: 0)
        for (int e = tid; e < p.npiece * 8 * 16; e += NT) {
            const int g = e >> 4, m = e & 15;
            float se = 0.f, so = 0.f;
            if (m < M && g < p.G) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kbb = 0; kbb < 4; ++kbb) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((((g * 4 + i) * (M + 1) + m) * 4 + kbb) * 16));
                        se = dot2(v[0], 0x3C003C00u, se);
                        so = dot2(v[1], 0x3C003C00u, so);
                        se = dot2(v[2], 0x3C003C00u, se);
                        so = dot2(v[3], 0x3C003C00u, so);
                    }
            }
            c0s[e] = 1024.f * se + 64.f * so;
            sxs[e] = se + so;
        }
        __syncthreads();
    }

    // ---- stream
    const int arow = min(n, M);  // A row of this lane (batch row n; the zero row past M) -- lane (n, kb) of the A operand
    float4_t yacc = {0.f, 0.f, 0.f, 0.f};
    float* ybuf = reinterpret_cast<float*>(smem + p.ybuf_off) + (size_t)wave * p.tiles_max * 256;  // [tile][m 16][n 16]
    for (int u = 0; u < nunit; ++u) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LDM * (RD - 1)) : "memory");
        const unsigned char* slot = smem + ring + (u % RD) * PIECE_B;
        const int tl = u / ppt, pi = kq + (u - tl * ppt) * p.ks;
        const int g0 = pi * 8;  // a piece is 8 groups; past K (last piece of a ragged row) the fragments and constants are zero
        const uint32_t zw = *reinterpret_cast<const uint32_t*>(slot + PIECE_W + 1024 + 4 * n);
        // the eight 16-byte weight chunks of this lane first: one LDS round trip for the whole piece instead of one per group
        u32x4 wq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 4 * j + kb;
            wq[j] = *reinterpret_cast<const u32x4*>(slot + n * 512 + (((q ^ n) & 15) | (q & 16)) * 16);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 w = wq[j];
            float4_t acc = {0.f, 0.f, 0.f, 0.f};
            const int gch = (g0 + j) * 4;  // (c4 * 4 + i) of the group's first fragment: c4 = g0 + j
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(smem + ((((gch + i) * (M + 1) + arow) * 4 + kb) * 16));
                const uint32_t ww = w[i], w8 = ww >> 8;
                const u32x4 b = {and_or(ww, 0x000F000Fu, 0x64006400u), and_or(ww, 0x00F000F0u, 0x54005400u),
                                 and_or(w8, 0x000F000Fu, 0x64006400u), and_or(w8, 0x00F000F0u, 0x54005400u)};
                acc = mfma16(a, b, acc);
            }
            const float scl = (float)*reinterpret_cast<const half_t*>(slot + PIECE_W + 16 * n + 2 * j);
            const float zf = (float)((zw >> (4 * j)) & 15u);
            const float4_t c0 = *reinterpret_cast<const float4_t*>(c0s + (g0 + j) * 16 + 4 * kb);
            const float4_t sx = *reinterpret_cast<const float4_t*>(sxs + (g0 + j) * 16 + 4 * kb);
#pragma unroll
            for (int r = 0; r < 4; ++r) yacc[r] = __builtin_fmaf(scl, __builtin_fmaf(-zf, sx[r], acc[r] - c0[r]), yacc[r]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of the slot has returned before it is overwritten
        request(u + RD);
        if (u - tl * ppt == ppt - 1) {  // last piece of the tile for this wave: park the partial sums
#pragma unroll
            for (int r = 0; r < 4; ++r) ybuf[tl * 256 + (4 * kb + r) * 16 + n] = yacc[r];
            yacc = float4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.ks > 1) __syncthreads();

    // ---- y: a wave (ks == 1) or the first wave of each owner (ks > 1, adding the ks partials) writes its tiles
    if (kq == 0) {
        for (int tl = 0; tl < ntile; ++tl) {
            const int row0 = (t0 + tl) * 16;
            for (int e = lane; e < M * 16; e += 64) {
                const int m = e >> 4, nn = e & 15;
                float s = 0.f;
                for (int k2 = 0; k2 < p.ks; ++k2) s += ybuf[(size_t)k2 * p.tiles_max * 256 + tl * 256 + m * 16 + nn];
                if (row0 + nn < p.N) p.y[(int64_t)m * p.N + row0 + nn] = (half_t)s;
            }
        }
    }
}

}  // namespace

bool awq_gemv_lds_supports(int M, int K, int N, int g) {
    if (M < 2 || M > 16 || N < 1 || K < 128 || K % 128 || g != 128 || (int64_t)M * K > 32768) return false;
    if ((int64_t)N * K / 2 >= ((int64_t)1 << 31) || (int64_t)N * (K / 128) * 2 >= ((int64_t)1 << 31)) return false;
    const int np = (K + 1023) / 1024;
    const size_t xf = (size_t)np * 32 * (M + 1) * 64, cg = (size_t)np * 8 * 16 * 4;
    return xf + 2 * cg + 8 * PIECE_B + 8 * 1024 <= 160 * 1024;  // eight waves, one piece in flight and one tile buffer each, at least
}

int awq_launch_gemv_lds(const uint16_t* x, const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* y,
                        int M, int K, int N, int g, int ZW, int ks, int depth, hipStream_t st) {
    if (!awq_gemv_lds_supports(M, K, N, g)) return AWQ_ERR_UNSUPPORTED;
    if (ZW * 8 < K / 128) return AWQ_ERR_BAD_SHAPE;
    LdsParams p;
    p.qweight = reinterpret_cast<const uint32_t*>(qweight);
    p.qzeros = reinterpret_cast<const uint32_t*>(qzeros);
    p.scales = reinterpret_cast<const half_t*>(scales);
    p.x = reinterpret_cast<const half_t*>(x);
    p.y = reinterpret_cast<half_t*>(y);
    p.M = M; p.K = K; p.N = N;
    p.KW = K / 8; p.ZW = ZW; p.SW = 8 * ZW;
    p.npiece = (K + 1023) / 1024;
    p.G = K / 128;
    p.tiles = (N + 15) / 16;
    // waves per tile: whole tiles per wave once there are enough tiles for the 1024 waves of a full grid, else the waves of a
    // block share a tile's pieces (4096 rows = 256 tiles: four waves each)
    if (ks != 1 && ks != 2 && ks != 4) ks = p.tiles >= 1280 ? 1 : (p.tiles >= 640 ? 2 : 4);
    if (ks > p.npiece) ks = p.npiece >= 2 ? 2 : 1;
    p.ks = ks;
    // eight waves with ONE piece in flight each beat four with two (one wave per SIMD is instruction-issue bound: every LDS round
    // trip and MFMA dependency stalls the only wave there is)
    const int NW = depth >= 2 ? 4 : 8, RD = depth >= 2 ? (depth > 3 ? 3 : depth) : 1;
    const int owners_per_block = NW / ks;
    int blocks = p.tiles < 256 ? p.tiles : 256;  // owner ids interleave the blocks: every CU gets its share of the owners
    const int owners = blocks * owners_per_block;
    p.tiles_base = p.tiles / owners;
    p.tiles_rem = p.tiles % owners;
    p.tiles_max = p.tiles_base + (p.tiles_rem ? 1 : 0);
    p.xf_bytes = p.npiece * 32 * (M + 1) * 64;
    p.cg_bytes = p.npiece * 8 * 16 * 4;
    p.ring_off = p.xf_bytes + 2 * p.cg_bytes;
    const int ybuf_bytes = NW * p.tiles_max * 1024;
    if ((size_t)p.ring_off + (size_t)NW * RD * PIECE_B + ybuf_bytes > 160 * 1024) return AWQ_ERR_UNSUPPORTED;
    p.ybuf_off = p.ring_off + NW * RD * PIECE_B;
    const size_t lds = (size_t)p.ybuf_off + ybuf_bytes;
#define AWQ_LDS_CASE(NWV, RDV)                                                                                                        \
    if (NW == NWV && RD == RDV) {                                                                                                      \
        static std::atomic<unsigned long long> opted{0};                                                                               \
        if (!awq_lds_opt_in(reinterpret_cast<const void*>(&awq_gemv_lds_kernel<NWV, RDV>), opted)) return AWQ_ERR_LAUNCH;              \
        hipLaunchKernelGGL((awq_gemv_lds_kernel<NWV, RDV>), dim3((unsigned)blocks), dim3(NWV * 64), lds, st, p);                       \
        return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;                                                              \
    }
    AWQ_LDS_CASE(8, 1) AWQ_LDS_CASE(4, 2) AWQ_LDS_CASE(4, 3)
#undef AWQ_LDS_CASE
    return AWQ_ERR_UNSUPPORTED;
}
