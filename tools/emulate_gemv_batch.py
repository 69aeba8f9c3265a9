#!/usr/bin/env python3
"""CPU model of csrc/gemv_batch.hip's index algebra (no GPU): the DMA lane -> (row, chunk) map with its global-side XOR
swizzle, the LDS image of a piece, the staging area of the activations with ITS swizzle, the fragment reads, the nibble decode order against the pair-permuted activations, the
MFMA operand / accumulator layouts (those of csrc/gemv_lds.hip, which the GPU suite pins), the K split over the waves of a
block with passes, the partial-tile reduction and the final store map.  It follows the kernel's expressions line by line and
compares the result with x @ dequant(W)^T; tests/test_gemv_batch_model.py runs it on ragged shapes."""
import numpy as np


def xs_f(m):
    return ((m & 3) << 2) | ((m >> 2) & 3)


def plan(M, K, N, form=0, rd_req=0, blocks_cap=256):
    """plan_batch() of csrc/gemv_batch.hip (form: 0 auto, 1 the activations through the LDS staging area, 2 direct fragment loads)"""
    MI = 2 if M > 16 else 1
    G = K // 128
    GW = 4
    wk = 1
    while wk < 8 and wk * GW < G:
        wk *= 2
    wt = 8 // wk
    passes = -(-G // (wk * GW))
    tiles = -(-N // 16)
    want = -(-tiles // wt)
    blocks = min(want, blocks_cap)  # (the launcher: 256 = one block per CU; the model lowers it to reach several tiles per owner)
    owners = blocks * wt
    tb, tr = tiles // owners, tiles % owners
    tiles_max = tb + (1 if tr else 0)
    ystage = wt * tiles_max * MI * 1024
    plain = 2 * MI * 1024 if wk > 1 else 0
    staged = max(min(M, 8) * 1024, plain)
    piece_b = 16 * GW * 64 + 1024 + 256
    xs = form != 2 and ystage + 8 * staged + 8 * piece_b <= 160 * 1024
    if form == 1 and not xs:
        return None
    return dict(MI=MI, GW=GW, wk=wk, wt=wt, passes=passes, blocks=blocks, tiles_base=tb, tiles_rem=tr, tiles_max=tiles_max,
                RD=rd_req or 2, G=G, XS=xs)


def decode_word(ww, z):
    """B fragment of one packed word: 8 halfs in the register order (n0, n4, n1, n5, n2, n6, n3, n7), the exact integers w - z"""
    nib = [(ww >> (4 * i)) & 15 for i in range(8)]
    order = [0, 4, 1, 5, 2, 6, 3, 7]
    return np.array([nib[i] - z for i in order], dtype=np.float16)


def run(x, qweight, qzeros, scales, form=0, rd_req=0, blocks_cap=None):
    M, K = x.shape
    N, KW = qweight.shape
    ZW = qzeros.shape[1]
    SW = scales.shape[1]
    p = plan(M, K, N, form, rd_req, blocks_cap or 256)
    MI, GW, wk, wt, RD, G, XS = p["MI"], p["GW"], p["wk"], p["wt"], p["RD"], p["G"], p["XS"]
    PIECE_W = 16 * GW * 64
    PIECE_B = PIECE_W + 1024 + 256
    CPR, RPI = 4 * GW, 64 // (4 * GW)
    rowbytes = KW * 4
    qwb = qweight.astype(np.uint32).view(np.uint8).reshape(-1)
    qzb = qzeros.astype(np.uint32).view(np.uint8).reshape(-1)
    scb = scales.astype(np.float16).view(np.uint8).reshape(-1)
    xh = x.astype(np.float16)
    y = np.full((M, N), np.nan, dtype=np.float32)
    lanes = np.arange(64)
    n_l, kq_l = lanes & 15, lanes >> 4
    for block in range(p["blocks"]):
        ystage = {}
        ring = {w: [bytearray(PIECE_B) for _ in range(RD)] for w in range(8)}
        info = {}
        for wave in range(8):
            wki, twi = wave % wk, wave // wk
            owner = twi * p["blocks"] + block
            t0 = owner * p["tiles_base"] + min(owner, p["tiles_rem"])
            ntile = p["tiles_base"] + (1 if owner < p["tiles_rem"] else 0)
            info[wave] = (wki, twi, t0, ntile)

        def request(wave, u):
            wki, twi, t0, ntile = info[wave]
            nunit = ntile * p["passes"]
            live = u < nunit
            uu = u if live else 0
            nt1 = max(ntile, 1)
            ps, tl = uu // nt1, uu % nt1
            g0 = (ps * wk + wki) * GW
            row0 = (t0 + tl) * 16 if live else 0
            slot = ring[wave][u % RD]
            for i in range(GW):
                for lane in range(64):
                    r = i * RPI + lane // CPR
                    pos = lane % CPR
                    c = (pos & 16) | ((pos ^ r) & 15)
                    byte = min(g0 * 64 + 16 * c, rowbytes - 16)
                    voff = (min(row0 + r, N - 1) * rowbytes + byte) if live else 0
                    slot[1024 * i + 16 * lane:1024 * i + 16 * lane + 16] = qwb[voff:voff + 16].tobytes()
            for lane in range(64):
                n = lane & 15
                r = min(row0 + n, N - 1)
                sbyte = min((2 * g0) & ~15, SW * 2 - 16)
                vs = r * SW * 2 + sbyte
                slot[PIECE_W + 16 * lane:PIECE_W + 16 * lane + 16] = scb[vs:vs + 16].tobytes()
                zword = min(g0 >> 3, ZW - 1)
                vz = (r * ZW + zword) * 4
                slot[PIECE_W + 1024 + 4 * lane:PIECE_W + 1024 + 4 * lane + 4] = qzb[vz:vz + 4].tobytes()

        def afrag(wave, ps):
            """afr[mi][u][c] per lane: 8 halfs in the permuted order, zero where invalid"""
            wki = info[wave][0]
            g0 = (ps * wk + wki) * GW
            out = np.zeros((MI, GW, 4, 64, 8), dtype=np.float16)
            if XS:
                # chunks of eight batch rows (rows 16 mi + 8 half ..) through an 8-row staging area: lane l of local row m - r0 fetches
                # chunk (l & 48) | ((l & 15) ^ f(m)) into chunk slot l; half 0: every lane takes what it reads, half 1: lanes 8-15 only
                xb = xh.view(np.uint8).reshape(M, K * 2)
                raw = np.zeros((MI, GW, 4, 64, 8), dtype=np.float16)
                for mi in range(MI):
                    for half in range(2):
                        r0 = 16 * mi + 8 * half
                        if r0 >= M:
                            continue
                        r1 = min(r0 + 8, M)
                        stage = np.zeros((8, 1024), dtype=np.uint8)
                        for m in range(r0, r1):
                            for lane in range(64):
                                j = (lane & 48) | ((lane & 15) ^ xs_f(m))
                                byte = min(256 * g0 + 16 * j, K * 2 - 16)
                                stage[m - r0, 16 * lane:16 * lane + 16] = xb[m, byte:byte + 16]
                        for lane in range(64):
                            n, kq = lane & 15, lane >> 4
                            if half == 1 and n < 8:
                                continue
                            m = min(max(16 * mi + n, r0), r1 - 1)
                            for u in range(GW):
                                for c in range(4):
                                    pos = 16 * u + ((4 * kq + c) ^ xs_f(m))
                                    raw[mi, u, c, lane] = stage[m - r0, 16 * pos:16 * pos + 16].view(np.float16)
            for mi in range(MI):
                for u in range(GW):
                    for c in range(4):
                        for lane in range(64):
                            n, kq = lane & 15, lane >> 4
                            m = min(16 * mi + n, M - 1)
                            if XS:
                                d = raw[mi, u, c, lane]
                            else:
                                kk = min(128 * (g0 + u) + 32 * kq + 8 * c, K - 8)
                                d = xh[m, kk:kk + 8]
                            valid = (16 * mi + n < M) and (g0 + u < G)
                            if valid:
                                out[mi, u, c, lane] = d[[0, 4, 1, 5, 2, 6, 3, 7]]
            return out

        units = {w: 0 for w in range(8)}
        for wave in range(8):
            for d in range(RD):
                request(wave, d)
        for ps in range(p["passes"]):
            A = {w: afrag(w, ps) for w in range(8)}
            for tl in range(p["tiles_max"]):
                pbuf = {}
                for wave in range(8):
                    wki, twi, t0, ntile = info[wave]
                    live = tl < ntile
                    if not live:
                        continue
                    g0 = (ps * wk + wki) * GW
                    u = units[wave]
                    slot = ring[wave][u % RD]
                    acc = np.zeros((MI, 16, 16), dtype=np.float32)  # D[m][n]
                    for uu in range(GW):
                        bfr = np.zeros((4, 64, 8), dtype=np.float16)
                        sc_lane = np.zeros(64, dtype=np.float32)
                        for lane in range(64):
                            n, kq = lane & 15, lane >> 4
                            zw = int(np.frombuffer(bytes(slot[PIECE_W + 1024 + 4 * lane:PIECE_W + 1024 + 4 * lane + 4]), dtype=np.uint32)[0])
                            sq = np.frombuffer(bytes(slot[PIECE_W + 16 * lane:PIECE_W + 16 * lane + 16]), dtype=np.float16)
                            q = 4 * uu + kq
                            off = n * 256 + (((q ^ n) & 15) * 16)
                            wq = np.frombuffer(bytes(slot[off:off + 16]), dtype=np.uint32)
                            gi = (g0 & 7) + uu
                            z = (zw >> (4 * gi)) & 15
                            sc_lane[lane] = np.float32(sq[(4 if (g0 & 4) else 0) + uu])
                            for c in range(4):
                                bfr[c, lane] = decode_word(int(wq[c]), z)
                        gacc = np.zeros((MI, 16, 16), dtype=np.float32)
                        for c in range(4):
                            # v_mfma_f32_16x16x32_f16: D[i][j] += sum_{kq, e} A[lane (i, kq)][e] * B[lane (j, kq)][e]
                            Bm = bfr[c].astype(np.float32).reshape(4, 16, 8)  # [kq][j][e]
                            for mi in range(MI):
                                Am = A[wave][mi, uu, c].astype(np.float32).reshape(4, 16, 8)  # [kq][i][e]
                                gacc[mi] += np.einsum("kie,kje->ij", Am, Bm)
                        # lane (n, kq) scales ITS accumulator elements D[4 kq + r][n] with the scale of row n (the same in all four kq)
                        sc_n = sc_lane[:16]
                        assert all(np.array_equal(sc_lane[16 * k:16 * k + 16], sc_n) for k in range(4))
                        acc += gacc * sc_n[None, None, :]
                    request(wave, u + RD)
                    units[wave] = u + 1
                    pbuf[wave] = acc
                for twi in range(wt):
                    ws = [twi * wk + j for j in range(wk)]
                    if ws[0] not in pbuf:
                        continue
                    s = pbuf[ws[0]].copy()
                    for w in ws[1:]:
                        s += pbuf[w]
                    key = (twi, tl)
                    ystage[key] = ystage[key] + s if ps > 0 else s
        for wave in range(8):
            wki, twi, t0, ntile = info[wave]
            for tl in range(wki, ntile, wk):
                row0 = (t0 + tl) * 16
                t = ystage[(twi, tl)]
                for mi in range(MI):
                    for ml in range(16):
                        for nn in range(16):
                            m = 16 * mi + ml
                            if m < M and row0 + nn < N:
                                assert np.isnan(y[m, row0 + nn]), "an output written twice"
                                y[m, row0 + nn] = t[mi, ml, nn]
    assert not np.isnan(y).any(), "an output never written"
    return y.astype(np.float16)


def reference(x, qweight, qzeros, scales):
    """x @ dequant^T with the reference's dequantised fp16 weights ((w - z) * s, awq/utils/packing_utils.py:98-100)"""
    N, KW = qweight.shape
    K = KW * 8
    q = qweight.astype(np.uint32)
    w = np.stack([(q >> (4 * i)) & 15 for i in range(8)], axis=-1).reshape(N, K).astype(np.int32)
    G = K // 128
    qz = qzeros.astype(np.uint32)
    z = np.stack([(qz >> (4 * i)) & 15 for i in range(8)], axis=-1).reshape(N, -1)[:, :G].astype(np.int32)
    s = scales[:, :G].astype(np.float16)
    wd = ((w - np.repeat(z, 128, axis=1)).astype(np.float16) * np.repeat(s, 128, axis=1)).astype(np.float16)
    return x.astype(np.float32) @ wd.astype(np.float32).T


def random_case(M, K, N, seed=0):
    rng = np.random.default_rng(seed)
    G = K // 128
    ZW = -(-G // 8)
    qweight = rng.integers(0, 2**32, size=(N, K // 8), dtype=np.uint64).astype(np.uint32)
    qzeros = rng.integers(0, 2**32, size=(N, ZW), dtype=np.uint64).astype(np.uint32)
    scales = np.zeros((N, ZW * 8), dtype=np.float16)
    scales[:, :G] = (rng.random((N, G)) * 0.02 + 0.005).astype(np.float16)
    x = (rng.standard_normal((M, K)) * 0.5).astype(np.float16)
    return x, qweight, qzeros, scales


if __name__ == "__main__":
    import sys

    cases = [(5, 512, 40, 0, 0), (8, 1024, 72, 1, 0), (16, 1280, 33, 2, 0), (20, 768, 48, 0, 0), (32, 1152, 100, 0, 0), (7, 2432, 24, 2, 0),
             (9, 4224, 20, 1, 0), (6, 1024, 200, 0, 1), (17, 4224, 88, 0, 2), (12, 8320, 56, 1, 1)]
    bad = 0
    for M, K, N, gw, cap in cases:
        x, qw, qz, sc = random_case(M, K, N, seed=M + K + N)
        y = run(x, qw, qz, sc, form=gw, blocks_cap=cap).astype(np.float32)
        ref = reference(x, qw, qz, sc)
        err = np.abs(y - ref).max()
        tol = 2e-3 * np.abs(ref).max() + 1e-3
        ok = err <= tol
        bad += not ok
        print(f"M={M} K={K} N={N} form={gw} plan={plan(M, K, N, gw, 0, cap or 256)} max err {err:.4g} (max |ref| {np.abs(ref).max():.3g}) {'ok' if ok else 'MISMATCH'}")
    sys.exit(1 if bad else 0)
