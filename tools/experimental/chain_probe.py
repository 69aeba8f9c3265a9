#!/usr/bin/env python3
"""Bring-up / timing probe of the persistent decode chain (csrc/gemv_chain.hip):
the Llama-2-7B-shape Linears as ONE dependent chain (qkv -> o (reads q) -> gate|up -> down (silu * up) -> next
qkv ...), checked link by link against the one-launch-per-Linear kernels, then timed against them."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def build_links(layers, dev, M, hidden=4096, inter=11008, seed=1, gated=False):
    from autoawq_amd.chain import ChainLink

    gen = torch.Generator(device=dev).manual_seed(seed)
    x0 = torch.randn((M, hidden), device=dev, generator=gen).half()
    links, meta = [], []
    for li in range(layers):
        for name, K, N, col0, gated in (("qkv", hidden, 3 * hidden, 0, False), ("o", hidden, hidden, 0, False),
                                        ("gate_up", hidden, 2 * inter, 0, False), ("down", inter, hidden, 0, gated)):
            qw, qz, sc = bench.rand_packed(K, N, 128, dev, gen)
            y = torch.zeros((M, N), dtype=torch.float16, device=dev)
            links.append(ChainLink(qw, sc, qz, x=x0 if not links else None, x_col0=col0, gated=gated, y=y))
            meta.append((name, K, N, gated))
    return x0, links, meta


def calibrate(x0, links, meta, ops):
    """Rescale every link's scales so that its output has rms ~1 (a 128-link chain of random matrices
    would overflow fp16 otherwise); returns the reference outputs of the sequential kernels."""
    x = x0
    refs = []
    for ln, (name, K, N, gated) in zip(links, meta):
        xin = ops.silu_and_mul(x[:, : 2 * K].contiguous()) if gated else x[:, :K].contiguous()
        y = ops.gemm_forward(xin, ln.qweight, ln.scales, ln.qzeros)
        rms = float(y.float().pow(2).mean().sqrt())
        if rms > 0:
            ln.scales.mul_(1.0 / rms)
            y = ops.gemm_forward(xin, ln.qweight, ln.scales, ln.qzeros)
        refs.append(y)
        x = y
    return refs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--M", type=int, default=1)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=11008)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--gated", action="store_true", help="down = silu(gate) * up (quadratic: long random chains blow up)")
    ap.add_argument("--trace-links", type=int, default=12)
    ap.add_argument("--trace", action="store_true", help="phase timeline per link (first launch after warm-up)")
    a = ap.parse_args()
    from autoawq_amd import _lib, ops
    from autoawq_amd.chain import DecodeChain

    _lib.lib()
    dev = torch.device("cuda", 0)
    x0, links, meta = build_links(a.layers, dev, a.M, a.hidden, a.inter, gated=a.gated)
    refs = calibrate(x0, links, meta, ops)
    chain = DecodeChain(links, M=a.M, trace=a.trace)
    print(f"chain: {len(links)} links, grid {_lib.lib().awq_chain_grid_blocks()} blocks, workspace {chain.workspace.numel() / 1e6:.1f} MB",
          flush=True)
    chain()
    torch.cuda.synchronize()
    st = chain.status()
    print(f"status after first launch: {st:#x}", flush=True)
    # every link against the one-launch kernel fed with the CHAIN's own previous output (a 128-link chain of
    # random matrices amplifies rounding differences, so only link-by-link comparisons mean anything)
    worst, nbad, x = 0.0, 0, x0
    for i, (ln, (name, K, N, gated)) in enumerate(zip(links, meta)):
        ref = ops.gemm_forward(x if gated else x[:, :K].contiguous(), ln.qweight, ln.scales, ln.qzeros,
                               flags=ops.X_GATED_SILU if gated else 0)
        d = (ln.y.float() - ref.float()).abs()
        tol = 2.0 ** -10 * ref.float().abs() + 2e-4 * float(ref.float().pow(2).mean().sqrt()) + 1e-6
        bad = int((~(d <= tol)).sum())
        worst = max(worst, float((d / tol).nan_to_num(1e9).max()))
        if bad and nbad < 6:
            idx = torch.nonzero(~(d <= tol))[:4].tolist()
            print(f"  link {i} ({name}): {bad}/{d.numel()} outside 1 ulp + 2e-4 rms, max err {float(d.max()):.4g}, first {idx}; "
                  f"rms {float(ref.float().pow(2).mean().sqrt()):.3g}", flush=True)
        nbad += 1 if bad else 0
        x = ln.y
    print(f"links off: {nbad}/{len(links)}  worst err/tol {worst:.3f}  final rms {float(x.float().pow(2).mean().sqrt()):.3g}", flush=True)
    first = [ln.y.clone() for ln in links]
    same = True
    for rep in range(10):
        chain()
        torch.cuda.synchronize()
        for i, (a_, ln) in enumerate(zip(first, links)):
            if not torch.equal(a_, ln.y):
                d = (a_.float() - ln.y.float()).abs()
                if same:
                    print(f"  replay {rep}: first differing link {i} ({meta[i][0]}): {int((d > 0).sum())} elements, max diff "
                          f"{float(d.max()):.4g}, at {torch.nonzero(d > 0)[:6].tolist()}", flush=True)
                same = False
                break
    print(f"10 replays bitwise identical: {same}; status {chain.status():#x}", flush=True)
    if a.trace:
        chain.trace.zero_()
        chain()
        torch.cuda.synchronize()
        tr = chain.trace.cpu().numpy().astype("float64")  # [links, G, 4 waves, 4 slots], 100 MHz ticks
        import numpy as np

        t0 = tr[tr > 0].min()
        us = lambda v: (v - t0) / 100.0

        def stat(v):
            v = v[v > 0]
            return "      -            " if v.size == 0 else f"{us(v.min()):7.2f} {us(np.median(v)):7.2f} {us(v.max()):7.2f}"

        print("us since first stamp: min / median / max over waves (wave 0 loader, 1-6 compute, 7-9 poll)")
        print("link | loader starts link   || compute: start       | weights + x ready    | mfma done            | folded / slab stored || poll: start          | slabs probed         | gather loads back    | x staged")
        for l in range(min(tr.shape[0], a.trace_links)):
            ld, c, pl = tr[l][:, 0], tr[l][:, 1:7], tr[l][:, 7:]
            print(f"{l:4d} | " + stat(ld[..., 0]) + " || " + " | ".join(stat(c[..., k]) for k in range(4)) + " || " +
                  " | ".join(stat(pl[..., k]) for k in (0, 1, 3, 2)))
        return
    if a.no_time or st:
        return
    nbytes = sum(bench.algorithmic_bytes(K, N, a.M, 128) for _, K, N, _ in meta)

    def timed(fn, label):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                fn()
            for _ in range(10):
                g.replay()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(a.steps):
                g.replay()
            e1.record(s)
            e1.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        print(f"{label}: {ms:.4f} ms/token  {1000 / ms:.1f} tok/s  {nbytes / ms / 1e6:.0f} GB/s = {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s",
              flush=True)
        return ms

    def sequential():
        x = x0
        for ln, (name, K, N, gated) in zip(links, meta):
            x = ops.gemm_forward(x if gated else x[:, :K], ln.qweight, ln.scales, ln.qzeros,
                                 flags=ops.X_GATED_SILU if gated else 0)

    from autoawq_amd.chain import ChainLink

    lean = [ChainLink(ln.qweight, ln.scales, ln.qzeros, x=ln.x, x_col0=ln.x_col0, gated=ln.gated,
                      y=ln.y if i == len(links) - 1 else None) for i, ln in enumerate(links)]
    chain2 = DecodeChain(lean, M=a.M)
    last = links[-1].y.clone()
    chain2()
    torch.cuda.synchronize()
    print(f"lean chain (only the last y materialised): final output identical {torch.equal(last, links[-1].y)}, status {chain2.status():#x}")
    timed(sequential, "one launch per Linear (dependent)")
    timed(chain.forward, "persistent chain, every y      ")
    timed(chain2.forward, "persistent chain, last y only  ")
    print(f"status at end: {chain.status():#x}", flush=True)




def debug_small():
    """dump the slabs of link 0 of a tiny chain and compare their slice sums with the reference"""
    import numpy as np
    from autoawq_amd import _lib, ops
    from autoawq_amd.chain import DecodeChain

    dev = torch.device("cuda", 0)
    x0, links, meta = build_links(1, dev, 1, 256, 384)
    refs = calibrate(x0, links, meta, ops)
    chain = DecodeChain(links[:1], M=1)
    chain()
    torch.cuda.synchronize()
    ws = chain.workspace.cpu().numpy()
    K, N = 256, 768
    tiles = 3
    # nsets = 2 -> rpb = 128 -> S = 2
    slab = ws[4096:4096 + 2 * tiles * 2048].view(np.uint32).reshape(2, tiles, 64, 8)
    vals = slab[..., 0::2].view(np.float32)      # [S, tiles, 64 quads, 4]
    tags = slab[..., 1::2]
    print("tags unique:", np.unique(tags))
    tot = vals.sum(0).reshape(tiles * 256)
    ref = refs[0][0].float().cpu().numpy()
    print("slab sums[:8]", tot[:8])
    print("ref      [:8]", ref[:8])
    print("y        [:8]", links[0].y[0, :8].float().cpu().numpy())
    print("max |slabsum - ref|", np.abs(tot[:N] - ref).max())


if __name__ == "__main__" and "--debug" in sys.argv:
    debug_small()
elif __name__ == "__main__":
    main()
