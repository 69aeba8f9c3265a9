#!/usr/bin/env python3
"""What do the decoder-block prologue / epilogue variants of the row-streaming kernel (awq_gemv_forward_ex: norm, residual,
silu pairs) cost over the plain kernel?  32 distinct weight sets per shape (cold weights), one hipGraph per variant."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from autoawq_amd import _lib, ops

    _lib.lib()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    st = torch.cuda.Stream(device=dev)
    H, I = 4096, 11008
    nw = torch.ones(H, dtype=torch.float16, device=dev)
    for name, K, N, variants in [("qkv", H, 3 * H, ["plain", "norm"]), ("o", H, H, ["plain", "res"]),
                                 ("gate_up", H, 2 * I, ["plain", "norm", "pairs", "norm+pairs"]), ("down", I, H, ["plain", "res"])]:
        nsets = 32
        sets = [bench.rand_packed_nk(K, N, bench.GROUP, dev, gen) for _ in range(nsets)]
        x = torch.randn((1, K), device=dev, generator=gen).half()
        res = torch.randn((1, N), device=dev, generator=gen).half()
        out = []
        for v in variants:
            kw = {}
            if "norm" in v:
                kw.update(norm_weight=nw[:K] if K <= H else torch.ones(K, dtype=torch.float16, device=dev), norm_eps=1e-5)
            if "res" in v:
                kw.update(add_residual=res)
            if "pairs" in v:
                kw.update(silu_pairs=True)

            def fn():
                for qw, qz, sc in sets:
                    if v == "plain":
                        ops.gemv_forward(x, qw, sc, qz, bench.GROUP)
                    else:
                        ops.gemv_forward_ex(x, qw, sc, qz, bench.GROUP, **kw)

            us = bench.graph_time(fn, st, reps=10, min_seconds=0.2) / nsets
            out.append(f"{v} {us:6.2f} us")
        print(f"{name:8s} {K:5d} -> {N:5d}: " + "   ".join(out), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
