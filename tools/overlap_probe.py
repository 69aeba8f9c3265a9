#!/usr/bin/env python3
"""Upper bound of what overlapping consecutive decode GEMVs can buy: bench.py's 128 launches per token,
issued round-robin on 1 / 2 / 3 streams inside ONE hipGraph (fork / join by events), with NO data
dependency between launches (every Linear has its own resident input).  Serial = bench.py's number.
If two streams do not beat one by a wide margin here, an in-kernel dependency (activation tags) cannot
either."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(nstreams, model, ops, steps=100, warmup=10):
    dev = torch.device("cuda", 0)
    main = torch.cuda.Stream(device=dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]
    streams = [main] + side
    outs = [None] * sum(len(l) for l in model)

    def step():
        ev = torch.cuda.Event()
        ev.record(main)
        for s in side:
            s.wait_event(ev)
        i = 0
        for layer in model:
            for lin in layer:
                with torch.cuda.stream(streams[i % nstreams]):
                    outs[i] = ops.gemm_forward(lin["x"], lin["qw"], lin["sc"], lin["qz"])
                i += 1
        for s in side:
            e = torch.cuda.Event()
            e.record(s)
            main.wait_event(e)

    with torch.cuda.stream(main):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            step()
        for _ in range(warmup):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(steps):
            g.replay()
        e1.record(main)
        e1.synchronize()
    ms = e0.elapsed_time(e1) / steps
    ref = [ops.gemm_forward(lin["x"], lin["qw"], lin["sc"], lin["qz"]) for layer in model for lin in layer]
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(a, b) else 1 for a, b in zip(outs, ref))
    return ms, bad


def main():
    from autoawq_amd import _lib, ops

    _lib.lib()
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev, 0, 1, bench.LAYERS)
    nbytes = sum(bench.algorithmic_bytes(l["K"], l["N"], 1, bench.GROUP) for layer in model for l in layer)
    for n in (1, 2, 3, 2, 1):
        ms, bad = run(n, model, ops)
        print(f"streams={n}: {ms:.4f} ms/token  {1000 / ms:.1f} tok/s  {nbytes / ms / 1e6:.0f} GB/s = "
              f"{nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s   outputs differing from a serial run: {bad}", flush=True)


if __name__ == "__main__":
    main()
