#!/usr/bin/env python3
"""Chained launches (AwqGemvEx.chain_*, ops.LaunchChain) on the headline's 128 Linears as ONE dependent chain:
correctness against the same chain run as ordinary launches on one stream (bit for bit), then hipGraph timing of
 (a) ordinary dependent launches, one stream   (b) chained launches, two graph branches   (c) the r03 headline (independent inputs)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CSRC = os.path.join(ROOT, "autoawq_amd", "csrc")
VARIANTS = [(0, "the product: sc1 (write-through) rows, sc1 activation loads, no fences"),
            (1, "consumer: ONE agent-scope acquire fence + plain loads instead of sc1 loads"),
            (2, "no polling at all (WRONG results: upper bound of what overlapping the launches can give)")]


def so(bits):
    return os.path.join(ROOT, "tools", "bin", f"libawq_hip_chainx{bits}.so")


def build():
    import subprocess

    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    others = [os.path.join(CSRC, "build", f[:-4] + ".o") for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != "gemv_rows.hip"]
    for bits, _ in VARIANTS:
        if bits == 0:
            continue
        obj = os.path.join(ROOT, "tools", "bin", f"gemv_rows_chainx{bits}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-fno-slp-vectorize", "-Wno-inline-asm", f"-DAWQ_CHAIN_DBG={bits}", "-DAWQ_BUILDING_LIB",
                               "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, "gemv_rows.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so(bits), obj] + others)
        os.remove(obj)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--variant", type=int, default=-1)
    a = ap.parse_args()
    if a.build_only:
        return build()
    if a.variant < 0:  # one child process per library variant
        import subprocess

        for bits, what in VARIANTS:
            print(f"==== AWQ_CHAIN_DBG={bits}: {what}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--variant", str(bits), "--layers", str(a.layers), "--reps", str(a.reps)], timeout=400)
        return
    from autoawq_amd import _lib

    if a.variant:
        _lib.LIB_PATH = so(a.variant)
    from autoawq_amd import ops

    _lib.lib()
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev, 0, 1, a.layers, layout="gemv")
    lins = [lin for layer in model for lin in layer]
    nbytes = sum(bench.algorithmic_bytes(l["K"], l["N"], 1, bench.GROUP) for l in lins)
    x0 = lins[0]["x"]
    x = x0
    for lin in lins:  # unit gain per link (bench.leg_decode_dependent)
        y = bench.forward_lin(ops, lin, x[:, : lin["K"]].contiguous())
        rms = float(y.float().pow(2).mean().sqrt())
        lin["sc"].mul_(1.0 / max(rms, 1e-6))
        x = bench.forward_lin(ops, lin, x[:, : lin["K"]].contiguous())

    def sequential(keep=None):
        t = x0
        for l in lins:
            t = ops.gemv_forward(t[:, : l["K"]], l["qw"], l["sc"], l["qz"], bench.GROUP)
            if keep is not None:
                keep.append(t)
        return t

    chain = ops.LaunchChain(dev, len(lins))

    def chained(keep=None):
        t = x0
        with chain:
            for l in lins:
                t = chain.gemv(t[:, : l["K"]], l["qw"], l["sc"], l["qz"], bench.GROUP)
                if keep is not None:
                    keep.append(t)
        return t

    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        ref, got = [], []
        sequential(ref)
        st.synchronize()
        t0 = time.perf_counter()
        chained(got)
        torch.cuda.synchronize()
        print(f"eager chained step: {time.perf_counter() - t0:.3f} s", flush=True)
        chain.check()
        bad = [i for i, (r, g) in enumerate(zip(ref, got)) if not torch.equal(r, g)]
        print(f"eager: links differing from the ordinary launches: {len(bad)} of {len(ref)} {bad[:8]}", flush=True)
        if bad:
            i = bad[0]
            d = (ref[i].float() - got[i].float()).abs()
            print(f"  first bad link {i} ({lins[i]['name']}): max abs diff {float(d.max())}, elements differing {int((d > 0).sum())} of {d.numel()}, nan {int(torch.isnan(got[i]).sum())}")
        for _ in range(3):  # replay counts advance; results must stay the same
            out = chained()
        torch.cuda.synchronize()
        print("eager x3 more: final output equal:", torch.equal(out, ref[-1]), flush=True)

    us_seq = bench.graph_time(sequential, st, reps=a.reps, min_seconds=0.3)
    print(f"(a) ordinary dependent launches: {us_seq:.1f} us/token  {1e6 / us_seq:.1f} tok/s  {nbytes / us_seq / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    # chained under capture (graph_time captures fn on `st`: LaunchChain forks its side stream off it)
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            out_g = chained()
        for _ in range(3):
            g.replay()
        st.synchronize()
        chain.check()
        print("graph: final output equal to the ordinary launches:", torch.equal(out_g, ref[-1]), flush=True)
        # a stale read of the PREVIOUS replay's activations would go unnoticed with a fixed input: new inputs per replay
        gen = torch.Generator(device=dev).manual_seed(77)
        keep0 = x0.clone()
        stale = 0
        for it in range(12):
            x0.copy_(torch.randn(x0.shape, device=dev, generator=gen).half())
            want = sequential()
            g.replay()
            st.synchronize()
            stale += 0 if torch.equal(out_g, want) else 1
        x0.copy_(keep0)
        print(f"graph, 12 replays with a new input each: {stale} differ from the ordinary launches", flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot, n = 0.0, 0
        while tot < 300.0:
            e0.record(st)
            for _ in range(a.reps):
                g.replay()
            e1.record(st)
            e1.synchronize()
            tot += e0.elapsed_time(e1)
            n += a.reps
        us_ch = tot * 1e3 / n
        chain.check()
        print("graph after", n + 3, "replays: final output equal:", torch.equal(out_g, ref[-1]), flush=True)
    print(f"(b) chained launches, two branches: {us_ch:.1f} us/token  {1e6 / us_ch:.1f} tok/s  {nbytes / us_ch / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    outs = [None] * len(lins)
    us_ind = bench.graph_time(lambda: bench.run_step(model, outs, ops, None), st, reps=a.reps, min_seconds=0.3)
    print(f"(c) independent inputs, one stream (r03 headline): {us_ind:.1f} us/token  {1e6 / us_ind:.1f} tok/s  {nbytes / us_ind / 1e3 / 8000:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
