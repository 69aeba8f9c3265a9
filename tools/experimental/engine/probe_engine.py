#!/usr/bin/env python3
"""GPU probe of the experimental persistent decode engine (gemv_engine.hip / engine.py in this directory): every op of a chain checked against
the fp32 product of the bit-exact dequantised weights ON THE ENGINE'S OWN INPUT of that op, then the 7B dependent chain timed
(one hipGraph replay per token, distinct weights per layer) beside one launch per Linear, per configuration.

    python tools/experimental/engine/probe_engine.py [--layers 32] [--check-only] [--time-only] [--flags inflight,consumers,slots ...]
"""
import argparse
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from autoawq_amd import ops  # noqa: E402
from engine import DecodeChain, engine_flags  # noqa: E402
from autoawq_amd.utils.packing import calculate_zeros_width  # noqa: E402

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(0)
lim = 0x7FFFFFFF
G = 128


def rand_nk(K, N, scale=1.0):
    zw = calculate_zeros_width(K, G)
    qw = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, device=dev, generator=gen)
    sc = ((torch.rand((N, zw * 8), device=dev, generator=gen) * 0.02 + 0.005) * scale).half()
    return qw, sc, qz


def normalise(chain_shapes, x0):
    """random Linears with unit gain per link (fp16 would overflow after a few links otherwise)"""
    lins, x = [], x0
    for K, N in chain_shapes:
        qw, sc, qz = rand_nk(K, N)
        y = ops.gemv_forward(x[:, :K].contiguous(), qw, sc, qz, G)
        rms = float(y.float().pow(2).mean().sqrt())
        sc = (sc.float() / max(rms, 1e-6)).half()
        x = ops.gemv_forward(x[:, :K].contiguous(), qw, sc, qz, G)
        lins.append((qw, sc, qz))
    return lins


def check_chain(name, shapes, flags=0):
    x0 = torch.randn((1, shapes[0][0]), device=dev, generator=gen).half()
    lins = normalise(shapes, x0)
    ch = DecodeChain(lins)
    t0 = time.time()
    ch.forward(x0.view(-1), flags)
    torch.cuda.synchronize()
    st = ch.status()
    bad = 0
    worst = 0.0
    xin = x0.view(-1)
    for i, ((K, N), (qw, sc, qz)) in enumerate(zip(shapes, lins)):
        Wt = ops.dequantize_weights_gemv(qw, sc, qz, G).float()  # [N, K]
        ref = Wt @ xin[:K].float()
        got = ch.outputs[i].float()
        tol = 2e-3 * ref.abs() + 2e-3 * ref.pow(2).mean().sqrt() + 1e-3
        err = ((got - ref).abs() / tol).max().item()
        worst = max(worst, err)
        if not (err <= 1.0):
            bad += 1
            if bad <= 3:
                d = (got - ref).abs()
                j = int(d.argmax())
                print(f"   op {i} ({K}->{N}): worst err/tol {err:.3g} at row {j}: got {got[j].item():.5g} ref {ref[j].item():.5g}; "
                      f"rows off {(d > tol).sum().item()} of {N}; nan {torch.isnan(got).sum().item()}")
        xin = ch.outputs[i]
    # replays: same result, epoch advances
    y1 = ch.outputs[-1].clone()
    for _ in range(5):
        ch.forward(x0.view(-1), flags)
    torch.cuda.synchronize()
    same = torch.equal(y1, ch.outputs[-1])
    print(f"{name}: {len(shapes)} ops, first launch {time.time() - t0:.2f} s, status {st[0]}/{st[1]}/{st[2]:#x} -> {ch.status()}, "
          f"ops off {bad}, worst err/tol {worst:.3f}, 5 replays bitwise equal {same}", flush=True)
    return bad == 0 and st[2] == 0 and same


def graph_us(fn, reps=20, warm=3):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(warm):
            g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def alg_bytes(K, N):
    return K * N // 2 + (K // G) * (N // 8) * 4 + (K // G) * N * 2 + K * 2 + N * 2


def time_7b(layers, flag_sets):
    layer = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]
    shapes = layer * layers
    x0 = torch.randn((1, 4096), device=dev, generator=gen).half()
    lins = normalise(shapes, x0)
    by = sum(alg_bytes(K, N) for K, N in shapes)

    def sequential():
        t = x0
        for (K, N), (qw, sc, qz) in zip(shapes, lins):
            t = ops.gemv_forward(t[:, :K], qw, sc, qz, G)
        return t

    y_seq = sequential()
    us = graph_us(sequential)
    print(f"7B chain, {layers} layers, {by / 1e6:.1f} MB: one launch per Linear {us:.1f} us/token = {by / us / 1e6:.3f} TB/s = "
          f"{by / us / 8e6:.3f} of 8 TB/s ({us / layers:.2f} us/layer)", flush=True)
    ch = DecodeChain(lins, keep_outputs=False)
    for fl in flag_sets:
        f = engine_flags(*fl)
        y = ch.forward(x0.view(-1), f)
        torch.cuda.synchronize()
        ch.check()
        d = (y.float() - y_seq.view(-1).float()).abs().max().item()
        us_e = graph_us(lambda: ch.forward(x0.view(-1), f))
        ch.check()
        print(f"   engine inflight/consumers/slots {fl}: {us_e:.1f} us/token = {by / us_e / 1e6:.3f} TB/s = {by / us_e / 8e6:.3f} of 8 TB/s "
              f"({us_e / layers:.2f} us/layer, x{us / us_e:.3f} vs launches); max |y - y_launches| {d:.3g}", flush=True)


def trace_7b(layers, fl):
    """where an op's time goes, per CU: medians over the CUs and over the layers 2 .. layers - 2, by op of the layer"""
    import numpy as np
    layer = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]
    shapes = layer * layers
    x0 = torch.randn((1, 4096), device=dev, generator=gen).half()
    lins = normalise(shapes, x0)
    ch = DecodeChain(lins, keep_outputs=False)
    f = engine_flags(*fl)
    for _ in range(3):
        ch.forward(x0.view(-1), f)
    torch.cuda.synchronize()
    ch.set_trace(True)
    ch.forward(x0.view(-1), f)
    torch.cuda.synchronize()
    ch.check()
    t = ch.trace.cpu().numpy().astype(np.float64) / 100.0  # us
    ch.set_trace(False)
    t0 = t[:, 0, 0].min()
    print(f"trace, flags {fl}: token took {t[:, -1, 6].max() - t0:.1f} us (first gather start -> last publish)")
    names = ["qkv", "o", "gate|up", "down"]
    print("op       | gather (poll+stage) | x -> regs | own rows      | bookkeep | wait others | publish || op period | loader lead at op start | "
          "spread of op end over CUs")
    for i, nm in enumerate(names):
        ops_i = [l * 4 + i for l in range(2, layers - 1)]
        g = np.median([t[:, o, 1] - t[:, o, 0] for o in ops_i])
        xl = np.median([t[:, o, 2] - t[:, o, 1] for o in ops_i])
        w1 = np.median([t[:, o, 3] - t[:, o, 2] for o in ops_i])
        cr = np.median([t[:, o, 4] - t[:, o, 3] for o in ops_i])
        wo = np.median([t[:, o, 5] - t[:, o, 4] for o in ops_i])
        pb = np.median([t[:, o, 6] - t[:, o, 5] for o in ops_i])
        per = np.median([t[:, o, 6] - t[:, o - 1, 6] for o in ops_i])
        lead = np.median([t[:, o, 0] - t[:, o, 7] for o in ops_i])
        spread = np.median([t[:, o, 6].max() - t[:, o, 6].min() for o in ops_i])
        print(f"{nm:8s} | {g:19.2f} | {xl:9.2f} | {w1:13.2f} | {cr:8.2f} | {wo:11.2f} | {pb:7.2f} || {per:9.2f} | {lead:23.2f} | {spread:8.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--flags", nargs="*", default=["0,0,0", "3,6,0", "1,6,0", "2,6,0,0,1", "2,3,0", "2,6,0,1", "2,6,0,2", "2,6,0,3", "2,6,0,7"])
    a = ap.parse_args()
    ok = True
    if not a.time_only:
        ok &= check_chain("one op 4096->512", [(4096, 512)])
        ok &= check_chain("one op 4096->4096", [(4096, 4096)])
        ok &= check_chain("two ops", [(4096, 4096), (4096, 1024)])
        ok &= check_chain("ragged", [(1024, 2050), (2048, 640), (512, 11008), (11008, 258), (256, 4096)])
        ok &= check_chain("7B layer", [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)])
        ok &= check_chain("7B x 4 layers, 3 consumers", [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)] * 4, engine_flags(2, 3, 0))
        ok &= check_chain("70B shards", [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)])
        ok &= check_chain("K 8192 / 6144 / 10240", [(8192, 6144), (6144, 10240), (10240, 512)])
        print("CHECK", "PASS" if ok else "FAIL", flush=True)
    if not a.check_only and ok:
        time_7b(a.layers, [tuple(int(v) for v in f.split(",")) for f in a.flags])
        if a.trace:
            trace_7b(a.layers, (2, 6, 0))
            trace_7b(a.layers, (2, 3, 0))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
