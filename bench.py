#!/usr/bin/env python3
"""bench.py -- decode tok/s @bs=1 + GEMV HBM GB/s on synthetic Llama-2-7B-shape AWQ int4 g128.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 launched by
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" = one decode token through every int4 Linear of the model (the hot path, BASELINE configs[1]):
per layer qkv (fused, 4096->12288), o (4096->4096), gate+up (fused, 4096->22016), down (11008->4096), 32
layers, batch 1, distinct random packed weights per layer (3.37 GB working set >> the 256 MiB Infinity Cache),
captured in ONE hipGraph and replayed.  Inputs are resident in HBM before the timed region.  `value` is that
leg and nothing else.  The Linears are in the reference's WQLinear_GEMV checkpoint format (`--layout gemv`, the
default since round 3: the format the reference itself recommends for batch 1, README.md:96-97; every output row is
K/2 contiguous bytes, so the decode kernel csrc/gemv_rows.hip needs no cross-CU split-K exchange); `config.layout`
names it and `by_layout` times the same step on WQLinear_GEMM and WQLinear_GEMVFast buffers.  N > 1: the same
model tensor-parallel over N GPUs (column-split qkv / gate+up, row-split o / down with one RCCL all-reduce each)
-- strong scaling; `--model 70b` = BASELINE configs[3].

Secondary objects on the same JSON line (N = 1, never `value`; each guarded so that the headline cannot
depend on them): `sustained` (the same graph for >= 1 s), `per_shape` (the four Linear shapes of the headline one by
one, each with its own roofline), `by_layout`, `gemm_bs` (configs[2]'s "bs=8 GEMM for 4096x11008": M = 1 .. 64, cold
weights, GEMM and GEMV layouts), `gemm_prefill` (configs[2]: M = 8 x 2048 = 16384, fused MFMA kernel vs HIP dequant
+ vendor GEMM, which one the module picks), `moe_bs4` (configs[4]), `decode_independent` (the headline's Linears with FIXED inputs: rounds 1-5's headline; since round 6 the headline itself has the TRUE
data dependencies: each consumes the previous one's output), `whole_model` (the fused decoder), and `cpu_baseline`
(the reference's CPU path restated in torch, per shape, M = 1 and 8, on this host's cores).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0    # measured-achievable copy rate (the same guide's chip table)
MFMA_PEAK_TF = 2500.0    # dense fp16 / bf16 MFMA (same file); AMD's 5 PF headline includes 2:1 sparsity
GROUP = 128
MODELS = {  # hidden, intermediate, layers, heads, kv heads
    "7b": dict(hidden=4096, inter=11008, layers=32, heads=32, kv_heads=32, name="Llama-2-7B"),
    "70b": dict(hidden=8192, inter=28672, layers=80, heads=64, kv_heads=8, name="Llama-3-70B"),
}
HIDDEN, INTER, LAYERS = 4096, 11008, 32  # the headline model (kept as names for tools/ that import them)
PMC_FILE = "r06_pmc_fetch_size.txt"      # rocprofv3 --pmc FETCH_SIZE pass of the headline command (tools/prof_r06.sh)
PMC_BS_FILE = "r06_pmc_gemm_bs.txt"      # the same counters for the batched-decode legs (tools/pmc_gemm_bs.py under rocprofv3)
KERNEL_OF_LAYOUT = {"gemv": "awq_gemv_rows_kernel", "gemm": "awq_gemv_mfma_kernel", "gemvfast": "awq_gemv_fast_kernel"}


def kernel_fingerprint():
    """sha1 over the sources libawq_hip.so is built from (csrc/*.hip, csrc/*.h, include/awq_hip.h): what a committed profile
    records, so that a counter file taken from an OLDER build of the kernels is recognised as such (there is no .git on the
    GPU box to ask for the head)."""
    import hashlib

    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "autoawq_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for f in files + [os.path.join(ROOT, "include", "awq_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch():
    """(mean HBM bytes per launch of the headline kernels from the committed FETCH_SIZE pass, note).  The pass is its own
    rocprofv3 run (tools/prof_r05.sh); its file records the kernel-source fingerprint it was taken at, and a file taken from
    other kernel sources than the ones in this tree is NOT reported as this build's traffic."""
    import re

    try:
        txt = open(os.path.join(ROOT, "profiles", PMC_FILE)).read()
        val = float(re.search(r"per launch \(weighted mean\): traffic ([0-9.]+) MB", txt).group(1)) * 1e6
    except (OSError, AttributeError):
        return None, "profiles/" + PMC_FILE + " not found"
    m = re.search(r"kernel source fingerprint: ([0-9a-f]+)", txt)
    if not m or m.group(1) != kernel_fingerprint():
        return None, (f"profiles/{PMC_FILE} was taken at kernel sources {m.group(1) if m else 'unrecorded'}, this tree is "
                      f"{kernel_fingerprint()}: {val / 1e6:.3f} MB per launch there, not reported as this build's traffic")
    return val, "kernel sources of the counter pass == this tree (" + m.group(1) + ")"


def pmc_traffic_gemm_bs():
    """{leg key: HBM bytes per call} of the batched-decode legs from the committed counter pass (profiles/PMC_BS_FILE: lines
    `<layout> M=<m>: traffic <x> MB ...`), or {} when the file is missing / was taken from other kernel sources."""
    import re

    try:
        txt = open(os.path.join(ROOT, "profiles", PMC_BS_FILE)).read()
    except OSError:
        return {}
    m = re.search(r"kernel source fingerprint: ([0-9a-f]+)", txt)
    if not m or m.group(1) != kernel_fingerprint():
        return {}
    return {(a, b): float(c) * 1e6 for a, b, c in re.findall(r"^(\w+) M=(\d+): traffic ([0-9.]+) MB", txt, re.M)}


def algorithmic_bytes(K, N, M, g, bias=False):
    """packed weights + zeros + scales read once, x read once, y written once."""
    return K * N // 2 + (K // g) * (N // 8) * 4 + (K // g) * N * 2 + M * K * 2 + M * N * 2 + (N * 2 if bias else 0)


def rand_packed_nk(K, N, g, dev, gen, fast=False):
    """Random buffers in the GEMV (qweight [N, K/8]) or GEMVFast (int16 [N/4, K]) layout."""
    from autoawq_amd.utils.packing import calculate_zeros_width

    lim = 0x7FFFFFFF
    zw = calculate_zeros_width(K, g)
    if fast:
        qw = torch.randint(-32768, 32767, (N // 4, K), dtype=torch.int16, device=dev, generator=gen)
        sc = (torch.rand((zw * 8, N), device=dev, generator=gen) * 0.02 + 0.005).half()
        qz = -(sc.float() * torch.randint(0, 16, (zw * 8, N), device=dev, generator=gen).float()).half()
        return qw, qz, sc
    qw = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((N, zw * 8), device=dev, generator=gen) * 0.02 + 0.005).half()
    return qw, qz, sc


def rand_packed(K, N, g, dev, gen):
    lim = 0x7FFFFFFF
    qw = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-lim - 1, lim, (K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((K // g, N), device=dev, generator=gen) * 0.02 + 0.005).half()
    return qw, qz, sc


def build_model(dev, rank, world, layers, seed=1234, layout="gemm", model="7b"):
    """Per-rank shard shapes (autoawq_amd.tp.llama_layer_bounds): attention split by whole KV heads with the
    query heads that attend to them, the MLP by whole quantisation groups of down's rows (uneven splits allowed:
    Llama-2-7B has 86 groups)."""
    from autoawq_amd.tp import llama_layer_bounds

    cfg = MODELS[model]
    H, I = cfg["hidden"], cfg["inter"]
    hd = H // cfg["heads"]
    b = llama_layer_bounds(cfg["heads"], cfg["kv_heads"], hd, I, GROUP, rank, world)
    nq, nkv, ni = b["q"][1] - b["q"][0], b["kv"][1] - b["kv"][0], b["mlp"][1] - b["mlp"][0]
    gen = torch.Generator(device=dev).manual_seed(seed + rank)
    shapes = [("qkv", H, nq + 2 * nkv, False), ("o", nq, H, True), ("gate_up", H, 2 * ni, False), ("down", ni, H, True)]
    net = []
    for _ in range(layers):
        layer = []
        for name, K, N, reduce_after in shapes:
            if layout == "gemm":
                qw, qz, sc = rand_packed(K, N, GROUP, dev, gen)
            else:
                qw, qz, sc = rand_packed_nk(K, N, GROUP, dev, gen, fast=(layout == "gemvfast"))
            x = torch.randn((1, K), device=dev, generator=gen).half()
            layer.append(dict(name=name, K=K, N=N, qw=qw, qz=qz, sc=sc, x=x, reduce=reduce_after and world > 1,
                              layout=layout))
        net.append(layer)
    return net, shapes


def run_step(model, outs, ops, allreduce, dependent=True):
    """One decode step over the int4 Linears.  dependent (the HEADLINE since round 6): every Linear consumes the
    previous one's output -- qkv -> o (reads the q columns) -> gate|up -> down (reads the first `inter` columns; a 128-link chain of
    random matrices with the quadratic silu * up in it blows up numerically) -> the next layer's qkv -- which is what a decode step
    is: one launch per Linear, each waiting for its producer (the slice a consumer takes is a view at batch 1, no copy launch).
    dependent=False: every Linear on its own fixed input (rounds 1-5's headline, now the `decode_independent` leg).
    allreduce: None (one GPU), or a callable summing a [1, hidden] fp16 tensor over the ranks in place."""
    i = 0
    t = model[0][0]["x"]
    for layer in model:
        for lin in layer:
            x = t[:, : lin["K"]] if dependent else lin["x"]
            if lin["layout"] == "gemm":
                y = ops.gemm_forward(x, lin["qw"], lin["sc"], lin["qz"])
            elif lin["layout"] == "gemv":
                y = ops.gemv_forward(x, lin["qw"], lin["sc"], lin["qz"], GROUP)
            else:
                y = ops.gemv_fast_forward(x, lin["qw"], lin["sc"], lin["qz"], GROUP)
            if lin["reduce"]:
                allreduce(y)
            outs[i] = y
            t = y
            i += 1


def unit_gain(model, ops, world=1):
    """Scale every Linear's scales so that its output has unit rms on a unit-rms input (random packed weights have a gain of ~10 per
    link: fp16 would overflow after a few links of the dependent chain).  Row-parallel Linears (their outputs are summed over the
    ranks) are scaled for a unit-rms SUM.  A GEMVFast Linear's zero terms -(s z) scale with its scales."""
    for layer in model:
        for lin in layer:
            y = forward_lin(ops, lin, lin["x"])
            rms = float(y.float().pow(2).mean().sqrt()) * (world ** 0.5 if lin["reduce"] else 1.0)
            f = 1.0 / max(rms, 1e-6)
            lin["sc"].mul_(f)
            if lin["layout"] == "gemvfast":
                lin["qz"].mul_(f)


def graph_time(fn, stream, reps, warm=2, min_seconds=0.0):
    """us per call of fn(): captured once, replayed `reps` times (at least `min_seconds`), HIP events on `stream`."""
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        for _ in range(warm):
            g.replay()
        stream.synchronize()
        total, n = 0.0, 0
        while True:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                g.replay()
            e1.record(stream)
            e1.synchronize()
            total += e0.elapsed_time(e1)
            n += reps
            if total >= min_seconds * 1e3:
                break
    del g
    return total * 1e3 / n


# ------------------------------------------------------------------------------------------ secondary legs

def leg_gemm_bs(dev, ops):
    """BASELINE configs[2] / north_star: "bs=8 GEMM for 4096x11008" -- HBM-bound (AI 30 flop/B): us per call over
    distinct matrices (> 256 MiB of them, so nothing is served by the Infinity Cache), GB/s, fraction of 8 TB/s."""
    K, N = 4096, 11008
    gen = torch.Generator(device=dev).manual_seed(5)
    nsets = 28  # 28 x 22.5 MB = 631 MB
    sets = [rand_packed(K, N, GROUP, dev, gen) for _ in range(nsets)]
    st = torch.cuda.Stream(device=dev)
    out = {"shape": f"{K}x{N} g{GROUP}", "weights": f"{nsets} distinct matrices ({nsets * K * N // 2 / 1e6:.0f} MB), one call each per replay",
           "unit": "us per call", "by_batch": {}}
    for M in (1, 8, 16, 32, 64):
        x = torch.randn((M, K), device=dev, generator=gen).half()

        def fn():
            for qw, qz, sc in sets:
                ops.gemm_forward(x, qw, sc, qz)

        us = graph_time(fn, st, reps=5) / nsets
        by = algorithmic_bytes(K, N, M, GROUP)
        out["by_batch"][str(M)] = {"us": us, "kernel": ops.last_kernel(),
                                   "roofline": {"bound": "hbm", "achieved": by / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": by / us / 1e3 / HBM_PEAK_GBS, "bytes_per_launch": by}}
    traffic = pmc_traffic_gemm_bs()  # HBM bytes per call from the committed FETCH_SIZE pass (null when taken at other kernel sources)
    for M in (1, 8, 64):
        r = out["by_batch"][str(M)]["roofline"]
        r["traffic"] = traffic.get(("gemm", str(M)))
        r["traffic_measured_in"] = "profiles/" + PMC_BS_FILE
    out["bs8"] = out["by_batch"]["8"]
    del sets
    torch.cuda.empty_cache()
    # the same matrix shape in the WQLinear_GEMV format (awq_gemv_forward: row-streaming kernel to 4 rows, the batched kernel from 5)
    sets = [rand_packed_nk(K, N, GROUP, dev, gen) for _ in range(nsets)]
    out["gemv_layout_by_batch"] = {}
    for M in (1, 2, 4, 5, 8, 12, 16, 24, 32, 48, 64, 96, 128):  # (from four rows at this shape csrc/gemv_batch.hip: one launch per <= 128 rows on the layout's own buffers)
        x = torch.randn((M, K), device=dev, generator=gen).half()

        def fn2():
            for qw, qz, sc in sets:
                ops.gemv_forward(x, qw, sc, qz, GROUP)

        us = graph_time(fn2, st, reps=5) / nsets
        by = algorithmic_bytes(K, N, M, GROUP)
        out["gemv_layout_by_batch"][str(M)] = {"us": us, "kernel": ops.last_kernel(),
                                               "roofline": {"bound": "hbm", "achieved": by / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                            "frac": by / us / 1e3 / HBM_PEAK_GBS, "bytes_per_launch": by}}
    for M in (1, 8, 64, 96, 128):
        r = out["gemv_layout_by_batch"][str(M)]["roofline"]
        r["traffic"] = traffic.get(("gemv", str(M)))
        r["traffic_measured_in"] = "profiles/" + PMC_BS_FILE
    out["gemv_layout_bs8"] = out["gemv_layout_by_batch"]["8"]  # north_star's "bs=8 ... 4096x11008" on the default decode layout
    del sets
    torch.cuda.empty_cache()
    # ... and in the WQLinear_GEMVFast format (round 5: the batched kernel reads this layout too)
    sets = [rand_packed_nk(K, N, GROUP, dev, gen, fast=True) for _ in range(nsets)]
    out["gemvfast_layout_by_batch"] = {}
    for M in (1, 8, 32, 64, 128):
        x = torch.randn((M, K), device=dev, generator=gen).half()

        def fn3():
            for qw, qz, sc in sets:
                ops.gemv_fast_forward(x, qw, sc, qz, GROUP)

        us = graph_time(fn3, st, reps=5) / nsets
        by = algorithmic_bytes(K, N, M, GROUP) + (K // GROUP) * N * 2 - (K // GROUP) * (N // 8) * 4  # fp16 zero terms instead of packed nibbles
        out["gemvfast_layout_by_batch"][str(M)] = {"us": us, "kernel": ops.last_kernel(),
                                                   "roofline": {"bound": "hbm", "achieved": by / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                                "frac": by / us / 1e3 / HBM_PEAK_GBS, "bytes_per_launch": by}}
    del sets
    torch.cuda.empty_cache()
    return out


def leg_gemm_prefill(dev, ops):
    """BASELINE configs[2]: bs=8 x seq=2048 -> M = 16384 on 4096x11008 (MFMA-bound, AI 2850 flop/B), and M = 4096 / 8192 next
    to it: the fused register-decoded kernel (csrc/gemm_regb.hip, what awq_gemm_forward's AUTO dispatch runs at these
    sizes), the LDS-tiled fused kernel of round 1, the reference's own two-pass route (HIP dequant + vendor fp16 GEMM,
    gemm.py:48-54), and what WQLinear_GEMM.forward dispatches to."""
    from autoawq_amd import WQLinear_GEMM

    K, N, M = 4096, 11008, 16384
    gen = torch.Generator(device=dev).manual_seed(6)
    qw, qz, sc = rand_packed(K, N, GROUP, dev, gen)
    x = torch.randn((M, K), device=dev, generator=gen).half()
    fl = 2.0 * M * K * N

    def timeit(fn, reps=5, batches=3):  # median of three batches: these MFMA-bound calls move the clock (DVFS) as they run
        fn()
        fn()
        out = []
        for _ in range(batches):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / reps)
        return statistics.median(out)

    for _ in range(10):  # bring the part to its sustained clock before the first measurement
        ops.gemm_forward(x, qw, sc, qz)
    us_f = timeit(lambda: ops.gemm_forward(x, qw, sc, qz))
    kernel = ops.last_kernel()
    us_t = timeit(lambda: ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=2)))
    us_2 = timeit(lambda: torch.matmul(x, ops.dequantize_weights(qw, sc, qz)))
    mod = WQLinear_GEMM(4, GROUP, K, N, False, dev)
    mod.qweight, mod.qzeros, mod.scales = qw, qz, sc
    us_m = timeit(lambda: mod(x))
    a = ops.gemm_forward(x[:2048], qw, sc, qz, flags=ops.gemm_flags(ops.KERNEL_REGB)).float()
    b = torch.matmul(x[:2048], ops.dequantize_weights(qw, sc, qz)).float()
    rel = float((a - b).abs().max() / b.abs().max())
    assert rel < 5e-3, f"fused prefill kernel disagrees with the two-pass route: {rel}"
    by_m = {}
    for m in (4096, 8192):
        xs = x[:m]
        f = 2.0 * m * K * N
        uf, u2 = timeit(lambda: ops.gemm_forward(xs, qw, sc, qz)), timeit(lambda: torch.matmul(xs, ops.dequantize_weights(qw, sc, qz)))
        by_m[str(m)] = {"fused_us": uf, "fused_tflops": f / uf / 1e6, "two_pass_us": u2, "two_pass_tflops": f / u2 / 1e6}

    def roof(us):
        return {"bound": "mfma", "achieved": fl / us / 1e6, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": fl / us / 1e6 / MFMA_PEAK_TF}

    # the same matrix in the WQLinear_GEMV checkpoint format: the N-major form of the same kernel on that layout's own buffers
    # (round 4: no GEMM-layout copy of the weights behind WQLinear_GEMV any more)
    del qw, qz, sc
    torch.cuda.empty_cache()
    nq, nz, ns = rand_packed_nk(K, N, GROUP, dev, gen)
    pre = ops.gemm_flags(kernel=ops.GEMV_KERNEL_PREFILL)
    us_nk = timeit(lambda: ops.gemv_forward(x, nq, ns, nz, GROUP, flags=pre))
    nk_kernel = ops.last_kernel()
    us_nk2 = timeit(lambda: torch.matmul(x, ops.dequantize_weights_gemv(nq, ns, nz, GROUP).t()))
    an = ops.gemv_forward(x[:2048], nq, ns, nz, GROUP, flags=pre).float()
    bn = torch.matmul(x[:2048], ops.dequantize_weights_gemv(nq, ns, nz, GROUP).t()).float()
    rel_nk = float((an - bn).abs().max() / bn.abs().max())
    assert rel_nk < 5e-3, f"GEMV-layout prefill kernel disagrees with dequantise + GEMM: {rel_nk}"
    nk_by_m = {str(m): 2.0 * m * K * N / timeit(lambda: ops.gemv_forward(x[:m], nq, ns, nz, GROUP, flags=pre)) / 1e6 for m in (4096, 8192)}
    # round 5, the module's default route: the packed nibbles transposed into a temporary of the call (csrc/repack.hip) + the fused MFMA
    # GEMM above on it -- two hand-written launches, no vendor GEMM, nothing resident
    us_rp = timeit(lambda: ops.gemv_prefill_repack(x, nq, ns, nz, GROUP))
    us_rk = timeit(lambda: ops.repack_gemv_to_gemm(nq, ns, nz, GROUP), reps=20)
    rp_by_m = {str(m): 2.0 * m * K * N / timeit(lambda: ops.gemv_prefill_repack(x[:m], nq, ns, nz, GROUP)) / 1e6 for m in (256, 1024, 4096)}
    ar = ops.gemv_prefill_repack(x[:2048], nq, ns, nz, GROUP).float()
    rel_rp = float((ar - bn).abs().max() / bn.abs().max())
    assert rel_rp < 5e-3, f"repack prefill route disagrees with dequantise + GEMM: {rel_rp}"
    from autoawq_amd import WQLinear_GEMV

    modv = WQLinear_GEMV(4, GROUP, K, N, False, dev)
    modv.qweight, modv.qzeros, modv.scales = nq, nz, ns
    us_mv = timeit(lambda: modv(x))

    # round 6, the same matrix in the WQLinear_GEMVFast format: the words transposed into a temporary (csrc/repack.hip) + the fused MFMA
    # GEMM in its FZ form (W = fp16(w s + qzeros), the format's own scales / fp16 zero terms) -- the role of
    # awq_v2_ext.gemm_forward_cuda_prefill (awq/modules/linear/gemv_fast.py:203-206); beside it dequantise + vendor GEMM
    from autoawq_amd import WQLinear_GEMVFast
    from autoawq_amd.modules.linear.gemv import prefill_route

    route_v = "repack" if prefill_route(M, K, N) == "hand" else "two_pass"
    del nq, nz, ns, modv
    torch.cuda.empty_cache()
    fq, fz, fs = rand_packed_nk(K, N, GROUP, dev, gen, fast=True)
    us_fz = timeit(lambda: ops.gemv_fast_prefill(x, fq, fs, fz, GROUP))
    fz_kernel = ops.last_kernel()
    us_fz2 = timeit(lambda: torch.matmul(x, ops.dequantize_weights_gemv_fast(fq, fs, fz, GROUP).t()))
    us_frk = timeit(lambda: ops.repack_gemvfast_to_gemm(fq), reps=20)
    af = ops.gemv_fast_prefill(x[:2048], fq, fs, fz, GROUP).float()
    bf = torch.matmul(x[:2048], ops.dequantize_weights_gemv_fast(fq, fs, fz, GROUP).t()).float()
    rel_fz = float((af - bf).abs().max() / bf.abs().max())
    assert rel_fz < 5e-3, f"GEMVFast prefill route disagrees with dequantise + GEMM: {rel_fz}"
    fz_by_m = {str(m): 2.0 * m * K * N / timeit(lambda: ops.gemv_fast_prefill(x[:m], fq, fs, fz, GROUP)) / 1e6 for m in (2048, 4096)}
    modf = WQLinear_GEMVFast(4, GROUP, K, N, False, dev)
    modf.qweight, modf.qzeros, modf.scales = fq, fz, fs
    x3 = x.view(8, M // 8, K)
    us_mf = timeit(lambda: modf(x3))

    return {"shape": f"{K}x{N} g{GROUP}, M={M} (bs 8 x seq 2048)", "flops": fl,
            "gemvfast_layout": {"fused": {"us": us_fz, "kernel": fz_kernel, "roofline": roof(us_fz), "repack_kernel_us": us_frk,
                                          "repack_kernel_gbs": K * N / us_frk / 1e3, "tflops_other_token_counts": fz_by_m,
                                          "vs_dequant_plus_gemm_max_rel": rel_fz},
                                "two_pass": {"us": us_fz2, "roofline": roof(us_fz2)},
                                "module": {"us": us_mf, "roofline": roof(us_mf),
                                           "route": modf.PREFILL_IMPL + " -> " + ("fused" if prefill_route(M, K, N) == "hand" else "two_pass")},
                                "what": "WQLinear_GEMVFast buffers (qweight int16 [N/4, K], fp16 zero terms): `fused` = awq_gemv_fast_prefill, two "
                                        "hand-written launches (round 6); `two_pass` = awq_dequantize_weights_gemv_fast + a dense vendor GEMM; the "
                                        "module's `auto` takes the faster one by shape and token count (modules/linear/gemv.py::prefill_route)"},
            "gemv_layout": {"fused_nk": {"us": us_nk, "kernel": nk_kernel, "roofline": roof(us_nk), "tflops_other_token_counts": nk_by_m,
                                         "vs_dequant_plus_gemm_max_rel": rel_nk},
                            "two_pass": {"us": us_nk2, "roofline": roof(us_nk2)},
                            "repack": {"us": us_rp, "roofline": roof(us_rp), "repack_kernel_us": us_rk,
                                       "repack_kernel_gbs": K * N / us_rk / 1e3, "tflops_other_token_counts": rp_by_m,
                                       "vs_dequant_plus_gemm_max_rel": rel_rp},
                            "module": {"us": us_mv, "roofline": roof(us_mv), "route": "auto -> " + route_v},
                            "what": "WQLinear_GEMV buffers (qweight [N, K/8]), no second copy of the weights: `repack` (csrc/repack.hip transposes "
                                    "the packed nibbles into a temporary, then the fused MFMA GEMM: hand-written end to end), `two_pass` "
                                    "(awq_dequantize_weights_gemv + a dense vendor GEMM), `fused_nk` (AWQ_GEMV_KERNEL_PREFILL: gemm_regb.hip, N-major "
                                    "form); the module's default since round 6 is `auto`: the faster of repack / two_pass by shape and token count "
                                    "(modules/linear/gemv.py::prefill_route, profiles/r06_prefill_routes.txt; ADVICE r05)"},
            "fused_mfma": {"us": us_f, "kernel": kernel, "roofline": roof(us_f)},
            "fused_lds_tiled_r01": {"us": us_t, "roofline": roof(us_t)}, "two_pass": {"us": us_2, "roofline": roof(us_2)},
            "module": {"us": us_m, "roofline": roof(us_m),
                       "route": "fused (awq_gemm_forward AUTO -> " + {ops.KERNEL_REGB: "gemm_regb", ops.KERNEL_TILED: "gemm_tiled"}.get(ops.auto_kernel(M, K, N, GROUP), "other") +
                                "); the dequant + vendor-GEMM route is opt-in only since round 3 (modules/linear/gemm.py PREFILL_IMPL)"},
            "other_token_counts": by_m,
            "fused_vs_two_pass_max_rel": rel}


def leg_prefill_attention(dev, ops):
    """BASELINE configs[2]'s attention (bs 8 x seq 2048, 32 heads x 128): the hand-written flash-style prefill kernel
    (csrc/prefill_attn.hip, the reference's flash_attn_func call attn.py:269-277) beside the vendor's scaled_dot_product_attention on
    the same tensors; MFMA roofline on the causal flops 2 * 2 * B * H * D * S (S + 1) / 2."""
    import torch.nn.functional as F

    B, S, H, D = 8, 2048, 32, 128
    gen = torch.Generator(device=dev).manual_seed(8)
    q = torch.randn((B, S, H, D), device=dev, generator=gen).half()
    kc = torch.randn((B, S, H, D), device=dev, generator=gen).half()
    vc = torch.randn((B, S, H, D), device=dev, generator=gen).half()
    fl = 4.0 * B * H * D * S * (S + 1) / 2

    def timeit(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    us = timeit(lambda: ops.prefill_attention(q, kc, vc, 0))
    qt, kt, vt = q.transpose(1, 2), kc.transpose(1, 2), vc.transpose(1, 2)
    us_v = timeit(lambda: F.scaled_dot_product_attention(qt, kt, vt, is_causal=True))
    a = ops.prefill_attention(q[:1], kc[:1], vc[:1], 0).float()
    b = F.scaled_dot_product_attention(qt[:1].float(), kt[:1].float(), vt[:1].float(), is_causal=True).transpose(1, 2)
    rel = float((a - b).abs().max() / b.abs().max())
    assert rel < 5e-3, f"prefill attention disagrees with fp32 attention: {rel}"
    out = {"shape": f"B {B} x S {S} x {H} heads x {D} (causal, start 0)", "flops": fl, "us": us, "kernel": "awq_prefill_attn_kernel",
           "roofline": {"bound": "mfma", "achieved": fl / us / 1e6, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": fl / us / 1e6 / MFMA_PEAK_TF},
           "vendor_sdpa_us": us_v, "vendor_sdpa_tflops": fl / us_v / 1e6, "vs_fp32_attention_max_rel": rel,
           "note": "what QuantAttentionFused runs for prefill steps at head_dim 128 (GQA, chunked prefill, ALiBi, soft cap in the same kernel)"}
    return out


def leg_moe_prefill(dev):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_moe

    r = bench_moe.run_prefill(dev=dev, verbose=False)
    r["what"] = ("Mixtral-8x7B-shape fused MoE MLP at 512 tokens (1024 pairs): routing (one launch), a counting sort of the pairs (one launch), ONE "
                 "grouped launch of the register-decoded MFMA GEMM per projection with the sort kept as an index list -- w1|w3 reads the tokens' rows "
                 "through it, w2 writes pair rows with the routing weight in its one rounding (awq_grouped_gemm_prefill_ex, modules/fused/moe.py)")
    return r


def leg_moe(dev):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_moe

    r = bench_moe.run(dev=dev, verbose=False)
    by, us = r["bytes"], r["us_per_block"]
    ug = r["gemm_layout_grouped_us"]
    return {"what": "Mixtral-8x7B-shape fused MoE MLP, bs=4, top-2 (BASELINE configs[4]); router excluded; the decode path of "
                    "fuse_mixtral(decode_layout='auto'): GEMV-layout twins of the expert stacks, every (token, expert) pair one batch-1 "
                    "call of the row-streaming kernel, all pairs in ONE launch per projection (awq_grouped_gemv_forward), silu * mul "
                    "and the routing weight in the launches' epilogues", "us_per_block": us, "kernel": r["kernel"],
            "experts_hit": r["experts_hit"], "checked_against": r["checked_against"],
            "roofline": {"bound": "hbm", "achieved": by / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / us / 1e3 / HBM_PEAK_GBS,
                         "bytes_per_block": by},
            "gemm_layout_grouped_kernel": {"us_per_block": ug, "frac": by / ug / 1e3 / HBM_PEAK_GBS,
                                           "what": "the checkpoint's own GEMM-layout stacks through awq_grouped_gemm_forward (rounds 1-5's path)"}}


def forward_lin(ops, lin, x):
    if lin["layout"] == "gemm":
        return ops.gemm_forward(x, lin["qw"], lin["sc"], lin["qz"])
    if lin["layout"] == "gemv":
        return ops.gemv_forward(x, lin["qw"], lin["sc"], lin["qz"], GROUP)
    return ops.gemv_fast_forward(x, lin["qw"], lin["sc"], lin["qz"], GROUP)


def leg_decode_independent(dev, ops, model, bytes_step):
    """Rounds 1-5's headline as a secondary leg: the same 128 launches, every Linear on its own FIXED input (no launch waits for the
    data of its predecessor; the launches still run one after the other on one stream)."""
    nl = sum(len(l) for l in model)
    outs = [None] * nl
    st = torch.cuda.Stream(device=dev)
    us = graph_time(lambda: run_step(model, outs, ops, None, dependent=False), st, reps=10, min_seconds=0.3)
    return {"what": "the headline's Linears with FIXED inputs per Linear (the headline of rounds 1-5): one launch each, no data dependency",
            "layout": model[0][0]["layout"], "ms_per_token": us / 1e3, "tok_s": 1e6 / us, "links": nl,
            "roofline": {"bound": "hbm", "achieved": bytes_step / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bytes_step / us / 1e3 / HBM_PEAK_GBS},
            "note": "round 6's persistent engine (one launch for the whole dependent chain) is correct and 0.69 x the launches' rate: "
                    "profiles/r06_engine_probe.txt, tools/experimental/engine/"}


def leg_whole_model(dev):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_decode_model

    # the decode layout (WQLinear_GEMV: fuse_llama(decode_layout="gemv")) and the checkpoint's own GEMM layout
    wm = bench_decode_model.run(contexts=(64, 2048), steps=48, dev=dev, verbose=False, check=True, layout="gemv")
    wg = bench_decode_model.run(contexts=(64, 2048), steps=48, dev=dev, verbose=False, check=True, layout="gemm")
    return {"unit": "tok/s", "layout": "gemv", "context_64": 1000.0 / wm[64], "context_2048": 1000.0 / wm[2048],
            "gemm_layout": {"context_64": 1000.0 / wg[64], "context_2048": 1000.0 / wg[2048]},
            "what": "synthetic 7B-shape fused decoder (32 blocks + lm_head) through modules/fused/decode.py::GraphedDecoder: one hipGraph "
                    "replay per token, one captured step per context-length bucket (256 / 1024 / 4096 ... rows; the attention launch is "
                    "sized for the bucket), five launches per block (qkv with the norm in its prologue, RoPE + append + attention, "
                    "o_proj + residual, gate|up with norm in and silu * mul out, down + residual); logits of the replayed graph "
                    "checked against the unfused module path before timing (HIP vs HIP: a consistency check, the parity tests "
                    "against the reference's logits are tests/test_decoder.py)",
            "published_reference": {"value": 198.848, "context": 64, "hardware": "RTX 4090", "source": "README.md:207 (BASELINE.md)"},
            "vs_published_ctx64": (1000.0 / wm[64]) / 198.848}


def cpu_baseline(layers_total):
    """AutoAWQ's own CPU path (dequantize_gemm + fp16 matmul, awq/modules/linear/gemm.py:71-79) restated in torch
    (oracle/awq_oracle.py, kind="port": /root/reference does not exist on the GPU box), on this host's cores.
    SURVEY.md 8(d): per shape, M = 1 and M = 8, dequant and matmul apart, plus the cached-dequant variant (matmul
    only).  Bounded sample: the three distinct Linear shapes of one layer, median of 3 after a warm-up."""
    from oracle import awq_oracle

    gen = torch.Generator().manual_seed(7)
    # 256 threads on this box OVERSUBSCRIBE these memory-bound torch ops (5x slower than 8): take the best of {8, 32, all}
    ncpu = os.cpu_count() or 1
    qw0, qz0, sc0 = rand_packed(HIDDEN, HIDDEN, GROUP, "cpu", gen)
    tries = {}
    for nthr in sorted({min(8, ncpu), min(32, ncpu), ncpu}):
        torch.set_num_threads(nthr)
        awq_oracle.torch_dequantize_gemm(qw0, qz0, sc0, GROUP)
        t0 = time.perf_counter()
        awq_oracle.torch_dequantize_gemm(qw0, qz0, sc0, GROUP)
        tries[nthr] = time.perf_counter() - t0
    torch.set_num_threads(min(tries, key=tries.get))

    def med(fn, n=3):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    per_shape = {}
    for K, N in [(HIDDEN, HIDDEN), (HIDDEN, INTER), (INTER, HIDDEN)]:
        qw, qz, sc = rand_packed(K, N, GROUP, "cpu", gen)
        t_dq = med(lambda: awq_oracle.torch_dequantize_gemm(qw, qz, sc, GROUP))
        W = awq_oracle.torch_dequantize_gemm(qw, qz, sc, GROUP)
        e = {"dequant_s": t_dq}
        for M in (1, 8):
            x = torch.randn((M, K), generator=gen).half()
            e[f"matmul_M{M}_s"] = med(lambda: torch.matmul(x, W), n=5)
            e[f"linear_M{M}_s"] = t_dq + e[f"matmul_M{M}_s"]
        per_shape[f"{K}x{N}"] = e
    a, b, c = (per_shape[f"{HIDDEN}x{HIDDEN}"], per_shape[f"{HIDDEN}x{INTER}"], per_shape[f"{INTER}x{HIDDEN}"])

    def layer_s(key):  # q, k, v, o + gate, up + down: the reference calls seven Linears per layer on CPU (no fused qkv there)
        return 4 * a[key] + 2 * b[key] + c[key]

    return {"value": 1.0 / (layer_s("linear_M1_s") * layers_total), "unit": "tok/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cores": ncpu, "threads": torch.get_num_threads(),  # os.cpu_count() of the GPU box, and the thread count used (the fastest tried)
            "thread_counts_tried_dequant_4096x4096_s": {str(k): v for k, v in tries.items()},
            "sample": f"the 3 distinct Linear shapes of 1 of {layers_total} layers, median of 3 (dequant) / 5 (matmul); a layer = 4 x "
                      f"{HIDDEN}x{HIDDEN} + 2 x {HIDDEN}x{INTER} + 1 x {INTER}x{HIDDEN} = {layer_s('linear_M1_s'):.2f} s; "
                      "torch-CPU restatement of dequantize_gemm + fp16 matmul",
            "per_shape": per_shape,
            "tok_s_M8_per_sequence": 1.0 / (layer_s("linear_M8_s") * layers_total),
            "cached_dequant_tok_s_M1": 1.0 / (layer_s("matmul_M1_s") * layers_total),
            "cached_dequant_tok_s_M8_per_sequence": 1.0 / (layer_s("matmul_M8_s") * layers_total)}


def leg_per_shape(dev, ops, model, shapes):
    """The four Linear shapes of the headline one by one: the 32 layers' instances of a shape (distinct weights: 0.28 - 1.5 GB,
    nothing comes from the Infinity Cache) captured in one hipGraph; us per launch (HIP events on the launch stream),
    algorithmic GB/s, fraction of 8 TB/s."""
    st = torch.cuda.Stream(device=dev)
    out = {}
    for idx, (name, K, N, _) in enumerate(shapes):
        lins = [layer[idx] for layer in model]

        def fn():
            for l in lins:
                forward_lin(ops, l, l["x"])

        us = graph_time(fn, st, reps=20, min_seconds=0.05) / len(lins)
        by = algorithmic_bytes(K, N, 1, GROUP)
        out[name] = {"K": K, "N": N, "us_per_launch": us, "bytes_per_launch": by, "kernel": ops.last_kernel(),
                     "roofline": {"bound": "hbm", "achieved": by / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / us / 1e3 / HBM_PEAK_GBS}}
    return out


def leg_by_layout(dev, ops, layers, skip):
    """The headline step (every int4 Linear of the model, batch 1, one hipGraph) on the other two checkpoint formats."""
    st = torch.cuda.Stream(device=dev)
    out = {}
    for layout in ("gemm", "gemv", "gemvfast"):
        if layout == skip:
            continue
        model, _ = build_model(dev, 0, 1, layers, layout=layout)
        by = sum(algorithmic_bytes(l["K"], l["N"], 1, GROUP) for layer in model for l in layer)
        if layout == "gemvfast":
            by += sum((l["K"] // GROUP) * l["N"] * 3 // 2 for layer in model for l in layer)
        outs = [None] * sum(len(l) for l in model)
        unit_gain(model, ops)
        us = graph_time(lambda: run_step(model, outs, ops, None), st, reps=20, min_seconds=0.3)
        kernels = {}
        for lin in model[0]:  # (one eager call per Linear of a layer, after the timed region: which kernel AUTO took at each shape)
            forward_lin(ops, lin, lin["x"])
            kernels[f'{lin["K"]}x{lin["N"]}'] = ops.last_kernel()
        out[layout] = {"tok_s": 1e6 / us, "ms_per_step": us / 1e3, "kernel": "/".join(sorted(set(kernels.values()))), "kernel_by_shape": kernels,
                       "roofline": {"bound": "hbm", "achieved": by / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / us / 1e3 / HBM_PEAK_GBS}}
        del model, outs
        torch.cuda.empty_cache()
    return out


def dry_run(a):
    """`bench.py --gpus N --dry-run` under torch.distributed.run on CPU: everything of the N > 1 path that is not a kernel."""
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE {world}"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    dev = torch.device("cpu")
    cfg = MODELS[a.model]
    layers = a.layers or cfg["layers"]
    model, shapes = build_model(dev, rank, world, layers, layout=a.layout, model=a.model)

    class NoLaunch:  # the three entry points run_step calls, shapes only
        @staticmethod
        def _y(x, n):
            return torch.full((x.shape[0], n), float(rank + 1), dtype=torch.float16)

        def gemm_forward(self, x, qw, sc, qz):
            return self._y(x, qw.shape[1] * 8)

        def gemv_forward(self, x, qw, sc, qz, g):
            return self._y(x, qw.shape[0])

        def gemv_fast_forward(self, x, qw, sc, qz, g):
            return self._y(x, qw.shape[0] * 4)

    from autoawq_amd.comm import make_collective

    allreduce, collective, oneshot = make_collective(cfg["hidden"], dev)
    assert oneshot is None and collective.startswith("RCCL all_reduce via torch.distributed"), collective
    outs = [None] * sum(len(l) for l in model)
    ops = NoLaunch()
    for _ in range(a.warmup):
        run_step(model, outs, ops, allreduce)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_step(model, outs, ops, allreduce)
    dist.barrier()
    t = torch.tensor([(time.perf_counter() - t0) * 1e3])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # every row-parallel output went through the collective: each element is the sum of (rank + 1) over the ranks
    reduced = [o for o, lin in zip(outs, (l for layer in model for l in layer)) if lin["reduce"]]
    ok = all(bool((o == world * (world + 1) / 2).all()) for o in reduced) and len(reduced) == 2 * layers
    tb = torch.tensor([float(sum(algorithmic_bytes(l["K"], l["N"], 1, GROUP) for layer in model for l in layer))])
    dist.all_reduce(tb)
    if rank == 0:
        print(json.dumps({"metric": f"decode tok/s @bs=1 (int4 linears), {a.model.upper()} AWQ-int4 g128", "value": None, "unit": "tok/s",
                          "dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": float(t.item()) / a.steps,
                          "scaling": "strong", "config": {"workload": "no launches: multi-rank plumbing only", "layers": layers,
                                                          "parallelism": f"tp{world}", "collective": collective,
                                                          "collectives_per_step": len(reduced), "collectives_summed_correctly": ok,
                                                          "shard_shapes_rank0": [[n, K, N] for n, K, N, _ in shapes],
                                                          "algorithmic_bytes_per_step_all_ranks": float(tb.item())}}), flush=True)
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", choices=sorted(MODELS), default="7b", help="7b = BASELINE configs[1] (the metric); 70b = configs[3] (TP=8)")
    ap.add_argument("--layers", type=int, default=0, help="debug only; the metric is quoted at the model's full depth")
    ap.add_argument("--layout", choices=["gemm", "gemv", "gemvfast"], default="gemv",
                    help="checkpoint format of the Linears: WQLinear_GEMV (default: the reference's own batch-1 format, README.md:96-97) / "
                         "WQLinear_GEMM / WQLinear_GEMVFast buffers")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-whole-model", action="store_true", help="skip the secondary whole-decoder figure")
    ap.add_argument("--no-secondary", action="store_true", help="headline leg (and cpu_baseline) only")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the multi-rank plumbing only (rendezvous over gloo, per-rank shard shapes, the collective's rank-"
                         "consistent fallback, barrier + max-over-ranks timing) with launches replaced by zero outputs; prints a line "
                         "marked dry_run whose value is null (tests/test_tp_cpu.py runs it with two processes)")
    a = ap.parse_args()
    if a.dry_run:
        return dry_run(a)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or (a.gpus == 1 and world == 1), f"--gpus {a.gpus} but WORLD_SIZE {world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from autoawq_amd import _lib, ops

    _lib.lib()
    cfg = MODELS[a.model]
    layers = a.layers or cfg["layers"]
    if world > 1 and a.layout == "gemvfast":
        raise SystemExit("tensor-parallel shards are implemented for the GEMM and GEMV layouts")
    model, shapes = build_model(dev, rank, world, layers, layout=a.layout, model=a.model)
    unit_gain(model, ops, world)  # (the headline is the DEPENDENT chain: unit gain per link keeps 128 links inside fp16)
    nl = sum(len(l) for l in model)
    outs = [None] * nl
    bytes_step = sum(algorithmic_bytes(l["K"], l["N"], 1, GROUP) for layer in model for l in layer)
    if a.layout == "gemvfast":  # zeros are stored as fp16 -(s*z): 2 bytes instead of half a byte per (group, column)
        bytes_step += sum((l["K"] // GROUP) * l["N"] * 3 // 2 for layer in model for l in layer)

    # the TP collective: the one-shot xGMI all-reduce of csrc/allreduce.hip (a plain kernel launch: the whole step stays one
    # hipGraph); RCCL through torch.distributed if its setup (CUDA-IPC mapping of the peers' buffers) fails
    allreduce, collective, oneshot = None, "none", None
    if world > 1:
        from autoawq_amd.comm import make_collective

        allreduce, collective, oneshot = make_collective(cfg["hidden"], dev)  # every rank takes the same branch
        if rank == 0:
            print(f"[bench] collective: {collective}", file=sys.stderr)

    stream = torch.cuda.Stream(device=dev)
    graph, used_graph, capture_note = None, False, None
    with torch.cuda.stream(stream):
        for _ in range(max(a.warmup, 3) if a.no_graph else 3):
            run_step(model, outs, ops, allreduce)
        stream.synchronize()
        if not a.no_graph:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    run_step(model, outs, ops, allreduce)
                used_graph = True
            except Exception as e:  # e.g. collective not capturable: fall back to eager launches, and SAY so
                capture_note = f"hipGraph capture failed ({type(e).__name__}: {str(e)[:200]}): every launch and all-reduce is issued eagerly"
                if rank == 0:
                    print(f"[bench] {capture_note}", file=sys.stderr)
                graph = None
        step = graph.replay if graph is not None else (lambda: run_step(model, outs, ops, allreduce))
        if dist is not None:
            dist.barrier()  # ranks leave the capture together: the one-shot all-reduce waits a bounded time (~1 s) for a peer
        for _ in range(a.warmup):
            step()
        stream.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.steps):
            step()
        e1.record(stream)
        e1.synchronize()
        torch.cuda.synchronize()
        ms_total = e0.elapsed_time(e1)
        if oneshot is not None:
            oneshot.check()  # a rank that gave up on a peer wrote NaN and raised the sticky word: the timing would be meaningless
        sustained = None
        if world == 1 and not a.no_secondary:  # the same replay for >= 1 s: a leg the GPU-busy sampler of the driver can see
            t_end, n_sus = time.perf_counter() + 1.2, 0
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record(stream)
            while time.perf_counter() < t_end:
                for _ in range(50):
                    step()
                n_sus += 50
                stream.synchronize()
            s1.record(stream)
            s1.synchronize()
            sustained = (s0.elapsed_time(s1), n_sus)
    if dist is not None:
        t = torch.tensor([ms_total], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_total = float(t.item())
        tb = torch.tensor([float(bytes_step)], device=dev)
        dist.all_reduce(tb)
        bytes_all = float(tb.item())
        dist.barrier()
    else:
        bytes_all = float(bytes_step)
    ms_step = ms_total / a.steps

    if rank == 0:
        tok_s = 1000.0 / ms_step
        launches = nl
        # dominant kernel = the fused int4 GEMV; the timed region holds nothing but its launches
        # (N=1), so its average launch duration (incl. inter-kernel gap) = step time / launches.
        achieved = (bytes_step / launches) / (ms_step * 1e-3 / launches) / 1e9  # this rank's GB/s
        shape_txt = ", ".join(f"{n} {K}->{N}" for n, K, N, _ in shapes)
        traffic, traffic_note = (pmc_traffic_per_launch() if (world == 1 and a.model == "7b" and a.layout == "gemv" and layers == cfg["layers"])
                                 else (None, "headline configuration only"))
        out = {
            "metric": f"decode tok/s @bs=1 (int4 linears), {a.model.upper()} AWQ-int4 g128", "value": tok_s, "unit": "tok/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f16 (int4 weights, f32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{cfg['name']}-shape AWQ int4 g128, GEMV bs=1 decode: {layers} layers x {{{shape_txt}}}"
                                   + (" per rank" if world > 1 else ""),
                       "layers": layers, "launches_per_step": launches, "hipgraph": used_graph, "layout": a.layout,
                       "chain": "dependent: every Linear consumes its predecessor's output (qkv -> o -> gate|up -> down -> next qkv), unit "
                                "gain per link; `decode_independent` has rounds 1-5's fixed-input form",
                       "layout_note": "packed tensors in the reference's WQLinear_" + {"gemm": "GEMM", "gemv": "GEMV", "gemvfast": "GEMVFast"}[a.layout] +
                                      " checkpoint format (awq/modules/linear/); utils/convert.py repacks between the three, bit-exactly",
                       "parallelism": f"tp{world}" if world > 1 else "single",
                       "collectives_per_step": sum(1 for layer in model for l in layer if l["reduce"]), "collective": collective,
                       "algorithmic_bytes_per_step_all_ranks": bytes_all, "kernel": ops.last_kernel()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the fraction of the measured-achievable copy rate beside the fraction of the spec peak
                         "achievable_copy_gbs": HBM_COPY_GBS, "frac_of_achievable_copy": achieved / HBM_COPY_GBS,
                         # HBM bytes per launch need the TCC fabric counters of a separate rocprofv3 --pmc pass
                         # (MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950): not measurable from inside this process
                         "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_unit": "bytes per launch",
                         "traffic_measured_in": "profiles/" + PMC_FILE + " (own rocprofv3 --pmc FETCH_SIZE pass of this command, x 2 gfx950 correction, calibrated on a linear read; the file names the git head it was taken at)",
                         "bytes_per_launch": bytes_step / launches, "avg_launch_us": ms_step * 1e3 / launches,
                         "kernel": KERNEL_OF_LAYOUT[a.layout] + " (4 shapes per layer: qkv, o, gate+up, down; `per_shape` has each one's own figure)",
                         "note": "achieved = algorithmic bytes per launch / average launch duration; duration = "
                                 "HIP-event-timed replay of the captured stream / launches, i.e. it contains the "
                                 "dispatch gap exactly as rocprofv3's back-to-back kernel durations do "
                                 "(profiles/r06_bench_kernel_trace_stats.txt)"},
        }
        if capture_note:
            out["config"]["capture_note"] = capture_note
        if sustained is not None:
            ms_s, n_s = sustained
            out["sustained"] = {"seconds": ms_s / 1e3, "steps": n_s, "ms_per_step": ms_s / n_s, "tok_s": 1000.0 * n_s / ms_s,
                                "GBps": bytes_step * n_s / ms_s / 1e6, "frac": bytes_step * n_s / ms_s / 1e6 / HBM_PEAK_GBS}
        full = world == 1 and layers == cfg["layers"] and a.model == "7b" and not a.no_secondary
        if full:
            del graph, outs
            legs = [("per_shape", lambda: leg_per_shape(dev, ops, model, shapes)),
                    ("decode_independent", lambda: leg_decode_independent(dev, ops, model, bytes_step)),
                    ("by_layout", lambda: (model.clear(), torch.cuda.empty_cache(), leg_by_layout(dev, ops, layers, a.layout))[2]),
                    ("gemm_bs", lambda: leg_gemm_bs(dev, ops)), ("gemm_prefill", lambda: leg_gemm_prefill(dev, ops)),
                    ("prefill_attention", lambda: leg_prefill_attention(dev, ops)),
                    ("moe_bs4", lambda: leg_moe(dev)), ("moe_prefill", lambda: leg_moe_prefill(dev))]
            if not a.no_whole_model:
                legs.append(("whole_model", lambda: (model.clear(), torch.cuda.empty_cache(), leg_whole_model(dev))[2]))
            for name, fn in legs:
                try:
                    out[name] = fn()
                except Exception as e:  # the headline line must not depend on a secondary leg
                    out[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                torch.cuda.empty_cache()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg["layers"])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
