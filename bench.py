#!/usr/bin/env python3
"""bench.py -- decode tok/s @bs=1 + GEMV HBM GB/s on synthetic Llama-2-7B-shape AWQ int4 g128.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 launched by
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" = one decode token through every int4 Linear of the model (the hot path): per layer
qkv (fused, 4096->12288), o (4096->4096), gate+up (fused, 4096->22016), down (11008->4096), 32
layers, batch 1, distinct random packed weights per layer (3.37 GB working set >> the 256 MiB
Infinity Cache), captured in ONE hipGraph and replayed.  Inputs are resident in HBM before the
timed region.  N > 1: the same model tensor-parallel over N GPUs (column-split qkv / gate+up,
row-split o / down with one RCCL all-reduce each) -- strong scaling.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured-achievable copy is 6290
HIDDEN, INTER, LAYERS, GROUP = 4096, 11008, 32, 128


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


def build_model(dev, rank, world, layers, seed=1234, layout="gemm"):
    """Per-rank shard shapes: column split of qkv / gate+up, whole-group row split of o / down."""
    from autoawq_amd.tp import split_even_units

    gen = torch.Generator(device=dev).manual_seed(seed + rank)
    # o_proj rows / qkv columns: split the 32 heads (128 columns each)
    hs, hc = split_even_units(HIDDEN // 128, world)[rank]
    # down rows: split the 86 groups; gate/up columns follow the same bounds
    gs, gc = split_even_units(INTER // GROUP, world)[rank]
    shapes = [("qkv", HIDDEN, 3 * hc * 128, False), ("o", hc * 128, HIDDEN, True),
              ("gate_up", HIDDEN, 2 * gc * GROUP, False), ("down", gc * GROUP, HIDDEN, True)]
    model = []
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
        model.append(layer)
    return model, shapes


def run_step(model, outs, ops, dist):
    i = 0
    for layer in model:
        for lin in layer:
            if lin["layout"] == "gemm":
                y = ops.gemm_forward(lin["x"], lin["qw"], lin["sc"], lin["qz"])
            elif lin["layout"] == "gemv":
                y = ops.gemv_forward(lin["x"], lin["qw"], lin["sc"], lin["qz"], GROUP)
            else:
                y = ops.gemv_fast_forward(lin["x"], lin["qw"], lin["sc"], lin["qz"], GROUP)
            if lin["reduce"]:
                dist.all_reduce(y)
            outs[i] = y
            i += 1


def cpu_baseline(layers_total):
    """AutoAWQ's own CPU path (dequantize_gemm + fp16 matmul, awq/modules/linear/gemm.py:71-79)
    restated in torch (oracle/awq_oracle.py, kind="port"), timed on this host's cores on a bounded
    sample: the 4 Linears of ONE layer, median of 3 after one warm-up."""
    from oracle import awq_oracle

    torch.set_num_threads(os.cpu_count() or 1)
    gen = torch.Generator().manual_seed(7)
    lins = []
    for K, N in [(HIDDEN, 3 * HIDDEN), (HIDDEN, HIDDEN), (HIDDEN, 2 * INTER), (INTER, HIDDEN)]:
        qw, qz, sc = rand_packed(K, N, GROUP, "cpu", gen)
        lins.append((torch.randn((1, K), generator=gen).half(), qw, qz, sc))
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        for x, qw, qz, sc in lins:
            awq_oracle.torch_linear_gemm(x, qw, qz, sc, GROUP)
        dt = time.perf_counter() - t0
        if it:
            times.append(dt)
    layer_s = statistics.median(times)
    return {"value": 1.0 / (layer_s * layers_total), "unit": "tok/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"4 Linears of 1 of {layers_total} layers, median of 3 ({layer_s:.2f} s/layer); "
                                       "torch-CPU restatement of dequantize_gemm + fp16 matmul"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--layers", type=int, default=LAYERS, help="debug only; the metric is quoted at 32")
    ap.add_argument("--layout", choices=["gemm", "gemv", "gemvfast"], default="gemm",
                    help="checkpoint format of the Linears: WQLinear_GEMM (default) / _GEMV / _GEMVFast buffers")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-whole-model", action="store_true", help="skip the secondary whole-decoder figure")
    a = ap.parse_args()

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
    if world > 1 and a.layout != "gemm":
        raise SystemExit("tensor-parallel shards are implemented for the GEMM layout")
    model, shapes = build_model(dev, rank, world, a.layers, layout=a.layout)
    nl = sum(len(l) for l in model)
    outs = [None] * nl
    bytes_step = sum(algorithmic_bytes(l["K"], l["N"], 1, GROUP) for layer in model for l in layer)
    if a.layout == "gemvfast":  # zeros are stored as fp16 -(s*z): 2 bytes instead of half a byte per (group, column)
        bytes_step += sum((l["K"] // GROUP) * l["N"] * 3 // 2 for layer in model for l in layer)

    stream = torch.cuda.Stream(device=dev)
    graph, used_graph = None, False
    with torch.cuda.stream(stream):
        for _ in range(max(a.warmup, 3) if a.no_graph else 3):
            run_step(model, outs, ops, dist)
        stream.synchronize()
        if not a.no_graph:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    run_step(model, outs, ops, dist)
                used_graph = True
            except Exception as e:  # e.g. collective not capturable: fall back to eager launches
                if rank == 0:
                    print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
                graph = None
        step = graph.replay if graph is not None else (lambda: run_step(model, outs, ops, dist))
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
        out = {
            "metric": "decode tok/s @bs=1 (int4 linears), 7B AWQ-int4 g128", "value": tok_s, "unit": "tok/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f16 (int4 weights, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "Llama-2-7B-shape AWQ int4 g128, GEMV bs=1 decode: 32 layers x "
                                   "{qkv 4096->12288, o 4096->4096, gate+up 4096->22016, down 11008->4096}",
                       "layers": a.layers, "launches_per_step": launches, "hipgraph": used_graph, "layout": a.layout,
                       "parallelism": f"tp{world}" if world > 1 else "single",
                       "algorithmic_bytes_per_step_all_ranks": bytes_all, "kernel": ops.last_kernel()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # HBM bytes per launch from the TCC fabric counters (own --pmc FETCH_SIZE pass,
                         # x2 x 1024 per MI355X_MICROARCH.md; profiles/r01_pmc_fetch_size.txt), N=1 shapes
                         "traffic": 26.980e6 if world == 1 and a.layers == LAYERS and a.layout == "gemm" else None,
                         "bytes_per_launch": bytes_step / launches, "avg_launch_us": ms_step * 1e3 / launches,
                         "kernel": "awq_gemv_mfma_kernel (4 shapes per layer: qkv, o, gate+up, down)",
                         "note": "achieved = algorithmic bytes per launch / average launch duration; duration = "
                                 "HIP-event-timed replay of the captured stream / launches, i.e. it contains the "
                                 "dispatch gap exactly as rocprofv3's back-to-back kernel durations do "
                                 "(profiles/r01_bench_kernel_trace_stats.txt)"},
        }
        if world == 1 and not a.no_whole_model and a.layers == LAYERS and a.layout == "gemm":
            # secondary figure (never `value`): the same shape as a WHOLE decoder -- fused blocks with
            # norms, RoPE + KV cache, attention, lm_head -- which is what the reference's README
            # tables measure (Vicuna-7B GEMV, bs=1, ctx/gen 64: 198.848 tok/s on an RTX 4090)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_decode_model

                del model, outs, graph
                torch.cuda.empty_cache()
                wm = bench_decode_model.run(contexts=(64, 2048), steps=48, dev=dev, verbose=False)
                out["whole_model"] = {"unit": "tok/s", "context_64": 1000.0 / wm[64], "context_2048": 1000.0 / wm[2048],
                                      "what": "synthetic 7B-shape fused decoder (32 blocks + lm_head), one hipGraph per token",
                                      "published_reference": {"value": 198.848, "context": 64, "hardware": "RTX 4090",
                                                              "source": "README.md:207 (BASELINE.md)"},
                                      "vs_published_ctx64": (1000.0 / wm[64]) / 198.848}
            except Exception as e:  # the headline line must not depend on this leg
                out["whole_model"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.layers)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
