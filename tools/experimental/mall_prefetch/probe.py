"""EXPERIMENTAL (round 4, never run on a GPU yet).  Two questions about the headline decode step (128 launches per token at
Llama-2-7B shapes, GEMV layout, 3.37 GB of packed weights, one hipGraph):

 1. What do the same launches cost when their weights sit in the 256 MiB Infinity Cache?  (ONE layer's buffers, 105 MB, used by
    all 32 layers of the captured step, against 32 distinct layers.)  If that is not clearly faster, stop here.
 2. If it is: a second graph branch on which a throw-away reader (prefetch.hip) pulls layer l + 1's buffers while layer l's four
    launches run on the main branch -- the HBM stream then runs through the dispatch gaps, the launches read from the cache.
    Reported for several reader sizes (blocks), against the plain step.

    gpurun --timeout 600 -- 'python tools/experimental/mall_prefetch/probe.py > gpurun_out/mall_prefetch.txt 2>&1'
"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "bin", "libmall_prefetch.so")


def build():
    import hashlib

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = os.path.join(HERE, "prefetch.hip")
    stamp = hashlib.sha1(open(src, "rb").read()).hexdigest()
    if os.path.exists(OUT) and os.path.exists(OUT + ".stamp") and open(OUT + ".stamp").read() == stamp:
        return
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT, src])
    open(OUT + ".stamp", "w").write(stamp)


def main():
    build()
    if "--build-only" in sys.argv:
        print("built", OUT)
        return 0
    import bench
    from autoawq_amd import ops

    lib = ctypes.CDLL(OUT)
    lib.awq_exp_prefetch.restype = ctypes.c_int
    lib.awq_exp_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    layers = 32
    model, shapes = bench.build_model(dev, 0, 1, layers, layout="gemv")
    bytes_step = sum(bench.algorithmic_bytes(l["K"], l["N"], 1, bench.GROUP) for layer in model for l in layer)
    outs = [None] * (layers * 4)
    st = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)

    def report(name, us):
        print(f"{name}: {us:.1f} us/token, {1e6 / us:.0f} tok/s, {bytes_step / us / 1e3:.0f} GB/s algorithmic")

    us_plain = bench.graph_time(lambda: bench.run_step(model, outs, ops, None), st, reps=10, min_seconds=0.3)
    report("32 distinct layers (the headline)", us_plain)
    same = [model[0]] * layers
    us_warm = bench.graph_time(lambda: bench.run_step(same, outs, ops, None), st, reps=10, min_seconds=0.3)
    report("one layer's buffers 32 times (cache-resident)", us_warm)

    def bufs(layer):
        return [t for lin in layer for t in (lin["qw"], lin["qz"], lin["sc"])]

    for blocks in (64, 128, 256, 512, 1024):
        def step():
            main_stream = torch.cuda.current_stream()
            for i, layer in enumerate(model):
                if i + 1 < layers:
                    ev = torch.cuda.Event()
                    ev.record(main_stream)
                    side.wait_event(ev)
                    for t in bufs(model[i + 1]):
                        rc = lib.awq_exp_prefetch(t.data_ptr(), t.numel() * t.element_size(), blocks, sink.data_ptr(), side.cuda_stream)
                        assert rc == 0, rc
                bench.run_step([layer], outs, ops, None)
            done = torch.cuda.Event()
            done.record(side)
            main_stream.wait_event(done)

        us = bench.graph_time(step, st, reps=10, min_seconds=0.3)
        report(f"next layer pulled by a {blocks}-block reader on a second branch", us)
    return 0


if __name__ == "__main__":
    sys.exit(main())
