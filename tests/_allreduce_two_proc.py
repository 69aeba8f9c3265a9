"""Two processes on ONE GPU all-reduce through CUDA-IPC mapped buffers (run by tests/test_gpu_allreduce_ipc.py)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from autoawq_amd.comm import OneShotAllReduce

    ar = OneShotAllReduce.from_process_group(max_halfs=8192)
    gen = torch.Generator(device="cuda").manual_seed(100 + rank)
    ok = True
    for it in range(20):
        x = torch.randn((8192,), device="cuda", generator=gen).half()
        mine = x.clone()
        gathered = [torch.empty_like(x).cpu() for _ in range(world)]
        dist.all_gather(gathered, mine.cpu())
        want = torch.zeros((8192,), dtype=torch.float32)
        for t in gathered:
            want += t.float()
        ar(x)
        torch.cuda.synchronize()
        ok = ok and torch.equal(x.cpu(), want.half())
        dist.barrier()
    # inside a hipGraph
    x = torch.ones((4096,), device="cuda", dtype=torch.float16) * (rank + 1)
    y = torch.empty_like(x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ar(x, y)
        s.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ar(x, y)
        for _ in range(10):
            dist.barrier()
            g.replay()
            s.synchronize()
            ok = ok and bool((y == sum(range(1, world + 1))).all())
    done, err = ar.status()
    print(f"rank {rank}: ok={ok} epochs={done} err={err}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok and err == 0 else 1)


if __name__ == "__main__":
    main()
