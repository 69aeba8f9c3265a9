import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from autoawq_amd import ops
from tools.sweep_gemv_rows import rand_nk, graph_us
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(0)
for K, N in ((4096, 11008), (4096, 4096), (4096, 12288), (8192, 7168), (8192, 1280)):
    nsets = max(4, min(96, (640 << 20) // (K * N // 2)))
    sets = [rand_nk(K, N, 128) for _ in range(nsets)]
    for M in (1, 2):
        x = torch.randn((M, K), device=dev, generator=gen).half()
        def run():
            for qw, qz, sc in sets:
                ops.gemv_forward(x, qw, sc, qz, 128)
        print(K, N, "M", M, ops.last_kernel(), round(graph_us(run, nsets), 2), "us", flush=True)
    del sets; torch.cuda.empty_cache()
