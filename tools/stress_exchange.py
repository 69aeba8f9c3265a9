import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
lim = 0x7FFFFFFF
def case(K, N, g, M, seed):
    gen = torch.Generator().manual_seed(seed)
    qw = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, generator=gen)
    qz = torch.randint(-lim - 1, lim, (K // g, N // 8), dtype=torch.int32, generator=gen)
    sc = (torch.rand((K // g, N), generator=gen) * 0.02 + 0.005).half()
    x = torch.randn((M, K), generator=gen).half()
    return qw.cuda(), qz.cuda(), sc.cuda(), x.cuda()
cases = [(2048, 2048, 2048, 300), (4096, 512, 128, 128), (512, 256, 128, 17), (4096, 4096, 128, 1), (1024, 8192, 128, 8), (4096, 11008, 128, 1), (256, 136 // 8 * 8, 32, 64)]
data = [case(*c, seed=i) for i, c in enumerate(cases)]
def run(i, fl=0):
    qw, qz, sc, x = data[i]
    return ops.gemm_forward(x, qw, sc, qz, flags=fl)
firsts = {}
bad = 0
import random
random.seed(0)
variants = [(0, 0), (0, ops.gemm_flags(ops.KERNEL_TILED, nlog=1)), (0, ops.gemm_flags(ops.KERNEL_TILED, nlog=2)), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0),
            (1, ops.gemm_flags(ops.KERNEL_TILED, nlog=1, splitk=3)), (3, ops.gemm_flags(ops.KERNEL_MFMA_GEMV, nlog=4, splitk=8, waves=4, unit=2))]
for it in range(6000):
    v = random.choice(variants)
    y = run(*v)
    if v not in firsts:
        firsts[v] = y.clone()
    elif not torch.equal(y, firsts[v]):
        d = (y.float() - firsts[v].float()).abs()
        idx = torch.nonzero(d > 0)
        bad += 1
        print(f"MISMATCH variant {v} (case {cases[v[0]]}) iter {it}: {idx.shape[0]} elems, max {float(d.max()):.4f}, nan {int(torch.isnan(y).sum())}, rows {sorted(set(idx[:,0].tolist()))[:8]} cols {sorted(set(idx[:,1].tolist()))[:20]}", flush=True)
        if bad > 15: break
torch.cuda.synchronize()
print("done", bad, "clean", ops.workspace_is_clean(torch.device("cuda")))
