"""Does a freshly hipMalloc'ed workspace (torch.cuda.empty_cache() first) misbehave on first use?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
lim = 0x7FFFFFFF
K, N, g, M = 2048, 2048, 2048, 300
FRESH = int(os.environ.get("FRESH", "1"))
gen = torch.Generator().manual_seed(5)
qw = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, generator=gen).cuda()
qz = torch.randint(-lim - 1, lim, (K // g, N // 8), dtype=torch.int32, generator=gen).cuda()
s = (torch.rand((K // g, N), generator=gen) * 0.02 + 0.005).half().cuda()
x = torch.randn((M, K), generator=gen).half().cuda()
fl = ops.gemm_flags(ops.KERNEL_TILED, nlog=1)
ref = ops.gemm_forward(x, qw, s, qz, flags=fl).clone()
bad = 0
for it in range(int(os.environ.get("ITERS", "300"))):
    ops.release_workspaces()
    if FRESH:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    ys = [ops.gemm_forward(x, qw, s, qz, flags=fl) for _ in range(3)]
    clean = ops.workspace_is_clean(x.device)
    for j, y in enumerate(ys):
        if not torch.equal(y, ref):
            d = (y.float() - ref.float()).abs()
            idx = torch.nonzero(d > 0)
            rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
            bad += 1
            print(f"MISMATCH iter {it} call {j}: {idx.shape[0]} elems max {float(d.max()):.4f} rows {rows[:12]} cols {cols[:40]} clean {clean}", flush=True)
    if bad > 10: break
print("done bad", bad)
