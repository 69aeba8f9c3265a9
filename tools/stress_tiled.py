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
    b = torch.randn((N,), generator=gen).half()
    return qw.cuda(), qz.cuda(), sc.cuda(), x.cuda(), b.cuda()
bad = 0
for (K, N, g, M) in [(512, 256, 128, 17), (1024, 384, 64, 64), (4096, 512, 128, 100), (512, 256, 128, 300)]:
    qw, qz, sc, x, b = case(K, N, g, M, K + N + M)
    ref = ops.gemm_forward(x, qw, sc, qz, b, flags=ops.gemm_flags(ops.KERNEL_NAIVE)).float()
    for bn in (1, 2):
        for sk in (0, 1, 3):
            fl = ops.gemm_flags(ops.KERNEL_TILED, nlog=bn, splitk=sk)
            first = ops.gemm_forward(x, qw, sc, qz, b, flags=fl)
            for it in range(300):
                y = ops.gemm_forward(x, qw, sc, qz, b, flags=fl)
                if not torch.equal(y, first):
                    d = (y.float() - first.float()).abs()
                    idx = torch.nonzero(d > 0)
                    bad += 1
                    print(f"MISMATCH K{K} N{N} M{M} bn{bn} sk{sk} iter {it}: {idx.shape[0]} elems, max {float(d.max()):.4f}, first idx {idx[:4].tolist()}, nan {int(torch.isnan(y).sum())}")
                    if bad > 20: sys.exit(1)
            err = float((first.float() - ref).abs().max() / ref.abs().max())
            print(f"K{K} N{N} M{M} bn{bn} sk{sk}: rel err vs naive {err:.2e}; clean {ops.workspace_is_clean(x.device)}")
# also interleave GEMV (M<=16) and tiled calls sharing the workspace
qw, qz, sc, x, b = case(4096, 4096, 128, 40, 5)
f1 = ops.gemm_forward(x[:1], qw, sc, qz); f2 = ops.gemm_forward(x, qw, sc, qz)
for it in range(300):
    a = ops.gemm_forward(x[:1], qw, sc, qz); c = ops.gemm_forward(x, qw, sc, qz)
    if not (torch.equal(a, f1) and torch.equal(c, f2)):
        bad += 1; print("MISMATCH interleaved", it)
print("done, mismatches:", bad)
