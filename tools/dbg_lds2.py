import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from autoawq_amd import ops
from tools.sweep_gemv_rows import rand_nk, gen, dev
K, N = 4096, 22016
qw, qz, sc = rand_nk(K, N, 128)
for M in (5, 8, 3):
    x = torch.randn((M, K), device=dev, generator=gen).half()
    for fl in (dict(), dict(splitk=2), dict(unit=2, splitk=1)):
        f = ops.gemm_flags(kernel=3, **fl)
        try:
            y1 = ops.gemv_forward(x, qw, sc, qz, 128, flags=f)
        except Exception as e:
            print(M, fl, str(e)[:60]); continue
        nd = 0
        for it in range(20):
            y = ops.gemv_forward(x, qw, sc, qz, 128, flags=f)
            nd += int((y != y1).sum())
        y2 = ops.gemv_forward(2 * x, qw, sc, qz, 128, flags=f)
        d = (y2.float() != 2 * y1.float())
        idx = d.nonzero()
        print(f"M{M} {fl}: repeat diffs {nd}; f(2x) != 2f(x) at {int(d.sum())} places", idx[:8].tolist(),
              [(float(y2[i, j]), 2 * float(y1[i, j])) for i, j in idx[:4].tolist()])
