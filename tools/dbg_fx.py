import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from autoawq_amd import ops
from test_gpu_parity import gemv_case
for K, N in ((4096, 22016), (8192, 8192), (4096, 12288)):
    g = 128
    qw, qz, sc, x4 = gemv_case(K, N, g, 1, seed=3 * K + N)
    x = x4[:1].cuda()
    rows = ops.gemm_flags(kernel=2)
    y = ops.gemv_forward(x, qw.cuda(), sc.cuda(), qz.cuda(), g, flags=rows)
    gu = torch.cat([y[:, 0::2], y[:, 1::2]], dim=1).contiguous()
    want = ops.silu_and_mul(gu)
    got = ops.gemv_forward_ex(x, qw.cuda(), sc.cuda(), qz.cuda(), g, silu_pairs=True)
    bad = (got.view(torch.int16) != want.view(torch.int16)).nonzero()
    print(K, N, "mismatches", bad.shape[0], "of", want.numel())
    for i in bad[:12, 1].tolist():
        print("  idx", i, "gate", float(y[0, 2 * i]), "up", float(y[0, 2 * i + 1]), "got", float(got[0, i]), "want", float(want[0, i]))
