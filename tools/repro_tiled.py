"""Replays the body of tests/test_gpu_parity.py::test_tiled_gemm_vs_oracle[1-300-2048-2048-2048]
in a loop (fresh device temporaries per call, workspace scans in between) and reports where a
result differs from the first one."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
lim = 0x7FFFFFFF
K, N, g, M = 2048, 2048, 2048, 300
bn = int(os.environ.get("BN", "1"))
SCAN = int(os.environ.get("SCAN", "1"))
TEMPS = int(os.environ.get("TEMPS", "1"))
iters = int(os.environ.get("ITERS", "300"))
gen = torch.Generator().manual_seed(5)
qw = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, generator=gen)
qz = torch.randint(-lim - 1, lim, (K // g, N // 8), dtype=torch.int32, generator=gen)
s = (torch.rand((K // g, N), generator=gen) * 0.02 + 0.005).half()
x = torch.randn((M, K), generator=gen).half()
bias = torch.randn((N,), generator=gen).half()
dev = (qw.cuda(), s.cuda(), qz.cuda(), x.cuda(), bias.cuda())
def up(t, i):
    return t.cuda() if TEMPS else dev[i]
if int(os.environ.get("BIGWS", "1")):
    ops.workspace(torch.device("cuda:0"), 16384 + (64 << 20))
ref = {}
bad = 0
def check(tag, y):
    global bad
    if tag not in ref:
        ref[tag] = y.clone()
        return
    if not torch.equal(y, ref[tag]):
        d = (y.float() - ref[tag].float()).abs()
        idx = torch.nonzero(d > 0)
        rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
        bad += 1
        print(f"MISMATCH {tag} iter {it}: {idx.shape[0]} elems max {float(d.max()):.4f} rows {rows[:12]} cols {cols[:40]}", flush=True)
        r0, c0 = rows[0], cols[0]
        print("   got ", y[r0, c0:c0 + 8].tolist(), "\n   want", ref[tag][r0, c0:c0 + 8].tolist(), flush=True)
for it in range(iters):
    for sk in (0, 1, 3):
        y = ops.gemm_forward(up(x, 3), up(qw, 0), up(s, 1), up(qz, 2), up(bias, 4), flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn, splitk=sk))
        check(f"s{sk}", y)
        if SCAN > 1: y.cpu()
    if SCAN: assert ops.workspace_is_clean(y.device)
    check("s3", ops.gemm_forward(up(x, 3), up(qw, 0), up(s, 1), up(qz, 2), up(bias, 4), flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn, splitk=3)))
    W = ops.dequantize_weights(up(qw, 0), up(s, 1), up(qz, 2))
    e = torch.zeros((M, K), dtype=torch.float16, device="cuda")
    ks = (torch.arange(M, device="cuda") * 7 + 3) % K
    e[torch.arange(M, device="cuda"), ks] = 1.0
    out = ops.gemm_forward(e, up(qw, 0), up(s, 1), up(qz, 2), flags=ops.gemm_flags(ops.KERNEL_TILED, nlog=bn))
    check("onehot", out)
    if not torch.equal(out, W[ks]):
        print("onehot != W[ks] at iter", it, flush=True)
    if bad > 10: break
torch.cuda.synchronize()
print("done bad", bad, "clean", ops.workspace_is_clean(torch.device("cuda")))
