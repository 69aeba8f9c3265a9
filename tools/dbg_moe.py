import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoawq_amd import ops
T, E, topk, K, N, g = 4, 8, 2, 256, 512, 128
gen = torch.Generator().manual_seed(T + E + K)
lim = 0x7FFFFFFF
qw = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, generator=gen)
qz = torch.randint(-lim - 1, lim, (E, K // g, N // 8), dtype=torch.int32, generator=gen)
sc = (torch.rand((E, K // g, N), generator=gen) * 0.02 + 0.005).half()
gen = torch.Generator().manual_seed(7)
x = torch.randn((T, K), generator=gen).half()
logits = torch.randn((T, E), generator=gen)
w, ids = ops.fused_topk(logits.cuda(), topk, True)
print("ids", ids.tolist())
for rows in (16, 8):
    s_ids, e_ids, npad = ops.moe_align_block_size(ids, rows, E)
    print(rows, "sorted", s_ids.tolist(), "experts", e_ids.tolist(), "npad", npad.tolist())
    y = ops.grouped_gemm_forward(x.cuda().view(T, 1, K), qw.cuda(), sc.cuda(), qz.cuda(), w, s_ids, e_ids, npad, False, block_rows=rows)
    torch.cuda.synchronize()
    for t in range(T):
        for j in range(topk):
            e = int(ids[t, j])
            ref = ops.gemm_forward(x[t:t+1].cuda(), qw[e].cuda(), sc[e].cuda(), qz[e].cuda()).float()
            err = float((y[t, j].float() - ref[0]).abs().max())
            # which expert would match?
            best = min(range(E), key=lambda ee: float((y[t, j].float() - ops.gemm_forward(x[t:t+1].cuda(), qw[ee].cuda(), sc[ee].cuda(), qz[ee].cuda()).float()[0]).abs().max()))
            print(f"  rows{rows} pair({t},{j}) expert {e}: err {err:.3f}  (closest expert {best})")
