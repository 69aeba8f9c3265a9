import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_moe
from bench_moe import E, H, I, g, topk, Stack
from autoawq_amd.modules.fused import moe
from autoawq_amd import ops
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(1); lim = 0x7FFFFFFF
def experts(K, N):
    s = Stack()
    s.qweight = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, device=dev, generator=gen)
    s.qzeros = torch.randint(-lim - 1, lim, (E, K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
    s.scales = (torch.rand((E, K // g, N), device=dev, generator=gen) * 0.004 + 0.001).half()
    return s
w1, w2 = experts(H, 2 * I), experts(I, H)
for T in (8, 16, 32, 64, 128, 256):
    x = torch.randn((T, H), device=dev, generator=gen).half(); logits = torch.randn((T, E), device=dev, generator=gen)
    res = {}
    for name, thr in (("blocks", 1 << 30), ("per_expert", 1)):
        moe.PREFILL_MIN_PAIRS = thr
        y = moe.apply_moe_weights(w1, w2, x, logits, topk, True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): y = moe.apply_moe_weights(w1, w2, x, logits, topk, True)
        torch.cuda.synchronize()
        res[name] = ((time.perf_counter() - t0) / 3 * 1e6, y)
    try:
        ops.check_workspaces(); ws = "ok"
    except Exception as e:
        ws = str(e)[:60]
    d = float((res["blocks"][1].float() - res["per_expert"][1].float()).abs().max() / res["per_expert"][1].float().abs().max())
    print(f"T={T}: blocks {res['blocks'][0]:.0f} us, per-expert {res['per_expert'][0]:.0f} us, rel diff {d:.2e}, nan {bool(torch.isnan(res['blocks'][1]).any())}, ws {ws}", flush=True)
