import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from autoawq_amd import ops
import bench
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for K, N in [(4096, 11008), (11008, 4096), (4096, 4096)]:
    qw, qz, sc = bench.rand_packed(K, N, 128, dev, gen)
    W = ops.dequantize_weights(qw, sc, qz)
    for M in (512, 2048, 4096, 16384):
        x = torch.randn((M, K), device=dev, generator=gen).half()
        ref = torch.matmul(x[:1024], W).float()
        out = {}
        for nlog in (1, 2, 3):
            f = ops.gemm_flags(ops.KERNEL_REGB, nlog=nlog)
            y = ops.gemm_forward(x, qw, sc, qz, flags=f)
            rel = float((y[:1024].float() - ref).abs().max() / ref.abs().max())
            y2 = ops.gemm_forward(x, qw, sc, qz, flags=f)
            us = timeit(lambda: ops.gemm_forward(x, qw, sc, qz, flags=f))
            out[nlog] = (us, 2.0 * M * K * N / us / 1e6, rel, bool(torch.equal(y, y2)))
        us2 = timeit(lambda: torch.matmul(x, W))
        print(f"{K}x{N} M={M}: " + "  ".join(f"nlog{n}: {v[0]:.1f} us {v[1]:.0f} TF rel {v[2]:.1e} repro {v[3]}" for n, v in out.items()) + f"  vendor-gemm-only {us2:.1f} us {2.0*M*K*N/us2/1e6:.0f} TF", flush=True)
