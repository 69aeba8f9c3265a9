"""EXPERIMENTAL (round 4, never run on a GPU yet): builds prefill_attn.hip into tools/bin/libprefill_attn.so, checks it on an
MI355X against an fp32 torch restatement of the reference's prefill attention (causal, GQA, optional ALiBi / soft cap, chunked
prefill), and times it beside torch's scaled_dot_product_attention (what autoawq_amd/modules/fused/attn.py calls today).

    gpurun --timeout 900 -- 'python tools/experimental/prefill_attention/probe.py > gpurun_out/prefill_attn.txt 2>&1'

Exit status 0 = every case within tolerance.  Nothing under autoawq_amd/ imports this file or the library it builds.
`emulate.py` (CPU, same directory) checks the kernel's index algebra and LDS layouts without a GPU."""
import ctypes
import math
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
BIN = os.path.join(ROOT, "tools", "bin")
# build variants, all checked and timed side by side (the kernel's #ifdef switches)
VARIANTS = {
    "compiler waits": [],
    "asm loads": ["-DAWQ_PATTN_ASM_LOADS"],
    "mfma rowsum": ["-DAWQ_PATTN_MFMA_ROWSUM"],
    "asm loads + mfma rowsum": ["-DAWQ_PATTN_ASM_LOADS", "-DAWQ_PATTN_MFMA_ROWSUM"],
    "asm loads + mfma rowsum + no-nans": ["-DAWQ_PATTN_ASM_LOADS", "-DAWQ_PATTN_MFMA_ROWSUM", "-fno-honor-nans"],
}


def lib_path(name):
    return os.path.join(BIN, "libprefill_attn_" + "".join(c if c.isalnum() else "_" for c in name) + ".so")


def build():
    """One library per variant under tools/bin/ (git-ignored, travels with a gpurun snapshot: build HERE first -- `--build-only` --
    and the GPU box spends no time compiling).  A stamp file holds the hash of source + flags: file times do not survive the copy."""
    import hashlib

    os.makedirs(BIN, exist_ok=True)
    src = os.path.join(HERE, "prefill_attn.hip")
    for name, extra in VARIANTS.items():
        out = lib_path(name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize",
               "-Wno-unused-function", "-Wno-inline-asm", "-Wno-unused-variable", "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "autoawq_amd", "csrc"), "-shared", "-o", out, src] + extra
        stamp = hashlib.sha1(open(src, "rb").read() + " ".join(cmd[1:]).replace(ROOT, "").encode()).hexdigest()
        if os.path.exists(out) and os.path.exists(out + ".stamp") and open(out + ".stamp").read() == stamp:
            continue
        subprocess.check_call(cmd)
        open(out + ".stamp", "w").write(stamp)


def reference(q, kc, vc, start, scale, softcap, slopes):
    """q [B, S, Hq, D], caches [B, Tmax, Hkv, D] -> [B, S, Hq, D] in fp32 (the arithmetic of attn.py:_attend_torch)"""
    B, S, Hq, D = q.shape
    end = start + S
    Hkv = kc.shape[2]
    k = kc[:B, :end].transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1).float()
    v = vc[:B, :end].transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1).float()
    s = torch.matmul(q.transpose(1, 2).float(), k.transpose(-1, -2)) * scale
    if softcap > 0:
        s = softcap * torch.tanh(s / softcap)
    qpos = start + torch.arange(S, device=q.device).view(-1, 1)
    kpos = torch.arange(end, device=q.device).view(1, -1)
    if slopes is not None:
        s = s + slopes.view(1, -1, 1, 1) * (kpos - qpos).float()
    s = s.masked_fill(kpos > qpos, float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2)


def main():
    if "--build-only" in sys.argv:
        build()
        print("built", ", ".join(os.path.basename(lib_path(n)) for n in VARIANTS))
        return 0
    build()
    fns = {}
    for name in VARIANTS:
        f = ctypes.CDLL(lib_path(name)).awq_exp_prefill_attention
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        fns[name] = f
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream

    def run(q, kc, vc, start, scale, softcap=0.0, slopes=None, variant="compiler waits"):
        B, S, Hq, D = q.shape
        out = torch.full_like(q, float("nan"))
        rc = fns[variant](q.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), B, S, Hq, kc.shape[2], D, kc.shape[1], start, scale, softcap,
                slopes.data_ptr() if slopes is not None else None, stream)
        return rc, out

    bad = 0
    gen = torch.Generator(device="cpu").manual_seed(0)
    cases = [  # B, S, start, Hq, Hkv, softcap, alibi
        (1, 128, 0, 4, 4, 0.0, False), (1, 1, 0, 2, 1, 0.0, False), (2, 200, 0, 8, 2, 0.0, False), (1, 70, 100, 4, 1, 0.0, False),
        (3, 333, 17, 4, 4, 0.0, False), (1, 2048, 0, 32, 32, 0.0, False), (1, 1000, 1048, 16, 2, 0.0, False),
        (1, 130, 60, 4, 2, 30.0, False), (2, 96, 33, 8, 8, 0.0, True), (1, 257, 0, 4, 4, 50.0, True), (1, 4096, 0, 8, 1, 0.0, False),
    ]
    for B, S, start, Hq, Hkv, softcap, alibi in cases:
        D, Tmax = 128, start + S + 37
        q = torch.randn((B, S, Hq, D), generator=gen).half().to(dev)
        kc = torch.randn((B + 1, Tmax, Hkv, D), generator=gen).half().to(dev)  # one spare batch entry, like a larger cache
        vc = torch.randn((B + 1, Tmax, Hkv, D), generator=gen).half().to(dev)
        kc[:, start + S:] = float("nan")  # rows past the context must never reach the result
        vc[:, start + S:] = float("nan")
        slopes = (2.0 ** (-8.0 * torch.arange(1, Hq + 1) / Hq)).float().to(dev) if alibi else None
        scale = D ** -0.5
        ref = reference(q, kc, vc, start, scale, softcap, slopes)
        for variant in fns:
            rc, out = run(q, kc, vc, start, scale, softcap, slopes, variant)
            if rc != 0:
                bad += 1
                print(f"B={B} S={S} start={start} Hq={Hq} Hkv={Hkv} [{variant}]: rc={rc}")
                continue
            err = (out.float() - ref).abs().max().item()
            # fp16 probabilities (rel 2^-11 each) under an fp32 sum + one fp16 rounding of the result (|o| <~ 4): 4e-3 absolute
            ok = bool(torch.isfinite(out).all()) and err < 4e-3
            bad += not ok
            print(f"B={B} S={S} start={start} Hq={Hq} Hkv={Hkv} cap={softcap} alibi={alibi} [{variant}]: max err {err:.3g} {'ok' if ok else 'MISMATCH'}")
            rc, out2 = run(q, kc, vc, start, scale, softcap, slopes, variant)
            if not torch.equal(out, out2):
                bad += 1
                print("   NOT reproducible run to run")

    # timing: Llama-2-7B prefill (MHA 32 x 128) and a 70B-style GQA shape, beside the vendor path
    import torch.nn.functional as F
    for B, S, Hq, Hkv in [(1, 2048, 32, 32), (8, 2048, 32, 32), (1, 8192, 32, 32), (4, 2048, 64, 8)]:
        q = torch.randn((B, S, Hq, 128), dtype=torch.float16, device=dev)
        kc = torch.randn((B, S, Hkv, 128), dtype=torch.float16, device=dev)
        vc = torch.randn((B, S, Hkv, 128), dtype=torch.float16, device=dev)
        flops = 4.0 * B * Hq * S * S * 128 / 2  # causal: half of the S x S scores

        def vendor():
            k = kc.transpose(1, 2)
            v = vc.transpose(1, 2)
            if Hq != Hkv:
                k = k.repeat_interleave(Hq // Hkv, dim=1)
                v = v.repeat_interleave(Hq // Hkv, dim=1)
            return F.scaled_dot_product_attention(q.transpose(1, 2), k, v, is_causal=True)

        calls = [(f"prefill_attn [{v}]", (lambda v=v: run(q, kc, vc, 0, 128 ** -0.5, variant=v))) for v in VARIANTS] + [("vendor sdpa", vendor)]
        for name, call in calls:
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                call()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            print(f"B={B} S={S} Hq={Hq} Hkv={Hkv} {name}: {ms:.3f} ms, {flops / ms / 1e9:.0f} TFLOP/s (causal flops)")
    print("FAILED" if bad else "ALL OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
