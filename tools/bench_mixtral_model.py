#!/usr/bin/env python3
"""Whole-model decode on synthetic Mixtral-8x7B-shape weights (32 blocks: fused qkv / o_proj in the GEMV layout, 8 experts of
4096 -> 2 x 14336 -> 4096 int4 g128, top-2; fp16 lm_head), ONE hipGraph replay per token (modules/fused/decode.py::GraphedDecoder):
tokens/s with the experts' GEMV-layout decode twins (what fuse_mixtral(decode_layout="auto") builds: round 6) and on the
checkpoint's GEMM-layout stacks alone.  The twins are REAL repacks of the random GEMM-layout stacks (build_decode_twins), and the
two paths' logits are compared before timing.

    gpurun -- python tools/bench_mixtral_model.py [--layers 32] [--batch 1,4] [--context 64]
"""
import argparse, os, sys
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, I, V, HEADS, KV, E, TOPK, G = 4096, 14336, 32000, 32, 8, 8, 2, 128


def run(layers=32, batches=(1, 4), context=64, steps=48, dev=None):
    from autoawq_amd.fuser import FusedCausalLM
    from autoawq_amd.modules.fused import moe as moe_mod
    from autoawq_amd.modules.fused.block import MixtralBlock
    from autoawq_amd.modules.fused.decode import GraphedDecoder
    from autoawq_amd.modules.fused.model import MixtralModel
    from autoawq_amd.modules.fused.norm import FasterTransformerRMSNorm
    from autoawq_amd.modules.linear import WQLinear_GEMM, WQLinear_GEMV
    from autoawq_amd.utils.packing import calculate_zeros_width

    dev = dev or torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    lim = 0x7FFFFFFF

    def gemv_linear(K, N):
        zw = calculate_zeros_width(K, G)
        m = WQLinear_GEMV(4, G, K, N, False, dev)
        m.qweight = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, device=dev, generator=gen)
        m.qzeros = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, device=dev, generator=gen)
        m.scales = (torch.rand((N, zw * 8), device=dev, generator=gen) * 0.004 + 0.001).half()
        return m

    def stack(K, N):  # what fuse_linears(..., operation=torch.stack) returns (awq/models/mixtral.py:130-158)
        m = WQLinear_GEMM(4, G, K, N, False, dev)
        m.qweight = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        m.qzeros = torch.randint(-lim - 1, lim, (E, K // G, N // 8), dtype=torch.int32, device=dev, generator=gen)
        m.scales = (torch.rand((E, K // G, N), device=dev, generator=gen) * 0.004 + 0.001).half()
        return m

    max_ctx = context + steps + 8
    ones = lambda: torch.ones(H, dtype=torch.float16, device=dev)
    blocks = []
    for _ in range(layers):
        gate = nn.Linear(H, E, bias=False).half().to(dev)
        ws, w2s = stack(H, 2 * I), stack(I, H)
        moe_mod.build_decode_twins(ws, w2s)
        blk = MixtralBlock(H, HEADS, KV, gemv_linear(H, (HEADS + 2 * KV) * 128), gemv_linear(H, H),
                           moe_mod.FusedSparseMoeBlock(TOPK, gate, ws, w2s), FasterTransformerRMSNorm(ones(), 1e-5),
                           FasterTransformerRMSNorm(ones(), 1e-5), dev, max_ctx, 1e6)
        blocks.append(blk)
    emb = nn.Embedding(V, H).half().to(dev)
    head = nn.Linear(H, V, bias=False).half().to(dev)
    lm = FusedCausalLM(MixtralModel(V, blocks, emb, FasterTransformerRMSNorm(ones(), 1e-5)), head)
    gb = torch.cuda.memory_allocated(dev) / 1e9
    print(f"Mixtral-8x7B-shape model, {layers} layers: {gb:.1f} GB resident (experts twice: GEMM-layout stacks + GEMV-layout decode twins)", flush=True)
    for blk in blocks:
        blk.attn.cache.k.normal_(generator=gen)
        blk.attn.cache.v.normal_(generator=gen)
    out = {}
    for batch in batches:
        res = {}
        for name, cap in (("twins", moe_mod.ROWS_MAX_PAIRS), ("gemm_layout", 0)):
            saved = moe_mod.ROWS_MAX_PAIRS
            moe_mod.ROWS_MAX_PAIRS = cap
            try:
                dec = GraphedDecoder(lm, batch=batch)
                tok = torch.randint(0, V, (batch, 1), device=dev, generator=torch.Generator(device=dev).manual_seed(7))
                dec.seek(context)
                logits = dec.step(tok).float().clone()
                s = dec.stream
                with torch.cuda.stream(s):
                    for _ in range(3):
                        dec.replay()
                    s.synchronize()
                    dec.seek(context)
                    s.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(steps):
                        dec.replay()
                    e1.record(s)
                    e1.synchronize()
                res[name] = (e0.elapsed_time(e1) / steps, logits)
                del dec
            finally:
                moe_mod.ROWS_MAX_PAIRS = saved
        rel = float((res["twins"][1] - res["gemm_layout"][1]).abs().max() / res["gemm_layout"][1].abs().max())
        assert rel < 3e-2, f"the two expert paths' logits differ by {rel}"
        mt, mg = res["twins"][0], res["gemm_layout"][0]
        out[batch] = (mt, mg)
        print(f"Mixtral-8x7B-shape whole-model decode, {layers} layers, batch {batch}, context {context}: decode twins {mt:.3f} ms/step = "
              f"{batch * 1000.0 / mt:.1f} tok/s; GEMM-layout stacks only {mg:.3f} ms/step = {batch * 1000.0 / mg:.1f} tok/s "
              f"(x {mg / mt:.2f}; logits of the two paths within {rel:.1e})", flush=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batch", default="1,4")
    ap.add_argument("--context", type=int, default=64)
    ap.add_argument("--steps", type=int, default=48)
    a = ap.parse_args()
    run(a.layers, tuple(int(b) for b in a.batch.split(",")), a.context, a.steps)
