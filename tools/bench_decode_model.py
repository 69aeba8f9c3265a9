#!/usr/bin/env python3
"""Whole-model decode on synthetic Llama-2-7B-shape weights: 32 fused blocks
(norm -> fused qkv int4 GEMV -> RoPE + KV append -> single-query attention -> o_proj -> add + norm ->
gate|up int4 GEMV -> down with silu*mul applied while staging), final norm, fp16 lm_head; ONE hipGraph
per decode step, positions in device tensors.  Prints tokens/s at several context lengths (the number
comparable with the reference's README decode tables, which are whole-model numbers)."""
import argparse, os, sys
import torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

H, I, V, HEADS, G = 4096, 11008, 32000, 32, 128


def run(layers=32, contexts=(64, 512, 2048), batch=1, steps=64, dev=None, verbose=True, check=True, layout="gemm",
        hidden=H, inter=I, vocab=V, heads=HEADS):
    """Returns {context: ms_per_token}.  check: before timing, the logits of the five-launch stream path (norms folded
    into the projections, fused RoPE + append + attention) must agree with the plain module path (separate norm, RoPE /
    append, attention, o_proj, MLP launches) on the same weights and cache."""
    from autoawq_amd.fuser import FusedCausalLM
    from autoawq_amd.modules.fused.block import LlamaLikeBlock
    from autoawq_amd.modules.fused.mlp import QuantFusedMLP
    from autoawq_amd.modules.fused.model import LlamaLikeModel
    from autoawq_amd.modules.fused.norm import FasterTransformerRMSNorm
    from autoawq_amd.modules.linear import WQLinear_GEMM

    dev = dev or torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    lim = 0x7FFFFFFF
    H, I, V, HEADS = hidden, inter, vocab, heads  # (tests run a small stack through the same code)

    def rand_linear(K, N):
        if layout == "gemv":  # the reference's WQLinear_GEMV checkpoint format: the decode layout since round 3
            from autoawq_amd.modules.linear import WQLinear_GEMV
            from autoawq_amd.utils.packing import calculate_zeros_width

            zw = calculate_zeros_width(K, G)
            m = WQLinear_GEMV(4, G, K, N, False, dev)
            m.qweight = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, device=dev, generator=gen)
            m.qzeros = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, device=dev, generator=gen)
            m.scales = (torch.rand((N, zw * 8), device=dev, generator=gen) * 0.004 + 0.001).half()
            return m
        m = WQLinear_GEMM(4, G, K, N, False, dev)
        m.qweight = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        m.qzeros = torch.randint(-lim - 1, lim, (K // G, N // 8), dtype=torch.int32, device=dev, generator=gen)
        m.scales = (torch.rand((K // G, N), device=dev, generator=gen) * 0.004 + 0.001).half()
        return m

    max_ctx = max(contexts) + steps + 8
    ones = lambda: torch.ones(H, dtype=torch.float16, device=dev)
    blocks = []
    for _ in range(layers):
        mlp = QuantFusedMLP(rand_linear(H, I), rand_linear(I, H), rand_linear(H, I))
        blocks.append(LlamaLikeBlock(H, HEADS, HEADS, rand_linear(H, 3 * H), rand_linear(H, H), mlp,
                                     FasterTransformerRMSNorm(ones(), 1e-5), FasterTransformerRMSNorm(ones(), 1e-5), dev, max_ctx))
    emb = nn.Embedding(V, H).half().to(dev)
    head = nn.Linear(H, V, bias=False).half().to(dev)
    lm = FusedCausalLM(LlamaLikeModel(V, blocks, emb, FasterTransformerRMSNorm(ones(), 1e-5)), head)
    from autoawq_amd.modules.fused.decode import GraphedDecoder

    dec = GraphedDecoder(lm, batch=batch)  # one hipGraph per length bucket (256 / 1024 / 4096 ... rows); positions on the device
    tok = torch.randint(0, V, (batch, 1), device=dev, generator=gen)
    out = {}
    for blk in blocks:  # pretend a context is cached: random K / V rows
        blk.attn.cache.k.normal_(generator=gen)
        blk.attn.cache.v.normal_(generator=gen)
    if check:
        dec.seek(40)
        fused = dec.step(tok).float().clone()          # captured + replayed: the path that is timed below
        torch.cuda.synchronize()
        assert dec.position == 41 and int(dec.pos.item()) == 41 and int(dec.len.item()) == 42
        for blk in blocks:
            blk.attn.start_pos = 40
            blk.attn.use_device_positions(None, None)
        lm.model.last_forward_num_tokens = 40
        saved = (LlamaLikeBlock.FOLD_NORMS_INTO_PROJECTIONS, type(blocks[0].attn).FUSE_ROPE_INTO_ATTENTION)
        LlamaLikeBlock.FOLD_NORMS_INTO_PROJECTIONS = False
        type(blocks[0].attn).FUSE_ROPE_INTO_ATTENTION = False
        try:
            plain = lm(tok).float()
        finally:
            LlamaLikeBlock.FOLD_NORMS_INTO_PROJECTIONS, type(blocks[0].attn).FUSE_ROPE_INTO_ATTENTION = saved
        for blk in blocks:
            blk.attn.use_device_positions(dec.pos, dec.len)
        rel = float((fused - plain).abs().max() / plain.abs().max())
        assert rel < 3e-2, f"fused decode path differs from the plain module path by {rel}"
        if layout == "gemv" and batch == 1:  # the five-launch path on the row-streaming kernel is the one that ran
            assert blocks[0]._can_fold_gemv(torch.empty((1, 1, H), dtype=torch.float16, device=dev))
    for ctx in contexts:
        dec.seek(ctx)
        dec.step(tok)                                  # captures this bucket's graph on first use
        s = dec.stream
        with torch.cuda.stream(s):
            for _ in range(3):
                dec.replay()
            s.synchronize()
            dec.seek(ctx)
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(steps):
                dec.replay()
            e1.record(s)
            e1.synchronize()
        ms = e0.elapsed_time(e1) / steps
        out[ctx] = ms
        if verbose:
            name = "7B" if H == 4096 else "hidden-%d" % H
            print(f"{name}-shape whole-model decode ({layout} layout), {layers} layers, batch {batch}, context {ctx}: {ms:.3f} ms/token = "
                  f"{batch * 1000.0 / ms:.1f} tok/s (one hipGraph replay per step, bucket <= {dec.bucket(ctx + 1)} rows)", flush=True)
    del dec
    del lm, blocks
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--contexts", default="64,512,2048")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--layout", choices=["gemm", "gemv"], default="gemm")
    a = ap.parse_args()
    run(a.layers, tuple(int(c) for c in a.contexts.split(",")), a.batch, a.steps, layout=a.layout)
