"""MixtralBlock / MixtralModel / fuse_mixtral (reference: awq/modules/fused/block.py:6-55, model.py:20-58, awq/models/mixtral.py:97-187)
on a tiny Mixtral-style model in the per-expert form AWQ checkpoints use.  The reference's own Mixtral classes need the old
transformers Mixtral modules (not importable with the installed transformers 5.x: "parity unpinned"), so the checker is the
oracle's restatement of every piece composed on the CPU: RMSNorm, int4 Linears, RoPE, causal attention, `moe_forward`
(SURVEY.md Appendix A.5)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

H, I, E, TOPK, HEADS, KV, D, V, LAYERS = 256, 512, 4, 2, 2, 1, 128, 64, 2
EPS = 1e-5


class _Norm(nn.Module):
    def __init__(self, gen):
        super().__init__()
        self.weight = nn.Parameter((1.0 + 0.1 * torch.randn(H, generator=gen)).half())
        self.variance_epsilon = EPS


class _Expert(nn.Module):
    def __init__(self):
        super().__init__()
        self.w1, self.w2, self.w3 = nn.Linear(H, I, bias=False), nn.Linear(I, H, bias=False), nn.Linear(H, I, bias=False)


class _Moe(nn.Module):
    def __init__(self):
        super().__init__()
        self.gate = nn.Linear(H, E, bias=False)
        self.experts = nn.ModuleList([_Expert() for _ in range(E)])
        self.top_k = TOPK


class _Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.q_proj, self.k_proj = nn.Linear(H, HEADS * D, bias=False), nn.Linear(H, KV * D, bias=False)
        self.v_proj, self.o_proj = nn.Linear(H, KV * D, bias=False), nn.Linear(HEADS * D, H, bias=False)


class _Layer(nn.Module):
    def __init__(self, gen):
        super().__init__()
        self.self_attn, self.block_sparse_moe = _Attn(), _Moe()
        self.input_layernorm, self.post_attention_layernorm = _Norm(gen), _Norm(gen)


class _Inner(nn.Module):
    def __init__(self, gen):
        super().__init__()
        self.embed_tokens = nn.Embedding(V, H)
        self.layers = nn.ModuleList([_Layer(gen) for _ in range(LAYERS)])
        self.norm = _Norm(gen)


class TinyMixtral(nn.Module):
    def __init__(self):
        super().__init__()
        gen = torch.Generator().manual_seed(5)
        torch.manual_seed(5)
        self.model = _Inner(gen)
        self.lm_head = nn.Linear(H, V, bias=False)
        self.config = SimpleNamespace(hidden_size=H, num_attention_heads=HEADS, num_key_value_heads=KV, vocab_size=V, rope_theta=10000.0)
        with torch.no_grad():
            for layer in self.model.layers:  # well separated router logits: the top-2 choice must not hinge on fp16 noise
                layer.block_sparse_moe.gate.weight.mul_(6.0)


def build():
    from autoawq_amd.checkpoint import AwqConfig, quantize_linears_rtn

    m = TinyMixtral().half().eval()
    done = quantize_linears_rtn(m, AwqConfig(version="gemm", modules_to_not_convert=["gate"]))
    assert len(done) == LAYERS * (4 + 3 * E)
    return m


def test_fuse_mixtral_structure_cpu():
    """Host logic only: the stacks the fuser builds (mixtral.py:130-158) and the declared layouts."""
    from autoawq_amd.fuser import fuse_mixtral
    from autoawq_amd.modules.fused.block import MixtralBlock
    from autoawq_amd.modules.fused.model import MixtralModel
    from autoawq_amd.modules.fused.moe import FusedSparseMoeBlock
    from autoawq_amd.modules.linear import WQLinear_GEMM, WQLinear_GEMV

    m = build()
    w1_0 = m.model.layers[0].block_sparse_moe.experts[1].w1.qweight.clone()
    w3_0 = m.model.layers[0].block_sparse_moe.experts[1].w3.qweight.clone()
    lm = fuse_mixtral(m, max_seq_len=32)
    assert isinstance(lm.model, MixtralModel) and len(lm.model.blocks) == LAYERS and lm.checkpoint_layout == "gemm"
    blk = lm.model.blocks[0]
    assert isinstance(blk, MixtralBlock) and isinstance(blk.moe, FusedSparseMoeBlock) and blk.moe.top_k == TOPK
    assert blk.moe.ws.qweight.shape == (E, H, 2 * I // 8) and blk.moe.w2s.qweight.shape == (E, I, H // 8)
    assert torch.equal(blk.moe.ws.qweight[1], torch.cat([w1_0, w3_0], dim=1))          # gate columns first, then up
    assert isinstance(blk.attn.qkv_proj, WQLinear_GEMV) and isinstance(blk.attn.o_proj, WQLinear_GEMV)  # "auto": attention on the decode layout
    assert blk.attn.qkv_proj.out_features == (HEADS + 2 * KV) * D and "gemv" in lm.decode_layout
    lm2 = fuse_mixtral(build(), max_seq_len=32, decode_layout=None)
    assert isinstance(lm2.model.blocks[0].attn.qkv_proj, WQLinear_GEMM) and lm2.decode_layout == "gemm"


def reference_logits(m, ids, oracle):
    """The whole model on the CPU from the oracle's pieces (fp16 roundings where the kernels return fp16): ids [B, S] -> [B, S, V]."""
    from oracle import decoder_oracle

    def lin(x, q):  # x [R, K] fp16 numpy, q a WQLinear_GEMM -> fp16 [R, N]
        y32, _ = oracle.linear_gemm(x, q.qweight.numpy(), q.qzeros.numpy(), q.scales.numpy(), q.group_size, None)
        return y32.astype(np.float16)

    B, S = ids.shape
    h = m.model.embed_tokens.weight.detach()[ids].numpy().astype(np.float16)          # [B, S, H]
    min_gap = np.inf
    for layer in m.model.layers:
        a, moe = layer.self_attn, layer.block_sparse_moe
        x = decoder_oracle.rmsnorm_reference(h, layer.input_layernorm.weight.detach().numpy(), EPS).reshape(B * S, H)
        q, k, v = lin(x, a.q_proj), lin(x, a.k_proj), lin(x, a.v_proj)
        qr, kr = decoder_oracle.rope_reference(torch.from_numpy(q).view(B, S, HEADS, D), torch.from_numpy(k).view(B, S, KV, D), 0, D, 64)
        kc, vc = kr.numpy(), v.reshape(B, S, KV, D)
        att = np.zeros((B, S, HEADS * D), np.float16)
        for s in range(S):
            att[:, s] = decoder_oracle.attention_reference(qr[:, s].numpy(), kc, vc, s + 1).reshape(B, -1).astype(np.float16)
        h = (h.astype(np.float32) + lin(att.reshape(B * S, -1), a.o_proj).reshape(B, S, H).astype(np.float32)).astype(np.float16)
        n2 = decoder_oracle.rmsnorm_reference(h, layer.post_attention_layernorm.weight.detach().numpy(), EPS).reshape(B * S, H)
        logits = (n2.astype(np.float32) @ moe.gate.weight.detach().float().numpy().T).astype(np.float16)
        srt = np.sort(logits.astype(np.float32), axis=-1)
        min_gap = min(min_gap, float((srt[:, -TOPK] - srt[:, -TOPK - 1]).min()))
        w1 = {k_: np.stack([np.concatenate([getattr(e.w1, k_).numpy(), getattr(e.w3, k_).numpy()], axis=1) for e in moe.experts])
              for k_ in ("qweight", "qzeros", "scales")}
        w2 = {k_: np.stack([getattr(e.w2, k_).numpy() for e in moe.experts]) for k_ in ("qweight", "qzeros", "scales")}
        y, _, _ = oracle.moe_forward(n2, logits, w1, w2, TOPK, 128)
        h = (h.astype(np.float32) + y.reshape(B, S, H).astype(np.float32)).astype(np.float16)
    hn = decoder_oracle.rmsnorm_reference(h, m.model.norm.weight.detach().numpy(), EPS)
    out = hn.astype(np.float32) @ m.lm_head.weight.detach().float().numpy().T
    return out, min_gap


@pytest.mark.gpu
@pytest.mark.parametrize("decode_layout", ["auto", None])
def test_fused_mixtral_prefill_then_graphed_decode_vs_oracle(oracle, decode_layout):
    """6 context tokens in one forward, then 3 tokens one at a time -- eagerly and through GraphedDecoder (one hipGraph replay per
    token: the MoE decode path reads no routing data back) -- against the oracle's composition of the same quantised model."""
    import copy

    from autoawq_amd.fuser import fuse_mixtral
    from autoawq_amd.modules.fused.decode import GraphedDecoder

    m = build()
    ids = torch.randint(0, V, (2, 9), generator=torch.Generator().manual_seed(9))
    ref, gap = reference_logits(copy.deepcopy(m), ids, oracle)
    assert gap > 0.05, f"router logits too close for a robust top-{TOPK} comparison: gap {gap}"
    rng = np.abs(ref).max()
    lm = fuse_mixtral(m.cuda(), max_seq_len=32, decode_layout=decode_layout)
    out = lm(ids[:, :6].cuda()).float().cpu().numpy()
    assert np.abs(out - ref[:, :6]).max() <= 2e-2 * rng
    for t in range(6, 9):
        step = lm(ids[:, t:t + 1].cuda()).float().cpu().numpy()
        assert np.abs(step[:, 0] - ref[:, t]).max() <= 2e-2 * rng, t
    dec = GraphedDecoder(lm, batch=2, buckets=(8, 16))
    out = dec.prefill(ids[:, :6].cuda()).float().cpu().numpy()
    assert np.abs(out - ref[:, :6]).max() <= 2e-2 * rng and dec.position == 6
    for t in range(6, 9):
        step = dec.step(ids[:, t:t + 1].cuda()).float().cpu().numpy()
        assert np.abs(step[:, 0] - ref[:, t]).max() <= 2e-2 * rng, t
    assert sorted(dec.graphs) == [8, 16]
