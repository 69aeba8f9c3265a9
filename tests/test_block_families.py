"""The other fused block families (reference: awq/modules/fused/block.py:122-544 -- QwenBlock, Gemma2LikeBlock, MPTBlock,
FalconDecoderLayer, Phi3Block; CohereBlock is not built and not shipped): constructor surface on the CPU; on the GPU each block's forward
against the reference's dataflow written out over the block's OWN sub-modules (norms, QuantAttentionFused, MLP -- whose parity
against the oracle / the reference's classes is pinned in tests/test_decoder.py and tests/test_gpu_parity.py), prefill then decode."""
import copy

import pytest
import torch
import torch.nn as nn

H, HEADS, KV, D, I, G = 256, 2, 1, 128, 512, 128


def _lin(K, N, dev, gen):
    from autoawq_amd import WQLinear_GEMM

    lim = 0x7FFFFFFF
    m = WQLinear_GEMM(4, G, K, N, False, dev)
    m.qweight = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, generator=gen).to(dev)
    m.qzeros = torch.randint(-lim - 1, lim, (K // G, N // 8), dtype=torch.int32, generator=gen).to(dev)
    m.scales = (torch.rand((K // G, N), generator=gen) * 0.004 + 0.001).half().to(dev)
    return m


def _rms(dev, gen):
    from autoawq_amd.modules.fused.norm import FasterTransformerRMSNorm

    return FasterTransformerRMSNorm((1.0 + 0.1 * torch.randn(H, generator=gen)).half().to(dev), 1e-5)


def _mlp(dev, gen):
    from autoawq_amd.modules.fused.mlp import QuantFusedMLP

    return QuantFusedMLP(_lin(H, I, dev, gen), _lin(I, H, dev, gen), _lin(H, I, dev, gen))


class _Ffn(nn.Module):  # MPT's ffn: up -> GELU -> down
    def __init__(self, dev, gen):
        super().__init__()
        self.up_proj, self.act, self.down_proj = _lin(H, I, dev, gen), nn.GELU(approximate="none"), _lin(I, H, dev, gen)

    def forward(self, x):
        return self.down_proj(self.act(self.up_proj(x)))


def build(family, dev, gen, kv=KV):
    from autoawq_amd.modules.fused import block as B

    qkv = lambda n_kv: _lin(H, (HEADS + 2 * n_kv) * D, dev, gen)
    if family == "qwen":
        return B.QwenBlock(H, HEADS, kv, qkv(kv), _lin(HEADS * D, H, dev, gen), _mlp(dev, gen), _rms(dev, gen), _rms(dev, gen), dev, 32,
                           q_norm=nn.LayerNorm(D).half().to(dev), k_norm=nn.LayerNorm(D).half().to(dev))
    if family == "gemma2":
        return B.Gemma2LikeBlock(H, HEADS, kv, qkv(kv), _lin(HEADS * D, H, dev, gen), _mlp(dev, gen), _rms(dev, gen), _rms(dev, gen),
                                 _rms(dev, gen), _rms(dev, gen), dev, 32, attn_logit_softcapping=30.0)
    if family == "phi3":
        return B.Phi3Block(H, HEADS, kv, qkv(kv), _lin(HEADS * D, H, dev, gen), _mlp(dev, gen), _rms(dev, gen), _rms(dev, gen), dev, 32)
    if family == "mpt":
        return B.MPTBlock(H, HEADS, qkv(HEADS), _lin(HEADS * D, H, dev, gen), _Ffn(dev, gen), nn.LayerNorm(H).half().to(dev),
                          nn.LayerNorm(H).half().to(dev), dev, 32)
    if family == "falcon_new":  # the new architecture hard-codes 8 KV heads (block.py:390): 8 query heads of 128 here
        return B.FalconDecoderLayer(1024, 8, _lin(1024, (8 + 16) * D, dev, gen), _lin(1024, 1024, dev, gen), _FalconMlp(dev, gen, 1024), dev, 32,
                                    ln_attn=nn.LayerNorm(1024).half().to(dev), ln_mlp=nn.LayerNorm(1024).half().to(dev), new_decoder_arch=True)
    if family == "falcon_old":
        return B.FalconDecoderLayer(H, HEADS, _lin(H, (HEADS + 2) * D, dev, gen), _lin(HEADS * D, H, dev, gen), _FalconMlp(dev, gen, H), dev, 32,
                                    input_layernorm=nn.LayerNorm(H).half().to(dev), new_decoder_arch=False)
    raise KeyError(family)


class _FalconMlp(nn.Module):
    def __init__(self, dev, gen, h):
        super().__init__()
        self.up, self.act, self.down = _lin(h, 2 * h, dev, gen), nn.GELU(), _lin(2 * h, h, dev, gen)

    def forward(self, x):
        return self.down(self.act(self.up(x)))


def test_block_family_surface_cpu():
    from autoawq_amd.modules.fused import block as B

    gen = torch.Generator().manual_seed(0)
    blk = build("gemma2", "cpu", gen)
    assert blk.attn.attn_logit_softcapping == 30.0 and blk.head_dim == D and [n for n, _ in blk.named_children()][:2] == ["norm_1", "attn"]
    q = build("qwen", "cpu", gen)
    assert q.attn.q_norm is not None and q.attn.k_norm is not None
    m = build("mpt", "cpu", gen)
    assert m.attn.use_alibi and m.n_kv_heads == 0 and hasattr(m, "ffn")
    fo = build("falcon_old", "cpu", gen)
    assert fo.attn.n_kv_heads == 1 and fo.attention_shapes["xqkv_view"] == (HEADS + 2, D) and not fo.new_decoder_arch
    assert not hasattr(B, "CohereBlock")  # (needs the interleaved rotary form: not built, and no stub class is shipped)
    with pytest.raises(NotImplementedError):
        B.Phi3Block(H, HEADS, KV, None, None, None, None, None, "cpu", 32, rope_scaling={"type": "longrope"})


def dataflow(family, blk, x):
    """The reference's forward of each family (block.py:178-188, 243-262, 356-368, 462-487, 535-544) over the block's sub-modules."""
    if family in ("qwen", "phi3"):
        a, _, _ = blk.attn.forward(hidden_states=blk.norm_1(x))
        h = x + a
        return h + blk.mlp.forward(blk.norm_2(h))
    if family == "gemma2":
        a, _, _ = blk.attn.forward(hidden_states=blk.norm_1(x))
        h = x + blk.norm_2(a)
        return h + blk.norm_4(blk.mlp(blk.norm_3(h)))
    if family == "mpt":
        a, _, _ = blk.attn.forward(hidden_states=blk.norm_1(x))
        h = x + a
        return h + blk.ffn.forward(blk.norm_2(h))
    if family == "falcon_new":
        a, _, _ = blk.attn.forward(hidden_states=blk.ln_attn(x))
        return (x + a) + blk.mlp.forward(blk.ln_mlp(x))
    ln = blk.input_layernorm(x)
    a, _, _ = blk.attn.forward(hidden_states=ln)
    return (x + a) + blk.mlp.forward(ln)


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["qwen", "gemma2", "phi3", "mpt", "falcon_new", "falcon_old"])
def test_block_families_follow_the_reference_dataflow(family):
    gen = torch.Generator().manual_seed(31)
    blk = build(family, "cuda", gen)
    ref = copy.deepcopy(blk)           # the same weights and an own KV cache for the written-out dataflow
    hs = blk.hidden_size
    with torch.inference_mode():
        x = (torch.randn((2, 5, hs), generator=gen) * 0.5).half().cuda()
        got, want = blk(x), dataflow(family, ref, x)
        assert got.shape == x.shape and torch.isfinite(got.float()).all()
        assert float((got.float() - want.float()).abs().max()) <= 2e-3 * float(want.float().abs().max()) + 1e-3
        for _ in range(2):                                                     # two decode steps through the caches
            x1 = (torch.randn((2, 1, hs), generator=gen) * 0.5).half().cuda()
            got, want = blk(x1), dataflow(family, ref, x1)
            assert float((got.float() - want.float()).abs().max()) <= 2e-3 * float(want.float().abs().max()) + 1e-3
        assert blk.attn.start_pos == 7 == ref.attn.start_pos
