"""Fused decoder block around the int4 Linears: RMSNorm (+ residual),
RoPE + KV append, single-query attention, LlamaLikeBlock / LlamaLikeModel.

CPU: the oracle's RoPE restatement against the reference's own RoPE class (tests/golden/
rope_golden.npz), host logic of the cache / input-id handling.  GPU: each kernel against the
oracle, then the whole fused model (prefill + token-by-token decode, eager and as a replayed
hipGraph) against the logits the reference computed on CPU from the same checkpoint.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from test_checkpoint import ckpt, skeleton


def ulp16(a):
    return np.maximum(np.abs(np.asarray(a, np.float64)), 2.0 ** -14) * 2.0 ** -10


# ------------------------------------------------------------------ CPU

@pytest.mark.parametrize("name", ["full", "partial", "late"])
def test_oracle_rope_matches_reference_class(name):
    from oracle import decoder_oracle

    g = golden("rope_golden")
    D, rot, start, S = [int(v) for v in g[f"{name}_meta"]]
    q, k = decoder_oracle.rope_reference(torch.from_numpy(g[f"{name}_xq"]), torch.from_numpy(g[f"{name}_xk"]), start, rot, 64)
    assert np.array_equal(q[..., :rot].numpy().view(np.uint16), g[f"{name}_q"].view(np.uint16))
    assert np.array_equal(k[..., :rot].numpy().view(np.uint16), g[f"{name}_k"].view(np.uint16))
    assert torch.equal(q[..., rot:], torch.from_numpy(g[f"{name}_xq"])[..., rot:])  # pass-through dims


def test_oracle_rmsnorm_close_to_transformers_rmsnorm():
    from oracle import decoder_oracle
    from transformers.models.llama.modeling_llama import LlamaRMSNorm

    gen = torch.Generator().manual_seed(0)
    x = (torch.randn((5, 256), generator=gen) * 3).half()
    norm = LlamaRMSNorm(256, eps=1e-5).half()
    norm.weight.data = (torch.rand(256, generator=gen) + 0.5).half()
    want = norm(x).detach().numpy().astype(np.float64)
    got = decoder_oracle.rmsnorm_reference(x.numpy(), norm.weight.detach().numpy(), 1e-5).astype(np.float64)
    assert (np.abs(got - want) <= 2 * ulp16(want)).all()  # HF rounds twice, the fused form once


def test_oracle_attention_matches_torch_sdpa():
    """The attention restatement against torch's own scaled_dot_product_attention (CPU, fp32) on a GQA
    case with a cache longer than the sequence: the definition flash_attn_with_kvcache implements."""
    from oracle import decoder_oracle
    import torch.nn.functional as F

    gen = torch.Generator().manual_seed(3)
    B, Hq, Hkv, T, Tmax = 2, 8, 2, 37, 50
    q = torch.randn((B, Hq, 128), generator=gen).half()
    kc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    vc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    want = F.scaled_dot_product_attention(q.float()[:, :, None, :],
                                          kc[:, :T].float().transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1),
                                          vc[:, :T].float().transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1))[:, :, 0].numpy()
    got = decoder_oracle.attention_reference(q.numpy(), kc.numpy(), vc.numpy(), T)
    assert np.abs(got - want).max() <= 1e-5


def test_oracle_prefill_attention_matches_torch_sdpa():
    """The prefill restatement (one causal row = one decode query over the first start + s + 1 cache rows) against torch's own
    scaled_dot_product_attention with the chunked-prefill mask (CPU, fp32): the definition flash_attn_func(causal=True) implements."""
    from oracle import decoder_oracle
    import torch.nn.functional as F

    gen = torch.Generator().manual_seed(1)
    B, S, Hq, Hkv, start = 2, 9, 4, 2, 5
    q = torch.randn((B, S, Hq, 128), generator=gen).half()
    kc = torch.randn((B, start + S + 3, Hkv, 128), generator=gen).half()
    vc = torch.randn((B, start + S + 3, Hkv, 128), generator=gen).half()
    ref = decoder_oracle.prefill_attention_reference(q.numpy(), kc.numpy(), vc.numpy(), start, list(range(S)))
    k = kc[:, :start + S].float().transpose(1, 2).repeat_interleave(2, dim=1)
    v = vc[:, :start + S].float().transpose(1, 2).repeat_interleave(2, dim=1)
    mask = torch.ones((S, start + S), dtype=torch.bool).tril(diagonal=start)
    want = F.scaled_dot_product_attention(q.float().transpose(1, 2), k, v, attn_mask=mask).transpose(1, 2).numpy()
    assert np.abs(want - ref).max() <= 1e-5


def test_input_id_and_cache_bookkeeping():
    from autoawq_amd.modules.fused.cache import WindowedCache
    from autoawq_amd.modules.fused.model import prepare_input_ids

    ids = torch.arange(10).reshape(1, 10)
    out, n = prepare_input_ids(ids, 0)
    assert out.shape[1] == 10 and n == 10
    out, n = prepare_input_ids(torch.arange(11).reshape(1, 11), 10)  # transformers passes the whole context
    assert out.shape[1] == 1 and int(out[0, 0]) == 10 and n == 11
    out, n = prepare_input_ids(torch.tensor([[7]]), 11)
    assert out.shape[1] == 1 and n == 12
    c = WindowedCache(2, 4, 2, 8, 16, "cpu")
    assert c.k.shape == (2, 16, 2, 8)
    c.k[:, :, :, :] = torch.arange(16, dtype=torch.float16).reshape(1, 16, 1, 1)
    c.v[:] = c.k
    assert c.k.data_ptr() == c.kv[0].data_ptr() and c.v.data_ptr() == c.kv[1].data_ptr()  # views of one allocation
    # forget the 5 oldest of 12 cached positions: the 7 live rows move to the front (rows past the new length are
    # never read by the attention kernels and are left as they are)
    assert c.roll_kv_n_steps(12, n=5) == 7
    assert [float(c.k[1, t, 1, 3]) for t in range(7)] == [5.0, 6.0, 7.0, 8.0, 9.0, 10.0, 11.0]
    assert torch.equal(c.v[:, :7], c.k[:, :7])
    assert c.roll_kv_n_steps(7, n=100) == 0 and c.roll_kv_n_steps(0, n=3) == 0
    vv, kk = c.get_kv(2, 3, 2)
    assert kk.shape == (2, 5, 2, 8) and vv.shape == (2, 5, 2, 8)
    c.update_kv(torch.full((2, 2, 2, 8), 3.0, dtype=torch.float16), torch.full((2, 2, 2, 8), 4.0, dtype=torch.float16), 2, 3, 2)
    assert float(c.k[1, 4, 0, 0]) == 4.0 and float(c.v[0, 3, 1, 7]) == 3.0 and float(c.k[0, 2, 0, 0]) == 7.0
    c.increase_batch_size(3)   # surviving sequences keep their rows
    assert c.k.shape == (3, 16, 2, 8) and float(c.k[1, 4, 0, 0]) == 4.0 and float(c.k[2, 0, 0, 0]) == 0.0
    c.decrease_batch_size(1)
    assert c.k.shape == (1, 16, 2, 8) and float(c.v[0, 3, 1, 7]) == 3.0


def test_step_plan_policy():
    """which ids are new and how much every block forgets (awq/utils/fused_utils.py:14-42)"""
    from autoawq_amd.modules.fused.model import StepPlan

    p = StepPlan(torch.arange(6).reshape(1, 6), 0)           # first prompt
    assert p.seqlen == 6 and p.tokens_seen == 6 and p.forget(0, 16) == 0
    p = StepPlan(torch.arange(7).reshape(1, 7), 6)           # whole context re-sent while decoding
    assert p.seqlen == 1 and p.tokens_seen == 7 and int(p.input_ids[0, 0]) == 6
    assert p.forget(6, 16) == 0 and p.forget(16, 16) == 100  # decode past the window: the 100 oldest go
    p = StepPlan(torch.arange(5).reshape(1, 5), 20)          # a new, shorter context: everything cached goes
    assert p.seqlen == 5 and p.forget(9, 16) == 9 and p.forget(0, 16) == 0


def test_fused_module_surfaces():
    """Constructor signatures the reference's fusers use (awq/models/llama.py:100-175)."""
    import inspect

    from autoawq_amd.modules.fused.attn import QuantAttentionFused
    from autoawq_amd.modules.fused.block import LlamaLikeBlock
    from autoawq_amd.modules.fused.model import LlamaLikeModel
    from autoawq_amd.modules.fused.norm import FasterTransformerRMSNorm

    assert list(inspect.signature(LlamaLikeBlock.__init__).parameters)[1:12] == [
        "hidden_size", "n_heads", "n_kv_heads", "qkv_layer", "o_proj", "mlp", "norm_1", "norm_2", "dev", "max_seq_len",
        "rope_theta"]
    assert list(inspect.signature(QuantAttentionFused.__init__).parameters)[1:8] == [
        "hidden_size", "n_heads", "n_kv_heads", "qkv_layer", "o_proj", "dev", "max_seq_len"]
    assert list(inspect.signature(LlamaLikeModel.__init__).parameters)[1:] == ["vocab_size", "blocks", "embedding", "norm"]
    assert list(inspect.signature(FasterTransformerRMSNorm.__init__).parameters)[1:] == ["weight", "eps"]


def test_alibi_and_attention_shapes_match_the_reference_classes():
    """ALiBi slopes / bias and get_attention_shapes against vectors the reference's own classes produced
    (tests/golden/make_golden_attention.py: attn.py:89-125, fused_utils.py:165-201), for the module and for the oracle."""
    from autoawq_amd.modules.fused.attn import ALiBi
    from autoawq_amd.utils.fused_utils import get_attention_shapes
    from oracle import decoder_oracle

    g = golden("attention_golden")
    for n in (8, 12, 40):
        slopes, bias = ALiBi.build_alibi_bias(n, 16)
        assert np.array_equal(slopes.numpy(), g[f"alibi_slopes_{n}"]) and np.array_equal(bias.numpy(), g[f"alibi_bias_{n}"])
        assert np.allclose(decoder_oracle.alibi_slopes_reference(n), g[f"alibi_slopes_{n}"], rtol=0, atol=0)
    for name in ("mha_interleaved", "gqa"):
        H, Hkv, D = (int(v) for v in g[f"{name}_meta"])
        sh = get_attention_shapes(None, H, Hkv, D)
        x = torch.from_numpy(g[f"{name}_xqkv"])
        v = x.view((2, 3) + tuple(sh["xqkv_view"]))
        for key in ("xq", "xk", "xv"):
            assert np.array_equal(sh[key + "_slice"](v).contiguous().numpy(), g[f"{name}_{key}"]), (name, key)
    custom = {"xqkv_view": (1,)}
    assert get_attention_shapes(custom, 4, 2, 8) is custom


def test_oracle_attention_score_modifiers():
    """soft-capping and ALiBi in the oracle against a direct restatement of what flash-attn computes with those arguments"""
    from oracle import decoder_oracle

    gen = torch.Generator().manual_seed(3)
    q = torch.randn((1, 4, 16), generator=gen).double().numpy()
    k = torch.randn((1, 9, 2, 16), generator=gen).double().numpy()
    v = torch.randn((1, 9, 2, 16), generator=gen).double().numpy()
    slopes = decoder_oracle.alibi_slopes_reference(4)
    got = decoder_oracle.attention_reference(q, k, v, 7, softcap=5.0, alibi_slopes=slopes)
    for h in range(4):
        s = (k[0, :7, h // 2] @ q[0, h]) * 16 ** -0.5
        s = 5.0 * np.tanh(s / 5.0) - slopes[h] * np.arange(6, -1, -1)
        p = np.exp(s - s.max())
        assert np.allclose(got[0, h], (p / p.sum()) @ v[0, :7, h // 2], rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ GPU kernels

@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from autoawq_amd import _lib, ops as _ops

    _lib.lib()
    return _ops


@pytest.mark.gpu
@pytest.mark.parametrize("M,H", [(1, 4096), (3, 256), (17, 5120), (2, 8192), (4, 136)])
def test_rmsnorm_vs_oracle(ops, M, H):
    from oracle import decoder_oracle

    gen = torch.Generator().manual_seed(M * 1000 + H)
    x = (torch.randn((M, H), generator=gen) * 2).half()
    w = (torch.rand(H, generator=gen) + 0.5).half()
    want = decoder_oracle.rmsnorm_reference(x.numpy(), w.numpy(), 1e-5).astype(np.float64)
    got = ops.rmsnorm(x.cuda(), w.cuda(), 1e-5).cpu().numpy().astype(np.float64)
    assert (np.abs(got - want) <= ulp16(want)).all()
    # residual form: r <- fp16(r + x) bit for bit, and the norm of that
    r = torch.randn((M, H), generator=gen).half()
    rd = r.cuda()
    got2 = ops.rmsnorm(x.cuda(), w.cuda(), 1e-5, residual=rd).cpu().numpy().astype(np.float64)
    rsum = (x.float() + r.float()).half()
    assert torch.equal(rd.cpu(), rsum)
    want2 = decoder_oracle.rmsnorm_reference(rsum.numpy(), w.numpy(), 1e-5).astype(np.float64)
    assert (np.abs(got2 - want2) <= ulp16(want2)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["full", "partial", "late"])
def test_rope_kv_append_vs_reference_rope(ops, name):
    """q, and the cache rows written, equal the reference RoPE class's output (<= 1 fp16 ulp: the
    complex multiply may or may not contract into an fma); untouched cache rows stay as they were;
    device-side start position gives the same result."""
    from autoawq_amd.modules.fused.attn import RoPE

    g = golden("rope_golden")
    D, rot, start, S = [int(v) for v in g[f"{name}_meta"]]
    xq, xk = torch.from_numpy(g[f"{name}_xq"]), torch.from_numpy(g[f"{name}_xk"])
    B, _, Hq, _ = xq.shape
    Hkv = xk.shape[2]
    xv = torch.randn((B, S, Hkv, D), generator=torch.Generator().manual_seed(1)).half()
    qkv = torch.cat([xq.reshape(B, S, -1), xk.reshape(B, S, -1), xv.reshape(B, S, -1)], dim=-1).cuda()
    rope = RoPE(rot, 64, "cuda", 10000.0)
    for use_dev in (False, True):
        kc = torch.full((B, 64, Hkv, D), 7.0, dtype=torch.float16, device="cuda")
        vc = torch.full((B, 64, Hkv, D), 9.0, dtype=torch.float16, device="cuda")
        pos = torch.tensor([start], dtype=torch.int32, device="cuda") if use_dev else None
        q = ops.rope_kv_append(qkv, kc, vc, rope.cos, rope.sin, 0 if use_dev else start, Hq, Hkv, D, rot, pos_dev=pos)
        qn, kn = q.cpu().numpy().astype(np.float64), kc[:, start:start + S].cpu().numpy().astype(np.float64)
        assert (np.abs(qn[..., :rot] - g[f"{name}_q"]) <= ulp16(g[f"{name}_q"])).all()
        assert (np.abs(kn[..., :rot] - g[f"{name}_k"]) <= ulp16(g[f"{name}_k"])).all()
        assert torch.equal(q[..., rot:].cpu(), xq[..., rot:]) and torch.equal(kc[:, start:start + S, :, rot:].cpu(), xk[..., rot:])
        assert torch.equal(vc[:, start:start + S].cpu(), xv)
        assert bool((kc[:, :start] == 7.0).all()) and bool((kc[:, start + S:] == 7.0).all())
        assert bool((vc[:, :start] == 9.0).all()) and bool((vc[:, start + S:] == 9.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("B,Hq,Hkv,T", [(1, 32, 32, 1), (1, 32, 32, 77), (2, 8, 4, 300), (1, 32, 8, 2048), (3, 16, 2, 129),
                                       (1, 32, 32, 4096), (2, 4, 4, 16)])
def test_decode_attention_vs_oracle(ops, B, Hq, Hkv, T):
    from oracle import decoder_oracle

    gen = torch.Generator().manual_seed(B * 7 + Hq + T)
    Tmax = max(T + 5, 64)
    q = torch.randn((B, Hq, 128), generator=gen).half()
    kc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    vc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    kc[:, T:] = 100.0  # rows past the length must not be read
    vc[:, T:] = 100.0
    want = decoder_oracle.attention_reference(q.numpy(), kc.numpy(), vc.numpy(), T)
    got = ops.decode_attention(q.cuda(), kc.cuda(), vc.cuda(), T).cpu().numpy().astype(np.float64)
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max() + 2 ** -11
    ln = torch.tensor([T], dtype=torch.int32, device="cuda")  # device-side length, launch sized for the whole cache
    got2 = ops.decode_attention(q.cuda(), kc.cuda(), vc.cuda(), 1, len_dev=ln, max_len=Tmax).cpu().numpy().astype(np.float64)
    assert np.abs(got2 - want).max() <= 2e-3 * np.abs(want).max() + 2 ** -11


@pytest.mark.gpu
@pytest.mark.parametrize("softcap,alibi", [(30.0, False), (0.0, True), (8.0, True)])
@pytest.mark.parametrize("B,Hq,Hkv,T", [(1, 32, 32, 77), (2, 8, 4, 300), (1, 32, 8, 2048), (3, 16, 2, 129), (2, 12, 12, 1)])
def test_decode_attention_softcap_and_alibi_vs_oracle(ops, B, Hq, Hkv, T, softcap, alibi):
    """awq_decode_attention_ex: the two score modifiers of flash_attn_with_kvcache (attn.py:286-302)"""
    from oracle import decoder_oracle

    gen = torch.Generator().manual_seed(B * 7 + Hq + T)
    Tmax = max(T + 5, 64)
    q = (torch.randn((B, Hq, 128), generator=gen) * 2).half()
    kc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    vc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    slopes = decoder_oracle.alibi_slopes_reference(Hq) if alibi else None
    want = decoder_oracle.attention_reference(q.numpy(), kc.numpy(), vc.numpy(), T, softcap=softcap, alibi_slopes=slopes)
    sl = torch.from_numpy(slopes).float().cuda() if alibi else None
    got = ops.decode_attention(q.cuda(), kc.cuda(), vc.cuda(), T, softcap=softcap, alibi_slopes=sl).cpu().numpy().astype(np.float64)
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max() + 2 ** -11
    ln = torch.tensor([T], dtype=torch.int32, device="cuda")
    got2 = ops.decode_attention(q.cuda(), kc.cuda(), vc.cuda(), 1, len_dev=ln, max_len=Tmax, softcap=softcap, alibi_slopes=sl)
    assert np.abs(got2.cpu().numpy().astype(np.float64) - want).max() <= 2e-3 * np.abs(want).max() + 2 ** -11
    plain = decoder_oracle.attention_reference(q.numpy(), kc.numpy(), vc.numpy(), T)
    if T > 1:
        assert np.abs(plain - want).max() > 1e-3, "the modifiers must change the result for this test to mean anything"


class _HeadRMSNorm(torch.nn.Module):
    """per-head RMSNorm over head_dim, the q_norm / k_norm of Qwen3 / Gemma3-style attention"""

    def __init__(self, dim, gen):
        super().__init__()
        self.weight = torch.nn.Parameter((torch.rand(dim, generator=gen) + 0.5).half(), requires_grad=False)

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * self.weight.float()).to(x.dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,Hq,Hkv,start,softcap,alibi", [
    (1, 2048, 32, 32, 0, 0.0, False),    # BASELINE configs[2]'s attention: 32 heads x d128 x 2048 tokens
    (2, 300, 8, 1, 0, 0.0, False),       # GQA 8:1, ragged S (partial last row block and KV tile)
    (1, 257, 16, 4, 1000, 0.0, False),   # chunked prefill: the new rows sit behind 1000 cached ones
    (3, 129, 4, 4, 63, 0.0, False),      # three sequences, an odd start
    (1, 1, 32, 8, 77, 0.0, False),       # one row (a decode step through the prefill kernel)
    (2, 200, 8, 2, 40, 30.0, False),     # soft cap (Gemma-2's attn_logit_softcapping)
    (1, 333, 12, 12, 0, 0.0, True),      # ALiBi slopes (MPT / Falcon)
    (2, 160, 8, 4, 500, 8.0, True),      # both, chunked
])
def test_prefill_attention_vs_oracle(ops, B, S, Hq, Hkv, start, softcap, alibi):
    """csrc/prefill_attn.hip (the reference's flash_attn_func call, attn.py:269-277) against oracle/decoder_oracle.py::
    prefill_attention_reference on sampled query rows (every row block boundary, the first and the last rows), and every output
    against a plain fp32 torch attention of the same tensors; cache rows past start + S are NaN-poisoned (never read); bitwise
    reproducible.  Tolerance: fp16 probabilities in the second product -> 2e-3 of the row's largest |value| + 2 fp16 ulps."""
    from oracle import decoder_oracle

    gen = torch.Generator().manual_seed(B * 1000 + S + Hq + start)
    Tmax = start + S + 70
    q = torch.randn((B, S, Hq, 128), generator=gen).half()
    kc = torch.randn((B + 1, Tmax, Hkv, 128), generator=gen).half()
    vc = torch.randn((B + 1, Tmax, Hkv, 128), generator=gen).half()
    kc[:, start + S:] = float("nan")
    vc[:, start + S:] = float("nan")
    slopes = (0.5 ** torch.arange(1, Hq + 1, dtype=torch.float32) * 4.0) if alibi else None
    qd, kd, vd = q.cuda(), kc.cuda(), vc.cuda()
    out = ops.prefill_attention(qd, kd, vd, start, softcap=softcap, alibi_slopes=slopes.cuda() if alibi else None)
    assert out.shape == q.shape and bool(torch.isfinite(out).all())
    assert torch.equal(out, ops.prefill_attention(qd, kd, vd, start, softcap=softcap, alibi_slopes=slopes.cuda() if alibi else None))
    rows = sorted(set(r for r in (0, 1, S - 1, S // 2, 31, 32, 63, 64, 127, 128, 129, 255, 256, 1023, 1024, 2047) if r < S))
    ref = decoder_oracle.prefill_attention_reference(q.numpy(), kc.numpy()[:B], vc.numpy()[:B], start, rows, softcap=softcap,
                                                     alibi_slopes=slopes.numpy() if alibi else None)
    got = out.cpu().numpy().astype(np.float64)[:, rows]
    tol = 2e-3 * np.abs(ref).max(axis=-1, keepdims=True) + 2 * ulp16(ref)
    assert (np.abs(got - ref) <= tol).all(), float(np.abs(got - ref).max())
    # every output element against fp32 torch attention on the GPU
    G = Hq // Hkv
    qf = qd.float().transpose(1, 2)
    kf = kd[:B, :start + S].float().transpose(1, 2).repeat_interleave(G, dim=1)
    vf = vd[:B, :start + S].float().transpose(1, 2).repeat_interleave(G, dim=1)
    sc = torch.matmul(qf, kf.transpose(-1, -2)) * (128 ** -0.5)
    if softcap:
        sc = softcap * torch.tanh(sc / softcap)
    qpos = start + torch.arange(S, device="cuda").view(-1, 1)
    kpos = torch.arange(start + S, device="cuda").view(1, -1)
    if alibi:
        sc = sc + slopes.cuda().view(1, -1, 1, 1) * (kpos - qpos).float()
    sc = sc.masked_fill(kpos > qpos, float("-inf"))
    full = torch.matmul(torch.softmax(sc, dim=-1), vf).transpose(1, 2)
    d = (out.float() - full).abs()
    assert bool((d <= 2e-3 * full.abs().amax(dim=-1, keepdim=True) + 2e-3).all()), float(d.max())


@pytest.mark.gpu
def test_prefill_attention_refuses_other_head_sizes(ops):
    from autoawq_amd import _lib

    q = torch.zeros((1, 4, 2, 64), dtype=torch.float16, device="cuda")
    kc = torch.zeros((1, 8, 2, 64), dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.AwqHipError) as ei:
        ops.prefill_attention(q, kc, kc.clone(), 0)
    assert ei.value.code == _lib.ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("feature", ["alibi", "softcap", "qk_norm", "custom_shapes", "alibi_head64"])
@torch.no_grad()
def test_quant_attention_fused_feature_surface(feature):
    """QuantAttentionFused with ALiBi, logit soft-capping, q / k norms and a custom `attention_shapes` (attn.py:159-203,243-302):
    a 6-token prefill followed by 3 single-token steps through the cache equals a plain fp32 causal attention over the whole
    sequence written out here from the reference's forward (slice -> norms -> rotate unless ALiBi -> scores * scale ->
    soft-cap -> + slope * (key - query) -> causal softmax -> values -> o_proj)."""
    from autoawq_amd.modules.fused.attn import ALiBi, QuantAttentionFused, RoPE
    from oracle import decoder_oracle

    gen = torch.Generator().manual_seed(hash(feature) % 1000)
    D = 64 if feature == "alibi_head64" else 128
    Hq, Hkv, hidden, B, T = 4, 2, 256, 2, 9
    width = (Hq + 2 * Hkv) * D
    qkv = torch.nn.Linear(hidden, width, bias=False).half().cuda()
    o = torch.nn.Linear(Hq * D, hidden, bias=False).half().cuda()
    with torch.no_grad():
        qkv.weight.copy_((torch.randn((width, hidden), generator=gen) * 0.08).half())
        o.weight.copy_((torch.randn((hidden, Hq * D), generator=gen) * 0.05).half())
    kw = {}
    shapes = None
    if feature.startswith("alibi"):
        kw["use_alibi"] = True
    if feature == "softcap":
        kw["attn_logit_softcapping"] = 4.0
    if feature == "qk_norm":
        kw["q_norm"], kw["k_norm"] = _HeadRMSNorm(D, gen).cuda(), _HeadRMSNorm(D, gen).cuda()
    if feature == "custom_shapes":  # Falcon-style rows: per KV group [its query heads, k, v]
        G = Hq // Hkv
        shapes = {"xqkv_view": (Hkv, G + 2, D),
                  "xq_slice": lambda x: x[:, :, :, :-2].reshape(x.shape[0], x.shape[1], -1, D),
                  "xk_slice": lambda x: x[:, :, :, -2],
                  "xv_slice": lambda x: x[:, :, :, -1]}
        kw["attention_shapes"] = shapes
    attn = QuantAttentionFused(hidden, Hq, Hkv, qkv, o, dev="cuda", max_seq_len=32, head_dim=D, **kw)
    x = torch.randn((B, T, hidden), generator=gen).half().cuda()

    # ---- the whole sequence at once, fp32
    rows = qkv(x)
    if shapes is not None:
        v5 = rows.view((B, T) + shapes["xqkv_view"])
        xq, xk, xv = shapes["xq_slice"](v5), shapes["xk_slice"](v5), shapes["xv_slice"](v5)
    else:
        v4 = rows.view(B, T, Hq + 2 * Hkv, D)
        xq, xk, xv = v4[:, :, :Hq], v4[:, :, Hq:Hq + Hkv], v4[:, :, Hq + Hkv:]
    if "q_norm" in kw:
        xq, xk = kw["q_norm"](xq), kw["k_norm"](xk)
    if not feature.startswith("alibi"):
        xq, xk = decoder_oracle.rope_reference(xq.cpu(), xk.cpu(), 0, D, 32)
        xq, xk = xq.cuda(), xk.cuda()
    qf = xq.float().transpose(1, 2)
    kf = xk.float().transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    vf = xv.float().transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    sc = qf @ kf.transpose(-1, -2) * D ** -0.5
    if feature == "softcap":
        sc = 4.0 * torch.tanh(sc / 4.0)
    pos = torch.arange(T, device="cuda")
    if feature.startswith("alibi"):
        sc = sc + ALiBi.gen_slopes(Hq).cuda() * (pos.view(1, -1) - pos.view(-1, 1)).float()
    sc = sc.masked_fill(pos.view(1, -1) > pos.view(-1, 1), float("-inf"))
    want = o((torch.softmax(sc, -1) @ vf).transpose(1, 2).reshape(B, T, -1).half()).float()

    got_prefill, _, _ = attn(x[:, :6])
    outs = [got_prefill.float()]
    for t in range(6, T):
        step, _, _ = attn(x[:, t:t + 1])
        outs.append(step.float())
    got = torch.cat(outs, dim=1)
    assert attn.start_pos == T
    err = (got - want).abs().max()
    assert float(err) <= 1e-2 * float(want.abs().max()) + 1e-3, float(err)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Hq,Hkv,pos", [(1, 32, 32, 0), (1, 32, 32, 300), (2, 8, 2, 63), (1, 32, 8, 2047), (3, 4, 4, 17)])
def test_fused_rope_append_attention_equals_the_two_kernels(ops, B, Hq, Hkv, pos):
    """awq_decode_attention_rope == awq_rope_kv_append followed by awq_decode_attention: the appended
    cache rows are bit-identical and nothing else in the caches is touched; the attention output agrees to
    the kernel tolerance (the two instantiations contract their fp32 multiply-adds differently); also with
    the position on the device."""
    from autoawq_amd.modules.fused.attn import RoPE

    gen = torch.Generator().manual_seed(pos + B)
    Tmax = 2048 + 8
    qkv = torch.randn((B, 1, (Hq + 2 * Hkv) * 128), generator=gen).half().cuda()
    base_k = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    base_v = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    rope = RoPE(128, Tmax, "cuda", 10000.0)
    k1, v1 = base_k.cuda(), base_v.cuda()
    q = ops.rope_kv_append(qkv, k1, v1, rope.cos, rope.sin, pos, Hq, Hkv, 128, 128)
    want = ops.decode_attention(q[:, 0], k1, v1, pos + 1)
    for use_dev in (False, True):
        k2, v2 = base_k.cuda(), base_v.cuda()
        pd = torch.tensor([pos], dtype=torch.int32, device="cuda") if use_dev else None
        got = ops.decode_attention_rope(qkv, k2, v2, rope.cos, rope.sin, 0 if use_dev else pos, Hq, Hkv, pos_dev=pd,
                                        max_len=Tmax if use_dev else None)
        assert torch.equal(k2, k1) and torch.equal(v2, v1)
        assert (got.float() - want.float()).abs().max() <= 2e-3 * want.float().abs().max() + 2 ** -11


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("K,N", [(4096, 12288), (512, 256), (11008, 4096)])
def test_norm_folded_into_projection(ops, M, K, N):
    """awq_gemm_forward_normed == awq_rmsnorm_forward (with residual add) + awq_gemm_forward: the residual
    stream bit for bit, the product to the kernel tolerance (the sum of squares is reduced in another order)."""
    from test_gpu_parity import fullrange_case

    qw, qz, s, _, _ = fullrange_case(K, N, 128, 1, seed=K + M, realistic=True)
    gen = torch.Generator().manual_seed(M + N)
    x = torch.randn((M, K), generator=gen).half().cuda()
    res = (torch.randn((M, K), generator=gen) * 2).half().cuda()
    w = (torch.rand(K, generator=gen) + 0.5).half().cuda()
    dq, ds, dz = qw.cuda(), s.cuda(), qz.cuda()
    for residual in (None, res):
        r2 = None if residual is None else residual.clone()
        normed = ops.rmsnorm(x, w, 1e-5, residual=r2)
        want = ops.gemm_forward(normed, dq, ds, dz).float()
        got, stream = ops.gemm_forward_normed(x, w, 1e-5, dq, ds, dz, residual=residual)
        assert ops.last_kernel() == "gemv_mfma"
        if residual is None:
            assert stream is None
        else:
            assert torch.equal(stream, r2) and torch.equal(residual, res)  # the input stream is not written
        err = (got.float() - want).abs()
        assert float(err.max()) <= 2e-3 * float(want.abs().max()) + 1e-3, float(err.max())
    with pytest.raises(Exception):
        ops.gemm_forward_normed(torch.zeros((5, K), dtype=torch.float16, device="cuda"), w, 1e-5, dq, ds, dz)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 3, 4])
@pytest.mark.parametrize("H,I", [(4096, 11008), (512, 1024)])
def test_residual_epilogue_hands_the_statistic_to_the_next_projection(ops, M, H, I):
    """awq_gemm_forward_ex: a row-parallel projection (o_proj / down) adds the residual stream in its
    epilogue and emits per-tile sums of squares; the next projection normalises from them without a
    statistic pass.  Stream: bit-identical to `h + proj(x)` in torch (two fp16 roundings); the partial
    sums match the stored values; the normed product matches rmsnorm + gemm within the kernel tolerance."""
    from test_gpu_parity import fullrange_case

    qw1, qz1, s1, _, _ = fullrange_case(I, H, 128, 1, seed=H + M, realistic=True)      # down: I -> H
    qw2, qz2, s2, _, _ = fullrange_case(H, 2 * I, 128, 1, seed=H + M + 1, realistic=True)  # next gate|up: H -> 2I
    gen = torch.Generator().manual_seed(M)
    gate_up = torch.randn((M, 2 * I), generator=gen).half().cuda()
    h = (torch.randn((M, H), generator=gen) * 2).half().cuda()
    w = (torch.rand(H, generator=gen) + 0.5).half().cuda()
    d1 = (qw1.cuda(), s1.cuda(), qz1.cuda())
    d2 = (qw2.cuda(), s2.cuda(), qz2.cuda())
    # reference chain with the separate kernels
    proj = ops.gemm_forward(gate_up, *d1, flags=ops.X_GATED_SILU)
    stream_ref = (proj.float() + h.float()).half()
    next_ref = ops.gemm_forward(ops.rmsnorm(stream_ref, w, 1e-5), *d2).float()
    # fused chain
    stream, ssq = ops.gemm_forward_ex(gate_up, *d1, flags=ops.X_GATED_SILU, add_residual=h, want_ssq=True)
    assert torch.equal(stream, stream_ref)
    tiles = (H + 255) // 256
    assert ssq.shape == (M, tiles)
    want_ssq = (stream.float() ** 2).reshape(M, tiles, -1).sum(-1)
    assert torch.allclose(ssq, want_ssq, rtol=1e-5, atol=1e-6)
    nxt, none = ops.gemm_forward_ex(stream, *d2, norm_weight=w, norm_eps=1e-5, ssq_in=ssq)
    assert none is None
    err = (nxt.float() - next_ref).abs()
    assert float(err.max()) <= 2e-3 * float(next_ref.abs().max()) + 1e-3, float(err.max())
    assert ops.workspace_is_clean(h.device)


# ------------------------------------------------------------------ GPU: the fused model

def _fused(version, max_seq_len=32):
    from autoawq_amd.checkpoint import load_quantized
    from autoawq_amd.fuser import fuse_llama

    model, _ = load_quantized(skeleton(), ckpt(version), device="cuda")
    return fuse_llama(model, max_seq_len=max_seq_len, decode_layout=None)  # the checkpoint's own layout


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["gemm", "gemv"])
def test_fused_model_prefill_then_decode_matches_reference_logits(version):
    """8 context tokens in one forward, then 4 tokens one at a time through the KV cache: the logits
    of every position equal the reference's full-sequence logits (causal attention: the same numbers)."""
    g = golden("tiny_llama_awq_gemm_outputs")
    ref = g["logits"]
    ids = torch.from_numpy(g["input_ids"]).cuda()
    lm = _fused(version)
    rng = np.abs(ref).max()
    out = lm(ids[:, :8]).float().cpu().numpy()
    assert np.abs(out - ref[:, :8]).max() <= 2e-2 * rng
    for t in range(8, 12):
        step = lm(ids[:, t:t + 1]).float().cpu().numpy()
        assert step.shape == (2, 1, 64)
        assert np.abs(step[:, 0] - ref[:, t]).max() <= 2e-2 * rng, t
    assert lm.model.blocks[0].attn.start_pos == 12


@pytest.mark.gpu
def test_fuse_llama_repacks_the_default_checkpoint_format_to_the_decode_layout_and_says_so():
    """`fuse_llama` default (decode_layout="auto"): a "gemm" checkpoint -- `AwqConfig.version`'s default -- is served on the
    WQLinear_GEMV layout (bit-exact integer repack), declared on the result; "gemv" checkpoints pass through; logits as before."""
    from autoawq_amd.checkpoint import load_quantized
    from autoawq_amd.fuser import fuse_llama
    from autoawq_amd.modules.linear import WQLinear_GEMM, WQLinear_GEMV

    g = golden("tiny_llama_awq_gemm_outputs")
    ref, ids = g["logits"], torch.from_numpy(g["input_ids"]).cuda()
    model, qc = load_quantized(skeleton(), ckpt("gemm"), device="cuda")
    assert qc.version == "gemm"
    lm = fuse_llama(model, max_seq_len=32)
    assert lm.checkpoint_layout == "gemm" and lm.decode_layout == "gemv"
    blk = lm.model.blocks[0]
    assert all(isinstance(m, WQLinear_GEMV) for m in (blk.attn.qkv_proj, blk.attn.o_proj, blk.mlp.down_proj)) and blk.mlp.gemv_layout
    assert not any(isinstance(m, WQLinear_GEMM) for m in lm.modules())
    rng = np.abs(ref).max()
    out = lm(ids[:, :8]).float().cpu().numpy()
    assert np.abs(out - ref[:, :8]).max() <= 2e-2 * rng
    step = lm(ids[:, 8:9]).float().cpu().numpy()
    assert np.abs(step[:, 0] - ref[:, 8]).max() <= 2e-2 * rng
    model2, _ = load_quantized(skeleton(), ckpt("gemv"), device="cuda")
    lm2 = fuse_llama(model2, max_seq_len=32)
    assert lm2.checkpoint_layout == "gemv" and lm2.decode_layout == "gemv"
    lm3 = _fused("gemm")
    assert lm3.checkpoint_layout == "gemm" and lm3.decode_layout == "gemm" and isinstance(lm3.model.blocks[0].attn.o_proj, WQLinear_GEMM)


@pytest.mark.gpu
def test_fused_decode_step_replayed_as_one_hipgraph():
    """Positions live in device tensors: ONE captured decode step is replayed for four tokens."""
    g = golden("tiny_llama_awq_gemm_outputs")
    ref = g["logits"]
    ids = torch.from_numpy(g["input_ids"]).cuda()
    lm = _fused("gemm")
    lm(ids[:, :8])
    pos = torch.tensor([8], dtype=torch.int32, device="cuda")
    ln = torch.tensor([9], dtype=torch.int32, device="cuda")
    for blk in lm.model.blocks:
        blk.attn.use_device_positions(pos, ln)
    tok = ids[:, 8:9].clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        lm(tok)  # warm-up (eager), then rewind
        for blk in lm.model.blocks:
            blk.attn.start_pos = 8
        lm.model.last_forward_num_tokens = 8
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            logits = lm(tok)
            pos.add_(1)
            ln.add_(1)
    pos.fill_(8)
    ln.fill_(9)
    rng = np.abs(ref).max()
    for t in range(8, 12):
        tok.copy_(ids[:, t:t + 1])
        graph.replay()
        torch.cuda.synchronize()
        assert np.abs(logits.float().cpu().numpy()[:, 0] - ref[:, t]).max() <= 2e-2 * rng, t
    assert int(pos.item()) == 12


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["gemm", "gemv"])
def test_graphed_decoder_buckets_match_reference_logits(version):
    """modules/fused/decode.py: prefill eagerly, then one hipGraph replay per token, the graph chosen by LENGTH BUCKET (bounds 10
    and 16 here, so the four steps cross a bucket boundary and use two graphs); every step's logits equal the reference's
    full-sequence logits, the device-side positions follow, and the first graph is reused when the decoder is sought back."""
    from autoawq_amd.modules.fused.decode import GraphedDecoder

    g = golden("tiny_llama_awq_gemm_outputs")
    ref = g["logits"]
    ids = torch.from_numpy(g["input_ids"]).cuda()
    lm = _fused(version)
    dec = GraphedDecoder(lm, batch=2, buckets=(10, 16))
    assert dec.bounds == [10, 16, 32] and dec.bucket(9) == 10 and dec.bucket(11) == 16 and dec.bucket(17) == 32
    rng = np.abs(ref).max()
    out = dec.prefill(ids[:, :8]).float().cpu().numpy()
    assert np.abs(out - ref[:, :8]).max() <= 2e-2 * rng and dec.position == 8
    for t in range(8, 12):
        step = dec.step(ids[:, t:t + 1]).float().cpu().numpy()
        assert step.shape == (2, 1, 64)
        assert np.abs(step[:, 0] - ref[:, t]).max() <= 2e-2 * rng, t
    assert sorted(dec.graphs) == [10, 16] and dec.position == 12 and int(dec.pos.item()) == 12 and int(dec.len.item()) == 13
    assert lm.model.blocks[0].attn.start_pos == 12
    dec.seek(8)   # the cache rows are still there: decode the same four tokens again through the cached graphs
    for t in range(8, 12):
        step = dec.step(ids[:, t:t + 1]).float().cpu().numpy()
        assert np.abs(step[:, 0] - ref[:, t]).max() <= 2e-2 * rng, t
    assert sorted(dec.graphs) == [10, 16]


@pytest.mark.gpu
@pytest.mark.parametrize("B,Hq,Hkv,T,bound", [(1, 32, 32, 65, 256), (1, 32, 32, 65, 2120), (1, 32, 8, 2049, 4096), (2, 8, 4, 300, 1024),
                                              (1, 32, 32, 1, 4096), (1, 32, 32, 1000, 1024), (3, 16, 2, 129, 130)])
def test_decode_attention_device_length_under_a_longer_bound(ops, B, Hq, Hkv, T, bound):
    """A launch sized for `bound` rows (a hipGraph's length bucket) with the true length on the device: the rows that exist are
    dealt to the launch's splits in the kernel; splits without rows contribute nothing."""
    from oracle import decoder_oracle

    gen = torch.Generator().manual_seed(B * 11 + Hq + T + bound)
    Tmax = bound + 3
    q = torch.randn((B, Hq, 128), generator=gen).half()
    kc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    vc = torch.randn((B, Tmax, Hkv, 128), generator=gen).half()
    kc[:, T:] = 100.0  # rows past the length must not be read
    vc[:, T:] = 100.0
    want = decoder_oracle.attention_reference(q.numpy(), kc.numpy(), vc.numpy(), T)
    ln = torch.tensor([T], dtype=torch.int32, device="cuda")
    got = ops.decode_attention(q.cuda(), kc.cuda(), vc.cuda(), 1, len_dev=ln, max_len=bound).cpu().numpy().astype(np.float64)
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max() + 2 ** -11


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["gemv", "gemm"])
def test_five_launch_decode_stack_matches_the_plain_module_path(layout):
    """tools/bench_decode_model.py on a small stack (2 layers, hidden 512, 4 heads of 128, intermediate 1024): the logits of the
    five-launch stream path (GEMV layout: awq_gemv_forward_ex with the norm / residual / silu-pairs prologue and epilogue;
    GEMM layout: awq_gemm_forward_ex) agree with the plain module path (separate norm, RoPE, attention, add and silu launches)
    on the same weights and cache, and the captured hipGraph replays."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_decode_model

    out = bench_decode_model.run(layers=2, contexts=(8, 40), steps=3, verbose=False, check=True, layout=layout,
                                 hidden=512, inter=1024, vocab=256, heads=4)
    assert set(out) == {8, 40} and all(v > 0 for v in out.values())
