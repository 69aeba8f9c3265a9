"""AWQ checkpoint format: config.json + safetensors <-> WQLinear_* modules.

Golden checkpoints under tests/golden/tiny_llama_awq_* were written by the REFERENCE
(tests/golden/make_golden_checkpoint.py: its pseudo_quantize_tensor, its from_linear packers, its
save format, its CPU forward).  CPU tests pin the loader, the writer and the RTN packer against
them bit for bit; the gpu-marked tests run the loaded model through the HIP kernels and compare
with the reference's logits.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden

VERSIONS = ["gemm", "gemv", "gemv_fast"]


def skeleton():
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=1,
                      num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=64,
                      tie_word_embeddings=False)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).half().eval()


def ckpt(version):
    return os.path.join(GOLDEN, f"tiny_llama_awq_{version}")


@pytest.mark.parametrize("version", VERSIONS)
def test_config_round_trip(version):
    from autoawq_amd.checkpoint import AwqConfig

    cfg = AwqConfig.from_pretrained(ckpt(version))
    assert (cfg.version, cfg.w_bit, cfg.q_group_size, cfg.zero_point, cfg.quant_method) == (version, 4, 128, True, "awq")
    with open(os.path.join(ckpt(version), "config.json")) as f:
        assert cfg.to_transformers_dict() == json.load(f)["quantization_config"]
    assert AwqConfig.from_dict(cfg.to_dict()).to_dict() == cfg.to_dict()
    assert AwqConfig.from_dict({"version": "GEMM"}).version == "gemm"
    assert AwqConfig.from_pretrained(GOLDEN).to_dict() == AwqConfig().to_dict()  # no config.json: defaults


@pytest.mark.parametrize("version", VERSIONS)
def test_load_reference_checkpoint(version):
    """Every tensor of the reference-written checkpoint lands in the module of the same name with
    the same bits; the Linears of the decoder layer become WQLinear_<version>; lm_head stays fp16."""
    from autoawq_amd.checkpoint import VERSIONS as CLS, load_quantized, read_state_dict

    model, cfg = load_quantized(skeleton(), ckpt(version))
    state = read_state_dict(ckpt(version))
    assert len(state) > 20
    got = model.state_dict()
    for k, v in state.items():
        assert k in got and got[k].dtype == v.dtype and torch.equal(got[k], v), k
    layer = model.model.layers[0]
    for lin in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj,
                layer.mlp.gate_proj, layer.mlp.up_proj, layer.mlp.down_proj):
        assert type(lin) is CLS[version]
        assert lin.group_size == 128 and lin.w_bit == 4
    assert type(model.lm_head) is torch.nn.Linear
    assert layer.mlp.down_proj.in_features == 512 and layer.mlp.down_proj.out_features == 256


@pytest.mark.parametrize("version", VERSIONS)
def test_rtn_packer_reproduces_reference_checkpoint(version):
    """pseudo_quantize_tensor + from_linear of this repo, applied to the same seeded skeleton,
    give the reference's packed tensors bit for bit (awq/quantize/quantizer.py:74-109,228-262)."""
    from autoawq_amd.checkpoint import AwqConfig, quantize_linears_rtn, read_state_dict

    model = skeleton()
    names = quantize_linears_rtn(model, AwqConfig(version=version))
    assert len(names) == 7
    state = read_state_dict(ckpt(version))
    got = model.state_dict()
    for k, v in state.items():
        assert torch.equal(got[k], v), k


def test_save_then_load_round_trip(tmp_path):
    from autoawq_amd.checkpoint import AwqConfig, load_quantized, quantize_linears_rtn, read_state_dict, save_quantized

    cfg = AwqConfig(version="gemm", modules_to_not_convert=["o_proj"])
    model = skeleton()
    quantize_linears_rtn(model, cfg)
    assert type(model.model.layers[0].self_attn.o_proj) is torch.nn.Linear
    save_quantized(model, cfg, str(tmp_path), shard_size="200KB")
    assert os.path.exists(tmp_path / "model.safetensors.index.json")
    model2, cfg2 = load_quantized(skeleton(), str(tmp_path))
    assert cfg2.modules_to_not_convert == ["o_proj"]
    assert type(model2.model.layers[0].self_attn.o_proj) is torch.nn.Linear
    a, b = model.state_dict(), model2.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert set(read_state_dict(str(tmp_path))) == set(a)


def test_loader_errors():
    from autoawq_amd.checkpoint import AwqConfig, load_quantized

    with pytest.raises(ValueError, match="group"):
        load_quantized(skeleton(), ckpt("gemm"), AwqConfig(q_group_size=64))
    with pytest.raises((ValueError, KeyError)):
        load_quantized(skeleton(), ckpt("gemv"), AwqConfig(version="gemm"))
    with pytest.raises(ValueError, match="unsupported AWQ version"):
        load_quantized(skeleton(), ckpt("gemm"), AwqConfig(version="marlin"))
    with pytest.raises(NotImplementedError):
        load_quantized(skeleton(), ckpt("gemm"), AwqConfig(w_bit=3))


def test_oracle_matches_reference_forward_on_checkpoint(oracle):
    """The reference module's own CPU forward on real quantised weights (q_proj of the golden
    checkpoint) == the oracle on the tensors read from disk."""
    from autoawq_amd.checkpoint import read_state_dict

    st = read_state_dict(ckpt("gemm"))
    g = golden("tiny_llama_awq_gemm_outputs")
    p = "model.layers.0.self_attn.q_proj."
    y, _ = oracle.linear_gemm(g["h"], st[p + "qweight"].numpy(), st[p + "qzeros"].numpy(), st[p + "scales"].numpy(), 128, None)
    ref = g["q_out"].astype(np.float64)
    assert np.abs(y - ref).max() <= 2.0 ** -10 * np.abs(ref).max() + 1e-6  # the reference rounds its output to fp16


@pytest.mark.parametrize("version", VERSIONS)
def test_fuse_qkv_on_loaded_checkpoint(version):
    from autoawq_amd.checkpoint import load_quantized
    from autoawq_amd.utils.fused_utils import fuse_qkv

    model, _ = load_quantized(skeleton(), ckpt(version))
    att = model.model.layers[0].self_attn
    fused = fuse_qkv(att, att.q_proj, att.k_proj, att.v_proj)
    assert fused.in_features == 256 and fused.out_features == 256 + 128 + 128  # 2 q heads + 1 k + 1 v of 128
    assert type(fused) is type(att.q_proj)


# ------------------------------------------------------------------ GPU: the loaded model runs on the HIP kernels

@pytest.mark.gpu
@pytest.mark.parametrize("version", VERSIONS)
def test_loaded_model_matches_reference_logits(version):
    """Whole-model forward (embedding, attention, RMSNorm by torch; every decoder Linear by the
    HIP kernels of its layout) against the logits the reference computed on CPU from the same
    checkpoint.  fp16 end to end on both sides: 2e-2 relative to the logit range."""
    from autoawq_amd import _lib, ops
    from autoawq_amd.checkpoint import load_quantized

    _lib.lib()
    g = golden("tiny_llama_awq_gemm_outputs")
    model, _ = load_quantized(skeleton(), ckpt(version), device="cuda")
    with torch.no_grad():
        logits = model(torch.from_numpy(g["input_ids"]).cuda()).logits.float().cpu().numpy()
        h = torch.from_numpy(g["h"]).cuda()
        hin = h[None] if version == "gemv_fast" else h  # WQLinear_GEMVFast takes 3-D input only (gemv_fast.py:190)
        q_out = model.model.layers[0].self_attn.q_proj(hin).float().cpu().numpy().reshape(3, -1)
        mlp_out = model.model.layers[0].mlp(hin).float().cpu().numpy().reshape(3, -1)
    assert ops.last_kernel() != ""
    ref = g["logits"]
    assert np.abs(logits - ref).max() <= 2e-2 * np.abs(ref).max(), np.abs(logits - ref).max()
    assert (logits.argmax(-1) == ref.argmax(-1)).mean() >= 0.9
    for got, want, what in ((q_out, g["q_out"], "q_proj"), (mlp_out, g["mlp_out"], "mlp")):
        err = np.abs(got - want).max()
        assert err <= 4e-3 * np.abs(want).max() + 2e-4, (what, err, np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("version", VERSIONS)
def test_fused_qkv_of_loaded_checkpoint_on_gpu(version):
    """fuse_qkv on real checkpoint tensors: one launch over the concatenated buffers gives what the
    three projections give (awq/utils/fused_utils.py:45-142; attn.py:243 call site)."""
    from autoawq_amd.checkpoint import load_quantized
    from autoawq_amd.utils.fused_utils import fuse_qkv

    model, _ = load_quantized(skeleton(), ckpt(version), device="cuda")
    att = model.model.layers[0].self_attn
    h = torch.randn((1, 5, 256), generator=torch.Generator().manual_seed(3)).half().cuda()
    with torch.no_grad():
        sep = torch.cat([att.q_proj(h), att.k_proj(h), att.v_proj(h)], dim=-1)
        fused = fuse_qkv(att, att.q_proj, att.k_proj, att.v_proj)
        out = fused(h)
    assert out.shape == sep.shape == (1, 5, 512)
    assert (out.float() - sep.float()).abs().max() <= 2e-3 * sep.float().abs().max()


@pytest.mark.parametrize("src", VERSIONS)
@pytest.mark.parametrize("dst", VERSIONS)
def test_layout_conversion_is_bit_exact(src, dst):
    """Repacking the reference-written checkpoint of one format gives the reference-written
    checkpoint of the other format, tensor for tensor (no floating-point work on the weights)."""
    from autoawq_amd.checkpoint import load_quantized, read_state_dict
    from autoawq_amd.utils.convert import convert_model

    model, _ = load_quantized(skeleton(), ckpt(src))
    n = convert_model(model, dst)
    assert n == (0 if src == dst else 7)
    want = read_state_dict(ckpt(dst))
    got = model.state_dict()
    for k, v in want.items():
        assert got[k].shape == v.shape and got[k].dtype == v.dtype and torch.equal(got[k], v), (src, dst, k)


@pytest.mark.gpu
def test_repack_on_load_runs_the_gemm_kernels():
    """A GEMV-format checkpoint repacked to the GEMM layout on the device gives the reference's
    logits through the GEMM-layout kernels."""
    from autoawq_amd.checkpoint import load_quantized
    from autoawq_amd.modules.linear import WQLinear_GEMM

    g = golden("tiny_llama_awq_gemm_outputs")
    model, cfg = load_quantized(skeleton(), ckpt("gemv"), device="cuda", repack="gemm")
    assert cfg.version == "gemv" and type(model.model.layers[0].mlp.down_proj) is WQLinear_GEMM
    assert model.model.layers[0].mlp.down_proj.qweight.is_cuda
    with torch.no_grad():
        logits = model(torch.from_numpy(g["input_ids"]).cuda()).logits.float().cpu().numpy()
    assert np.abs(logits - g["logits"]).max() <= 2e-2 * np.abs(g["logits"]).max()


@pytest.mark.parametrize("K,N,g", [(256, 64, 128), (1408, 96, 128), (512, 40, 64), (384, 72, 32), (11008, 32, 128), (256, 256, 256)])
def test_layout_conversion_cycle_on_random_tensors(K, N, g):
    """gemm -> gemv -> gemv_fast -> gemm on random integers, zero points and scales (padded group counts:
    86 groups -> 88 columns, g = 64 / 32 round the zeros width up to 2 / 4 words): every hop preserves the
    logical (w, z, s) and the cycle returns the original buffers bit for bit."""
    from autoawq_amd.utils.convert import convert_linear, pack_linear, unpack_linear

    assert N % 4 == 0 and K % 64 == 0, "GEMVFast needs N % 4 == 0 and K % 64 == 0"
    gen = torch.Generator().manual_seed(K + N + g)
    G = K // g
    w = torch.randint(0, 16, (N, K), dtype=torch.int32, generator=gen)
    z = torch.randint(0, 16, (N, G), dtype=torch.int32, generator=gen)
    s = (torch.rand((N, G), generator=gen) * 0.02 + 0.004).half()
    bias = torch.randn(N, generator=gen).half()
    a = pack_linear("gemm", w, z, s, bias, K, N, g)
    b = convert_linear(a, "gemv")
    c = convert_linear(b, "gemv_fast")
    d = convert_linear(c, "gemm")
    for m in (a, b, c, d):
        w2, z2, s2, b2 = unpack_linear(m)
        assert torch.equal(w2, w) and torch.equal(z2, z) and torch.equal(s2, s) and torch.equal(b2, bias)
    assert torch.equal(d.qweight, a.qweight) and torch.equal(d.qzeros, a.qzeros) and torch.equal(d.scales, a.scales)
    assert convert_linear(a, "gemm") is a
