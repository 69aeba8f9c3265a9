#!/usr/bin/env python3
"""Generate tests/golden/tiny_llama_awq_{gemm,gemv,gemv_fast}/ by RUNNING the reference.

A 1-layer Llama skeleton (transformers, random init, seed 0) is quantised with the reference's own
arithmetic -- `AwqQuantizer.pseudo_quantize_tensor` (awq/quantize/quantizer.py:74-109) and
`WQLinear_*.from_linear` (awq/modules/linear/gemm.py:171-251, gemv.py:77-154,
gemv_fast.py:127-183), the same calls `_apply_quant` makes (quantizer.py:228-262) -- and written the
way `save_quantized` writes it (awq/models/base.py:274-319: config.json with the transformers
form of AwqConfig + `save_torch_state_dict`).  For the GEMM version the logits of the reference
model (its CPU naive branch, awq/modules/linear/gemm.py:71-79) on a fixed prompt are stored too.

Run by hand in the build container (the only place /root/reference exists):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_checkpoint.py
"""
import json
import os
import shutil
import sys
import types

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
from huggingface_hub import save_torch_state_dict
from transformers import LlamaConfig, LlamaForCausalLM

import awq.modules.linear.gemm as ref_gemm
from awq.models._config import AwqConfig
from awq.modules.linear.gemm import WQLinear_GEMM
from awq.modules.linear.gemv import WQLinear_GEMV
from awq.modules.linear.gemv_fast import WQLinear_GEMVFast
from awq.quantize.quantizer import AwqQuantizer
from awq.utils.module import get_named_linears, set_op_by_name

ref_gemm.TRITON_AVAILABLE = False  # CPU naive branch
ref_gemm.user_has_been_warned = True
HERE = os.path.dirname(os.path.abspath(__file__))
G = 128


def skeleton():
    cfg = LlamaConfig(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=1,
                      num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=64,
                      tie_word_embeddings=False)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).half().eval()


for version, cls in (("gemm", WQLinear_GEMM), ("gemv", WQLinear_GEMV), ("gemv_fast", WQLinear_GEMVFast)):
    model = skeleton()
    qself = types.SimpleNamespace(w_bit=4, group_size=G, zero_point=True)
    with torch.no_grad():
        for layer in model.model.layers:
            for name, lin in get_named_linears(layer).items():
                lin.weight.data, scales, zeros = AwqQuantizer.pseudo_quantize_tensor(qself, lin.weight.data)
                if version == "gemm":
                    scales, zeros = scales.t().contiguous(), zeros.t().contiguous()
                q = cls.from_linear(linear=lin, w_bit=4, group_size=G, init_only=False, scales=scales, zeros=zeros)
                set_op_by_name(layer, name, q)
    out = os.path.join(HERE, f"tiny_llama_awq_{version}")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    qc = AwqConfig(q_group_size=G, w_bit=4, version=version, zero_point=True)
    d = model.config.to_dict()
    d["quantization_config"] = qc.to_transformers_dict()
    with open(os.path.join(out, "config.json"), "w") as f:
        json.dump(d, f, indent=2, sort_keys=True, default=str)
    save_torch_state_dict(state_dict=model.state_dict(), save_directory=out, max_shard_size="300KB" if version == "gemm" else "5GB",
                          safe_serialization=True, force_contiguous=True)
    if version == "gemm":  # reference forward on CPU
        ids = (torch.arange(24).reshape(2, 12) * 7 + 3) % 64
        with torch.no_grad():
            logits = model(ids).logits
            h = torch.randn(3, 256, generator=torch.Generator().manual_seed(1)).half()
            mlp_out = model.model.layers[0].mlp(h)
            q_out = model.model.layers[0].self_attn.q_proj(h)
        np.savez_compressed(os.path.join(HERE, "tiny_llama_awq_gemm_outputs.npz"), input_ids=ids.numpy(),
                            logits=logits.float().numpy(), h=h.numpy(), mlp_out=mlp_out.float().numpy(),
                            q_out=q_out.float().numpy())
    print(version, sorted(os.listdir(out)), sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)))

# RoPE of the fused attention (awq/modules/fused/attn.py:27-87), run as the reference runs it
from awq.modules.fused.attn import RoPE  # noqa: E402

g = torch.Generator().manual_seed(5)
cases = {}
for name, (D, rot, start, S) in {"full": (128, 128, 3, 5), "partial": (128, 64, 0, 4), "late": (128, 128, 57, 1)}.items():
    xq = torch.randn((2, S, 4, D), generator=g).half()
    xk = torch.randn((2, S, 2, D), generator=g).half()
    rope = RoPE(rot, 64, "cpu", 10000.0)
    q, k = rope.forward(xq, xk, start, S, partial=rot < D)
    cases.update({f"{name}_xq": xq.numpy(), f"{name}_xk": xk.numpy(), f"{name}_q": q.numpy(), f"{name}_k": k.numpy(),
                  f"{name}_meta": np.array([D, rot, start, S])})
np.savez_compressed(os.path.join(HERE, "rope_golden.npz"), **cases)
print("rope golden", sorted(cases)[:4])
