"""Fuse a loaded Llama-style model into fused decoder blocks (reference:
`LlamaFuser.fuse_transformer`, awq/models/llama.py:100-175): per decoder layer fuse q|k|v into one
Linear (`fuse_qkv`), wrap the norms in FasterTransformerRMSNorm, the MLP in QuantFusedMLP (gate|up
in one launch), and build LlamaLikeBlock / LlamaLikeModel around them."""
import torch
import torch.nn as nn

from .modules.fused.block import LlamaLikeBlock
from .modules.fused.mlp import QuantFusedMLP
from .modules.fused.model import LlamaLikeModel
from .modules.fused.norm import FasterTransformerRMSNorm
from .modules.linear import WQLinear_GEMM, WQLinear_GEMV
from .utils.fused_utils import fuse_qkv


class FusedCausalLM(nn.Module):
    """`logits = lm_head(fused_model(input_ids).last_hidden_state)`: what `LlamaForCausalLM.forward`
    does with the fused model swapped in, without depending on transformers' internals."""

    def __init__(self, model, lm_head):
        super().__init__()
        self.model = model
        self.lm_head = lm_head

    @torch.inference_mode()
    def forward(self, input_ids):
        return self.lm_head(self.model(input_ids).last_hidden_state)


def fuse_llama(hf_model, max_seq_len=2048, decode_layout=None):
    """decode_layout="gemv": the quantized Linears are repacked (same integers, bit-exact) into the WQLinear_GEMV layout
    before fusing -- the layout the reference itself recommends for batch 1 (README.md:96-97) and the one the row-streaming
    decode kernel reads; prefill-sized calls run the fused MFMA GEMM kernels on a cached GEMM-layout copy.  A declared
    option like the reference's ExLlama repack at load time (awq/modules/linear/exllama.py:66-79); None keeps the
    checkpoint's layout."""
    cfg = hf_model.config
    if decode_layout is not None:
        from .utils.convert import convert_model

        if decode_layout.lower() != "gemv":
            raise ValueError("fuse_llama: decode_layout must be None or 'gemv'")
        convert_model(hf_model.model, "gemv")
    blocks = []
    for layer in hf_model.model.layers:
        dev = next(iter(layer.state_dict().values())).device
        att = layer.self_attn
        qkv = fuse_qkv(layer, att.q_proj, att.k_proj, att.v_proj)
        norm_1 = FasterTransformerRMSNorm(layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon)
        norm_2 = FasterTransformerRMSNorm(layer.post_attention_layernorm.weight,
                                          layer.post_attention_layernorm.variance_epsilon)
        mlp = layer.mlp
        if isinstance(mlp.gate_proj, (WQLinear_GEMM, WQLinear_GEMV)):
            mlp = QuantFusedMLP(mlp.gate_proj, mlp.down_proj, mlp.up_proj)
        rope_theta = getattr(cfg, "rope_theta", None)
        if rope_theta is None:
            rope_theta = (getattr(cfg, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
        blocks.append(LlamaLikeBlock(hidden_size=cfg.hidden_size, n_heads=cfg.num_attention_heads,
                                     n_kv_heads=cfg.num_key_value_heads, qkv_layer=qkv, o_proj=att.o_proj, mlp=mlp,
                                     norm_1=norm_1, norm_2=norm_2, dev=dev, max_seq_len=max_seq_len,
                                     rope_theta=rope_theta, head_dim=getattr(cfg, "head_dim", None)))
    fused = LlamaLikeModel(cfg.vocab_size, blocks, hf_model.model.embed_tokens,
                           FasterTransformerRMSNorm(hf_model.model.norm.weight, hf_model.model.norm.variance_epsilon))
    return FusedCausalLM(fused, hf_model.lm_head)
