"""Fuse a loaded Llama-style model into fused decoder blocks (reference:
`LlamaFuser.fuse_transformer`, awq/models/llama.py:100-175): per decoder layer fuse q|k|v into one
Linear (`fuse_qkv`), wrap the norms in FasterTransformerRMSNorm, the MLP in QuantFusedMLP (gate|up
in one launch), and build LlamaLikeBlock / LlamaLikeModel around them."""
import torch
import torch.nn as nn

from .modules.fused.block import LlamaLikeBlock, MixtralBlock
from .modules.fused.mlp import QuantFusedMLP
from .modules.fused.model import LlamaLikeModel, MixtralModel
from .modules.fused.moe import FusedSparseMoeBlock
from .modules.fused.norm import FasterTransformerRMSNorm
from .modules.linear import WQLinear_GEMM, WQLinear_GEMV
from .utils.fused_utils import fuse_linears, fuse_qkv


class FusedCausalLM(nn.Module):
    """`logits = lm_head(fused_model(input_ids).last_hidden_state)`: what `LlamaForCausalLM.forward`
    does with the fused model swapped in, without depending on transformers' internals."""

    def __init__(self, model, lm_head, checkpoint_layout=None, decode_layout=None):
        super().__init__()
        self.model = model
        self.lm_head = lm_head
        # declared: the layout the checkpoint came in and the one its Linears were repacked to for decoding (fuse_llama)
        self.checkpoint_layout = checkpoint_layout
        self.decode_layout = decode_layout

    @torch.inference_mode()
    def forward(self, input_ids):
        return self.lm_head(self.model(input_ids).last_hidden_state)


def _layout_of(model):
    kinds = {type(m) for m in model.modules() if isinstance(m, (WQLinear_GEMM, WQLinear_GEMV))}
    from .modules.linear import WQLinear_GEMVFast

    if any(isinstance(m, WQLinear_GEMVFast) for m in model.modules()):
        kinds.add(WQLinear_GEMVFast)
    if len(kinds) != 1:
        return None
    return {WQLinear_GEMM: "gemm", WQLinear_GEMV: "gemv", WQLinear_GEMVFast: "gemv_fast"}[kinds.pop()]


def fuse_llama(hf_model, max_seq_len=2048, decode_layout="auto"):
    """decode_layout:
      "auto" (default since round 4)  a checkpoint in the DEFAULT format (`AwqConfig.version == "gemm"`,
              awq/models/_config.py:8-27) is repacked -- same integers, bit-exact (utils/convert.py, pinned against
              reference-written checkpoints in all three formats) -- into the WQLinear_GEMV layout before fusing: the layout
              the reference itself recommends for batch 1 (README.md:96-97) and the one the row-streaming decode kernel reads
              (bs = 1: 0.50 of the HBM roofline against 0.37 on the GEMM layout).  Checkpoints that already are "gemv" or
              "gemv_fast" are left as they are.  What happened is DECLARED on the result: `.checkpoint_layout`,
              `.decode_layout`.  A GEMV-layout module serves every batch size from its OWN buffers: up to 2 - 4 rows (by K) the row-streaming kernel,
              from there to 96 rows the batched-decode kernel (csrc/gemv_batch.hip), above that a temporary GEMM-layout transpose of the
              packed words (csrc/repack.hip, freed after the call) under the fused MFMA GEMM -- no second resident copy;
      "gemv"  repack whatever the checkpoint's layout is;
      None    keep the checkpoint's layout (the reference's ExLlama repack at load time is the same kind of declared option:
              awq/modules/linear/exllama.py:66-79)."""
    cfg = hf_model.config
    src = _layout_of(hf_model.model)
    target = None
    if decode_layout is not None:
        from .utils.convert import convert_model

        mode = decode_layout.lower()
        if mode not in ("auto", "gemv"):
            raise ValueError("fuse_llama: decode_layout must be 'auto', 'gemv' or None")
        if mode == "gemv" or src == "gemm":
            convert_model(hf_model.model, "gemv")
            target = "gemv"
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
    return FusedCausalLM(fused, hf_model.lm_head, checkpoint_layout=src, decode_layout=target or src)


def fuse_mixtral(hf_model, max_seq_len=2048, decode_layout="auto"):
    """`MixtralFuser.fuse_transformer` (awq/models/mixtral.py:97-187) for a Mixtral-style model in the per-expert form AWQ
    checkpoints are written in (`model.layers[i].block_sparse_moe.{gate, top_k, experts[e].{w1, w2, w3}}`): per layer fuse
    q|k|v, wrap the norms, and -- when the experts are WQLinear_GEMM -- concatenate each expert's w1|w3 on N, stack the experts
    on a leading dim (`fuse_linears`, mixtral.py:130-151) and put a FusedSparseMoeBlock over the stacks (`:153-158`).
    decode_layout as in `fuse_llama` for the attention projections.  The expert stacks stay in the GEMM layout (the stacked
    `[E, K, N/8]` tensors of the reference: prefill-sized token counts and more than `moe.ROWS_MAX_PAIRS` pairs read those);
    with a decode layout they also get GEMV-layout TWINS (`moe.build_decode_twins`: a second resident copy of the experts, w1|w3
    with its gate / up rows interleaved), which decode-sized steps run through the row-streaming kernel -- Mixtral shape, bs = 4:
    114 us per block against 132 us on the GEMM-layout stacks.  decode_layout=None: no repack, no twins."""
    cfg = hf_model.config
    src = _layout_of(hf_model.model)
    target = src
    blocks = []
    for layer in hf_model.model.layers:
        dev = next(iter(layer.state_dict().values())).device
        att = layer.self_attn
        if decode_layout is not None:
            mode = decode_layout.lower()
            if mode not in ("auto", "gemv"):
                raise ValueError("fuse_mixtral: decode_layout must be 'auto', 'gemv' or None")
            if mode == "gemv" or isinstance(att.q_proj, WQLinear_GEMM):
                from .utils.convert import convert_linear

                for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
                    setattr(att, name, convert_linear(getattr(att, name), "gemv"))
                target = "gemv (attention projections) + gemm (experts)"
        qkv = fuse_qkv(layer, att.q_proj, att.k_proj, att.v_proj)
        norm_1 = FasterTransformerRMSNorm(layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon)
        norm_2 = FasterTransformerRMSNorm(layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon)
        moe = layer.block_sparse_moe
        if isinstance(moe.experts[0].w1, WQLinear_GEMM):
            w1w3 = [fuse_linears([e.w1, e.w3], dev) for e in moe.experts]
            moe = FusedSparseMoeBlock(top_k=moe.top_k, gate=moe.gate, ws=fuse_linears(w1w3, dev, dim=0, operation=torch.stack),
                                      w2s=fuse_linears([e.w2 for e in moe.experts], dev, dim=0, operation=torch.stack))
            if decode_layout is not None and moe.ws.qweight.is_cuda:
                from .modules.fused.moe import build_decode_twins

                build_decode_twins(moe.ws, moe.w2s)
                target = "gemv (attention projections, expert twins for decode) + gemm (experts)"
        rope_theta = getattr(cfg, "rope_theta", None)
        if rope_theta is None:
            rope_theta = (getattr(cfg, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
        blocks.append(MixtralBlock(hidden_size=cfg.hidden_size, n_heads=cfg.num_attention_heads, n_kv_heads=cfg.num_key_value_heads,
                                   qkv_layer=qkv, o_proj=att.o_proj, moe=moe, norm_1=norm_1, norm_2=norm_2, dev=dev,
                                   max_seq_len=max_seq_len, rope_theta=rope_theta))
    fused = MixtralModel(cfg.vocab_size, blocks, hf_model.model.embed_tokens,
                         FasterTransformerRMSNorm(hf_model.model.norm.weight, hf_model.model.norm.variance_epsilon))
    return FusedCausalLM(fused, hf_model.lm_head, checkpoint_layout=src, decode_layout=target)
