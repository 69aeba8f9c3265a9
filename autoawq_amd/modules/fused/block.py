"""LlamaLikeBlock for MI355X (reference: awq/modules/fused/block.py:57-119): same constructor, same
dataflow `h = x + attn(norm_1(x)); out = h + mlp(norm_2(h))`.  The first residual add and norm_2
are one launch (`FasterTransformerRMSNorm.forward(x, residual=...)`); the sum lands in the buffer
`o_proj` allocated, so the caller's `hidden_states` is never written."""
import torch
import torch.nn as nn

from ... import ops
from .attn import QuantAttentionFused
from .norm import FasterTransformerRMSNorm


class MixtralBlock(nn.Module):
    """Reference: awq/modules/fused/block.py:6-55 -- same constructor, same dataflow
    `h = x + attn(norm_1(x)); out = h + moe(norm_2(h))`, `moe` a FusedSparseMoeBlock (modules/fused/moe.py: routing in one launch,
    grouped int4 GEMMs over the experts that were hit, no routing data read back below the prefill threshold, so a decode step
    is hipGraph-capturable).  As in LlamaLikeBlock the first residual add rides on norm_2 (`FasterTransformerRMSNorm(x,
    residual=...)`: one launch, the sum lands in o_proj's own output buffer)."""

    def __init__(self, hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, moe, norm_1, norm_2, dev, max_seq_len, rope_theta):
        super().__init__()
        self.n_heads = n_heads
        self.n_kv_heads = n_kv_heads
        self.hidden_size = hidden_size
        self.norm_1 = norm_1.to(dev)
        self.attn = QuantAttentionFused(self.hidden_size, self.n_heads, self.n_kv_heads, qkv_layer, o_proj, dev=dev,
                                        max_seq_len=max_seq_len, use_alibi=False, rope_theta=rope_theta).to(dev)
        self.norm_2 = norm_2.to(dev)
        self.moe = moe
        self.device = dev

    def forward(self, hidden_states):
        norm_out = self.norm_1(hidden_states)
        attn_output, _, _ = self.attn.forward(hidden_states=norm_out)
        if (isinstance(self.norm_2, FasterTransformerRMSNorm) and attn_output.dtype == hidden_states.dtype == torch.float16
                and attn_output.is_contiguous() and hidden_states.is_contiguous()):
            h = attn_output  # becomes hidden_states + attn_output, in place in o_proj's own output buffer
            normed = self.norm_2(hidden_states, residual=h)
        else:
            h = hidden_states.to(attn_output.device) + attn_output
            normed = self.norm_2(h)
        return h + self.moe.forward(normed)


class LlamaLikeBlock(nn.Module):
    def __init__(self, hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, mlp, norm_1, norm_2, dev, max_seq_len,
                 rope_theta=10000, partial_rotary_factor=1.0, use_alibi=False, head_dim=None):
        super().__init__()
        self.n_heads = n_heads
        self.n_kv_heads = n_kv_heads
        self.head_dim = head_dim if head_dim else hidden_size // n_heads
        self.hidden_size = hidden_size
        self.norm_1 = norm_1.to(dev)
        self.attn = QuantAttentionFused(self.hidden_size, self.n_heads, self.n_kv_heads, qkv_layer, o_proj, dev=dev,
                                        max_seq_len=max_seq_len, use_alibi=use_alibi, rope_theta=rope_theta,
                                        partial_rotary_factor=partial_rotary_factor, head_dim=head_dim).to(dev)
        self.norm_2 = norm_2.to(dev)
        self.mlp = mlp.to(dev)
        self.device = dev

    def forward(self, hidden_states):
        norm_out = self.norm_1(hidden_states)
        attn_output, _, _ = self.attn.forward(hidden_states=norm_out)
        if (isinstance(self.norm_2, FasterTransformerRMSNorm) and attn_output.dtype == hidden_states.dtype == torch.float16
                and attn_output.is_contiguous() and hidden_states.is_contiguous()):
            h = attn_output  # becomes hidden_states + attn_output, in place in o_proj's own output buffer
            normed = self.norm_2(hidden_states, residual=h)
        else:
            h = hidden_states.to(attn_output.device) + attn_output
            normed = self.norm_2(h)
        return h + self.mlp.forward(normed)

    FOLD_NORMS_INTO_PROJECTIONS = True  # decode batches (rows <= 4), GEMM-layout qkv and gate|up

    def _can_fold(self, h):
        from ..linear.gemm import WQLinear_GEMM
        from .mlp import QuantFusedMLP

        rows = h.numel() // h.shape[-1]
        return (self.FOLD_NORMS_INTO_PROJECTIONS and rows <= 4 and h.shape[1] == 1 and h.dtype == torch.float16
                and isinstance(self.attn.qkv_proj, WQLinear_GEMM) and isinstance(self.mlp, QuantFusedMLP)
                and not self.mlp.gemv_layout and self.attn.qkv_proj.in_features % 32 == 0)

    def _can_fold_gemv(self, h):
        """GEMV-layout decode (one row): the four projections run the row-streaming kernel with the block's norm, residual add
        and silu * mul in their prologue / epilogue (ops.gemv_forward_ex)."""
        from ..linear.gemv import WQLinear_GEMV
        from .mlp import QuantFusedMLP

        if getattr(self, "_gemv_fold_refused", False) or not self.FOLD_NORMS_INTO_PROJECTIONS:
            return False
        a, m = self.attn, self.mlp
        if not (h.numel() == h.shape[-1] and h.dtype == torch.float16 and isinstance(m, QuantFusedMLP) and m.gemv_layout
                and m.activation is torch.nn.functional.silu
                and all(isinstance(l, WQLinear_GEMV) and l.bias is None for l in (a.qkv_proj, a.o_proj, m.down_proj))):
            return False
        # what awq_gemv_forward_ex takes (include/awq_hip.h): a wave covers whole rows (K <= 16384), groups of 128 k, and with a
        # residual at most 64 rows per wave; every projection of the block must qualify, or the block runs the separate launches
        # (the norm prologue, which qkv and gate|up carry, stops at K = 12288: eight 1-KiB slots per wave leave no registers for it)
        return (all(l.in_features <= 16384 and l.group_size % 128 == 0 and l.out_features <= 65536 and l.in_features >= 128
                    for l in (a.qkv_proj, a.o_proj, m.down_proj)) and 2 * m.intermediate_size <= 131072
                and a.qkv_proj.in_features <= 12288 and m.in_features <= 12288)

    def _forward_stream_gemv(self, h):
        B, S, H = h.shape
        q, o, d, n1, n2 = self.attn.qkv_proj, self.attn.o_proj, self.mlp.down_proj, self.norm_1, self.norm_2
        h0 = h.reshape(1, H)
        xqkv = ops.gemv_forward_ex(h0, q.qweight, q.scales, q.qzeros, q.group_size, norm_weight=n1.weight, norm_eps=n1.variance_epsilon)
        heads, _, _ = self.attn.forward_qkv(xqkv.reshape(B, S, -1), apply_o_proj=False)
        h1 = ops.gemv_forward_ex(heads.reshape(1, -1), o.qweight, o.scales, o.qzeros, o.group_size, add_residual=h0)
        pw, ps, pz = self.mlp.gate_up_pairs()
        act = ops.gemv_forward_ex(h1, pw, ps, pz, self.mlp.group_size, norm_weight=n2.weight, norm_eps=n2.variance_epsilon, silu_pairs=True)
        h2 = ops.gemv_forward_ex(act, d.qweight, d.scales, d.qzeros, d.group_size, add_residual=h1)
        return h2.reshape(B, S, H)

    def forward_stream(self, x, h, ssq=None):
        """Residual stream kept by the caller (LlamaLikeModel): `h` is the stream, `x` a previous MLP output
        that has not been added yet (or None), `ssq` the per-tile sums of squares of `h` if the projection
        that produced it handed them over.  Decode batches run FIVE launches per block: qkv (norm folded
        into its staging), attention (RoPE + cache append inside), o_proj (residual add + sums of squares in
        its epilogue), gate|up (norm from those sums), down (silu*mul while staging, residual add + sums of
        squares for the next block).  Returns (pending x | None, stream, ssq | None)."""
        if self._can_fold_gemv(h):
            # FIVE launches on the GEMV layout as well: qkv (norm in its prologue), attention, o_proj (+ residual), gate|up
            # (norm in, silu * mul out), down (+ residual)
            from ..._lib import AwqHipError

            start = self.attn.start_pos
            try:
                return None, self._forward_stream_gemv(h if x is None else h + x), None
            except AwqHipError as e:  # a shape the row-streaming kernel does not take (K > 16384, group size < 128): separate launches
                if getattr(e, "code", 0) != -3 or self.attn.start_pos != start:
                    raise
                self._gemv_fold_refused = True
        if self._can_fold(h):
            B, S, H = h.shape
            q, o, d = self.attn.qkv_proj, self.attn.o_proj, self.mlp.down_proj
            n1, n2 = self.norm_1, self.norm_2
            if ssq is not None and x is None:
                xqkv, _ = ops.gemm_forward_ex(h.reshape(B * S, H), q.qweight, q.scales, q.qzeros, q.bias,
                                              norm_weight=n1.weight, norm_eps=n1.variance_epsilon, ssq_in=ssq)
            elif x is None:
                xqkv, _ = ops.gemm_forward_normed(h.reshape(B * S, H), n1.weight, n1.variance_epsilon, q.qweight, q.scales,
                                                  q.qzeros, q.bias)
            else:
                xqkv, hn = ops.gemm_forward_normed(x.reshape(B * S, H), n1.weight, n1.variance_epsilon, q.qweight, q.scales,
                                                   q.qzeros, q.bias, residual=h.reshape(B * S, H))
                h = hn.reshape(B, S, H)
            from ..linear.gemm import WQLinear_GEMM

            chain = isinstance(o, WQLinear_GEMM) and isinstance(d, WQLinear_GEMM) and self.mlp.activation is torch.nn.functional.silu
            if chain:
                heads, _, _ = self.attn.forward_qkv(xqkv.reshape(B, S, -1), apply_o_proj=False)
                h1, ssq1 = ops.gemm_forward_ex(heads.reshape(B * S, -1), o.qweight, o.scales, o.qzeros, o.bias,
                                               add_residual=h.reshape(B * S, H), want_ssq=True)
                qw, sc, qz = self.mlp._gate_up_fused()
                gate_up, _ = ops.gemm_forward_ex(h1, qw, sc, qz, norm_weight=n2.weight, norm_eps=n2.variance_epsilon, ssq_in=ssq1)
                h2, ssq2 = ops.gemm_forward_ex(gate_up, d.qweight, d.scales, d.qzeros, d.bias, flags=ops.X_GATED_SILU,
                                               add_residual=h1, want_ssq=True)
                return None, h2.reshape(B, S, H), ssq2
            attn_output, _, _ = self.attn.forward_qkv(xqkv.reshape(B, S, -1))
            qw, sc, qz = self.mlp._gate_up_fused()
            gate_up, hn = ops.gemm_forward_normed(attn_output.reshape(B * S, H), n2.weight, n2.variance_epsilon, qw, sc, qz,
                                                  residual=h.reshape(B * S, H))
            return self.mlp.forward(attn_output, gate_up=gate_up), hn.reshape(B, S, H), None
        norm_out = self.norm_1(h) if x is None else self.norm_1(x, residual=h)
        attn_output, _, _ = self.attn.forward(hidden_states=norm_out)
        normed = self.norm_2(attn_output, residual=h)
        return self.mlp.forward(normed), h, None


# ---------------------------------------------------------------------------------------------------------------------------
# The other block families of awq/modules/fused/block.py (round 4): the same constructors and dataflows over this package's
# QuantAttentionFused (whose feature surface -- ALiBi, logit soft-capping, q / k norms, custom attention shapes -- round 3 built and
# pinned).  They run the plain module path (separate norm / attention / MLP launches); the five-launch folded decode path is
# LlamaLikeBlock's.  Not built (and no class of that name is shipped: round 5): CohereBlock (block.py:264-320) -- it asks for the INTERLEAVED rotary form (`is_neox=False`), which
# awq_rope_kv_append does not implement -- and Phi-3's `rope_scaling` (long-rope factors): both raise instead of computing
# something else.

def _add(a, b):
    return a.to(b.device) + b


class QwenBlock(nn.Module):
    """awq/modules/fused/block.py:122-188 (Qwen2 / Qwen3: q_norm / k_norm on the heads, optional separate head_dim)."""

    def __init__(self, hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, mlp, norm_1, norm_2, dev, max_seq_len,
                 rope_theta=10000, partial_rotary_factor=1.0, use_alibi=False, head_dim=None, q_norm=None, k_norm=None):
        super().__init__()
        self.n_heads, self.n_kv_heads, self.hidden_size = n_heads, n_kv_heads, hidden_size
        self.head_dim = head_dim if head_dim else hidden_size // n_heads
        self.norm_1 = norm_1.to(dev)
        self.attn = QuantAttentionFused(hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, dev=dev, max_seq_len=max_seq_len,
                                        use_alibi=use_alibi, rope_theta=rope_theta, partial_rotary_factor=partial_rotary_factor,
                                        head_dim=head_dim, q_norm=q_norm, k_norm=k_norm).to(dev)
        self.norm_2 = norm_2.to(dev)
        self.mlp = mlp.to(dev)
        self.device = dev

    def forward(self, hidden_states):
        attn_output, _, _ = self.attn.forward(hidden_states=self.norm_1(hidden_states))
        h = _add(hidden_states, attn_output)
        return h + self.mlp.forward(self.norm_2(h))


class Gemma2LikeBlock(nn.Module):
    """awq/modules/fused/block.py:190-262 (Gemma-2: a norm before AND after attention and MLP, logit soft-capping)."""

    def __init__(self, hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, mlp, norm_1, norm_2, norm_3, norm_4, dev, max_seq_len,
                 rope_theta=10000, partial_rotary_factor=1.0, use_alibi=False, head_dim=None, attn_logit_softcapping=None):
        super().__init__()
        self.n_heads, self.n_kv_heads, self.hidden_size = n_heads, n_kv_heads, hidden_size
        self.head_dim = head_dim if head_dim else hidden_size // n_heads
        self.norm_1 = norm_1.to(dev)
        self.attn = QuantAttentionFused(hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, dev=dev, max_seq_len=max_seq_len,
                                        use_alibi=use_alibi, rope_theta=rope_theta, partial_rotary_factor=partial_rotary_factor,
                                        head_dim=head_dim, attn_logit_softcapping=attn_logit_softcapping or 0.0).to(dev)
        self.norm_2 = norm_2.to(dev)
        self.norm_3 = norm_3.to(dev)
        self.mlp = mlp.to(dev)
        self.norm_4 = norm_4.to(dev)
        self.device = dev

    def forward(self, hidden_states):
        attn_output, _, _ = self.attn.forward(hidden_states=self.norm_1(hidden_states))
        h = _add(hidden_states, self.norm_2(attn_output))
        return h + self.norm_4(self.mlp(self.norm_3(h)))


class Phi3Block(nn.Module):
    """awq/modules/fused/block.py:489-544 (Phi-3: the checkpoint's qkv_proj is already fused, the MLP's gate_up_proj too)."""

    def __init__(self, hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, mlp, norm_1, norm_2, dev, max_seq_len,
                 rope_theta=10000, rope_scaling=None, use_alibi=False, head_dim=None):
        super().__init__()
        if rope_scaling is not None:
            raise NotImplementedError("Phi3Block: rope_scaling (long-rope factors) is not implemented by awq_rope_kv_append")
        self.n_heads, self.n_kv_heads, self.hidden_size = n_heads, n_kv_heads, hidden_size
        self.head_dim = head_dim if head_dim else hidden_size // n_heads
        self.norm_1 = norm_1.to(dev)
        self.attn = QuantAttentionFused(hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, dev=dev, max_seq_len=max_seq_len,
                                        use_alibi=use_alibi, rope_theta=rope_theta, head_dim=head_dim).to(dev)
        self.norm_2 = norm_2.to(dev)
        self.mlp = mlp.to(dev)
        self.device = dev

    def forward(self, hidden_states):
        attn_output, _, _ = self.attn.forward(hidden_states=self.norm_1(hidden_states))
        h = _add(hidden_states, attn_output)
        return h + self.mlp.forward(self.norm_2(h))


class MPTBlock(nn.Module):
    """awq/modules/fused/block.py:322-368 (MPT: multi-head attention with ALiBi, no rotation; LayerNorms; `ffn`)."""

    def __init__(self, hidden_size, n_heads, qkv_layer, o_proj, mpt_mlp, norm_1, norm_2, dev, max_seq_len):
        super().__init__()
        self.n_heads, self.n_kv_heads, self.hidden_size = n_heads, 0, hidden_size
        self.norm_1 = norm_1
        self.attn = QuantAttentionFused(hidden_size, n_heads, 0, qkv_layer, o_proj, dev=dev, max_seq_len=max_seq_len, use_alibi=True).to(dev)
        self.norm_2 = norm_2
        self.ffn = mpt_mlp.to(dev)
        self.device = dev

    def forward(self, hidden_states):
        attn_output, _, _ = self.attn.forward(hidden_states=self.norm_1(hidden_states))
        h = _add(hidden_states, attn_output)
        return h + self.ffn.forward(self.norm_2(h))


class FalconDecoderLayer(nn.Module):
    """awq/modules/fused/block.py:371-487 (Falcon: attention and MLP in PARALLEL on the block input; new architecture: 8 KV heads and a
    LayerNorm per branch; old architecture: multi-query attention, one shared LayerNorm, custom attention shapes)."""

    def __init__(self, hidden_size, n_heads, qkv_layer, o_proj, mlp, dev, max_seq_len, input_layernorm=None, ln_attn=None, ln_mlp=None,
                 new_decoder_arch=True):
        super().__init__()
        self.n_heads, self.hidden_size, self.new_decoder_arch = n_heads, hidden_size, new_decoder_arch
        self.n_kv_heads = 8 if new_decoder_arch else 0
        head_dim = hidden_size // n_heads
        shapes = None
        if not new_decoder_arch:  # one shared K / V head behind the query heads of the fused row (block.py:428-460)
            shapes = {"xqkv_view": (n_heads + 2, head_dim),
                      "xq_slice": lambda xqkv: xqkv[:, :, :-2], "xk_slice": lambda xqkv: xqkv[:, :, [-2]], "xv_slice": lambda xqkv: xqkv[:, :, [-1]],
                      "xq_view": (n_heads, head_dim), "xk_view": (1, head_dim), "xv_view": (1, head_dim), "xk_reshape": (1, head_dim // 8, 8),
                      "single_xq_view": (n_heads, head_dim), "single_xk_view": (1, head_dim), "single_xv_view": (1, head_dim)}
        self.attention_shapes = shapes
        self.attn = QuantAttentionFused(hidden_size, n_heads, self.n_kv_heads if new_decoder_arch else 1, qkv_layer, o_proj, dev=dev,
                                        max_seq_len=max_seq_len, use_alibi=False, attention_shapes=shapes).to(dev)
        if new_decoder_arch:
            self.ln_attn, self.ln_mlp = ln_attn, ln_mlp
        else:
            self.input_layernorm = input_layernorm
        self.mlp = mlp
        self.device = dev

    def forward(self, hidden_states):
        if self.new_decoder_arch:
            attn_in, mlp_in = self.ln_attn(hidden_states), self.ln_mlp(hidden_states)
        else:
            attn_in = mlp_in = self.input_layernorm(hidden_states)
        attn_output, _, _ = self.attn.forward(hidden_states=attn_in)
        return _add(hidden_states, attn_output) + self.mlp.forward(mlp_in)
