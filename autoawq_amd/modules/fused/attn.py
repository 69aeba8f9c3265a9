"""QuantAttentionFused for MI355X (reference: awq/modules/fused/attn.py:127-312).

Same constructor arguments and the same forward contract: `forward(hidden_states)` runs the fused
qkv projection, rotary embedding, the KV-cache update, attention and `o_proj`, advances
`start_pos`, and returns `(attn_output, attention_weight, past_key_value)`.

MI355X-first differences:
  * RoPE + cache update is ONE kernel on the fused qkv output (`awq_rope_kv_append`) instead of
    a complex-multiply in torch followed by two slice assignments (attn.py:54-87, cache.py:40-45);
  * decode (one new token per sequence) runs `awq_decode_attention` over the cache -- the role
    flash_attn_with_kvcache has in the reference (attn.py:286-302).  Prefill (seqlen > 1) uses
    torch's scaled_dot_product_attention, where the reference calls flash_attn_func (:277-285):
    vendor attention is plumbing here, the int4 projections are the product;
  * `use_device_positions(pos, length)`: start position and length live in device int32 tensors so
    that one captured hipGraph can be replayed for every decode step.
  * the feature surface of attn.py:89-125,159-203 rides on the same kernels: `attention_shapes` (how the fused row is viewed and
    sliced, awq/utils/fused_utils.py:165-201) and `q_norm` / `k_norm` are applied to the projection's output, which is then put
    back into the q | k | v head order the kernels read; ALiBi models skip the rotation (rotary width 0 in the append kernel)
    and pass their slopes, `attn_logit_softcapping` its cap, to `awq_decode_attention_ex` -- what the reference hands
    flash_attn_with_kvcache as `alibi_slopes=` / `softcap=` (attn.py:286-302).  Prefill with either modifier computes the
    scores in torch (fp32), where the reference calls flash_attn_func with the same arguments (attn.py:277-285)."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from .cache import WindowedCache


class RoPE(nn.Module):
    """cos / sin tables [max_seq_len, rotary_dim / 2] fp32: the real and imaginary parts of
    `precompute_freqs_cis` (attn.py:37-43)."""

    def __init__(self, head_dim, max_seq_len, device, rope_theta):
        super().__init__()
        self.head_dim = head_dim
        freqs = 1.0 / (rope_theta ** (torch.arange(0, head_dim, 2)[: (head_dim // 2)].float() / head_dim))
        t = torch.arange(max_seq_len)
        freqs = torch.outer(t, freqs).float()
        self.register_buffer("cos", freqs.cos().contiguous().to(device), persistent=False)
        self.register_buffer("sin", freqs.sin().contiguous().to(device), persistent=False)


class ALiBi(nn.Module):
    """Slopes [n_heads] and the bias row of the last query [1, n_heads, 1, max_seq_len] (attn.py:89-125): head h adds
    slope_h * (key position - query position) to its scores.  `forward(scores, seqlen)` like the reference's."""

    def __init__(self, n_heads, max_seq_len, device, alibi_bias_max=8):
        super().__init__()
        slopes, bias = self.build_alibi_bias(n_heads, max_seq_len, alibi_bias_max=alibi_bias_max)
        self.slopes = nn.Parameter(slopes.float().to(device), requires_grad=False)
        self.bias = nn.Parameter(bias.float().to(device), requires_grad=False)

    @staticmethod
    def gen_slopes(n_heads, alibi_bias_max=8):
        pow2 = 1 << max(0, math.ceil(math.log2(n_heads)))
        exponents = torch.arange(1, pow2 + 1, dtype=torch.float32) * (alibi_bias_max / pow2)
        slopes = torch.pow(2.0, exponents).reciprocal()
        if pow2 != n_heads:  # the odd exponents first, then the even ones, cut to n_heads
            slopes = torch.cat([slopes[1::2], slopes[0::2]])[:n_heads]
        return slopes.view(1, n_heads, 1, 1)

    @staticmethod
    def build_alibi_bias(n_heads, seq_len, alibi_bias_max=8, dtype=torch.float32):
        slopes = ALiBi.gen_slopes(n_heads, alibi_bias_max)
        distance = torch.arange(1 - seq_len, 1, dtype=torch.int32).view(1, 1, 1, seq_len)
        return slopes.reshape(n_heads).to(dtype), (distance * slopes).to(dtype)

    def forward(self, scores, seqlen):
        scores += self.bias[..., :seqlen]
        return scores


class QuantAttentionFused(nn.Module):
    FUSE_ROPE_INTO_ATTENTION = True  # decode steps, head_dim 128, full rotary
    PREFILL_KERNEL = True            # prefill steps, head_dim 128: csrc/prefill_attn.hip (False: the vendor's attention, for A/B runs)

    def __init__(self, hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, dev, max_seq_len=2048, use_alibi=False,
                 attention_shapes=None, rope_theta=10000, partial_rotary_factor=1.0, head_dim=None,
                 attn_logit_softcapping=0.0, q_norm=None, k_norm=None, **kwargs):
        super().__init__()
        from ...utils.fused_utils import get_attention_shapes

        self.hidden_size = hidden_size
        self.n_heads = n_heads
        self.n_kv_heads = n_kv_heads if n_kv_heads != 0 else n_heads
        self.n_kv_groups = n_heads // self.n_kv_heads
        self.head_dim = head_dim if head_dim is not None else hidden_size // n_heads
        self.qkv_proj = qkv_layer
        self.o_proj = o_proj
        self.start_pos = 0
        self.use_alibi = bool(use_alibi)
        self.q_norm, self.k_norm = q_norm, k_norm
        self.attn_logit_softcapping = float(attn_logit_softcapping or 0.0)
        self._custom_shapes = attention_shapes is not None
        self.attention_shapes = get_attention_shapes(attention_shapes, n_heads, n_kv_heads, self.head_dim)
        self.cache_batch_size = int(os.getenv("AWQ_BATCH_SIZE", "1"))
        if kwargs.get("max_length") is not None:
            max_seq_len = kwargs["max_length"]
        self.max_seq_len = max_seq_len
        self.is_hf_transformers = False
        self.rope_theta = rope_theta
        self.cache = WindowedCache(self.cache_batch_size, n_heads, self.n_kv_heads, self.head_dim, max_seq_len, dev)
        self.partial_rotary_factor = partial_rotary_factor
        if self.use_alibi:  # no rotation (attn.py:189-194,254): the append kernel runs with rotary width 0
            self.alibi = ALiBi(n_heads, max_seq_len, dev)
            self.rotary_dim = 0
            self.rope = RoPE(2, 1, dev, rope_theta)  # (tables the kernel never reads)
            self.is_neox = False
        else:
            self.alibi = None
            self.rotary_dim = int(self.head_dim * partial_rotary_factor)
            self.rope = RoPE(self.rotary_dim, max_seq_len, dev, rope_theta)
            self.is_neox = True
        self._pos_dev = None
        self._len_dev = None

    def use_device_positions(self, pos_dev, len_dev):
        """pos_dev / len_dev: int32 device tensors of one element (shared by all layers) holding the
        start position and the cache length after the append; the caller advances them."""
        self._pos_dev, self._len_dev = pos_dev, len_dev

    def _len_bound(self):
        """What a decode launch with a DEVICE-side length is sized for: the cache window, or the tighter bound a caller that
        knows the context on the host has set (`decode_len_bound`, modules/fused/decode.py: one hipGraph per length bucket)."""
        b = getattr(self, "decode_len_bound", None)
        return self.max_seq_len if b is None else max(1, min(int(b), self.max_seq_len))

    def _resize_cache(self, bsz):
        if bsz != self.cache_batch_size:
            if bsz > self.cache_batch_size:
                self.cache.increase_batch_size(bsz)
            else:
                self.cache.decrease_batch_size(bsz)
            self.cache_batch_size = bsz
            self.start_pos = 0  # the reference resets on a batch-size change (attn.py:208-218)

    def forward(self, hidden_states, *args, **kwargs):
        bsz, seqlen, _ = hidden_states.shape
        self._resize_cache(bsz)
        xqkv = self.qkv_proj(hidden_states)
        if xqkv.dtype != torch.float16:
            xqkv = xqkv.half()
        return self.forward_qkv(xqkv)

    def _qkv_rows(self, xqkv):
        """The projection's output as the kernels read it: [B, S, (Hq + 2 Hkv) * D], q heads | k heads | v heads.  Custom
        `attention_shapes` and q / k norms are applied on the reference's views (attn.py:244-253) and the three tensors joined
        again; without them (the default view IS that order) the tensor passes through untouched."""
        if not self._custom_shapes and self.q_norm is None and self.k_norm is None:
            return xqkv
        B, S, _ = xqkv.shape
        sh = self.attention_shapes
        v = xqkv.view((B, S) + tuple(sh["xqkv_view"]))
        xq, xk, xv = sh["xq_slice"](v), sh["xk_slice"](v), sh["xv_slice"](v)
        xq = xq.reshape(B, S, self.n_heads, self.head_dim)
        xk = xk.reshape(B, S, self.n_kv_heads, self.head_dim)
        xv = xv.reshape(B, S, self.n_kv_heads, self.head_dim)
        if self.q_norm is not None:
            xq = self.q_norm(xq)
        if self.k_norm is not None:
            xk = self.k_norm(xk)
        return torch.cat([xq.reshape(B, S, -1), xk.reshape(B, S, -1), xv.reshape(B, S, -1)], dim=-1).to(torch.float16).contiguous()

    def _attend_torch(self, q, k, v, first_q_pos):
        """q [B, Hq, S, D], k / v [B, Hq, T, D]: causal attention with the score modifiers, fp32 (prefill with soft-capping
        or ALiBi; decode for head shapes the hand-written kernel does not take)."""
        scores = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (self.head_dim ** -0.5)
        if self.attn_logit_softcapping:
            scores = self.attn_logit_softcapping * torch.tanh(scores / self.attn_logit_softcapping)
        qpos = first_q_pos + torch.arange(q.shape[2], device=q.device).view(-1, 1)
        kpos = torch.arange(k.shape[2], device=q.device).view(1, -1)
        if self.alibi is not None:
            scores = scores + self.alibi.slopes.view(1, -1, 1, 1) * (kpos - qpos).float()
        scores = scores.masked_fill(kpos > qpos, float("-inf"))
        return torch.matmul(torch.softmax(scores, dim=-1), v.float()).to(torch.float16)

    def forward_qkv(self, xqkv, apply_o_proj=True):
        """Everything after the qkv projection (the caller may have produced `xqkv` [B, S, (Hq + 2 Hkv) D]
        with the norm folded into the projection).  apply_o_proj=False returns the attention heads' output
        in place of attn_output (the caller runs o_proj with the residual add in its epilogue)."""
        bsz, seqlen, _ = xqkv.shape
        self._resize_cache(bsz)
        xqkv = self._qkv_rows(xqkv)
        device_pos = self._pos_dev is not None and seqlen == 1
        # the hand-written decode kernels take head_dim 128 and 1, 2, 4 or 8 query heads per KV head; other
        # shapes (e.g. 24 / 8 heads, 28 / 4) take the vendor path below instead of failing at the first decode step
        native = self.head_dim == 128 and self.n_kv_groups in (1, 2, 4, 8)
        modified = self.alibi is not None or self.attn_logit_softcapping > 0  # scores carry a bias or a cap
        if seqlen == 1 and native and self.rotary_dim == 128 and self.FUSE_ROPE_INTO_ATTENTION and not modified:
            # decode: rotation, cache append and attention in ONE launch (awq_decode_attention_rope)
            out = ops.decode_attention_rope(xqkv, self.cache.k, self.cache.v, self.rope.cos, self.rope.sin, self.start_pos,
                                            self.n_heads, self.n_kv_heads, pos_dev=self._pos_dev if device_pos else None,
                                            max_len=self._len_bound() if device_pos else None)
            attention_weight = out.reshape(bsz, 1, -1)
            attn_output = self.o_proj(attention_weight) if apply_o_proj else attention_weight
            self.start_pos += 1
            return attn_output, attention_weight, [torch.zeros(1, 1, self.start_pos, 1)]
        xq = ops.rope_kv_append(xqkv, self.cache.k, self.cache.v, self.rope.cos, self.rope.sin, self.start_pos,
                                self.n_heads, self.n_kv_heads, self.head_dim, self.rotary_dim,
                                pos_dev=self._pos_dev if device_pos else None)
        if seqlen > 1 and self.head_dim == 128 and self.n_heads % self.n_kv_heads == 0 and self.PREFILL_KERNEL:
            # prefill: the hand-written flash-style kernel (csrc/prefill_attn.hip; the reference's flash_attn_func call, attn.py:269-277),
            # chunked prefill, GQA, ALiBi and the soft cap included -- no mask tensor, no repeat_interleave copy of the cache
            out = ops.prefill_attention(xq, self.cache.k, self.cache.v, self.start_pos, softcap=self.attn_logit_softcapping,
                                        alibi_slopes=self.alibi.slopes if self.alibi is not None else None)
            output = out.reshape(bsz, seqlen, -1)
        elif seqlen > 1:  # other head sizes: the vendor's attention (fp32 matmul softmax when the scores carry a bias or a cap)
            end = self.start_pos + seqlen
            q = xq.transpose(1, 2)                                  # [B, Hq, S, D]
            k = self.cache.k[:bsz, :end].transpose(1, 2)            # [B, Hkv, T, D]
            v = self.cache.v[:bsz, :end].transpose(1, 2)
            if self.n_kv_groups > 1:
                k = k.repeat_interleave(self.n_kv_groups, dim=1)
                v = v.repeat_interleave(self.n_kv_groups, dim=1)
            if modified:
                out = self._attend_torch(q, k, v, self.start_pos)
            elif self.start_pos == 0:
                out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
            else:  # chunked prefill: query s sees cache rows <= start_pos + s
                mask = torch.ones((seqlen, end), dtype=torch.bool, device=q.device).tril(diagonal=self.start_pos)
                out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
            output = out.transpose(1, 2).reshape(bsz, seqlen, -1)
        else:
            if not native:  # the hand-written kernel is specialised (head_dim, GQA ratio); other shapes take the vendor path
                k = self.cache.k[:bsz, : self.start_pos + 1].transpose(1, 2)
                v = self.cache.v[:bsz, : self.start_pos + 1].transpose(1, 2)
                if self.n_kv_groups > 1:
                    k = k.repeat_interleave(self.n_kv_groups, dim=1)
                    v = v.repeat_interleave(self.n_kv_groups, dim=1)
                if modified:
                    output = self._attend_torch(xq.transpose(1, 2), k, v, self.start_pos).transpose(1, 2).reshape(bsz, 1, -1)
                else:
                    output = F.scaled_dot_product_attention(xq.transpose(1, 2), k, v).transpose(1, 2).reshape(bsz, 1, -1)
            else:
                out = ops.decode_attention(xq[:, 0], self.cache.k, self.cache.v, self.start_pos + 1,
                                           len_dev=self._len_dev if device_pos else None, max_len=self._len_bound(),
                                           softcap=self.attn_logit_softcapping,
                                           alibi_slopes=self.alibi.slopes if self.alibi is not None else None)
                output = out.reshape(bsz, 1, -1)
        attention_weight = output
        attn_output = self.o_proj(attention_weight) if apply_o_proj else attention_weight
        self.start_pos += seqlen
        past_key_value = [torch.zeros(1, 1, self.start_pos, 1)]
        return attn_output, attention_weight, past_key_value
