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
Not carried over (raise NotImplementedError): ALiBi, q/k norms, logit soft-capping."""
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


class QuantAttentionFused(nn.Module):
    FUSE_ROPE_INTO_ATTENTION = True  # decode steps, head_dim 128, full rotary

    def __init__(self, hidden_size, n_heads, n_kv_heads, qkv_layer, o_proj, dev, max_seq_len=2048, use_alibi=False,
                 attention_shapes=None, rope_theta=10000, partial_rotary_factor=1.0, head_dim=None,
                 attn_logit_softcapping=0.0, q_norm=None, k_norm=None, **kwargs):
        super().__init__()
        if use_alibi or q_norm is not None or k_norm is not None or attn_logit_softcapping:
            raise NotImplementedError("ALiBi, q/k norms and logit soft-capping are not built for gfx950 yet")
        if attention_shapes is not None:
            raise NotImplementedError("custom attention_shapes: only the q | k | v head layout is built")
        self.hidden_size = hidden_size
        self.n_heads = n_heads
        self.n_kv_heads = n_kv_heads if n_kv_heads != 0 else n_heads
        self.n_kv_groups = n_heads // self.n_kv_heads
        self.head_dim = head_dim if head_dim is not None else hidden_size // n_heads
        self.qkv_proj = qkv_layer
        self.o_proj = o_proj
        self.start_pos = 0
        self.use_alibi = False
        self.cache_batch_size = int(os.getenv("AWQ_BATCH_SIZE", "1"))
        if kwargs.get("max_length") is not None:
            max_seq_len = kwargs["max_length"]
        self.max_seq_len = max_seq_len
        self.is_hf_transformers = False
        self.rope_theta = rope_theta
        self.cache = WindowedCache(self.cache_batch_size, n_heads, self.n_kv_heads, self.head_dim, max_seq_len, dev)
        self.partial_rotary_factor = partial_rotary_factor
        self.rotary_dim = int(self.head_dim * partial_rotary_factor)
        self.rope = RoPE(self.rotary_dim, max_seq_len, dev, rope_theta)
        self.is_neox = True
        self._pos_dev = None
        self._len_dev = None

    def use_device_positions(self, pos_dev, len_dev):
        """pos_dev / len_dev: int32 device tensors of one element (shared by all layers) holding the
        start position and the cache length after the append; the caller advances them."""
        self._pos_dev, self._len_dev = pos_dev, len_dev

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

    def forward_qkv(self, xqkv, apply_o_proj=True):
        """Everything after the qkv projection (the caller may have produced `xqkv` [B, S, (Hq + 2 Hkv) D]
        with the norm folded into the projection).  apply_o_proj=False returns the attention heads' output
        in place of attn_output (the caller runs o_proj with the residual add in its epilogue)."""
        bsz, seqlen, _ = xqkv.shape
        self._resize_cache(bsz)
        device_pos = self._pos_dev is not None and seqlen == 1
        # the hand-written decode kernels take head_dim 128 and 1, 2, 4 or 8 query heads per KV head; other
        # shapes (e.g. 24 / 8 heads, 28 / 4) take the vendor path below instead of failing at the first decode step
        native = self.head_dim == 128 and self.n_kv_groups in (1, 2, 4, 8)
        if seqlen == 1 and native and self.rotary_dim == 128 and self.FUSE_ROPE_INTO_ATTENTION:
            # decode: rotation, cache append and attention in ONE launch (awq_decode_attention_rope)
            out = ops.decode_attention_rope(xqkv, self.cache.k, self.cache.v, self.rope.cos, self.rope.sin, self.start_pos,
                                            self.n_heads, self.n_kv_heads, pos_dev=self._pos_dev if device_pos else None,
                                            max_len=self.max_seq_len if device_pos else None)
            attention_weight = out.reshape(bsz, 1, -1)
            attn_output = self.o_proj(attention_weight) if apply_o_proj else attention_weight
            self.start_pos += 1
            return attn_output, attention_weight, [torch.zeros(1, 1, self.start_pos, 1)]
        xq = ops.rope_kv_append(xqkv, self.cache.k, self.cache.v, self.rope.cos, self.rope.sin, self.start_pos,
                                self.n_heads, self.n_kv_heads, self.head_dim, self.rotary_dim,
                                pos_dev=self._pos_dev if device_pos else None)
        if seqlen > 1:
            end = self.start_pos + seqlen
            q = xq.transpose(1, 2)                                  # [B, Hq, S, D]
            k = self.cache.k[:bsz, :end].transpose(1, 2)            # [B, Hkv, T, D]
            v = self.cache.v[:bsz, :end].transpose(1, 2)
            if self.n_kv_groups > 1:
                k = k.repeat_interleave(self.n_kv_groups, dim=1)
                v = v.repeat_interleave(self.n_kv_groups, dim=1)
            if self.start_pos == 0:
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
                output = F.scaled_dot_product_attention(xq.transpose(1, 2), k, v).transpose(1, 2).reshape(bsz, 1, -1)
            else:
                out = ops.decode_attention(xq[:, 0], self.cache.k, self.cache.v, self.start_pos + 1,
                                           len_dev=self._len_dev if device_pos else None, max_len=self.max_seq_len)
                output = out.reshape(bsz, 1, -1)
        attention_weight = output
        attn_output = self.o_proj(attention_weight) if apply_o_proj else attention_weight
        self.start_pos += seqlen
        past_key_value = [torch.zeros(1, 1, self.start_pos, 1)]
        return attn_output, attention_weight, past_key_value
