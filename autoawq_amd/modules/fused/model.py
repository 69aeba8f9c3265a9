"""LlamaLikeModel for MI355X (reference: awq/modules/fused/model.py:60-109 and
awq/utils/fused_utils.py:14-42): embedding -> fused blocks -> final norm, with the reference's
input-id slicing (`prepare_input_ids`) and cache rolling (`prepare_cache`).  `forward` returns an
object with `.last_hidden_state` (transformers' BaseModelOutputWithPast when available)."""
from types import SimpleNamespace

import torch
import torch.nn as nn


def prepare_cache(blocks, seqlen):
    for block in blocks:
        start_pos = block.attn.start_pos
        exceeded = start_pos + seqlen > block.attn.max_seq_len
        if seqlen > 1 and (exceeded or start_pos > 0):  # new context: drop what is cached
            block.attn.start_pos = block.attn.cache.roll_kv_n_steps(start_pos, n=start_pos)
        elif seqlen == 1 and exceeded:  # decoding past the window: roll 100 positions out
            block.attn.start_pos = block.attn.cache.roll_kv_n_steps(start_pos, n=100)


def prepare_input_ids(input_ids, last_forward_num_tokens):
    num_input_tokens = input_ids.shape[-1]
    num_new_tokens = num_input_tokens
    if num_input_tokens != 1:
        num_new_tokens = num_input_tokens - last_forward_num_tokens
        if num_new_tokens == 1:  # transformers >= 4.35 passes the whole context while decoding
            input_ids = input_ids[:, -1:]
    return input_ids, last_forward_num_tokens + num_new_tokens


class LlamaLikeModel(nn.Module):
    def __init__(self, vocab_size, blocks, embedding, norm):
        super().__init__()
        self.vocab_size = vocab_size
        self.embedding = embedding
        self.blocks = nn.ModuleList(blocks)
        self.norm = norm
        self.last_forward_num_tokens = 0

    @property
    def embed_tokens(self):
        return self.embedding

    @property
    def layers(self):
        return self.blocks

    def _stream_mode(self, h):
        from .block import LlamaLikeBlock
        from .norm import FasterTransformerRMSNorm

        return (h.dtype == torch.float16 and h.is_contiguous() and isinstance(self.norm, FasterTransformerRMSNorm)
                and all(isinstance(b, LlamaLikeBlock) and isinstance(b.norm_1, FasterTransformerRMSNorm)
                        and isinstance(b.norm_2, FasterTransformerRMSNorm) for b in self.blocks))

    @torch.inference_mode()
    def forward(self, input_ids, *args, **kwargs):
        input_ids, self.last_forward_num_tokens = prepare_input_ids(input_ids, self.last_forward_num_tokens)
        _bsz, seqlen = input_ids.shape
        prepare_cache(self.blocks, seqlen)
        h = self.embedding(input_ids)
        if self._stream_mode(h):
            x, ssq = None, None
            for layer in self.blocks:  # residual adds and norms ride on the projections (block.forward_stream)
                x, h, ssq = layer.forward_stream(x, h, ssq)
            h = self.norm(x, residual=h) if x is not None else self.norm(h)
        else:
            for layer in self.blocks:
                h = layer(h)
            h = self.norm(h)
        try:
            from transformers.modeling_outputs import BaseModelOutputWithPast

            return BaseModelOutputWithPast(last_hidden_state=h, past_key_values=None, hidden_states=(), attentions=())
        except Exception:  # pragma: no cover
            return SimpleNamespace(last_hidden_state=h)
