"""LlamaLikeModel for MI355X (reference: awq/modules/fused/model.py:60-109 and
awq/utils/fused_utils.py:14-42): embedding -> fused blocks -> final norm, with the reference's
input-id slicing (`prepare_input_ids`) and cache rolling (`prepare_cache`).  `forward` returns an
object with `.last_hidden_state` (transformers' BaseModelOutputWithPast when available)."""
from types import SimpleNamespace

import torch
import torch.nn as nn


class StepPlan:
    """What one `forward` call means for the caches: which ids are new, and how many cached positions every
    block has to forget first.  Same policy as the reference (awq/utils/fused_utils.py:14-42 -- a new multi-token
    context forgets everything cached, a decode step that would run past the window forgets the 100 oldest
    positions; transformers >= 4.35 re-sends the whole context while decoding, only its last id is new), stated
    as data instead of as two loops over the blocks."""

    ROLL = 100

    def __init__(self, input_ids, tokens_seen):
        n = input_ids.shape[-1]
        fresh = n if n == 1 else n - tokens_seen
        if n != 1 and fresh == 1:
            input_ids = input_ids[:, -1:]
        self.input_ids = input_ids
        self.tokens_seen = tokens_seen + fresh
        self.seqlen = input_ids.shape[-1]

    def forget(self, start_pos, max_seq_len):
        """positions block must drop before this step, given its cache fill"""
        over = start_pos + self.seqlen > max_seq_len
        if self.seqlen > 1:
            return start_pos if (over or start_pos > 0) else 0
        return self.ROLL if over else 0

    def apply(self, blocks):
        for block in blocks:
            a = block.attn
            n = self.forget(a.start_pos, a.max_seq_len)
            if n:
                a.start_pos = a.cache.roll_kv_n_steps(a.start_pos, n=n)


def prepare_cache(blocks, seqlen):
    """reference name (fused_utils.py:14-25), for callers that roll the caches themselves"""
    plan = StepPlan.__new__(StepPlan)
    plan.seqlen = seqlen
    plan.apply(blocks)


def prepare_input_ids(input_ids, last_forward_num_tokens):
    """reference name (fused_utils.py:28-42)"""
    plan = StepPlan(input_ids, last_forward_num_tokens)
    return plan.input_ids, plan.tokens_seen


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
        plan = StepPlan(input_ids, self.last_forward_num_tokens)
        input_ids, self.last_forward_num_tokens = plan.input_ids, plan.tokens_seen
        plan.apply(self.blocks)
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


class MixtralModel(LlamaLikeModel):
    """Reference: awq/modules/fused/model.py:20-58 -- the same step bookkeeping over MixtralBlock layers; `forward` returns
    transformers' MoeModelOutputWithPast when it is importable (empty router_logits, like the reference)."""

    @torch.inference_mode()
    def forward(self, input_ids, *args, **kwargs):
        out = super().forward(input_ids, *args, **kwargs)
        try:
            from transformers.modeling_outputs import MoeModelOutputWithPast

            return MoeModelOutputWithPast(last_hidden_state=out.last_hidden_state, past_key_values=None, hidden_states=(),
                                          attentions=(), router_logits=())
        except Exception:  # pragma: no cover
            return out
