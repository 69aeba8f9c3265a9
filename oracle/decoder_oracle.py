"""CPU restatements for the fused-decoder-block kernels.  TEST INFRASTRUCTURE ONLY (see awq_oracle.py).

  * rope_reference        follows RoPE.forward / precompute_freqs_cis, awq/modules/fused/attn.py:27-87,
                          op for op (complex multiply in fp32, cast back) -- PINNED against
                          tests/golden/rope_golden.npz, which the reference's own RoPE class produced;
  * rmsnorm_reference     fp32 `x * rsqrt(mean(x^2) + eps) * w`, one rounding.  The reference's kernel
                          (awq_ext.layernorm_forward_cuda, awq/modules/fused/norm.py:33-36) lives in the
                          un-vendored autoawq-kernels package: PARITY UNPINNED for this function; it is
                          cross-checked against transformers' LlamaRMSNorm in tests/test_decoder.py;
  * attention_reference   softmax(q k^T * scale) v over cache rows [0, seq_len) in fp64: the definition
                          flash_attn_with_kvcache implements (awq/modules/fused/attn.py:291-302; flash-attn
                          is a third-party dependency, absent here): PARITY UNPINNED, anchored on the whole-
                          model logits of tests/golden/tiny_llama_awq_gemm_outputs.npz instead.
"""
import numpy as np
import torch


def precompute_freqs_cis(dim, end, theta=10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
    t = torch.arange(end)
    freqs = torch.outer(t, freqs).float()
    return torch.polar(torch.ones_like(freqs), freqs)


def rope_reference(xq, xk, start_pos, rotary_dim, max_seq_len, theta=10000.0):
    """xq [B, S, Hq, D], xk [B, S, Hkv, D] fp16 torch tensors -> rotated copies (attn.py:54-87).
    With partial rotary the reference RETURNS only the rotated slice [..., :rotary_dim] (its
    concatenation with the pass-through dims at :83-85 is assigned to a dead variable); the full
    head -- rotated slice followed by the untouched dims, the transformers convention -- is what a
    KV cache of head_dim needs and what this function returns.  The golden vectors pin the slice."""
    partial = rotary_dim < xq.shape[-1]
    if partial:
        xq, xq_pass = xq[..., :rotary_dim], xq[..., rotary_dim:]
        xk, xk_pass = xk[..., :rotary_dim], xk[..., rotary_dim:]
    seqlen = xq.shape[1]
    freqs_cis = precompute_freqs_cis(rotary_dim, max_seq_len, theta)[start_pos: start_pos + seqlen]
    xq_ = torch.view_as_complex(xq.float().reshape(*xq.shape[:-1], 2, -1).transpose(-2, -1).contiguous())
    xk_ = torch.view_as_complex(xk.float().reshape(*xk.shape[:-1], 2, -1).transpose(-2, -1).contiguous())
    fc = freqs_cis.view(1, seqlen, 1, -1)
    xq_out = torch.view_as_real(xq_ * fc).transpose(-2, -1).flatten(3).type_as(xq)
    xk_out = torch.view_as_real(xk_ * fc).transpose(-2, -1).flatten(3).type_as(xk)
    if partial:
        xq_out = torch.cat((xq_out, xq_pass), dim=-1)
        xk_out = torch.cat((xk_out, xk_pass), dim=-1)
    return xq_out, xk_out


def rmsnorm_reference(x, w, eps):
    """x [..., H] fp16, w [H] fp16 numpy -> fp16 numpy."""
    xf = np.asarray(x, np.float32)
    inv = 1.0 / np.sqrt((xf.astype(np.float64) ** 2).mean(-1, keepdims=True) + eps)
    return (xf * inv.astype(np.float32) * np.asarray(w, np.float32)).astype(np.float16)


def alibi_slopes_reference(n_heads, alibi_bias_max=8):
    """ALiBi.gen_slopes (awq/modules/fused/attn.py:101-111): [n_heads] float64."""
    import math

    n2 = 2 ** math.ceil(math.log2(n_heads))
    m = np.arange(1, n2 + 1, dtype=np.float32) * np.float32(alibi_bias_max / n2)
    slopes = (1.0 / np.power(np.float32(2), m)).astype(np.float64)
    if n2 != n_heads:
        slopes = np.concatenate([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes


def attention_reference(q, k_cache, v_cache, seq_len, scale=None, softcap=0.0, alibi_slopes=None):
    """q [B, Hq, D], caches [B, Tmax, Hkv, D] numpy fp16 -> [B, Hq, D] float64.  softcap / alibi_slopes: the two score modifiers
    the reference hands flash_attn_with_kvcache (attn.py:286-302): s := softcap * tanh(s / softcap), then the bias
    slope_h * (t - (seq_len - 1)) of ALiBi.build_alibi_bias (attn.py:113-121: arange(1 - seq_len, 1) * slope)."""
    q = np.asarray(q, np.float64)
    B, Hq, D = q.shape
    Hkv = k_cache.shape[2]
    G = Hq // Hkv
    scale = D ** -0.5 if scale is None else scale
    out = np.zeros((B, Hq, D))
    for b in range(B):
        for h in range(Hq):
            k = np.asarray(k_cache[b, :seq_len, h // G], np.float64)
            v = np.asarray(v_cache[b, :seq_len, h // G], np.float64)
            s = k @ q[b, h] * scale
            if softcap:
                s = softcap * np.tanh(s / softcap)
            if alibi_slopes is not None:
                s = s + alibi_slopes[h] * np.arange(1 - seq_len, 1)
            p = np.exp(s - s.max())
            out[b, h] = (p / p.sum()) @ v
    return out


def prefill_attention_reference(q, k_cache, v_cache, start, rows, scale=None, softcap=0.0, alibi_slopes=None):
    """The prefill step of the fused attention -- flash_attn_func(xq, keys, values, causal=True, alibi_slopes=..., softcap=...),
    awq/modules/fused/attn.py:269-277 -- for the sampled query rows `rows` of q [B, S, Hq, D] (numpy fp16, after RoPE) over caches
    [B, Tmax, Hkv, D] that hold rows 0 .. start + S - 1: query row s is ONE query token at position start + s, i.e. exactly
    attention_reference over the first start + s + 1 cache rows (the causal mask), with the same score modifiers.
    Returns [B, len(rows), Hq, D] float64."""
    q = np.asarray(q)
    out = np.zeros((q.shape[0], len(rows), q.shape[2], q.shape[3]))
    for i, s_ in enumerate(rows):
        out[:, i] = attention_reference(q[:, s_], k_cache, v_cache, start + int(s_) + 1, scale=scale, softcap=softcap, alibi_slopes=alibi_slopes)
    return out
