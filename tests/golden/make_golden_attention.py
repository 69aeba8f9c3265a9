#!/usr/bin/env python3
"""Golden vectors for the QuantAttentionFused feature surface, produced by RUNNING the reference classes in the build container
(never imported by tests, smoke() or bench.py):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_attention.py

Pins  awq/modules/fused/attn.py:89-125   ALiBi.gen_slopes / build_alibi_bias  (slopes and the bias row of the last query)
      awq/utils/fused_utils.py:165-201   get_attention_shapes: how a fused qkv row is viewed and sliced into q / k / v for the
                                         n_kv_heads == 0 (interleaved [3, heads, dim]) and the grouped-query layouts
"""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from awq.modules.fused.attn import ALiBi  # noqa: E402
from awq.utils.fused_utils import get_attention_shapes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for n in (8, 12, 40):
    slopes, bias = ALiBi.build_alibi_bias(n, 16)
    out[f"alibi_slopes_{n}"] = slopes.numpy()
    out[f"alibi_bias_{n}"] = bias.numpy()
g = torch.Generator().manual_seed(11)
for name, (H, Hkv, D) in {"mha_interleaved": (4, 0, 8), "gqa": (4, 2, 8)}.items():
    shapes = get_attention_shapes(None, H, Hkv, D)
    width = 3 * H * D if Hkv == 0 else (H + 2 * Hkv) * D
    xqkv = torch.randn((2, 3, width), generator=g).half()
    v = xqkv.view((2, 3) + shapes["xqkv_view"])
    out[f"{name}_xqkv"] = xqkv.numpy()
    out[f"{name}_meta"] = np.array([H, Hkv, D])
    for key in ("xq", "xk", "xv"):
        out[f"{name}_{key}"] = shapes[f"{key}_slice"](v).contiguous().numpy()
np.savez_compressed(os.path.join(HERE, "attention_golden.npz"), **out)
print(sorted(out))
