#!/usr/bin/env python3
"""Generates tests/golden/route_a_calls.npz: every call the UNMODIFIED reference modules make into the `awq_ext` / `awq_v2_ext`
shims (function name, positional arguments, keyword arguments) together with the oracle-backed result, recorded while
tests/test_route_a.py runs them in THIS container (where /root/reference exists and no GPU does).  tests/test_gpu_route_a_replay.py
replays the calls against the HIP kernels on the GPU box (where the reference tree does not exist) -- VERDICT r05 item 7.

    python tests/golden/make_route_a_calls.py

Call sites covered (reference file:line): awq/modules/linear/gemm.py:51-58,100-102 (gemm_forward_cuda, dequantize_weights_cuda
forward and backward), gemv.py:168-180 (gemv_forward_cuda, gemmv2_forward_cuda), gemv_fast.py:191-206 (gemv_forward_cuda_decode,
gemm_forward_cuda_prefill), fused/mlp.py:37-62 (gemm_forward_cuda with five positional arguments), fused/moe.py:60-89,129-133
(grouped_gemm_forward, silu_and_mul, moe_alig_block_size), fused/norm.py:33-36 (layernorm_forward_cuda)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)


def main():
    import test_route_a

    test_route_a.RECORD = []
    rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", os.path.join(TESTS, "test_route_a.py")])
    assert rc == 0, "tests/test_route_a.py must pass while its calls are recorded"
    calls = test_route_a.RECORD
    arrays, manifest = {}, []

    def put(key, t):
        arrays[key] = t.numpy() if t.dtype != torch.bfloat16 else t.float().numpy()
        return {"t": key, "dtype": str(t.dtype).replace("torch.", "")}

    for i, c in enumerate(calls):
        e = {"mod": c["mod"], "fn": c["fn"], "args": [], "kwargs": {k: (v if not torch.is_tensor(v) else None) for k, v in c["kwargs"].items()}}
        for j, a in enumerate(c["args"]):
            if torch.is_tensor(a):
                e["args"].append(put(f"c{i}_a{j}", a))
            elif a is None:
                e["args"].append({"none": True})
            else:
                e["args"].append({"v": a if isinstance(a, (bool, int, float)) else float(a)})
        e["ret"] = put(f"c{i}_ret", c["ret"]) if torch.is_tensor(c["ret"]) else None
        e["mutated"] = {str(j): put(f"c{i}_m{j}", t) for j, t in c["mutated"].items()}
        manifest.append(e)
    arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    out = os.path.join(HERE, "route_a_calls.npz")
    np.savez_compressed(out, **arrays)
    by = {}
    for e in manifest:
        by[e["mod"] + "." + e["fn"]] = by.get(e["mod"] + "." + e["fn"], 0) + 1
    print(f"{len(manifest)} calls -> {out} ({os.path.getsize(out) / 1e6:.2f} MB):", by)


if __name__ == "__main__":
    main()
