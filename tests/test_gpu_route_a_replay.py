"""Route A on the HIP kernels: the calls the UNMODIFIED reference modules make into the `awq_ext` /
`awq_v2_ext` shims -- recorded in the build container, where /root/reference exists, by tests/golden/make_route_a_calls.py while
tests/test_route_a.py ran `WQLinear_GEMM / _GEMV / _GEMVFast`, `QuantFusedMLP`, `apply_moe_weights`, `FasterTransformerRMSNorm`
over the shims with an oracle back end -- REPLAYED here against autoawq_amd.awq_ext / awq_v2_ext on the GPU: the same function
names, the same positional arguments and keyword arguments, tensors moved to the device; results compared with the oracle-backed
ones recorded beside them.  Fails when a reference call form (awq/modules/linear/gemm.py:51-58,100-102, gemv.py:168-180,
gemv_fast.py:191-206, fused/mlp.py:37-62, fused/moe.py:60-89,129-133, fused/norm.py:33-36) stops being accepted."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "route_a_calls.npz")
BIT_EXACT = {"dequantize_weights_cuda", "moe_alig_block_size"}
EXPECTED_FORMS = {"awq_ext.gemm_forward_cuda", "awq_ext.dequantize_weights_cuda", "awq_ext.gemv_forward_cuda", "awq_ext.gemmv2_forward_cuda",
                  "awq_v2_ext.gemv_forward_cuda_decode", "awq_v2_ext.gemm_forward_cuda_prefill", "awq_ext.moe_alig_block_size",
                  "awq_ext.grouped_gemm_forward", "awq_ext.silu_and_mul", "awq_ext.layernorm_forward_cuda"}


def _load():
    d = np.load(FIXTURE)
    return d, json.loads(bytes(d["manifest"]).decode())


def _close(y, ref, ulps=8):
    y, ref = np.asarray(y, np.float64), np.asarray(ref, np.float64)
    rms = float(np.sqrt(np.mean(ref ** 2))) if ref.size else 0.0
    tol = ulps * np.maximum(np.abs(ref), 2.0 ** -14) * 2.0 ** -10 + 2e-3 * np.abs(ref) + 2e-3 * rms
    return bool((np.abs(y - ref) <= tol).all())


def test_fixture_covers_every_reference_call_form():
    _, manifest = _load()
    assert {e["mod"] + "." + e["fn"] for e in manifest} >= EXPECTED_FORMS
    # the fused MLP passes split_k_iters as the FIFTH positional argument (mlp.py:41,49-62); the backward passes seven (gemm.py:100-102)
    assert any(e["fn"] == "gemm_forward_cuda" and len(e["args"]) == 5 for e in manifest)
    assert any(e["fn"] == "dequantize_weights_cuda" and len(e["args"]) == 7 for e in manifest)


@pytest.mark.parametrize("i", range(len(_load()[1]) if os.path.exists(FIXTURE) else 0))
def test_reference_call_replayed_on_hip(i):
    from autoawq_amd import awq_ext, awq_v2_ext

    d, manifest = _load()
    e = manifest[i]
    mod = {"awq_ext": awq_ext, "awq_v2_ext": awq_v2_ext}[e["mod"]]
    args = []
    for a in e["args"]:
        if "t" in a:
            args.append(torch.from_numpy(np.ascontiguousarray(d[a["t"]])).cuda())
        elif "none" in a:
            args.append(None)
        else:
            args.append(a["v"])
    ret = getattr(mod, e["fn"])(*args, **{k: v for k, v in e["kwargs"].items() if v is not None})
    torch.cuda.synchronize()
    checked = 0
    pairs = []
    if e["ret"] is not None:
        assert torch.is_tensor(ret), f'{e["fn"]}: the reference expects a tensor back'
        pairs.append((ret, d[e["ret"]["t"]], "return value"))
    for j, m in e["mutated"].items():
        pairs.append((args[int(j)], d[m["t"]], f"argument {j} (written by the call)"))
    for got, want, what in pairs:
        assert tuple(got.shape) == tuple(want.shape), (e["fn"], what, tuple(got.shape), want.shape)
        g = got.cpu().numpy()
        if e["fn"] in BIT_EXACT or not np.issubdtype(want.dtype, np.floating):
            assert np.array_equal(g, want), f'{e["mod"]}.{e["fn"]}: {what} differs from the reference-side result'
        else:
            assert _close(g, want), f'{e["mod"]}.{e["fn"]}: {what} outside tolerance (max err {np.abs(g.astype(np.float64) - want.astype(np.float64)).max()})'
        checked += 1
    assert checked >= 1, f'{e["fn"]}: nothing to compare'
