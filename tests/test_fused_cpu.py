"""Host logic of the fused / MoE path that needs no GPU: block alignment and routing."""
import numpy as np
import torch


def test_moe_align_docstring_example_and_random_vs_oracle(oracle):
    from autoawq_amd import ops

    # worked example of awq/modules/fused/moe.py:111-119 (experts numbered 1..4 there -> 0..3 + 1 unused)
    ids = torch.tensor([[2, 3, 4], [1, 2, 4], [1, 3, 4], [1, 2, 3]], dtype=torch.int32)
    s, e, n = ops.moe_align_block_size(ids, 4, 5)
    want_s, want_e, want_n = oracle.moe_align(ids.numpy(), 5, 4)
    assert int(n) == want_n == 16
    assert np.array_equal(s.numpy()[:16], [3, 6, 9, 12, 0, 4, 10, 12, 1, 7, 11, 12, 2, 5, 8, 12])
    assert np.array_equal(s.numpy(), want_s)
    assert np.array_equal(e.numpy()[: want_n // 4], want_e[: want_n // 4])
    gen = torch.Generator().manual_seed(0)
    for T, E, k, blk in [(4, 8, 2, 16), (1, 8, 2, 16), (37, 8, 2, 16), (64, 16, 4, 16), (5, 3, 1, 8)]:
        ids = torch.stack([torch.randperm(E, generator=gen)[:k] for _ in range(T)]).to(torch.int32)
        s, e, n = ops.moe_align_block_size(ids, blk, E)
        ws, we, wn = oracle.moe_align(ids.numpy(), E, blk)
        assert int(n) == wn and s.dtype == torch.int32 and e.dtype == torch.int32
        assert s.numel() == T * k + E * (blk - 1) and e.numel() == T * k + E
        assert np.array_equal(s.numpy(), ws)
        assert np.array_equal(e.numpy()[: wn // blk], we[: wn // blk])


def test_fused_topk_matches_reference_rocm_branch():
    from autoawq_amd import ops

    logits = torch.randn((6, 8), generator=torch.Generator().manual_seed(1))
    w, ids = ops.fused_topk(logits, 2, True)
    p = torch.softmax(logits.float(), -1)
    tw, ti = torch.topk(p, 2, -1)
    assert torch.equal(ids.long(), ti) and torch.allclose(w, tw / tw.sum(-1, keepdim=True))
    assert torch.allclose(w.sum(-1), torch.ones(6))


def test_fuse_linears_builds_the_mixtral_expert_stacks():
    """awq/models/mixtral.py:131-151: per expert cat(w1, w3) on N, then torch.stack over the experts
    (awq/utils/fused_utils.py:145-162).  Pure buffer plumbing, so it runs on CPU tensors."""
    from autoawq_amd import WQLinear_GEMM
    from autoawq_amd.utils.fused_utils import fuse_linears

    E, K, I, g = 4, 256, 384, 128
    gen = torch.Generator().manual_seed(3)

    def rand(Kd, Nd):
        m = WQLinear_GEMM(4, g, Kd, Nd, False, "cpu")
        m.qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (Kd, Nd // 8), dtype=torch.int32, generator=gen)
        m.qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (Kd // g, Nd // 8), dtype=torch.int32, generator=gen)
        m.scales = torch.rand((Kd // g, Nd), generator=gen).half()
        return m

    experts = [(rand(K, I), rand(K, I), rand(I, K)) for _ in range(E)]
    keep = [[(m.qweight.clone(), m.qzeros.clone(), m.scales.clone()) for m in e] for e in experts]
    fused = [fuse_linears([w1, w3], "cpu") for w1, w3, _ in experts]
    for f, (k1, k3, _) in zip(fused, keep):
        assert f.out_features == 2 * I and f.in_features == K and f.bias is None
        assert torch.equal(f.qweight, torch.cat([k1[0], k3[0]], 1))   # gate columns first, then up (moe.py:73-76)
        assert torch.equal(f.qzeros, torch.cat([k1[1], k3[1]], 1)) and torch.equal(f.scales, torch.cat([k1[2], k3[2]], 1))
    assert all(not hasattr(m, "qweight") for e in experts for m in e[:2])
    ws = fuse_linears(fused, "cpu", dim=0, operation=torch.stack)
    w2s = fuse_linears([e[2] for e in experts], "cpu", dim=0, operation=torch.stack)
    assert ws.qweight.shape == (E, K, 2 * I // 8) and ws.qzeros.shape == (E, K // g, 2 * I // 8) and ws.scales.shape == (E, K // g, 2 * I)
    assert w2s.qweight.shape == (E, I, K // 8) and w2s.qzeros.shape == (E, I // g, K // 8) and w2s.scales.shape == (E, I // g, K)
    for e in range(E):
        assert torch.equal(ws.qweight[e, :, I // 8:], keep[e][1][0]) and torch.equal(w2s.scales[e], keep[e][2][2])
    assert set(ws.state_dict()) == {"qweight", "qzeros", "scales"}


def test_fused_mlp_keeps_one_copy_of_gate_up():
    """QuantFusedMLP (awq/modules/fused/mlp.py:14-70): the six registered buffers the reference names are views into the
    ONE [gate | up] concatenation the fused projection reads -- same state_dict, half the resident gate/up memory; loading,
    moving and re-assigning them keeps the fused tensors in step."""
    from autoawq_amd import WQLinear_GEMM
    from autoawq_amd.modules.fused.mlp import QuantFusedMLP

    gen = torch.Generator().manual_seed(4)
    lim = 0x7FFFFFFF

    def lin(K, N):
        m = WQLinear_GEMM(4, 128, K, N, False, "cpu")
        m.qweight = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, generator=gen)
        m.qzeros = torch.randint(-lim - 1, lim, (K // 128, N // 8), dtype=torch.int32, generator=gen)
        m.scales = torch.rand((K // 128, N), generator=gen).half()
        return m

    gate, up, down = lin(256, 512), lin(256, 512), lin(512, 256)
    gq, uq, us = gate.qweight.clone(), up.qweight.clone(), up.scales.clone()
    mlp = QuantFusedMLP(gate, down, up)
    fq, fs, fz = mlp._gate_up_fused()
    assert torch.equal(fq, torch.cat([gq, uq], 1)) and fq.shape == (256, 128)
    for name in ("gate_proj_qweight", "up_proj_qweight"):
        assert getattr(mlp, name).untyped_storage().data_ptr() == fq.untyped_storage().data_ptr()   # views, not copies
    sd = mlp.state_dict()
    assert {"gate_proj_qweight", "gate_proj_scales", "gate_proj_qzeros", "up_proj_qweight", "up_proj_scales", "up_proj_qzeros"} <= set(sd)
    assert torch.equal(sd["up_proj_qweight"], uq) and torch.equal(sd["up_proj_scales"], us)
    # state_dict hands out tensors with storages of their own (safetensors' save_file rejects shared memory)
    assert len({sd[k].untyped_storage().data_ptr() for k in sd if k.startswith(("gate_proj_", "up_proj_"))}) == 6
    import os, tempfile
    from safetensors.torch import load_file, save_file
    with tempfile.TemporaryDirectory() as tmp:
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(tmp, "mlp.safetensors"))
        assert torch.equal(load_file(os.path.join(tmp, "mlp.safetensors"))["up_proj_qweight"], uq)
    # load into another instance: the copy lands in the fused tensors
    other = QuantFusedMLP(lin(256, 512), lin(512, 256), lin(256, 512))
    other.load_state_dict(sd)
    assert torch.equal(other._gate_up_fused()[0], fq) and torch.equal(other._gate_up_fused()[1], fs)
    # module-wide conversions re-fuse
    other = other.to(torch.device("cpu"))
    assert other.gate_proj_qweight.untyped_storage().data_ptr() == other._gate_up_fused()[0].untyped_storage().data_ptr()
    assert torch.equal(other._gate_up_fused()[0], fq)
    # a caller that re-assigns a registered buffer (what a loader may do) is picked up at the next use
    new_up = torch.randint(-lim - 1, lim, (256, 64), dtype=torch.int32, generator=gen)
    other.up_proj_qweight = new_up
    assert torch.equal(other._gate_up_fused()[0], torch.cat([gq, new_up], 1))
    assert other.up_proj_qweight.untyped_storage().data_ptr() == other._gate_up_fused()[0].untyped_storage().data_ptr()


def test_fused_mlp_gate_up_pairs_interleave_rows_gemv_layout():
    """QuantFusedMLP.gate_up_pairs (GEMV layout, decode): row 2 i = gate row i, row 2 i + 1 = up row i in all three buffers -- the
    form `awq_gemv_forward_ex(..., AWQ_GEMV_EX_SILU_PAIRS)` reads; built lazily from the registered buffers, dropped when they move."""
    from autoawq_amd import WQLinear_GEMV
    from autoawq_amd.modules.fused.mlp import QuantFusedMLP
    from autoawq_amd.utils.packing import calculate_zeros_width

    gen = torch.Generator().manual_seed(6)
    lim = 0x7FFFFFFF

    def lin(K, N):
        zw = calculate_zeros_width(K, 128)
        m = WQLinear_GEMV(4, 128, K, N, False, "cpu")
        m.qweight = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, generator=gen)
        m.qzeros = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, generator=gen)
        m.scales = torch.rand((N, zw * 8), generator=gen).half()
        return m

    gate, up, down = lin(256, 48), lin(256, 48), lin(384, 256)
    gq, uq, gs, us, gz, uz = (t.clone() for t in (gate.qweight, up.qweight, gate.scales, up.scales, gate.qzeros, up.qzeros))
    mlp = QuantFusedMLP(gate, down, up)
    assert mlp.gemv_layout and mlp._pairs is None
    pq, ps, pz = mlp.gate_up_pairs()
    assert pq.shape == (96, 32) and pq.is_contiguous()
    for pairs, g_, u_ in ((pq, gq, uq), (ps, gs, us), (pz, gz, uz)):
        assert torch.equal(pairs[0::2], g_) and torch.equal(pairs[1::2], u_)
    assert mlp.gate_up_pairs()[0] is pq                      # cached
    mlp.up_proj_qweight = uq.flip(0).contiguous()            # a loader re-assigns a buffer: the fused tensors and the pairs follow
    mlp._gate_up_fused()
    assert torch.equal(mlp.gate_up_pairs()[0][1::2], uq.flip(0))


def test_derived_weight_caches_follow_in_place_loads():
    """`QuantFusedMLP.gate_up_pairs` (the interleaved gate / up copy of the five-launch decode) was keyed on data
    pointers only; `load_state_dict` / `.copy_()` write IN PLACE (same pointers), after which the decode path and the registered
    buffers disagreed silently.  The key now carries the tensors' versions.  (The other derived cache of round 3, the GEMM-layout
    prefill copy of GEMV / GEMVFast modules, no longer exists: prefill reads the module's own buffers.)"""
    from autoawq_amd import WQLinear_GEMV
    from autoawq_amd.modules.fused.mlp import QuantFusedMLP
    from autoawq_amd.modules.linear import gemv as gemv_mod
    from autoawq_amd.modules.linear import gemv_fast as gemv_fast_mod
    from autoawq_amd.utils.packing import calculate_zeros_width

    assert not hasattr(gemv_mod, "_gemm_layout_copy") and not hasattr(gemv_fast_mod, "_gemm_layout_copy")
    gen = torch.Generator().manual_seed(16)
    lim = 0x7FFFFFFF

    def lin(K, N):
        zw = calculate_zeros_width(K, 128)
        m = WQLinear_GEMV(4, 128, K, N, False, "cpu")
        m.qweight = torch.randint(-lim - 1, lim, (N, K // 8), dtype=torch.int32, generator=gen)
        m.qzeros = torch.randint(-lim - 1, lim, (N, zw), dtype=torch.int32, generator=gen)
        m.scales = torch.rand((N, zw * 8), generator=gen).half()
        return m

    mlp = QuantFusedMLP(lin(256, 48), lin(384, 256), lin(256, 48))
    other = QuantFusedMLP(lin(256, 48), lin(384, 256), lin(256, 48))
    p0 = mlp.gate_up_pairs()[0].clone()
    assert mlp.gate_up_pairs()[0] is mlp.gate_up_pairs()[0]             # cached while nothing changes
    mlp.load_state_dict(other.state_dict())
    pq, ps, pz = mlp.gate_up_pairs()
    assert torch.equal(pq[0::2], other.gate_proj_qweight) and torch.equal(pq[1::2], other.up_proj_qweight)
    assert torch.equal(ps[0::2], other.gate_proj_scales) and torch.equal(pz[1::2], other.up_proj_qzeros)
    assert not torch.equal(pq, p0)
    mlp.gate_proj_scales.mul_(2)                                        # any in-place write, not only load_state_dict
    assert torch.equal(mlp.gate_up_pairs()[1][0::2], mlp.gate_proj_scales)


def test_gemm_stack_to_gemv_twin_holds_the_same_weights(oracle):
    """utils/convert.py::gemm_stack_to_gemv (the MoE decode twins of round 6, built from the stacked GEMM-layout experts of
    awq/models/mixtral.py:130-158): an integer repack -- every expert's dequantised rows equal the GEMM-layout expert's
    dequantised columns bit for bit (both through the oracle), w1|w3 with its halves interleaved as (gate_j, up_j) row pairs."""
    import numpy as np
    import torch

    from autoawq_amd.utils.convert import gemm_stack_to_gemv

    E, K, N, g = 3, 256, 192, 128
    gen = torch.Generator().manual_seed(4)
    lim = 2 ** 31 - 1
    qw = torch.randint(-lim - 1, lim, (E, K, N // 8), dtype=torch.int32, generator=gen)
    qz = torch.randint(-lim - 1, lim, (E, K // g, N // 8), dtype=torch.int32, generator=gen)
    sc = (torch.rand((E, K // g, N), generator=gen) * 0.02 + 0.005).half()
    plain = gemm_stack_to_gemv(qw, qz, sc)
    pairs = gemm_stack_to_gemv(qw, qz, sc, interleave_halves=True)
    assert plain.qweight.shape == (E, N, K // 8) and plain.group_size == g and pairs.pairs and not plain.pairs
    for e in range(E):
        W = oracle.dequant_gemm(qw[e].numpy(), qz[e].numpy(), sc[e].numpy(), g)                                   # [K, N]
        Wt = oracle.dequant_gemv(plain.qweight[e].numpy(), plain.qzeros[e].numpy(), plain.scales[e].numpy(), g)    # [K, N]
        assert np.array_equal(np.asarray(Wt).view(np.uint16), np.asarray(W).view(np.uint16))
        Wp = oracle.dequant_gemv(pairs.qweight[e].numpy(), pairs.qzeros[e].numpy(), pairs.scales[e].numpy(), g)
        want = np.stack([W[:, : N // 2], W[:, N // 2:]], axis=2).reshape(K, N)
        assert np.array_equal(np.asarray(Wp).view(np.uint16), np.ascontiguousarray(want).view(np.uint16))
