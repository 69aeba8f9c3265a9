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
