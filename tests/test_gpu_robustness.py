"""Split-K exchange and workspace hardening (VERDICT r1 "robustness"): concurrent streams, a long mixed-shape
stress run, workspaces that grow under a captured hipGraph, and the error word that a reducer raises when it
gives up -- which somebody now looks at."""
import pytest
import torch

from test_gpu_parity import fullrange_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from autoawq_amd import _lib, ops as _ops

    _lib.lib()
    return _ops


def _case(K, N, M, seed):
    qw, qz, s, x, bias = fullrange_case(K, N, 128, M, seed=seed, realistic=True)
    return x.cuda(), qw.cuda(), s.cuda(), qz.cuda(), bias.cuda()


def test_two_streams_concurrently(ops):
    """Each stream owns its workspace (keyed by stream): two streams hammering different shapes at the same time
    give bitwise the results of a quiet run, and both workspaces end clean."""
    a = _case(4096, 4096, 1, 1)
    b = _case(4096, 11008, 4, 2)
    ref_a, ref_b = ops.gemm_forward(*a), ops.gemm_forward(*b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(200):
        with torch.cuda.stream(s1):
            outs_a.append(ops.gemm_forward(*a))
        with torch.cuda.stream(s2):
            outs_b.append(ops.gemm_forward(*b))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref_a) for o in outs_a) and all(torch.equal(o, ref_b) for o in outs_b)
    for s in (s1, s2):
        with torch.cuda.stream(s):
            assert ops.workspace_is_clean(a[0].device)
    ops.check_workspaces()


def test_thousand_mixed_calls(ops):
    """1000 calls over shapes that take every exchange path (decode split-K, tiled split-K, no split), in random
    order on one stream: every result bitwise equal to the first one of its shape, workspace clean, no error word."""
    shapes = [(4096, 4096, 1), (4096, 12288, 1), (11008, 4096, 1), (4096, 22016, 2), (4096, 4096, 8), (4096, 11008, 16),
              (2048, 2048, 40), (4096, 4096, 100), (1024, 8192, 3), (512, 256, 128)]
    cases = [_case(K, N, M, seed=10 + i) for i, (K, N, M) in enumerate(shapes)]
    first = [ops.gemm_forward(*c) for c in cases]
    torch.cuda.synchronize()
    gen = torch.Generator().manual_seed(0)
    order = torch.randint(0, len(cases), (1000,), generator=gen).tolist()
    bad = 0
    for n, i in enumerate(order):
        y = ops.gemm_forward(*cases[i])
        if n % 50 == 49 or n == len(order) - 1:  # compare in batches: keep the stream busy back to back
            torch.cuda.synchronize()
        bad += 0 if torch.equal(y, first[i]) else 1
    assert bad == 0
    assert ops.workspace_is_clean(cases[0][0].device)
    ops.check_workspaces()


def test_workspace_growth_keeps_captured_graph_valid(ops):
    """A graph captured on a fresh stream holds the pointer of that stream's first (small) workspace.  A later,
    larger call grows the workspace: the old one is retired, not freed, and the replay still lands in memory that
    is initialised and owned."""
    x, qw, s, qz, _ = _case(4096, 4096, 1, 5)
    big = _case(4096, 22016, 16, 6)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ref = ops.gemm_forward(x, qw, s, qz)
        st.synchronize()
        before = ops._current_workspace(x.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            y = ops.gemm_forward(x, qw, s, qz)
        ops.workspace(x.device, before.numel() * 4)  # force growth
        after = ops._current_workspace(x.device)
        assert after.data_ptr() != before.data_ptr() and any(w.buf.data_ptr() == before.data_ptr() for w in ops._retired)
        for _ in range(3):
            ops.gemm_forward(*big)  # traffic through the new workspace in between
            y.zero_()
            g.replay()
            st.synchronize()
            assert torch.equal(y, ref)
    ops.check_workspaces()


def test_error_word_is_noticed(ops):
    """Raise the control word by hand (what a reducer does when it gives up): the blocking check reports it and
    re-initialises; the asynchronous watch reports it within a bounded number of calls."""
    from autoawq_amd import _lib

    c = _case(4096, 4096, 1, 7)
    ops.gemm_forward(*c)
    ws = ops._current_workspace(c[0].device)
    ws[:4].view(torch.int32).fill_(1)
    with pytest.raises(_lib.AwqHipError):
        ops.check_workspaces()
    assert ops.workspace_is_clean(c[0].device)
    ws[:4].view(torch.int32).fill_(1)
    with pytest.raises(_lib.AwqHipError):
        for _ in range(4 * ops._CHECK_EVERY):
            ops.gemm_forward(*c)
            torch.cuda.synchronize()
    assert ops.workspace_is_clean(c[0].device)
    ref = ops.gemm_forward(*c)
    assert torch.equal(ref, ops.gemm_forward(*c))
    ops.check_workspaces()


def test_workspace_first_created_inside_inference_mode_is_usable_outside():
    """The fused model's forward runs under `torch.inference_mode()`; a split-K workspace first created there used to hold
    inference tensors, and its error-word read-back -- every 64th call, here from an ordinary context -- raised (found by the
    round-4 suite once a test that decodes under inference mode ran before the others)."""
    from autoawq_amd import ops

    gen = torch.Generator().manual_seed(3)
    lim = 0x7FFFFFFF
    K, N = 512, 256
    qw = torch.randint(-lim - 1, lim, (K, N // 8), dtype=torch.int32, generator=gen).cuda()
    qz = torch.randint(-lim - 1, lim, (K // 128, N // 8), dtype=torch.int32, generator=gen).cuda()
    sc = (torch.rand((K // 128, N), generator=gen) * 0.02 + 0.005).half().cuda()
    x = torch.randn((1, K), generator=gen).half().cuda()
    s = torch.cuda.Stream()   # a stream nobody has used: its workspace does not exist yet
    with torch.cuda.stream(s):
        with torch.inference_mode():
            first = ops.gemm_forward(x, qw, sc, qz).clone()
        for _ in range(3 * 64 + 5):   # crosses the read-back interval several times, outside inference mode
            y = ops.gemm_forward(x, qw, sc, qz)
        s.synchronize()
    assert torch.equal(y, first)
    ops.check_workspaces()
