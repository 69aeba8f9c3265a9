"""The persistent decode chain (csrc/gemv_chain.hip, autoawq_amd/chain.py): a run of dependent decode-sized
WQLinear_GEMM projections in ONE launch.  Every link is checked against the CPU oracle fed with the chain's
own previous output (so each stage is validated at its own scale, not through a whole-chain tolerance), and
against the one-launch-per-Linear kernel; planner / error paths run without a GPU."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import assert_product_close
from test_gpu_parity import fullrange_case


def _link_arrays(shapes, M=1, gated_idx=()):
    from autoawq_amd import _lib

    arr = (_lib.AwqChainLink * len(shapes))()
    for i, (K, N) in enumerate(shapes):
        a = arr[i]
        a.qweight = a.scales = a.qzeros = 0x10000
        a.K, a.N, a.group_size = K, N, 128
        a.x, a.x_stride, a.x_from = (0x10000, K, -1) if i == 0 else (None, 0, i - 1)
        a.flags = 1 if i in gated_idx else 0
        a.y = 0x10000 if i == len(shapes) - 1 else None
    return arr


def test_chain_planner_without_a_gpu():
    """awq_chain_build is host-only: sizes, alignment and the shapes it refuses."""
    from autoawq_amd import _lib

    L = _lib.lib()
    shapes = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)] * 2
    arr = _link_arrays(shapes, gated_idx=(3, 7))
    nb = L.awq_chain_plan_bytes(len(shapes))
    plan = (ctypes.c_uint8 * nb)()
    need = ctypes.c_size_t(0)
    assert L.awq_chain_build(arr, len(shapes), 1, None, 0, plan, nb, ctypes.byref(need)) == 0
    assert 4096 < need.value < (64 << 20)
    assert L.awq_chain_build(arr, len(shapes), 1, 0x40000000, need.value - 1, plan, nb, ctypes.byref(need)) == -4   # workspace too small
    assert L.awq_chain_build(arr, len(shapes), 1, 0x40000010, need.value, plan, nb, ctypes.byref(need)) == -2       # not 256-byte aligned
    assert L.awq_chain_build(arr, len(shapes), 1, 0x40000000, need.value, plan, nb, ctypes.byref(need)) == 0
    import struct

    magic, n_int, grid, M = struct.unpack_from("<IIII", bytes(plan), 0)
    assert magic == 0x41575143 and M == 1 and grid % 8 == 0 and n_int >= len(shapes)
    assert grid == L.awq_chain_grid_blocks()
    assert L.awq_chain_build(arr, len(shapes), 9, None, 0, plan, nb, ctypes.byref(need)) == -3      # M <= 8
    bad = _link_arrays([(4096, 4096), (4096, 4096)])
    bad[1].x_from = 0
    bad[1].x_col0 = 64                                                                                # not a multiple of 128
    assert L.awq_chain_build(bad, 2, 1, None, 0, plan, nb, ctypes.byref(need)) == -1
    bad = _link_arrays([(4096, 4096), (8192, 4096)])                                                  # consumes more columns than link 0 makes
    assert L.awq_chain_build(bad, 2, 1, None, 0, plan, nb, ctypes.byref(need)) == -1
    bad = _link_arrays([(4096, 4096), (4096, 4096)])
    bad[1].group_size = 64                                                                            # groups of 128 rows only
    assert L.awq_chain_build(bad, 2, 1, None, 0, plan, nb, ctypes.byref(need)) == -3
    assert L.awq_chain_forward(None, None, None, 0, None) == -6
    assert L.awq_chain_workspace_init(None, 0, None) == -6


def _build_chain(hidden, inter, layers, M, gated, seed, with_bias=False, with_residual=False):
    from autoawq_amd.chain import ChainLink

    gen = torch.Generator().manual_seed(seed)
    x0 = torch.randn((M, hidden), generator=gen).half()
    links, meta = [], []
    for li in range(layers):
        for name, K, N, g_ in (("qkv", hidden, 3 * hidden, False), ("o", hidden, hidden, False),
                               ("gate_up", hidden, 2 * inter, False), ("down", inter, hidden, gated)):
            qw, qz, sc, _, bias = fullrange_case(K, N, 128, 1, seed=seed + 7 * len(links), realistic=True)
            sc = (sc.float() * (6.0 / (K ** 0.5))).half()  # keep the chain's amplitude near 1 through many links
            b = (bias * 0.1).half() if with_bias and name in ("qkv", "down") else None
            res = (torch.randn((M, N), generator=gen) * 0.1).half() if with_residual and name in ("o", "down") else None
            links.append(dict(qw=qw, qz=qz, sc=sc, bias=b, res=res, gated=g_))
            meta.append((name, K, N))
    dev_links = []
    for i, ln in enumerate(links):
        y = torch.zeros((M, meta[i][2]), dtype=torch.float16, device="cuda")
        dev_links.append(ChainLink(ln["qw"].cuda(), ln["sc"].cuda(), ln["qz"].cuda(), bias=None if ln["bias"] is None else ln["bias"].cuda(),
                                   x=x0.cuda() if i == 0 else None, gated=ln["gated"], y=y,
                                   add_residual=None if ln["res"] is None else ln["res"].cuda()))
    return x0, links, meta, dev_links


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,inter,M,gated", [(768, 1536, 1, True), (1024, 2816, 2, True), (1024, 2816, 5, False),
                                                  (768, 1536, 8, True), (2048, 5632, 1, True)])
def test_chain_links_vs_oracle(oracle, hidden, inter, M, gated):
    """Every link of a two-layer chain vs the CPU oracle (input = the chain's own previous output): bias, residual
    epilogue, the silu(gate) * up staging, K-slices that straddle tiles, batch rows 1..8."""
    from autoawq_amd import ops
    from autoawq_amd.chain import DecodeChain

    x0, links, meta, dev_links = _build_chain(hidden, inter, 2, M, gated, seed=hidden + M, with_bias=True, with_residual=True)
    chain = DecodeChain(dev_links, M=M)
    chain()
    torch.cuda.synchronize()
    assert chain.status() == 0
    x = x0.numpy()
    for i, (ln, (name, K, N), dl) in enumerate(zip(links, meta, dev_links)):
        xin = oracle.silu_and_mul(x[:, : 2 * K]) if ln["gated"] else np.ascontiguousarray(x[:, :K])
        bias = None if ln["bias"] is None else ln["bias"].numpy()
        y32, _ = oracle.linear_gemm(xin, ln["qw"].numpy(), ln["qz"].numpy(), ln["sc"].numpy(), 128, bias)
        W = oracle.dequant_gemm(ln["qw"].numpy(), ln["qz"].numpy(), ln["sc"].numpy(), 128)
        wsig = oracle.weight_rounding_sigma(xin, W)
        got = dl.y.cpu().numpy()
        if ln["res"] is not None:  # fp16(fp16(x W + b) + residual): compare before the second rounding's ulp
            want = (y32.astype(np.float16).astype(np.float32) + ln["res"].numpy().astype(np.float32))
            ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10
            assert_product_close(got.astype(np.float64), want, f"link {i} {name} (+residual)", wsigma=wsig + ulp)
        else:
            assert_product_close(got.astype(np.float64), y32, f"link {i} {name}", wsigma=wsig)
        # and the one-launch kernel on the same input: same exact-integer arithmetic, another summation order
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        ref = ops.gemm_forward(xd[:, : 2 * K].contiguous() if ln["gated"] else xd[:, :K].contiguous(), dl.qweight, dl.scales, dl.qzeros,
                               dl.bias, flags=ops.X_GATED_SILU if ln["gated"] else 0)
        mag = ref.float().abs()
        if ln["res"] is not None:  # an ulp of the projection before the residual is added survives the cancellation
            mag = mag + dl.add_residual.float().abs()
            ref = (ref.float() + dl.add_residual.float()).half()
            mag = mag + ref.float().abs()
        d = (dl.y.float() - ref.float()).abs()
        tol = 2.0 ** -9 * mag + 3e-4 * float(ref.float().pow(2).mean().sqrt()) + 1e-6
        assert bool((d <= tol).all()), (i, name, float((d / tol).max()))
        x = got


@pytest.mark.gpu
def test_chain_7b_shapes_replay_and_graph():
    """Two layers of the Llama-2-7B shapes (what bench.py times): link by link against the one-launch kernels,
    bitwise identical across launches and under hipGraph replay (the epoch lives on the device), a lean chain that
    materialises only the last result gives the same bits, nothing times out."""
    from autoawq_amd import ops
    from autoawq_amd.chain import ChainLink, DecodeChain

    x0, links, meta, dev_links = _build_chain(4096, 11008, 2, 1, True, seed=3)
    chain = DecodeChain(dev_links, M=1)
    chain()
    torch.cuda.synchronize()
    assert chain.status() == 0
    x = x0.cuda()
    for i, (ln, (name, K, N), dl) in enumerate(zip(links, meta, dev_links)):
        ref = ops.gemm_forward(x if ln["gated"] else x[:, :K].contiguous(), dl.qweight, dl.scales, dl.qzeros,
                               flags=ops.X_GATED_SILU if ln["gated"] else 0)
        d = (dl.y.float() - ref.float()).abs()
        tol = 2.0 ** -10 * ref.float().abs() + 2e-4 * float(ref.float().pow(2).mean().sqrt()) + 1e-6
        assert bool((d <= tol).all()), (i, name, float((d / tol).max()))
        x = dl.y
    first = [dl.y.clone() for dl in dev_links]
    for _ in range(20):
        chain()
    torch.cuda.synchronize()
    assert all(torch.equal(a, dl.y) for a, dl in zip(first, dev_links))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain()
        for dl in dev_links:
            dl.y.zero_()
        for _ in range(5):
            g.replay()
        s.synchronize()
    assert all(torch.equal(a, dl.y) for a, dl in zip(first, dev_links)) and chain.status() == 0
    lean = [ChainLink(dl.qweight, dl.scales, dl.qzeros, x=dl.x, gated=dl.gated, y=dl.y if i == len(dev_links) - 1 else None)
            for i, dl in enumerate(dev_links)]
    dev_links[-1].y.zero_()
    chain2 = DecodeChain(lean, M=1)
    chain2()
    torch.cuda.synchronize()
    assert chain2.status() == 0 and torch.equal(first[-1], dev_links[-1].y)


@pytest.mark.gpu
def test_chain_rejects_what_it_cannot_run():
    from autoawq_amd import _lib
    from autoawq_amd.chain import ChainLink, DecodeChain

    qw, qz, sc, x, _ = fullrange_case(512, 256, 128, 1, seed=1, realistic=True)   # K = 512: fewer than six K groups
    y = torch.zeros((1, 256), dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.AwqHipError):
        DecodeChain([ChainLink(qw.cuda(), sc.cuda(), qz.cuda(), x=x.cuda(), y=y)], M=1)
    with pytest.raises(_lib.AwqHipError):
        DecodeChain([], M=1)
