"""CPU-side checks of the drop-in boundary: C ABI symbols, module surface, packers, and the rule
that the product never touches the oracle.  No GPU needed (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


def _header_functions():
    src = open(os.path.join(ROOT, "include", "awq_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(awq_[a-z0-9_]+)\s*\(", src)))  # every declared function


def test_header_declares_and_library_exports_same_symbols():
    from autoawq_amd import _lib

    if not _lib.available():
        import __graft_entry__

        __graft_entry__.build()
    names = _header_functions()
    assert names, "no functions parsed from include/awq_hip.h"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    h = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"libawq_hip.so does not export {n}"
    L = _lib.lib()
    assert L.awq_hip_abi_version() == 1
    assert L.awq_hip_error_string(-1).decode().startswith("bad shape")
    # pure host-side validation paths (no kernel is launched for these)
    assert L.awq_gemm_forward(None, None, None, None, None, None, 1, 128, 12, 128, None, 0, 0, None) == -1  # N % 8
    assert L.awq_gemm_forward(None, None, None, None, None, None, 1, 100, 16, 64, None, 0, 0, None) == -1   # K % g
    assert L.awq_gemm_forward(None, None, None, None, None, None, 0, 128, 16, 128, None, 0, 0, None) == 0   # M == 0
    assert L.awq_gemm_forward(None, None, None, None, None, None, 1, 128, 16, 128, None, 0, 0, None) == -6  # NULL
    assert L.awq_gemm_workspace_bytes(1, 4096, 4096, 128) > 16384
    assert L.awq_gemm_workspace_init(None, 0, None) == -6


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under autoawq_amd/ may reference it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "autoawq_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle|awq_oracle|libawq_oracle", txt, re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_wqlinear_gemm_surface_matches_reference_checkpoint_layout():
    from autoawq_amd import WQLinear_GEMM

    g = golden("packed_K512_N64_g128")
    m = WQLinear_GEMM(4, 128, 512, 64, True, "cpu")
    sd = m.state_dict()
    assert list(sd.keys()) == ["qweight", "qzeros", "scales", "bias"]
    assert tuple(sd["qweight"].shape) == g["gemm_qweight"].shape and sd["qweight"].dtype == torch.int32
    assert tuple(sd["qzeros"].shape) == g["gemm_qzeros"].shape and sd["qzeros"].dtype == torch.int32
    assert tuple(sd["scales"].shape) == g["gemm_scales"].shape and sd["scales"].dtype == torch.float16
    assert list(dict(m.named_parameters())) == []  # buffers, not Parameters
    assert "in_features=512, out_features=64, bias=True, w_bit=4, group_size=128" in repr(m)
    assert WQLinear_GEMM(4, -1, 256, 64, False, "cpu").group_size == 256  # gemm.py:128
    assert WQLinear_GEMM(4, 128, 256, 64, False, "cpu").bias is None
    with pytest.raises(NotImplementedError):
        WQLinear_GEMM(3, 128, 256, 64, False, "cpu")
    with pytest.raises(AssertionError):
        WQLinear_GEMM(4, 128, 200, 64, False, "cpu")
    # state-dict round trip with reference-produced buffers
    m.load_state_dict({"qweight": torch.from_numpy(g["gemm_qweight"]), "qzeros": torch.from_numpy(g["gemm_qzeros"]),
                       "scales": torch.from_numpy(g["gemm_scales"]), "bias": torch.from_numpy(g["bias"])})
    assert np.array_equal(m.qweight.numpy(), g["gemm_qweight"])


@pytest.mark.parametrize("name", ["packed_K512_N64_g128", "packed_K256_N32_g64", "packed_K128_N32_g32"])
def test_from_linear_packs_bit_identically_to_reference(name):
    """Vectorised packer == reference Python-loop packer (awq/modules/linear/gemm.py:171-251)."""
    from autoawq_amd import WQLinear_GEMM

    g = golden(name)
    gs = int(g["group_size"])
    K, N = g["w_int"].shape
    lin = torch.nn.Linear(K, N, bias=True).half()
    lin.weight.data = torch.from_numpy(g["lin_weight"])
    lin.bias.data = torch.from_numpy(g["bias"])
    scales = torch.from_numpy(g["gemm_scales"])                      # [G, N] fp16
    zeros = torch.from_numpy(g["z_int"].astype(np.float32))          # [G, N]
    m = WQLinear_GEMM.from_linear(lin, 4, gs, False, scales, zeros)
    assert np.array_equal(m.qweight.numpy(), g["gemm_qweight"])
    assert np.array_equal(m.qzeros.numpy(), g["gemm_qzeros"])
    assert np.array_equal(m.scales.numpy().view(np.uint16), g["gemm_scales"].view(np.uint16))
    assert np.array_equal(m.bias.detach().numpy(), g["bias"])
    empty = WQLinear_GEMM.from_linear(lin, 4, gs, init_only=True)
    assert int(empty.qweight.abs().sum()) == 0


def test_cpu_tensors_fail_loudly():
    from autoawq_amd import WQLinear_GEMM
    from autoawq_amd._lib import AwqHipError

    m = WQLinear_GEMM(4, 128, 256, 64, False, "cpu")
    with pytest.raises(AwqHipError):
        m(torch.randn(1, 1, 256))
    # empty batch never reaches a kernel (gemm.py:44-45)
    out = m(torch.randn(0, 5, 256))
    assert out.shape == (0, 5, 64) and out.dtype == torch.float32


def test_round3_entry_points_fail_loudly_on_cpu_tensors():
    """No CPU fallback behind the new entry points either: the block-fusion GEMV, the score-modified attention, the MoE prefill
    path and the one-shot all-reduce wrapper raise on non-HIP tensors instead of computing something else."""
    from autoawq_amd import ops
    from autoawq_amd._lib import AwqHipError
    from autoawq_amd.comm import OneShotAllReduce
    from autoawq_amd.modules.fused import moe

    K, N, zw = 256, 16, 2
    x = torch.zeros((1, K), dtype=torch.float16)
    qw, qz, sc = torch.zeros((N, K // 8), dtype=torch.int32), torch.zeros((N, zw), dtype=torch.int32), torch.zeros((N, zw * 8), dtype=torch.float16)
    with pytest.raises(AwqHipError):
        ops.gemv_forward_ex(x, qw, sc, qz, 128, norm_weight=torch.ones(K, dtype=torch.float16))
    q = torch.zeros((1, 4, 128), dtype=torch.float16)
    kv = torch.zeros((1, 8, 4, 128), dtype=torch.float16)
    with pytest.raises(AwqHipError):
        ops.decode_attention(q, kv, kv, 4, softcap=30.0)

    class Stack:
        pass
    w1, w2 = Stack(), Stack()
    w1.qweight, w1.qzeros, w1.scales = torch.zeros((2, 128, 32), dtype=torch.int32), torch.zeros((2, 1, 32), dtype=torch.int32), torch.zeros((2, 1, 256), dtype=torch.float16)
    w2.qweight, w2.qzeros, w2.scales = torch.zeros((2, 128, 16), dtype=torch.int32), torch.zeros((2, 1, 16), dtype=torch.int32), torch.zeros((2, 1, 128), dtype=torch.float16)
    with pytest.raises(AwqHipError):  # 512 tokens x top-1 pairs: the prefill path
        moe.apply_moe_weights(w1, w2, torch.zeros((512, 128), dtype=torch.float16), torch.zeros((512, 2)), 1, True)
    with pytest.raises((AwqHipError, RuntimeError, AssertionError)):
        OneShotAllReduce.local_group(2, 64, device="cpu")


# ------------------------------------------------------------------ WQLinear_GEMV surface (CPU)

@pytest.mark.parametrize("name", ["packed_K512_N64_g128", "packed_K256_N32_g64", "packed_K128_N32_g32"])
def test_gemv_module_surface_and_packer_match_reference(name):
    """Buffers, shapes, dtypes and the vectorised packer == reference gemv.py packer outputs."""
    from autoawq_amd import WQLinear_GEMV
    from autoawq_amd.utils.packing import calculate_zeros_width

    g = golden(name)
    gs = int(g["group_size"])
    K, N = g["w_int"].shape
    m = WQLinear_GEMV(4, gs, K, N, True, "cpu")
    sd = m.state_dict()
    assert list(sd.keys()) == ["qweight", "qzeros", "scales", "bias"]
    assert tuple(sd["qweight"].shape) == g["gemv_qweight"].shape and sd["qweight"].dtype == torch.int32
    assert tuple(sd["qzeros"].shape) == g["gemv_qzeros"].shape and sd["qzeros"].dtype == torch.int32
    assert tuple(sd["scales"].shape) == g["gemv_scales"].shape and sd["scales"].dtype == torch.float16
    assert m.split_k_iters == 8 and list(dict(m.named_parameters())) == []
    assert calculate_zeros_width(4096, 128) == 4 and calculate_zeros_width(11008, 128) == 11
    assert calculate_zeros_width(8192, 64) == 16 and calculate_zeros_width(4096, 32) == 16
    with pytest.raises(NotImplementedError):
        calculate_zeros_width(4096, 16)
    lin = torch.nn.Linear(K, N, bias=True).half()
    lin.weight.data = torch.from_numpy(g["lin_weight"])
    lin.bias.data = torch.from_numpy(g["bias"])
    G = K // gs
    scales = torch.from_numpy(g["gemv_scales"][:, :G].copy())                 # [N, G] fp16
    zeros = torch.from_numpy(g["z_int"].T.astype(np.float32).copy())          # [N, G]
    p = WQLinear_GEMV.from_linear(lin, 4, gs, False, scales, zeros)
    assert np.array_equal(p.qweight.numpy(), g["gemv_qweight"])
    assert np.array_equal(p.qzeros.numpy(), g["gemv_qzeros"])
    assert np.array_equal(p.scales.numpy().view(np.uint16), g["gemv_scales"].view(np.uint16))
    assert int(WQLinear_GEMV.from_linear(lin, 4, gs, init_only=True).qweight.abs().sum()) == 0
    from autoawq_amd._lib import AwqHipError
    with pytest.raises(AwqHipError):
        p(torch.randn(1, 1, K))


@pytest.mark.parametrize("name", ["packed_K512_N64_g128", "packed_K256_N32_g64", "packed_K128_N32_g32"])
def test_gemvfast_module_surface_and_packer_match_reference(name):
    """WQLinear_GEMVFast: buffers / dtypes and the closed-form packer == reference pack_intweight."""
    from autoawq_amd import WQLinear_GEMVFast

    g = golden(name)
    gs = int(g["group_size"])
    K, N = g["w_int"].shape
    m = WQLinear_GEMVFast(4, gs, K, N, True, "cpu")
    sd = m.state_dict()
    assert list(sd.keys()) == ["qweight", "scales", "qzeros", "bias"]
    assert tuple(sd["qweight"].shape) == g["fast_qweight"].shape and sd["qweight"].dtype == torch.int16
    assert tuple(sd["scales"].shape) == g["fast_scales"].shape and sd["scales"].dtype == torch.float16
    assert tuple(sd["qzeros"].shape) == g["fast_qzeros"].shape and sd["qzeros"].dtype == torch.float16
    assert m.interleave == 4 and m.split_k_iters == 8
    lin = torch.nn.Linear(K, N, bias=True).half()
    lin.weight.data = torch.from_numpy(g["lin_weight"])
    lin.bias.data = torch.from_numpy(g["bias"])
    G = K // gs
    scales = torch.from_numpy(g["gemv_scales"][:, :G].copy())
    zeros = torch.from_numpy(g["z_int"].T.astype(np.float32).copy())
    p = WQLinear_GEMVFast.from_linear(lin, 4, gs, False, scales, zeros)
    assert np.array_equal(p.qweight.numpy(), g["fast_qweight"])
    assert np.array_equal(p.scales.numpy().view(np.uint16), g["fast_scales"].view(np.uint16))
    assert np.array_equal(p.qzeros.numpy().view(np.uint16), g["fast_qzeros"].view(np.uint16))
    with pytest.raises(ValueError):
        p(torch.randn(4, K))  # 3-D input required (gemv_fast.py:190)


def test_no_wide_buffer_store_with_sgpr_soffset():
    """Regression guard for profiles/r01_store_hazard.txt: in the ISA hipcc generates for the two kernels
    that use buffer stores, every 12/16-byte store keeps soffset = 0 (with an SGPR soffset the compiler
    inserts no wait states before the data registers are rewritten and gfx950 reads them late)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    csrc = os.path.join(ROOT, "autoawq_amd", "csrc")
    rc, res = mod.main([os.path.join(csrc, "gemm_tiled.hip"), os.path.join(csrc, "gemv_mfma.hip"), os.path.join(csrc, "gemm_skinny.hip")])
    assert rc == 0, res
    assert sum(stores for _, stores, _ in res) > 200  # the audit really saw the exchange stores


def test_hand_counted_asm_loads_are_hazard_safe():
    """Regression guard for the two traps recorded in DESIGN.md 3.1f: in the ISA of the register-decoded kernels every
    asm block with a buffer load opens with s_nop 4, an LDS-DMA block sets M0 itself, and nothing goes through scratch."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    csrc = os.path.join(ROOT, "autoawq_amd", "csrc")
    for f, min_blocks in (("gemm_regb.hip", 10), ("gemm_skinny.hip", 10), ("prefill_attn.hip", 10)):
        name, blocks, bad = mod.audit_asm_loads(os.path.join(csrc, f))
        assert not bad, (name, bad[:5])
        assert blocks >= min_blocks, (name, blocks)


def test_gemv_layout_auto_dispatch_table_host_only():
    """awq_gemv_auto_kernel (host only): the row-streaming kernel at batches 1 - 2 (3 while K <= 6144, 4 while K <= 2048), from there the
    batched kernel (round 5, group size 128), the 16-row tile kernel otherwise -- DESIGN.md 3.0 / 3.0c."""
    from autoawq_amd import _lib

    q = lambda M, K, N, g=128: _lib.lib().awq_gemv_auto_kernel(M, K, N, g)
    ROWS, LDS, TILE = 2, 3, 1
    for K, N in [(4096, 4096), (4096, 12288), (4096, 22016), (11008, 4096), (8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)]:
        assert q(1, K, N) == ROWS, (K, N)
    assert q(2, 4096, 12288) == ROWS and q(2, 11008, 4096) == ROWS   # batch 2: ahead of the tile kernel on every 7B shape (r03 sweep)
    BATCH = 5  # round 5: csrc/gemv_batch.hip from five rows at group size 128, any M in one call; four rows while K > 2048, three while K > 6144
    assert q(3, 4096, 22016) == ROWS and q(4, 1024, 8192) == ROWS and q(4, 2048, 4096) == ROWS and q(3, 3584, 8192) == ROWS
    assert q(4, 4096, 11008) == BATCH and q(4, 11008, 4096) == BATCH and q(3, 8192, 1280) == BATCH and q(3, 11008, 4096) == BATCH
    assert q(4, 4096, 11008, 64) == TILE and q(3, 8192, 1280, 64) == TILE and q(4, 4096, 4096, 4096) == ROWS  # other group sizes: as before
    assert q(8, 4096, 11008) == BATCH and q(8, 4096, 22016) == BATCH and q(16, 4096, 11008) == BATCH and q(8, 4096, 4096) == BATCH
    assert q(5, 11008, 4096) == BATCH and q(32, 8192, 1280) == BATCH and q(64, 4096, 11008) == BATCH and q(100, 3584, 8192) == BATCH
    assert q(8, 4096, 11008, 64) == TILE and q(17, 4096, 11008, 64) == -1  # other group sizes: the 16-row kernels (the wrapper chunks)
    assert q(1, 4096, 4096, 64) == TILE       # groups below 128: the tile kernel
    assert q(0, 4096, 4096) == -1 and q(1, 4100, 4096) == -1


def test_row_streaming_kernel_never_touches_registers_with_loads_in_flight():
    """csrc/gemv_rows.hip requests ring slots in asm blocks and releases them with hand-counted waits that NAME the registers.
    hipcc believes a destination is written when the request block ends, so it may copy or reuse it while the load is still
    in flight (it did, for a drain phase with two alternative slot orders: v_mov of ring registers in front of the wait).
    tools/isa_audit.py walks every path from each request to the wait that releases it in the generated ISA."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    name, checked, bad = mod.audit_inflight_regs(os.path.join(ROOT, "autoawq_amd", "csrc", "gemv_rows.hip"))
    assert not bad, (name, bad[:5])
    assert checked >= 100, checked


def test_hand_counted_waits_of_the_register_decoded_kernels_hold_in_the_generated_isa():
    """csrc/gemm_regb.hip, gemm_skinny.hip, gemv_lds.hip and gemv_rows.hip (the headline kernel) issue their weight / activation loads in inline asm and wait with
    HAND-COUNTED `s_waitcnt vmcnt(N)` (vector-memory operations retire in issue order).  tools/isa_audit.py::audit_vmcnt
    simulates the vector-memory queue over the control-flow graph of the generated ISA -- every request enters it, a wait
    retires all but the N youngest, both sides of every branch are followed, loops run to their steady state -- and reports
    any instruction that names a VGPR whose load is still in the queue: a count that is too lax, a compiler copy or reuse of
    an in-flight register, an output of an asm block that shares the register of an address operand a later load of the
    block still reads.  (A wait that is too STRICT only costs time and is not an error here.)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for f, least in (("gemm_regb.hip", 6), ("gemm_skinny.hip", 2), ("gemv_lds.hip", 3), ("gemv_rows.hip", 64), ("gemv_batch.hip", 12),
                     ("prefill_attn.hip", 2)):
        from autoawq_amd.csrc import build as hip_build

        name, kernels, visits, bad = mod.audit_vmcnt(os.path.join(ROOT, "autoawq_amd", "csrc", f), flags=tuple(hip_build.EXTRA.get(f, [])))
        assert not bad, (name, bad[:5])
        assert kernels >= least and visits > 0, (name, kernels, visits)


def test_counted_wait_kernels_use_no_scratch():
    """The kernels whose vector-memory waits are counted by hand must not touch scratch memory: a spill is a vector-memory operation hipcc
    places where it likes -- it would sit in the same in-order queue and break every count after it (and it is slow: round 5's first
    GEMVFast form of gemv_batch spilled 600-800 bytes per lane at MI = 2 and ran 2.8x slower than the GEMV form).  hipcc's own resource
    report, per instantiation."""
    import re
    import subprocess

    from autoawq_amd.csrc import build as hip_build

    for f in ("gemv_batch.hip", "gemv_rows.hip", "gemv_lds.hip", "gemm_regb.hip", "gemm_skinny.hip", "prefill_attn.hip"):
        cmd = [hip_build.HIPCC] + hip_build.FLAGS + hip_build.EXTRA.get(f, []) + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
                                                                                  os.path.join(ROOT, "autoawq_amd", "csrc", f), "-o", os.devnull]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        sizes = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
        assert sizes and max(sizes) == 0, (f, sizes)


def test_auto_dispatch_table_host_only():
    """awq_gemm_auto_kernel is a host-only query (no launch, no GPU): which kernel awq_gemm_forward's AUTO dispatch takes
    for the BASELINE shapes, by token count -- the table DESIGN.md section 1 (row a4/a5) describes."""
    from autoawq_amd import _lib, ops

    L = _lib.lib()
    q = lambda M, K, N, g=128: L.awq_gemm_auto_kernel(M, K, N, g)
    for K, N in [(4096, 4096), (4096, 11008), (11008, 4096), (4096, 12288), (4096, 22016)]:
        for M in (1, 2, 4, 8):
            assert q(M, K, N) == ops.KERNEL_MFMA_GEMV
        for M in (17, 32, 64):  # above 32 rows the batched kernel has four reducer blocks per tile, all of which must be
            # resident at once (one 128 KB block per CU): 86 tiles x 4 > 256 -> the LDS-tiled kernel
            assert q(M, K, N) == (ops.KERNEL_TILED if (M > 32 and N > 16384) else ops.KERNEL_SKINNY)
        for M in (65, 128):
            assert q(M, K, N) == ops.KERNEL_TILED
        assert q(16384, K, N) == ops.KERNEL_REGB
    # 9 .. 16 rows: the batched kernel where it is ahead (up to 8191 rows, fewer than 64 column tiles), else the decode kernel
    assert q(16, 4096, 11008) == ops.KERNEL_SKINNY and q(9, 4096, 4096) == ops.KERNEL_SKINNY
    assert q(16, 11008, 4096) == ops.KERNEL_MFMA_GEMV and q(12, 4096, 22016) == ops.KERNEL_MFMA_GEMV
    # the prefill kernel once 128 x 256 tiles give every CU a block
    assert q(512, 4096, 11008) == ops.KERNEL_TILED and q(768, 4096, 11008) == ops.KERNEL_REGB
    assert q(1024, 11008, 4096) == ops.KERNEL_TILED and q(2048, 11008, 4096) == ops.KERNEL_REGB
    # groups of 32 rows, K not a multiple of 64: neither register-decoded kernel
    assert q(32, 4096, 4096, 32) == ops.KERNEL_TILED and q(4096, 4096, 4096, 32) == ops.KERNEL_TILED
    # shapes only the naive kernel takes (N % 32 != 0 at decode sizes), invalid shapes
    assert q(1, 256, 40) == ops.KERNEL_NAIVE
    assert q(0, 4096, 4096) == -1 and q(4, 4096, 4100) == -1


def test_prefill_route_policy_is_a_pure_function_of_shape_and_rows():
    """modules/linear/gemv.py::prefill_route / prefill_min_rows (round 6): the hand-written pair only where it measured
    ahead of or level with dequantise + dense GEMM (profiles/r06_prefill_routes.txt) -- from 3072 rows on matrices at least as wide
    as tall; the batched-decode kernel (one launch per <= 128 rows) up to 256 rows while one pass covers K, 192 beyond.  Both modules and the awq_v2_ext shim use it."""
    from autoawq_amd.modules.linear.gemv import PREFILL_MIN_ROWS, WQLinear_GEMV, prefill_min_rows, prefill_route
    from autoawq_amd.modules.linear.gemv_fast import WQLinear_GEMVFast

    assert WQLinear_GEMV.PREFILL_IMPL == "auto" and WQLinear_GEMVFast.PREFILL_IMPL == "auto"
    assert prefill_route(16384, 4096, 11008) == "hand" and prefill_route(4096, 4096, 11008) == "hand" and prefill_route(3072, 4096, 4096) == "hand"
    assert prefill_route(2048, 4096, 11008) == "two_pass" and prefill_route(128, 4096, 11008) == "two_pass"
    assert prefill_route(16384, 11008, 4096) == "two_pass"  # the tall (down-projection) shape: the vendor GEMM runs at 0.60 of peak there
    assert prefill_min_rows(4096) == PREFILL_MIN_ROWS == 257 and prefill_min_rows(11008) == 193 and prefill_min_rows(8192) == 193
