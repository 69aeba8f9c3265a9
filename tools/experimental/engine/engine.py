"""EXPERIMENT (not shipped): a chain of dependent batch-1 `WQLinear_GEMV` projections in ONE persistent launch (gemv_engine.hip here).

The reference runs the dependent projections of a decoder layer as one `awq_ext.gemv_forward_cuda` launch each
(awq/modules/linear/gemv.py:177-180 from awq/modules/fused/block.py:108-119, fused/mlp.py:46-62).  `DecodeChain` takes
the (qweight, scales, qzeros) buffers of such a run of Linears -- op i consumes the first K_i elements of op i-1's output --
and executes all of them with `awq_engine_forward`.  No CPU fallback: the HIP library must be there.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from autoawq_amd import _lib  # noqa: E402
from autoawq_amd.ops import _ptr, _require_gpu, _stream  # noqa: E402

SO = os.path.join(HERE, "libawq_engine.so")
_eng = None


def build(force=False):
    """hipcc the experiment into its own shared object next to this file (cross-compiles without a GPU)"""
    src = os.path.join(HERE, "gemv_engine.hip")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "awq_engine.h"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                               "-Wno-inline-asm", "-I" + os.path.join(ROOT, "include"), src, "-o", SO])
    return SO


def englib():
    global _eng
    if _eng is None:
        _lib.lib()
        h = ctypes.CDLL(build())
        V, I64 = ctypes.c_void_p, ctypes.c_int64
        h.awq_engine_describe.restype, h.awq_engine_describe.argtypes = ctypes.c_int, [V] * 5 + [I64] * 4 + [ctypes.c_uint64]
        h.awq_engine_granule_bytes.restype, h.awq_engine_granule_bytes.argtypes = ctypes.c_size_t, [I64]
        h.awq_engine_ctrl_bytes.restype, h.awq_engine_ctrl_bytes.argtypes = ctypes.c_size_t, []
        h.awq_engine_forward.restype, h.awq_engine_forward.argtypes = ctypes.c_int, [V, V, I64, I64, I64, V, V, ctypes.c_uint32, V]
        _eng = h
    return _eng

OP_BYTES = 64


def engine_flags(inflight=0, consumers=0, ring_slots=0, dbg=0, no_thin=0):
    """dbg (measurement only, results are wrong by design): 1 = no arithmetic, 2 = the gather accepts any tag, 4 = no weight DMA"""
    return (inflight & 0xF) | ((consumers & 0xF) << 4) | ((ring_slots & 0xF) << 8) | ((dbg & 0xF) << 12) | ((no_thin & 1) << 16)


class DecodeChain:
    """linears: sequence of (qweight [N, K/8] i32, scales [N, 8 ZW] f16, qzeros [N, ZW] i32) in GEMV layout, group_size 128.
    keep_outputs: every op's output is also written as a plain fp16 vector (`self.outputs[i]`), else only the last one."""

    def __init__(self, linears, group_size=128, keep_outputs=True):
        L = englib()
        if not linears:
            raise _lib.AwqHipError("DecodeChain: no Linears")
        dev = linears[0][0].device
        self.linears = [tuple(t for t in lin) for lin in linears]  # keeps the buffers alive
        self.shapes = []
        table = np.zeros((len(linears), OP_BYTES), dtype=np.uint8)
        goff, prev_n = 0, None
        self.outputs = []
        for i, (qw, sc, qz) in enumerate(linears):
            _require_gpu(qw, sc, qz)
            N, K = qw.shape[0], qw.shape[1] * 8
            ZW = qz.shape[1]
            if sc.shape != (N, 8 * ZW) or qz.shape[0] != N or qw.dtype != torch.int32 or qz.dtype != torch.int32 or sc.dtype != torch.float16:
                raise _lib.AwqHipError(f"DecodeChain: op {i} is not a GEMV-layout Linear")
            if not (qw.is_contiguous() and sc.is_contiguous() and qz.is_contiguous()):
                raise _lib.AwqHipError(f"DecodeChain: op {i} has non-contiguous buffers")
            if prev_n is not None and K > prev_n:
                raise _lib.AwqHipError(f"DecodeChain: op {i} reads {K} activations but op {i - 1} produces {prev_n}")
            last = i == len(linears) - 1
            y = torch.empty(N, dtype=torch.float16, device=dev) if (keep_outputs or last) else None
            self.outputs.append(y)
            rc = L.awq_engine_describe(table[i].ctypes.data_as(ctypes.c_void_p), _ptr(qw), _ptr(sc), _ptr(qz), _ptr(y), K, N,
                                       group_size, ZW, goff)
            _lib.check(rc, f"awq_engine_describe(op {i}: {K} -> {N})")
            goff += (L.awq_engine_granule_bytes(N) + 255) // 256 * 256
            prev_n = N
            self.shapes.append((K, N))
        self.table = torch.from_numpy(table).to(dev)
        self.granules = torch.zeros(goff, dtype=torch.uint8, device=dev)
        self.ctrl = torch.zeros(L.awq_engine_ctrl_bytes() // 4, dtype=torch.int32, device=dev)
        self.ctrl[0] = 1
        self.max_n = max(n for _, n in self.shapes)
        self.max_k = max(k for k, _ in self.shapes)
        self.k0 = self.shapes[0][0]

    def forward(self, x, flags=0):
        """x: fp16 [K_0] (or [1, K_0]) on the device; returns the last op's output [N_last] (a persistent buffer)."""
        _require_gpu(x)
        if x.dtype != torch.float16 or x.numel() != self.k0 or not x.is_contiguous():
            raise _lib.AwqHipError("DecodeChain.forward expects a contiguous fp16 vector of K_0 elements")
        rc = englib().awq_engine_forward(_ptr(x), _ptr(self.table), len(self.shapes), self.max_n, self.max_k, _ptr(self.granules),
                                           _ptr(self.ctrl), flags, _stream())
        _lib.check(rc, "awq_engine_forward")
        return self.outputs[-1]

    def set_trace(self, on=True):
        """debug: the next launches write [CU][op][8] wall-clock stamps (100 MHz ticks) into `self.trace`"""
        if on:
            self.trace = torch.zeros((256, len(self.shapes), 8), dtype=torch.int64, device=self.ctrl.device)
            ptr = self.trace.data_ptr()
        else:
            self.trace, ptr = None, 0
        lo, hi = ptr & 0xFFFFFFFF, ptr >> 32
        self.ctrl[8] = lo - (1 << 32) if lo >= (1 << 31) else lo
        self.ctrl[9] = hi
        return self.trace

    def status(self):
        """(epoch, blocks done, sticky error word) -- synchronises"""
        c = self.ctrl.cpu().tolist()
        return c[0], c[1], c[2] & 0xFFFFFFFF

    def check(self):
        e = self.status()[2]
        if e:
            raise _lib.AwqHipError(f"DecodeChain: a bounded spin gave up (code {e:#x}); outputs are not valid")
