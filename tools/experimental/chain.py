"""DecodeChain: a run of dependent decode-sized WQLinear_GEMM projections in ONE persistent launch.

Host-side mirror of the C ABI's `awq_chain_*` entry points (include/awq_hip.h, csrc/gemv_chain.hip).
Replaces consecutive `awq_ext.gemm_forward_cuda` / `gemv_forward_cuda` calls of a decode step
(awq/modules/linear/gemm.py:56-58, awq/modules/fused/mlp.py:37-62) where each call consumes the
previous call's output: o_proj -> gate|up -> down -> next block's qkv_proj.  PyTorch supplies the
device memory and the stream; the library neither allocates nor copies.
"""
import ctypes

import torch

from . import _lib


class ChainLink:
    """One projection of a chain.  `x`: fp16 [M, K] tensor (link 0 only).  Later links read the previous
    link's output: K columns starting at `x_col0`, or 2K columns [gate | up] with `gated=True`
    (staged as silu(gate) * up).  `y`: fp16 [M, N] tensor that receives the result, or None when the
    output is only consumed inside the chain."""

    def __init__(self, qweight, scales, qzeros, bias=None, x=None, x_col0=0, gated=False, y=None, add_residual=None):
        self.qweight, self.scales, self.qzeros, self.bias = qweight, scales, qzeros, bias
        self.x, self.x_col0, self.gated, self.y, self.add_residual = x, x_col0, gated, y, add_residual


class DecodeChain:
    def __init__(self, links, M=1, trace=False):
        L = _lib.lib()
        if not links:
            raise _lib.AwqHipError("DecodeChain: no links")
        dev = links[0].qweight.device
        self.device, self.M, self.links = dev, int(M), list(links)
        arr = (_lib.AwqChainLink * len(links))()
        self._keep = []
        for i, ln in enumerate(links):
            for t in (ln.qweight, ln.scales, ln.qzeros, ln.bias, ln.x, ln.y, ln.add_residual):
                if t is not None and (not t.is_cuda or not t.is_contiguous()):
                    raise _lib.AwqHipError("DecodeChain: tensors must be contiguous and on a HIP device")
            K, N = ln.qweight.shape[0], ln.qweight.shape[1] * 8
            G = ln.qzeros.shape[0]
            a = arr[i]
            a.qweight, a.scales, a.qzeros = ln.qweight.data_ptr(), ln.scales.data_ptr(), ln.qzeros.data_ptr()
            a.bias = ln.bias.data_ptr() if ln.bias is not None else None
            a.K, a.N, a.group_size = K, N, K // G
            if i == 0:
                if ln.x is None or ln.x.dtype != torch.float16 or ln.x.shape != (self.M, K):
                    raise _lib.AwqHipError("DecodeChain: link 0 needs x = fp16 [M, K]")
                a.x, a.x_stride, a.x_from = ln.x.data_ptr(), K, -1
            else:
                a.x, a.x_stride, a.x_from = None, 0, i - 1
            a.x_col0 = int(ln.x_col0)
            a.flags = 1 if ln.gated else 0
            if ln.y is not None and (ln.y.dtype != torch.float16 or ln.y.shape != (self.M, N)):
                raise _lib.AwqHipError("DecodeChain: y must be fp16 [M, N]")
            a.y = ln.y.data_ptr() if ln.y is not None else None
            if ln.add_residual is not None and (ln.add_residual.dtype != torch.float16 or ln.add_residual.shape != (self.M, N)):
                raise _lib.AwqHipError("DecodeChain: add_residual must be fp16 [M, N]")
            a.add_residual = ln.add_residual.data_ptr() if ln.add_residual is not None else None
        self._arr = arr
        nplan = L.awq_chain_plan_bytes(len(links))
        self._plan_host = (ctypes.c_uint8 * nplan)()
        need = ctypes.c_size_t(0)
        _lib.check(L.awq_chain_build(arr, len(links), self.M, None, 0, self._plan_host, nplan, ctypes.byref(need)),
                   "awq_chain_build (size query)")
        with torch.cuda.device(dev):
            self.workspace = torch.empty(int(need.value), dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(L.awq_chain_workspace_init(self.workspace.data_ptr(), self.workspace.numel(), st),
                       "awq_chain_workspace_init")
            _lib.check(L.awq_chain_build(arr, len(links), self.M, self.workspace.data_ptr(), self.workspace.numel(),
                                         self._plan_host, nplan, ctypes.byref(need)), "awq_chain_build")
            self.trace = None
            if trace:  # debug: per-(link, block, wave) phase stamps (tools/chain_probe.py --trace)
                import struct

                n_int, grid = struct.unpack_from("<II", self._plan_host, 4)
                self.trace = torch.zeros((n_int, grid, 10, 4), dtype=torch.int64, device=dev)
                struct.pack_into("<Q", self._plan_host, 40, self.trace.data_ptr())
            self.plan_dev = torch.frombuffer(bytearray(self._plan_host), dtype=torch.uint8).to(dev)
            torch.cuda.current_stream().synchronize()

    def forward(self):
        """Launch the chain on the current stream (hipGraph-capturable)."""
        with torch.cuda.device(self.device):
            rc = _lib.lib().awq_chain_forward(self.plan_dev.data_ptr(), self._plan_host, self.workspace.data_ptr(),
                                              self.workspace.numel(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "awq_chain_forward")

    __call__ = forward

    @staticmethod
    def grid_blocks():
        return int(_lib.lib().awq_chain_grid_blocks())

    def status(self):
        """Synchronises the current stream; 0 = healthy, else the OR of give-up codes (bit 31: aborted)."""
        err = ctypes.c_uint32(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().awq_chain_status(self.workspace.data_ptr(), torch.cuda.current_stream().cuda_stream,
                                                   ctypes.byref(err)), "awq_chain_status")
        return int(err.value)
