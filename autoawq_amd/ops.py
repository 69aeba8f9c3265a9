"""Host-side operators: torch tensors in, torch tensors out, work done by libawq_hip.so.

PyTorch is plumbing here (device memory, current stream); every computation is a hand-written
gfx950 kernel reached through the C ABI of include/awq_hip.h.
"""
import torch

from . import _lib

# kernel-selection flags (mirror include/awq_hip.h)
KERNEL_AUTO, KERNEL_NAIVE, KERNEL_VALU, KERNEL_MFMA_GEMV, KERNEL_TILED = 0, 1, 2, 3, 4
FLAG_TWO_PASS = 1 << 16
FLAG_NO_NT = 1 << 17


def gemm_flags(kernel=0, nlog=0, splitk=0, two_pass=False, no_nt=False, waves=0, unit=0):
    """nlog: VALU -> log2 column lanes (2..4); MFMA_GEMV -> packed words per lane (2|4);
    waves / unit: MFMA_GEMV waves per block and 16-row sets per wave iteration."""
    f = (kernel & 0xF) | ((nlog & 0xF) << 4) | ((splitk & 0xFF) << 8) | ((waves & 0xF) << 24) | ((unit & 0xF) << 20)
    if two_pass:
        f |= FLAG_TWO_PASS
    if no_nt:
        f |= FLAG_NO_NT
    return f


def _require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.AwqHipError(
                "autoawq_amd kernels run on a HIP device (MI355X); got a tensor on "
                f"'{t.device}'. There is no CPU fallback.")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


_workspaces = {}


def workspace(device, nbytes):
    """Split-K workspace, one per (device, stream), grown on demand and prepared once by
    awq_gemm_workspace_init (control words zero, exchange region = all-ones sentinel); every
    kernel that uses it restores that state."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _lib.check(_lib.lib().awq_gemm_workspace_init(_ptr(ws), ws.numel(), _stream()), "awq_gemm_workspace_init")
        _workspaces[key] = ws
    return ws


def _current_workspace(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    return _workspaces.get(key)


def workspace_is_clean(device):
    """True when the workspace is back in its initial state: error flag / control words zero and the
    split-K exchange region (first half after the control words) all-ones again."""
    ws = _current_workspace(device)
    if ws is None:
        return True
    half = ((ws.numel() - 16384) // 2) & ~255
    return (not bool(ws[:16384].any().item())) and bool((ws[16384:16384 + half] == 0xFF).all().item())


def unpack_int4(q):
    """[rows, words] int32 -> [rows, 8*words] uint8 logical nibbles (awq_unpack_int4)."""
    _require_gpu(q)
    q = q.contiguous()
    out = torch.empty((q.shape[0], q.shape[1] * 8), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().awq_unpack_int4(_ptr(q), _ptr(out), q.shape[0], q.shape[1], _stream()), "awq_unpack_int4")
    return out


def dequantize_weights(qweight, scales, qzeros):
    """GEMM-layout buffers -> fp16 W [K, N] (awq_dequantize_weights)."""
    _require_gpu(qweight, scales, qzeros)
    qweight, scales, qzeros = qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    K, N = qweight.shape[0], qweight.shape[1] * 8
    G = qzeros.shape[0]
    if G == 0 or K % G:
        raise _lib.AwqHipError("dequantize_weights: qzeros rows do not divide in_features")
    out = torch.empty((K, N), dtype=torch.float16, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _lib.check(_lib.lib().awq_dequantize_weights(_ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(out), K, N,
                                                     K // G, _stream()), "awq_dequantize_weights")
    return out


def gemm_forward(x2d, qweight, scales, qzeros, bias=None, flags=0):
    """y [M, N] fp16 = x2d [M, K] fp16 @ dequant(GEMM-layout buffers) (+ bias) (awq_gemm_forward)."""
    _require_gpu(x2d, qweight, scales, qzeros, bias)
    if x2d.dtype != torch.float16:
        raise _lib.AwqHipError("gemm_forward expects fp16 activations")
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    N = qweight.shape[1] * 8
    G = qzeros.shape[0]
    if qweight.shape[0] != K or G == 0 or K % G:
        raise _lib.AwqHipError(f"gemm_forward: shape mismatch x{tuple(x2d.shape)} qweight{tuple(qweight.shape)}")
    y = torch.empty((M, N), dtype=torch.float16, device=x2d.device)
    if M == 0:
        return y
    L = _lib.lib()
    with torch.cuda.device(x2d.device):
        need = L.awq_gemm_workspace_bytes(M, K, N, K // G)
        ws = workspace(x2d.device, need) if need else None
        rc = L.awq_gemm_forward(_ptr(x2d), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(y), M, K, N,
                                K // G, _ptr(ws), ws.numel() if ws is not None else 0, flags, _stream())
    _lib.check(rc, "awq_gemm_forward")
    return y


def gemv_forward(x2d, qweight, scales, qzeros, group_size, flags=0):
    """GEMV layout (qweight [N, K/8], qzeros [N, ZW], scales [N, 8*ZW]): y [M, N] fp16 = x2d @ W^T
    (awq_gemv_forward).  M is processed in chunks that fit the kernel (<= 16 rows and the LDS)."""
    _require_gpu(x2d, qweight, scales, qzeros)
    if x2d.dtype != torch.float16:
        raise _lib.AwqHipError("gemv_forward expects fp16 activations")
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    N, ZW = qweight.shape[0], qzeros.shape[1]
    if qweight.shape[1] * 8 != K or scales.shape != (N, ZW * 8):
        raise _lib.AwqHipError(f"gemv_forward: shape mismatch x{tuple(x2d.shape)} qweight{tuple(qweight.shape)} "
                               f"qzeros{tuple(qzeros.shape)} scales{tuple(scales.shape)}")
    y = torch.empty((M, N), dtype=torch.float16, device=x2d.device)
    if M == 0:
        return y
    L = _lib.lib()
    chunk = 16
    while chunk > 1 and L.awq_gemv_lds_bytes(chunk, K, ZW) > 160 * 1024:
        chunk //= 2
    with torch.cuda.device(x2d.device):
        for m0 in range(0, M, chunk):
            m1 = min(M, m0 + chunk)
            rc = L.awq_gemv_forward(_ptr(x2d[m0:m1]), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y[m0:m1]),
                                    m1 - m0, K, N, group_size, ZW, flags, _stream())
            _lib.check(rc, "awq_gemv_forward")
    return y


def dequantize_weights_gemv(qweight, scales, qzeros, group_size):
    """GEMV-layout buffers -> fp16 W^T [N, K] (awq_dequantize_weights_gemv)."""
    _require_gpu(qweight, scales, qzeros)
    qweight, scales, qzeros = qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    N, K = qweight.shape[0], qweight.shape[1] * 8
    out = torch.empty((N, K), dtype=torch.float16, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _lib.check(_lib.lib().awq_dequantize_weights_gemv(_ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(out), K, N,
                                                          group_size, qzeros.shape[1], _stream()),
                   "awq_dequantize_weights_gemv")
    return out


def has_tiled_gemm():
    """True once the fused LDS-tiled MFMA GEMM (large M) is built into the library."""
    return True


def last_kernel():
    return _lib.lib().awq_hip_last_kernel().decode()
