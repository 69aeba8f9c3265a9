"""Host-side operators: torch tensors in, torch tensors out, work done by libawq_hip.so.

PyTorch is plumbing here (device memory, current stream); every computation is a hand-written
gfx950 kernel reached through the C ABI of include/awq_hip.h.
"""
import ctypes

import torch

from . import _lib

# kernel-selection flags (mirror include/awq_hip.h)
KERNEL_AUTO, KERNEL_NAIVE, KERNEL_VALU, KERNEL_MFMA_GEMV, KERNEL_TILED, KERNEL_REGB, KERNEL_SKINNY = 0, 1, 2, 3, 4, 5, 6
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


_workspaces = {}   # (device index, stream handle) -> _Workspace
_retired = []      # outgrown workspaces: a hipGraph captured earlier may still hold their pointers
_retired_raw = []  # the same for the MoE / attention scratch tensors
_CHECK_EVERY = 64  # calls between two asynchronous read-backs of a workspace's error word


class _Workspace:
    """One split-K workspace (control words + exchange + scratch) of one stream, with an asynchronous
    watch on its error word: the kernels never hang -- a reducer that gives up waiting raises word 0
    and uses what is there (csrc/gemv_mfma.hip) -- so somebody has to look at that word.  Every
    `_CHECK_EVERY` calls a 4-byte device->pinned-host copy is queued behind the work; the next call
    that finds the copy complete inspects it, re-initialises the workspace and raises."""

    def __init__(self, device, nbytes):
        # (created OUTSIDE inference mode whatever the caller's mode is: a workspace first needed inside `torch.inference_mode()` --
        #  the fused model's forward -- would otherwise hold inference tensors, and the read-back below, issued later from an
        #  ordinary context, would raise "Inplace update to inference tensor outside InferenceMode")
        with torch.inference_mode(False):
            self.buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.calls = 0
        self.event = None
        self.init()

    def init(self):
        _lib.check(_lib.lib().awq_gemm_workspace_init(_ptr(self.buf), self.buf.numel(), _stream()), "awq_gemm_workspace_init")

    def numel(self):
        return self.buf.numel()

    def poll(self):
        """non-blocking: look at a finished read-back, start a new one now and then (never while capturing)"""
        if torch.cuda.is_current_stream_capturing():
            return
        if self.event is not None and self.event.query():
            self.event = None
            if int(self.host[0]) != 0:
                self.host.zero_()
                self.init()
                raise _lib.AwqHipError("a split-K reducer gave up waiting for its partial sums on this workspace "
                                       "(control word != 0): results since the last check are unreliable; the "
                                       "workspace has been re-initialised")
        self.calls += 1
        if self.event is None and self.calls % _CHECK_EVERY == 0:
            self.host.copy_(self.buf[:4].view(torch.int32), non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()


def workspace(device, nbytes):
    """Split-K workspace, one per (device, stream), grown on demand and prepared once by
    awq_gemm_workspace_init (control words zero, exchange region = all-ones sentinel); every kernel
    that uses it restores that state.  A workspace that has to grow is RETIRED, not freed: a hipGraph
    captured earlier on the stream keeps replaying into the old buffer (`release_workspaces()` drops
    them all once no such graph is alive)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired.append(ws)
        ws = _Workspace(device, nbytes)
        _workspaces[key] = ws
    ws.poll()
    return ws.buf


def _current_workspace(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _workspaces.get(key)
    return None if ws is None else ws.buf


def check_workspaces():
    """Blocking form of the error watch: synchronises and raises if any workspace's error word is set
    (after re-initialising it)."""
    torch.cuda.synchronize()
    bad = []
    for key, ws in list(_workspaces.items()) + [(None, w) for w in _retired]:
        if int(ws.buf[:4].view(torch.int32).item()) != 0:
            bad.append(key)
            with torch.cuda.device(ws.buf.device):
                ws.init()
    for d in (_moe_workspaces,):
        for key, buf in d.items():
            if int(buf[:4].view(torch.int32).item()) != 0:
                bad.append(key)
                with torch.cuda.device(buf.device):
                    _lib.check(_lib.lib().awq_gemm_workspace_init(_ptr(buf), buf.numel(), _stream()), "awq_gemm_workspace_init")
    torch.cuda.synchronize()
    if bad:
        raise _lib.AwqHipError(f"split-K error word set on workspace(s) {bad}: a reducer gave up waiting; re-initialised")


def release_workspaces():
    """Free every workspace (current and retired).  Only when no captured hipGraph that used them is
    going to be replayed again."""
    torch.cuda.synchronize()
    _workspaces.clear()
    _moe_workspaces.clear()
    _attn_workspaces.clear()
    del _retired[:]
    del _retired_raw[:]


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


X_GATED_SILU = 1 << 18  # AWQ_GEMM_FLAG_X_GATED_SILU


def gemm_forward(x2d, qweight, scales, qzeros, bias=None, flags=0):
    """y [M, N] fp16 = x2d [M, K] fp16 @ dequant(GEMM-layout buffers) (+ bias) (awq_gemm_forward).
    With flags | X_GATED_SILU, x2d is [M, 2K] = [gate | up] and the kernel applies silu(gate) * up
    while staging (M <= 16)."""
    _require_gpu(x2d, qweight, scales, qzeros, bias)
    if x2d.dtype != torch.float16:
        raise _lib.AwqHipError("gemm_forward expects fp16 activations")
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    if flags & X_GATED_SILU:
        K //= 2
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


def grouped_gemm_prefill(x_sorted, qweight, scales, qzeros, seg_offsets, flags=0):
    """MoE prefill (awq_grouped_gemm_prefill): x_sorted [P, K] fp16 = the (token, expert) pairs' rows sorted by expert,
    seg_offsets [E + 1] int32 on the device, qweight [E, K, N/8] / scales [E, K/g, N] / qzeros [E, K/g, N/8] the stacked
    GEMM-layout experts -> y [P, N].  No routing data is read back: capturable."""
    _require_gpu(x_sorted, qweight, scales, qzeros, seg_offsets)
    if x_sorted.dtype != torch.float16 or seg_offsets.dtype != torch.int32:
        raise _lib.AwqHipError("grouped_gemm_prefill expects fp16 activations and int32 offsets")
    x_sorted, qweight, scales, qzeros = x_sorted.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    P, K = x_sorted.shape
    E, N, G = qweight.shape[0], qweight.shape[2] * 8, qzeros.shape[1]
    if qweight.shape[1] != K or scales.shape != (E, G, N) or qzeros.shape != (E, G, N // 8) or seg_offsets.numel() != E + 1 or K % G:
        raise _lib.AwqHipError(f"grouped_gemm_prefill: shape mismatch x{tuple(x_sorted.shape)} qweight{tuple(qweight.shape)} "
                               f"scales{tuple(scales.shape)} qzeros{tuple(qzeros.shape)} seg{tuple(seg_offsets.shape)}")
    y = torch.empty((P, N), dtype=torch.float16, device=x_sorted.device)
    with torch.cuda.device(x_sorted.device):
        rc = _lib.lib().awq_grouped_gemm_prefill(_ptr(x_sorted), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y),
                                                 _ptr(seg_offsets.contiguous()), P, E, K, N, K // G, flags, _stream())
    _lib.check(rc, "awq_grouped_gemm_prefill")
    return y


GROUPED_PREFILL_GATHER_X, GROUPED_PREFILL_SCATTER_Y = 1 << 19, 1 << 20


def moe_sort_pairs(topk_ids, num_experts):
    """(order [P] int32, seg_offsets [E + 1] int32): the pairs of topk_ids [T, topk] grouped by expert as an index list, one
    launch (awq_moe_sort_pairs) == torch.argsort(topk_ids.flatten(), stable=True) + the experts' row ranges."""
    _require_gpu(topk_ids)
    ids = topk_ids.contiguous().view(-1)
    if ids.dtype != torch.int32:
        raise _lib.AwqHipError("moe_sort_pairs expects int32 topk_ids")
    order = torch.empty_like(ids)
    seg = torch.empty((num_experts + 1,), dtype=torch.int32, device=ids.device)
    with torch.cuda.device(ids.device):
        _lib.check(_lib.lib().awq_moe_sort_pairs(_ptr(ids), _ptr(order), _ptr(seg), ids.numel(), num_experts, _stream()),
                   "awq_moe_sort_pairs")
    return order, seg


def grouped_gemm_prefill_ex(x, qweight, scales, qzeros, seg_offsets, row_map, x_div=1, gather=False, scatter=False,
                            pair_weights=None, flags=0, zero_init=False):
    """awq_grouped_gemm_prefill_ex: the grouped prefill GEMM with the sort kept as an index list.  gather: x [T, K] holds the
    tokens, sorted row r reads row row_map[r] / x_div; else x [P, K] is sorted.  scatter: y row row_map[r] (pair order) instead
    of r; pair_weights [P] fp32 by pair index multiply the fp32 product before its one rounding."""
    _require_gpu(x, qweight, scales, qzeros, seg_offsets, row_map, pair_weights)
    if x.dtype != torch.float16 or seg_offsets.dtype != torch.int32 or row_map.dtype != torch.int32:
        raise _lib.AwqHipError("grouped_gemm_prefill_ex expects fp16 activations and int32 offsets / row map")
    x, qweight, scales, qzeros = x.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    P, K = row_map.numel(), x.shape[1]
    E, N, G = qweight.shape[0], qweight.shape[2] * 8, qzeros.shape[1]
    if qweight.shape[1] != K or scales.shape != (E, G, N) or qzeros.shape != (E, G, N // 8) or seg_offsets.numel() != E + 1 or K % G:
        raise _lib.AwqHipError(f"grouped_gemm_prefill_ex: shape mismatch x{tuple(x.shape)} qweight{tuple(qweight.shape)} "
                               f"scales{tuple(scales.shape)} qzeros{tuple(qzeros.shape)} seg{tuple(seg_offsets.shape)}")
    if (gather and x.shape[0] * x_div < P) or (not gather and x.shape[0] != P):
        raise _lib.AwqHipError(f"grouped_gemm_prefill_ex: x{tuple(x.shape)} does not cover {P} pairs (x_div {x_div})")
    w = pair_weights.contiguous().float().view(-1) if pair_weights is not None else None
    if w is not None and w.numel() != P:
        raise _lib.AwqHipError("grouped_gemm_prefill_ex: one routing weight per pair")
    y = (torch.zeros if zero_init else torch.empty)((P, N), dtype=torch.float16, device=x.device)
    f = flags | (GROUPED_PREFILL_GATHER_X if gather else 0) | (GROUPED_PREFILL_SCATTER_Y if scatter else 0)
    with torch.cuda.device(x.device):
        rc = _lib.lib().awq_grouped_gemm_prefill_ex(_ptr(x), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y),
                                                    _ptr(seg_offsets.contiguous()), _ptr(row_map.contiguous()), _ptr(w), P, x_div, E, K, N,
                                                    K // G, f, _stream())
    _lib.check(rc, "awq_grouped_gemm_prefill_ex")
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
    if (flags & 0xF) == GEMV_KERNEL_PREFILL:
        # explicit only: ONE call of the register-decoded MFMA GEMM on this layout's own buffers (any M); AUTO keeps the
        # 16-row chunks of the decode kernels (include/awq_hip.h: the kernel is latency-bound below ~2000 rows)
        with torch.cuda.device(x2d.device):
            rc = L.awq_gemv_forward(_ptr(x2d), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y), M, K, N, group_size, ZW, flags,
                                    _stream())
        _lib.check(rc, "awq_gemv_forward")
        return y
    kern = flags & 0xF
    if kern == GEMV_KERNEL_BATCH or (kern == 0 and L.awq_gemv_auto_kernel(M, K, N, group_size) == GEMV_KERNEL_BATCH):
        # round 5: the batched kernel (csrc/gemv_batch.hip) takes ANY M in one call (launches of <= 128 rows inside the library)
        with torch.cuda.device(x2d.device):
            rc = L.awq_gemv_forward(_ptr(x2d), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y), M, K, N, group_size, ZW, flags,
                                    _stream())
        _lib.check(rc, "awq_gemv_forward")
        return y
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


def gemv_auto_kernel(M, K, N, group_size):
    """Which AWQ_GEMV_KERNEL_* awq_gemv_forward's AUTO dispatch takes for this shape (host only; -1: none)."""
    return int(_lib.lib().awq_gemv_auto_kernel(M, K, N, group_size))


GEMV_EX_SILU_PAIRS = 1  # include/awq_hip.h AWQ_GEMV_EX_SILU_PAIRS
GEMV_KERNEL_PREFILL = 4  # include/awq_hip.h AWQ_GEMV_KERNEL_PREFILL
GEMV_KERNEL_BATCH = 5    # include/awq_hip.h AWQ_GEMV_KERNEL_BATCH


def gemv_forward_ex(x2d, qweight, scales, qzeros, group_size, norm_weight=None, norm_eps=0.0, add_residual=None,
                    silu_pairs=False):
    """GEMV-layout decode projection with the decoder block's prologue / epilogue in the same launch
    (awq_gemv_forward_ex): norm_weight normalises x on its way into registers; add_residual stores
    fp16(fp16(W x) + add_residual); silu_pairs: rows (2 i, 2 i + 1) are (gate_i, up_i), returns [1, N / 2] =
    silu(gate) * up.  Raises AwqHipError with code AWQ_ERR_UNSUPPORTED (-3) for shapes the row-streaming kernel
    does not take (M > 1, K > 16384): the caller then runs the separate launches."""
    _require_gpu(x2d, qweight, scales, qzeros, norm_weight, add_residual)
    if x2d.dtype != torch.float16:
        raise _lib.AwqHipError("gemv_forward_ex expects fp16 activations")
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    N, ZW = qweight.shape[0], qzeros.shape[1]
    if qweight.shape[1] * 8 != K or scales.shape != (N, ZW * 8):
        raise _lib.AwqHipError(f"gemv_forward_ex: shape mismatch x{tuple(x2d.shape)} qweight{tuple(qweight.shape)} "
                               f"qzeros{tuple(qzeros.shape)} scales{tuple(scales.shape)}")
    n_out = N // 2 if silu_pairs else N
    y = torch.empty((M, n_out), dtype=torch.float16, device=x2d.device)
    if add_residual is not None and (add_residual.shape != y.shape or add_residual.dtype != torch.float16
                                     or not add_residual.is_contiguous()):
        raise _lib.AwqHipError("gemv_forward_ex: add_residual must be a contiguous fp16 [M, N] tensor")
    if norm_weight is not None and (norm_weight.dtype != torch.float16 or norm_weight.numel() != K):
        raise _lib.AwqHipError("gemv_forward_ex: norm_weight must be fp16 [K]")
    with torch.cuda.device(x2d.device):
        e = _lib.AwqGemvEx()
        e.struct_bytes = ctypes.sizeof(_lib.AwqGemvEx)
        e.flags = GEMV_EX_SILU_PAIRS if silu_pairs else 0
        e.x, e.qweight, e.scales, e.qzeros, e.y = _ptr(x2d), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y)
        e.M, e.K, e.N, e.group_size, e.zeros_width = M, K, N, group_size, ZW
        e.stream = _stream()
        e.norm_weight = _ptr(norm_weight.contiguous()) if norm_weight is not None else None
        e.norm_eps = float(norm_eps)
        e.add_residual = _ptr(add_residual)
        rc = _lib.lib().awq_gemv_forward_ex(ctypes.byref(e))
    _lib.check(rc, "awq_gemv_forward_ex")
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


def repack_gemv_to_gemm(qweight, scales, qzeros, group_size):
    """GEMV-layout buffers -> (qweight [K, N/8], scales [K/g, N], qzeros [K/g, N/8]) in the GEMM layout: the same integers and
    scales, bit for bit (awq_repack_gemv_to_gemm, csrc/repack.hip).  The outputs are fresh temporaries."""
    _require_gpu(qweight, scales, qzeros)
    qweight, scales, qzeros = qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    N, K = qweight.shape[0], qweight.shape[1] * 8
    G = K // group_size
    qw = torch.empty((K, N // 8), dtype=torch.int32, device=qweight.device)
    sc = torch.empty((G, N), dtype=torch.float16, device=qweight.device)
    qz = torch.empty((G, N // 8), dtype=torch.int32, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _lib.check(_lib.lib().awq_repack_gemv_to_gemm(_ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(qw), _ptr(sc), _ptr(qz), K, N,
                                                      group_size, qzeros.shape[1], _stream()), "awq_repack_gemv_to_gemm")
    return qw, sc, qz


def gemv_prefill_repack(x2d, qweight, scales, qzeros, group_size, bias=None):
    """Prefill-sized batches on the GEMV layout's own buffers: transpose the packed nibbles into a temporary of this call
    (repack_gemv_to_gemm) and run the fused MFMA GEMM on it (gemm_forward) -- two hand-written launches, nothing resident."""
    qw, sc, qz = repack_gemv_to_gemm(qweight, scales, qzeros, group_size)
    return gemm_forward(x2d, qw, sc, qz, bias)


def silu_and_mul(gate_up, out=None):
    """[..., 2d] fp16 = [gate | up] -> [..., d] = silu(gate) * up (awq_silu_and_mul)."""
    _require_gpu(gate_up)
    gate_up = gate_up.contiguous()
    d = gate_up.shape[-1] // 2
    if out is None:
        out = torch.empty(gate_up.shape[:-1] + (d,), dtype=torch.float16, device=gate_up.device)
    rows = gate_up.numel() // (2 * d) if d else 0
    with torch.cuda.device(gate_up.device):
        _lib.check(_lib.lib().awq_silu_and_mul(_ptr(gate_up), _ptr(out), rows, d, _stream()), "awq_silu_and_mul")
    return out


_moe_workspaces = {}


def grouped_gemm_forward(x, qweight, scales, qzeros, topk_weights, sorted_token_ids, expert_ids,
                         num_tokens_post_padded, mul_weights, split_k_iters=8, block_rows=16, zero_init=False, x_gated=False):
    """awq_ext.grouped_gemm_forward semantics (awq/modules/fused/moe.py:60-89): x [T, 1 | topk, K] fp16,
    stacked GEMM-layout expert tensors [E, ...]; returns [T, topk, N] fp16.  Nothing is read back
    to the host: the routing tensors are consumed on the device.  block_rows = the block size the
    routing tensors were aligned with (16 like the reference, or 8: decode-sized batches)."""
    _require_gpu(x, qweight, scales, qzeros, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_padded)
    T, topk = topk_weights.shape
    E, K, NW = qweight.shape
    N = NW * 8
    G = qzeros.shape[1]
    x = x.contiguous()
    x_div = topk if x.shape[1] == 1 else 1
    # x_gated: x holds [gate | up] rows of 2 K halves; silu(gate) * up is applied while the kernel stages its activations
    if x.shape[0] * x.shape[1] * x_div != T * topk or x.shape[-1] != (2 * K if x_gated else K):
        raise _lib.AwqHipError(f"grouped_gemm_forward: x{tuple(x.shape)} does not match {T} tokens x top-{topk}")
    qweight, scales, qzeros = qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    sorted_token_ids = sorted_token_ids.contiguous()
    # zero_init: rows of pairs that no block covers (expert-parallel routing places only the local experts' pairs) read as 0
    y = (torch.zeros if zero_init else torch.empty)((T, topk, N), dtype=torch.float16, device=x.device)
    # expert_ids has the reference's capacity (one entry per pair + E, moe.py:121-123); the blocks that can exist are bounded by
    # the padded row capacity of sorted_token_ids: that bound sizes the grid and the split-K exchange
    max_blocks = max(1, min(expert_ids.numel(), sorted_token_ids.numel() // block_rows))
    L = _lib.lib()
    with torch.cuda.device(x.device):
        need = L.awq_grouped_gemm_workspace_bytes(max_blocks, K, N)
        key = (x.device.index, _stream())
        ws = _moe_workspaces.get(key)
        if ws is None or ws.numel() < need:
            if ws is not None:
                _retired_raw.append(ws)  # a captured graph may still replay into it
            ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
            _lib.check(L.awq_gemm_workspace_init(_ptr(ws), ws.numel(), _stream()), "awq_gemm_workspace_init")
            _moe_workspaces[key] = ws
        w = topk_weights.contiguous().float() if mul_weights else None
        rc = L.awq_grouped_gemm_forward_ex(_ptr(x), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y),
                                           _ptr(sorted_token_ids), _ptr(expert_ids), _ptr(num_tokens_post_padded),
                                           _ptr(w), T * topk, x_div, block_rows, max_blocks, E, K, N, K // G, _ptr(ws),
                                           ws.numel(), X_GATED_SILU if x_gated else 0, _stream())
    _lib.check(rc, "awq_grouped_gemm_forward")
    return y


def grouped_gemv_forward(x, qweight, scales, qzeros, topk_ids, group_size, topk_weights=None, silu_pairs=False, parts=0,
                         zero_init=False, first_expert=0):
    """MoE decode on GEMV-layout expert stacks (awq_grouped_gemv_forward; the decode-sized calls of
    awq/modules/fused/moe.py:60-89): x [T, K] (every token feeds its topk pairs) or [T * topk, K] (one row per pair) fp16,
    qweight [E, N, K/8] i32, qzeros [E, N, ZW] i32, scales [E, N, 8 ZW] fp16, topk_ids [T, topk] i32 on the device.
    Returns [T, topk, N] (N / 2 with silu_pairs: rows (2 j, 2 j + 1) of every expert = (gate_j, up_j), silu(gate) * up
    written by the launch); topk_weights [T, topk] fp32 multiplies each pair's row before its one rounding.  One launch of
    the row-streaming kernel for all pairs; the stack holds experts [first_expert, first_expert + E) of topk_ids' global ids
    and pairs of other experts are skipped (expert-parallel shards; zero_init: their rows read as 0)."""
    _require_gpu(x, qweight, scales, qzeros, topk_ids, topk_weights)
    if x.dtype != torch.float16 or topk_ids.dtype != torch.int32:
        raise _lib.AwqHipError("grouped_gemv_forward expects fp16 activations and int32 topk_ids")
    T, topk = topk_ids.shape
    E, N, KW = qweight.shape
    K, ZW = KW * 8, qzeros.shape[2]
    P = T * topk
    x = x.contiguous()
    if x.dim() != 2 or x.shape[1] != K or x.shape[0] not in (T, P):
        raise _lib.AwqHipError(f"grouped_gemv_forward: x{tuple(x.shape)} does not match {T} tokens x top-{topk}, K = {K}")
    x_div = topk if x.shape[0] == T and topk > 1 else 1
    if qzeros.shape[:2] != (E, N) or scales.shape != (E, N, ZW * 8):
        raise _lib.AwqHipError(f"grouped_gemv_forward: shape mismatch qweight{tuple(qweight.shape)} qzeros{tuple(qzeros.shape)} "
                               f"scales{tuple(scales.shape)}")
    qweight, scales, qzeros, topk_ids = qweight.contiguous(), scales.contiguous(), qzeros.contiguous(), topk_ids.contiguous()
    w = topk_weights.contiguous().float() if topk_weights is not None else None
    if w is not None and w.numel() != P:
        raise _lib.AwqHipError("grouped_gemv_forward: topk_weights must hold one weight per pair")
    y = (torch.zeros if zero_init else torch.empty)((T, topk, N // 2 if silu_pairs else N), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().awq_grouped_gemv_forward(_ptr(x), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y), _ptr(topk_ids),
                                                 _ptr(w), P, x_div, E, first_expert, K, N, group_size, ZW,
                                                 GEMV_EX_SILU_PAIRS if silu_pairs else 0, parts, _stream())
    _lib.check(rc, "awq_grouped_gemv_forward")
    return y


def moe_route(gating_output, topk, renormalize, block_size, first_expert=0, num_local=None):
    """softmax + top-k (+ renormalise) + block alignment in ONE launch (awq_moe_route): returns
    (topk_weights [T, k] fp32, topk_ids [T, k] i32, sorted_token_ids, expert_ids, num_tokens_post_padded).
    block_size = 0: routing only (the last three are None).
    first_expert / num_local (expert parallel, awq_moe_route_local): only the pairs of experts
    [first_expert, first_expert + num_local) are placed, expert_ids are relative to first_expert; topk_ids stay global."""
    _require_gpu(gating_output)
    g = gating_output.float().contiguous()
    T, E = g.shape
    nl = E if num_local is None else num_local
    dev = g.device
    w = torch.empty((T, topk), dtype=torch.float32, device=dev)
    ids = torch.empty((T, topk), dtype=torch.int32, device=dev)
    if block_size == 0:  # routing only: no alignment pass (the row-streaming MoE decode path runs pairs, not blocks)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().awq_moe_route_local(_ptr(g), _ptr(w), _ptr(ids), None, None, None, T, E, topk,
                                                      1 if renormalize else 0, 0, first_expert, nl, _stream()), "awq_moe_route_local")
        return w, ids, None, None, None
    sorted_ids = torch.empty((T * topk + nl * (block_size - 1),), dtype=torch.int32, device=dev)
    expert_ids = torch.empty((T * topk + nl,), dtype=torch.int32, device=dev)
    npad = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().awq_moe_route_local(_ptr(g), _ptr(w), _ptr(ids), _ptr(sorted_ids), _ptr(expert_ids), _ptr(npad), T, E,
                                                  topk, 1 if renormalize else 0, block_size, first_expert, nl, _stream()),
                   "awq_moe_route_local")
    return w, ids, sorted_ids, expert_ids, npad


def moe_align_block_size(topk_ids, block_size, num_experts):
    """Device-side (no host read) restatement of moe_align_block_size (awq/modules/fused/moe.py:94-134):
    returns (sorted_token_ids [numel + E*(block-1)] padded with `numel`, expert_ids [numel + E] one per
    block (entries past the used blocks are 0), num_tokens_post_padded [1]) as int32 tensors."""
    flat = topk_ids.reshape(-1).to(torch.int64)
    numel = flat.numel()
    dev = flat.device
    # one-hot sum instead of torch.bincount: bincount reads its output size back to the host
    counts = (flat.unsqueeze(1) == torch.arange(num_experts, device=dev).unsqueeze(0)).sum(0)
    padded = (counts + block_size - 1) // block_size * block_size
    pad_end = torch.cumsum(padded, 0)
    pad_start = pad_end - padded
    raw_start = torch.cumsum(counts, 0) - counts
    order = torch.argsort(flat, stable=True)                 # pairs grouped by expert, original order kept
    e_sorted = flat[order]
    rank = torch.arange(numel, device=dev) - raw_start[e_sorted]
    pos = pad_start[e_sorted] + rank
    sorted_ids = torch.full((numel + num_experts * (block_size - 1),), numel, dtype=torch.int32, device=dev)
    sorted_ids[pos] = order.to(torch.int32)
    nblk = numel + num_experts  # capacity used by the reference (moe.py:121-123)
    blk_first_row = torch.arange(nblk, device=dev) * block_size
    expert_ids = torch.searchsorted(pad_end, blk_first_row, right=True).clamp_(max=num_experts - 1).to(torch.int32)
    return sorted_ids, expert_ids, pad_end[-1:].to(torch.int32)


def fused_topk(gating_output, topk, renormalize):
    """awq/modules/fused/moe.py:137-171 (the branch the reference itself takes on ROCm)."""
    routing = torch.softmax(gating_output, dim=-1, dtype=torch.float32)
    w, ids = torch.topk(routing, topk, dim=-1)
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return w, ids.to(torch.int32)


def gemv_fast_forward(x2d, qweight, scales, qzeros, group_size, flags=0):
    """GEMVFast layout (qweight int16 [N/4, K], scales / qzeros fp16 [8*ZW, N]): y [M, N] fp16
    (awq_gemv_fast_forward).  AUTO: the batched kernel (csrc/gemv_batch.hip, GEMVFast form) from five rows -- at every batch size while
    2048 < K <= 4096 -- in one call; otherwise the 16-row kernel (csrc/gemv_fast.hip) in chunks of <= 16 rows."""
    _require_gpu(x2d, qweight, scales, qzeros)
    if x2d.dtype != torch.float16:
        raise _lib.AwqHipError("gemv_fast_forward expects fp16 activations")
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    N, GP = qweight.shape[0] * 4, scales.shape[0]
    if qweight.shape[1] != K or scales.shape != (GP, N) or qzeros.shape != (GP, N):
        raise _lib.AwqHipError(f"gemv_fast_forward: shape mismatch x{tuple(x2d.shape)} qweight{tuple(qweight.shape)} "
                               f"scales{tuple(scales.shape)}")
    y = torch.empty((M, N), dtype=torch.float16, device=x2d.device)
    if M == 0:
        return y
    L = _lib.lib()
    kern = flags & 0xF
    if (kern == GEMV_KERNEL_BATCH or (kern == 0 and M >= 5)):
        # round 5: the batched kernel in its GEMVFast form takes ANY M in one call (launches of <= 128 rows inside the library); shapes it
        # refuses (group sizes other than 128) fall through to the 16-row kernel in chunks
        with torch.cuda.device(x2d.device):
            rc = L.awq_gemv_fast_forward(_ptr(x2d), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y), M, K, N, group_size, GP, flags, _stream())
        if rc != _lib.ERR_UNSUPPORTED or kern == GEMV_KERNEL_BATCH:
            _lib.check(rc, "awq_gemv_fast_forward")
            return y
    chunk = 16
    while chunk > 1 and L.awq_gemv_fast_lds_bytes_c(chunk, K, group_size) > 160 * 1024:
        chunk //= 2
    with torch.cuda.device(x2d.device):
        for m0 in range(0, M, chunk):
            m1 = min(M, m0 + chunk)
            rc = L.awq_gemv_fast_forward(_ptr(x2d[m0:m1]), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y[m0:m1]),
                                         m1 - m0, K, N, group_size, GP, flags, _stream())
            _lib.check(rc, "awq_gemv_fast_forward")
    return y


def gemv_fast_prefill(x2d, qweight, scales, qzeros, group_size, flags=0):
    """GEMVFast layout, prefill-sized token counts: y [M, N] = x @ W^T, W = fp16(w * s + qzeros), by TWO hand-written launches
    (awq_gemv_fast_prefill: the packed words transposed into a temporary of the call, then the register-decoded MFMA GEMM with this
    format's scales / fp16 zero terms) -- the role of awq_v2_ext.gemm_forward_cuda_prefill (awq/modules/linear/gemv_fast.py:203-206).
    Raises AwqHipError with code ERR_UNSUPPORTED for shapes the fused kernel does not take (K % 64, group_size % 64, N % 8)."""
    _require_gpu(x2d, qweight, scales, qzeros)
    if x2d.dtype != torch.float16:
        raise _lib.AwqHipError("gemv_fast_prefill expects fp16 activations")
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    N, GP = qweight.shape[0] * 4, scales.shape[0]
    if qweight.shape[1] != K or scales.shape != (GP, N) or qzeros.shape != (GP, N):
        raise _lib.AwqHipError(f"gemv_fast_prefill: shape mismatch x{tuple(x2d.shape)} qweight{tuple(qweight.shape)} scales{tuple(scales.shape)}")
    y = torch.empty((M, N), dtype=torch.float16, device=x2d.device)
    if M == 0:
        return y
    tmp = torch.empty((K, max(N // 8, 1)), dtype=torch.int32, device=x2d.device)  # GEMM-layout words of the same integers: a temporary of the call
    with torch.cuda.device(x2d.device):
        rc = _lib.lib().awq_gemv_fast_prefill(_ptr(x2d), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(y), _ptr(tmp), M, K, N, group_size, GP,
                                              flags, _stream())
    _lib.check(rc, "awq_gemv_fast_prefill")
    return y


def repack_gemvfast_to_gemm(qweight):
    """GEMVFast words int16 [N/4, K] -> GEMM-layout words int32 [K, N/8] (awq_repack_gemvfast_to_gemm), bit-exact"""
    _require_gpu(qweight)
    qweight = qweight.contiguous()
    N, K = qweight.shape[0] * 4, qweight.shape[1]
    out = torch.empty((K, N // 8), dtype=torch.int32, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _lib.check(_lib.lib().awq_repack_gemvfast_to_gemm(_ptr(qweight), _ptr(out), K, N, _stream()), "awq_repack_gemvfast_to_gemm")
    return out


def dequantize_weights_gemv_fast(qweight, scales, qzeros, group_size):
    """GEMVFast-layout buffers -> fp16 W^T [N, K] (awq_dequantize_weights_gemv_fast)."""
    _require_gpu(qweight, scales, qzeros)
    qweight, scales, qzeros = qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    N, K = qweight.shape[0] * 4, qweight.shape[1]
    out = torch.empty((N, K), dtype=torch.float16, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _lib.check(_lib.lib().awq_dequantize_weights_gemv_fast(_ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(out), K, N,
                                                               group_size, _stream()), "awq_dequantize_weights_gemv_fast")
    return out


def auto_kernel(M, K, N, group_size):
    """KERNEL_* the AUTO dispatch of gemm_forward takes for this shape (host-only query, no launch)."""
    return _lib.lib().awq_gemm_auto_kernel(M, K, N, group_size)


def last_kernel():
    return _lib.lib().awq_hip_last_kernel().decode()


# ---- fused decoder block (csrc/decoder.hip)
def rmsnorm(x, weight, eps, residual=None, out=None):
    """out = rmsnorm(x) * weight (awq_rmsnorm_forward); with `residual` (same shape, updated in place):
    residual += x, out = rmsnorm(residual) * weight."""
    _require_gpu(x, weight, residual)
    if x.dtype != torch.float16 or weight.dtype != torch.float16:
        raise _lib.AwqHipError("rmsnorm expects fp16 tensors")
    x = x.contiguous()
    H = x.shape[-1]
    M = x.numel() // H if H else 0
    if out is None:
        out = torch.empty_like(x)
    if residual is not None and (not residual.is_contiguous() or residual.shape != x.shape or residual.dtype != x.dtype):
        raise _lib.AwqHipError("rmsnorm: residual must be a contiguous fp16 tensor of x's shape")
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().awq_rmsnorm_forward(_ptr(x), _ptr(residual), _ptr(weight.contiguous()), _ptr(out), M, H,
                                                  float(eps), _stream()), "awq_rmsnorm_forward")
    return out


def rope_kv_append(qkv, k_cache, v_cache, cos, sin, start_pos, n_heads, n_kv_heads, head_dim, rotary_dim, pos_dev=None):
    """qkv [B, S, (Hq + 2 Hkv) * D] fp16 -> rotated q [B, S, Hq, D]; rotated k and v are written into
    the caches [B, Tmax, Hkv, D] at rows start_pos .. start_pos + S - 1 (awq_rope_kv_append)."""
    _require_gpu(qkv, k_cache, v_cache, cos, sin, pos_dev)
    B, S = qkv.shape[0], qkv.shape[1]
    qkv = qkv.contiguous()
    q = torch.empty((B, S, n_heads, head_dim), dtype=torch.float16, device=qkv.device)
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()) or k_cache.shape[0] < B:
        raise _lib.AwqHipError("rope_kv_append: caches must be contiguous [>=B, Tmax, Hkv, D]")
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.lib().awq_rope_kv_append(_ptr(qkv), _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(cos), _ptr(sin),
                                                 _ptr(pos_dev), int(start_pos), B, S, n_heads, n_kv_heads, head_dim,
                                                 rotary_dim, k_cache.shape[1], _stream()), "awq_rope_kv_append")
    return q


def prefill_attention(q, k_cache, v_cache, start_pos, scale=None, softcap=0.0, alibi_slopes=None):
    """q [B, S, Hq, 128] fp16 (after RoPE), caches [>= B, Tmax, Hkv, 128] fp16 already holding rows 0 .. start_pos + S - 1 ->
    [B, S, Hq, 128]: causal attention of the S new rows over the cache (awq_prefill_attention, csrc/prefill_attn.hip; the
    reference's flash_attn_func call, attn.py:269-277).  Raises AwqHipError code -3 for head sizes other than 128."""
    _require_gpu(q, k_cache, v_cache, alibi_slopes)
    if q.dtype != torch.float16 or k_cache.dtype != torch.float16 or v_cache.dtype != torch.float16:
        raise _lib.AwqHipError("prefill_attention expects fp16 tensors")
    B, S, Hq, D = q.shape
    Tmax, Hkv = k_cache.shape[1], k_cache.shape[2]
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()) or k_cache.shape[0] < B or k_cache.shape != v_cache.shape:
        raise _lib.AwqHipError("prefill_attention: caches must be contiguous [>=B, Tmax, Hkv, D]")
    q = q.contiguous()
    out = torch.empty_like(q)
    if alibi_slopes is not None:
        alibi_slopes = alibi_slopes.to(torch.float32).contiguous()
        if alibi_slopes.numel() != Hq:
            raise _lib.AwqHipError("prefill_attention: alibi_slopes must have one entry per query head")
    if scale is None:
        scale = D ** -0.5
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().awq_prefill_attention(_ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(out), B, S, Hq, Hkv, D, Tmax, int(start_pos),
                                                    float(scale), float(softcap), _ptr(alibi_slopes), _stream()), "awq_prefill_attention")
    return out


_attn_workspaces = {}


def decode_attention(q, k_cache, v_cache, seq_len, scale=None, len_dev=None, max_len=None, softcap=0.0, alibi_slopes=None):
    """q [B, Hq, 128] fp16, caches [B, Tmax, Hkv, 128] -> [B, Hq, 128]: attention of ONE query token per
    sequence over cache rows [0, seq_len) (awq_decode_attention; with logit soft-capping / ALiBi slopes [Hq] fp32:
    awq_decode_attention_ex)."""
    _require_gpu(q, k_cache, v_cache, len_dev, alibi_slopes)
    B, Hq, D = q.shape
    Hkv, Tmax = k_cache.shape[2], k_cache.shape[1]
    q = q.contiguous()
    out = torch.empty_like(q)
    L = _lib.lib()
    key = (q.device.index if q.device.index is not None else torch.cuda.current_device(), _stream())
    need = L.awq_decode_attention_workspace_bytes(B, Hq)
    ws = _attn_workspaces.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _retired_raw.append(ws)
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)
        _attn_workspaces[key] = ws
    if scale is None:
        scale = D ** -0.5
    with torch.cuda.device(q.device):
        if softcap or alibi_slopes is not None:
            if alibi_slopes is not None:
                alibi_slopes = alibi_slopes.to(torch.float32).contiguous()
                if alibi_slopes.numel() != Hq:
                    raise _lib.AwqHipError("decode_attention: alibi_slopes must hold one slope per query head")
            _lib.check(L.awq_decode_attention_ex(_ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(out), _ptr(len_dev), int(seq_len),
                                                 int(max_len if max_len is not None else seq_len), B, Hq, Hkv, D, Tmax,
                                                 float(scale), float(softcap), _ptr(alibi_slopes), _ptr(ws), ws.numel(),
                                                 _stream()), "awq_decode_attention_ex")
        else:
            _lib.check(L.awq_decode_attention(_ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(out), _ptr(len_dev), int(seq_len),
                                              int(max_len if max_len is not None else seq_len), B, Hq, Hkv, D, Tmax,
                                              float(scale), _ptr(ws), ws.numel(), _stream()), "awq_decode_attention")
    return out


def decode_attention_rope(qkv, k_cache, v_cache, cos, sin, start_pos, n_heads, n_kv_heads, pos_dev=None, max_len=None,
                          scale=None):
    """One launch for a decode step: qkv [B, 1, (Hq + 2 Hkv) * 128] (or [B, ...]) -> attention output
    [B, Hq, 128]; rotates q / k, appends k / v at row start_pos and attends over rows [0, start_pos]
    (awq_decode_attention_rope; bit-identical to rope_kv_append + decode_attention)."""
    _require_gpu(qkv, k_cache, v_cache, cos, sin, pos_dev)
    B = qkv.shape[0]
    D = k_cache.shape[3]
    qkv = qkv.contiguous()
    out = torch.empty((B, n_heads, D), dtype=torch.float16, device=qkv.device)
    L = _lib.lib()
    key = (qkv.device.index if qkv.device.index is not None else torch.cuda.current_device(), _stream())
    need = L.awq_decode_attention_workspace_bytes(B, n_heads)
    ws = _attn_workspaces.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _retired_raw.append(ws)
        ws = torch.empty(need, dtype=torch.uint8, device=qkv.device)
        _attn_workspaces[key] = ws
    if scale is None:
        scale = D ** -0.5
    with torch.cuda.device(qkv.device):
        _lib.check(L.awq_decode_attention_rope(_ptr(qkv), _ptr(k_cache), _ptr(v_cache), _ptr(cos), _ptr(sin), _ptr(out),
                                               _ptr(pos_dev), int(start_pos),
                                               int(max_len if max_len is not None else start_pos + 1), B, n_heads,
                                               n_kv_heads, D, k_cache.shape[1], float(scale), _ptr(ws), ws.numel(),
                                               _stream()), "awq_decode_attention_rope")
    return out


def gemm_forward_normed(x2d, norm_weight, eps, qweight, scales, qzeros, bias=None, residual=None, flags=0):
    """(y, stream) with y [M, N] = rmsnorm(x2d (+ residual)) * norm_weight @ W and stream = fp16(x2d + residual)
    (None without a residual): awq_gemm_forward_normed, decode batches (M <= 4) of GEMM-layout weights."""
    _require_gpu(x2d, norm_weight, qweight, scales, qzeros, bias, residual)
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    N = qweight.shape[1] * 8
    G = qzeros.shape[0]
    if x2d.dtype != torch.float16 or qweight.shape[0] != K or G == 0 or K % G or norm_weight.numel() != K:
        raise _lib.AwqHipError("gemm_forward_normed: fp16 x [M, K], norm weight [K], GEMM-layout buffers expected")
    y = torch.empty((M, N), dtype=torch.float16, device=x2d.device)
    res_out = None
    if residual is not None:
        if residual.shape != x2d.shape or residual.dtype != torch.float16 or not residual.is_contiguous():
            raise _lib.AwqHipError("gemm_forward_normed: residual must be a contiguous fp16 tensor of x's shape")
        res_out = torch.empty_like(residual)
    L = _lib.lib()
    with torch.cuda.device(x2d.device):
        need = L.awq_gemm_workspace_bytes(M, K, N, K // G)
        ws = workspace(x2d.device, need)
        rc = L.awq_gemm_forward_normed(_ptr(x2d), _ptr(residual), _ptr(res_out), _ptr(norm_weight.contiguous()), float(eps),
                                       _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(y), M, K, N, K // G,
                                       _ptr(ws), ws.numel(), flags, _stream())
    _lib.check(rc, "awq_gemm_forward_normed")
    return y, res_out


def gemm_forward_ex(x2d, qweight, scales, qzeros, bias=None, flags=0, norm_weight=None, norm_eps=0.0, ssq_in=None,
                    add_residual=None, want_ssq=False):
    """Decode-sized projection with the decoder block's prologue / epilogue (awq_gemm_forward_ex):
    norm_weight (+ ssq_in [M, tiles] from the producing call, else a statistic pass) normalises x while
    staging; add_residual stores y = fp16(fp16(x W + bias) + add_residual); want_ssq returns the per-tile
    sums of squares of y.  Returns (y, ssq | None)."""
    _require_gpu(x2d, qweight, scales, qzeros, bias, norm_weight, ssq_in, add_residual)
    x2d, qweight, scales, qzeros = x2d.contiguous(), qweight.contiguous(), scales.contiguous(), qzeros.contiguous()
    M, K = x2d.shape
    if flags & X_GATED_SILU:
        K //= 2
    N = qweight.shape[1] * 8
    G = qzeros.shape[0]
    if x2d.dtype != torch.float16 or qweight.shape[0] != K or G == 0 or K % G:
        raise _lib.AwqHipError("gemm_forward_ex: fp16 x [M, K] and GEMM-layout buffers expected")
    L = _lib.lib()
    y = torch.empty((M, N), dtype=torch.float16, device=x2d.device)
    ssq = torch.empty((M, L.awq_gemm_ex_ssq_tiles(N)), dtype=torch.float32, device=x2d.device) if want_ssq else None
    if add_residual is not None and (add_residual.shape != y.shape or add_residual.dtype != torch.float16
                                     or not add_residual.is_contiguous()):
        raise _lib.AwqHipError("gemm_forward_ex: add_residual must be a contiguous fp16 [M, N] tensor")
    if ssq_in is not None and (ssq_in.dtype != torch.float32 or ssq_in.shape[0] != M or not ssq_in.is_contiguous()):
        raise _lib.AwqHipError("gemm_forward_ex: ssq_in must be a contiguous fp32 [M, tiles] tensor")
    with torch.cuda.device(x2d.device):
        need = L.awq_gemm_workspace_bytes(M, K, N, K // G)
        ws = workspace(x2d.device, need)
        e = _lib.AwqGemmEx()
        e.struct_bytes = ctypes.sizeof(_lib.AwqGemmEx)
        e.flags = flags
        e.x, e.qweight, e.scales, e.qzeros, e.bias, e.y = _ptr(x2d), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(bias), _ptr(y)
        e.M, e.K, e.N, e.group_size = M, K, N, K // G
        e.workspace, e.workspace_bytes, e.stream = _ptr(ws), ws.numel(), _stream()
        e.norm_weight = _ptr(norm_weight.contiguous()) if norm_weight is not None else None
        e.norm_eps = float(norm_eps)
        e.residual_in = e.residual_out = None
        e.ssq_in = _ptr(ssq_in)
        e.ssq_in_tiles = ssq_in.shape[1] if ssq_in is not None else 0
        e.add_residual = _ptr(add_residual)
        e.ssq_out = _ptr(ssq)
        rc = L.awq_gemm_forward_ex(ctypes.byref(e))
    _lib.check(rc, "awq_gemm_forward_ex")
    return y, ssq
