"""Converters between the three packed layouts of the same quantised weights.

The reference only ever packs from an fp16 nn.Linear (`from_linear`: awq/modules/linear/
gemm.py:171-251, gemv.py:77-154, gemv_fast.py:127-183).  Going from one PACKED layout to another
needs no floating-point arithmetic on the weights: unpack the nibbles (Appendix A.1 / A.3 / A.4 of
SURVEY.md), transpose, repack.  Scales move as they are (transposed / zero-padded); the fp16 zero
term of the GEMVFast layout is `-(scales * zeros)` computed exactly like `gemv_fast.py:175-181`.
Everything runs on whatever device the buffers live on (plain torch integer ops), so a checkpoint
loaded in the GEMM format can be repacked on the GPU at load time.

`tests/test_checkpoint.py` pins every direction bit for bit against reference-written checkpoints
of the same weights in all three formats.
"""
import torch

from ..modules.linear import WQLinear_GEMM, WQLinear_GEMV, WQLinear_GEMVFast
from .packing import AWQ_ORDER, GEMV_ORDER, calculate_zeros_width, pack_intweight_fast, pack_rows_int4, pack_zeros_nk

_CLASSES = {"gemm": WQLinear_GEMM, "gemv": WQLinear_GEMV, "gemv_fast": WQLinear_GEMVFast}


def _unpack_rows(q, order):
    """int32 [R, C] -> int32 [R, 8C] with nibble i of word c at column 8c + order[i]."""
    R, C = q.shape
    out = torch.empty((R, C, 8), dtype=torch.int32, device=q.device)
    for i, o in enumerate(order):
        out[:, :, o] = (q >> (4 * i)) & 0xF
    return out.reshape(R, C * 8)


def _unpack_fast(q16):
    """int16 [N/4, K] (GEMVFast) -> int32 [N, K]: inverse of pack_intweight_fast."""
    R, K = q16.shape
    v = q16.to(torch.int32) & 0xFFFF
    nib = torch.stack([(v >> (4 * j)) & 0xF for j in range(4)], dim=-1)  # r, (b, i, h, t), j
    nib = nib.reshape(R, K // 64, 4, 2, 8, 4)                            # r, b, i, h, t, j
    return nib.permute(0, 2, 1, 3, 5, 4).reshape(R * 4, K)               # r, i, b, h, j, t


def unpack_linear(m):
    """Any WQLinear_* -> (w [N, K] int32 in 0..15, z [N, G] int32, s [N, G] fp16, bias)."""
    K, N, g = m.in_features, m.out_features, m.group_size
    G = K // g
    if isinstance(m, WQLinear_GEMM):
        w = _unpack_rows(m.qweight, AWQ_ORDER).t().contiguous()
        z = _unpack_rows(m.qzeros, AWQ_ORDER).t().contiguous()
        s = m.scales.t().contiguous()
    elif isinstance(m, WQLinear_GEMV):
        w = _unpack_rows(m.qweight, GEMV_ORDER)
        z = _unpack_rows(m.qzeros, GEMV_ORDER)[:, :G].contiguous()
        s = m.scales[:, :G].contiguous()
    elif isinstance(m, WQLinear_GEMVFast):
        w = _unpack_fast(m.qweight)
        s = m.scales[:G].t().contiguous()
        # qzeros = -(s * z) rounded to fp16: z = round(-qzeros / s) recovers the integer (|error| << 0.5)
        qz = m.qzeros[:G].t().float()
        z = torch.round(-qz / s.float()).clamp_(0, 15).to(torch.int32)
        z = torch.where(s.float() == 0, torch.zeros_like(z), z)
    else:
        raise TypeError(f"unpack_linear: unsupported module {type(m).__name__}")
    return w, z, s, m.bias


def pack_linear(version, w, z, s, bias, in_features, out_features, group_size):
    """(w [N, K], z [N, G], s [N, G]) -> a WQLinear_<version> holding exactly these values."""
    cls = _CLASSES[version]
    dev = w.device
    m = cls(4, group_size, in_features, out_features, bias is not None, dev)
    G = s.shape[1]
    if version == "gemm":
        m.qweight = pack_rows_int4(w.t().contiguous(), AWQ_ORDER)
        m.qzeros = pack_rows_int4(z.t().contiguous(), AWQ_ORDER)
        m.scales = s.t().contiguous().half()
    else:
        zw = calculate_zeros_width(in_features, m.group_size)
        padded = torch.zeros((out_features, zw * 8), dtype=torch.float16, device=dev)
        padded[:, :G] = s
        if version == "gemv":
            m.qweight = pack_rows_int4(w, GEMV_ORDER)
            m.qzeros = pack_zeros_nk(z, zw)
            m.scales = padded
        else:
            m.qweight = pack_intweight_fast(w.contiguous())
            m.scales = padded.t().contiguous()
            qz = torch.zeros_like(padded)
            qz[:, :G] = -(padded[:, :G] * z.to(torch.int32).to(torch.float32)).to(torch.float16)
            m.qzeros = qz.t().contiguous()
    if bias is not None:
        m.bias = bias.clone()
    return m


class ExpertStackGemv:
    """GEMV-layout twin of a stacked GEMM-layout expert tensor set (what `fuse_linears(..., operation=torch.stack)` returns,
    awq/utils/fused_utils.py:145-162): qweight [E, N, K/8] i32, qzeros [E, N, ZW] i32, scales [E, N, 8 ZW] fp16."""

    def __init__(self, qweight, qzeros, scales, group_size, pairs):
        self.qweight, self.qzeros, self.scales, self.group_size, self.pairs = qweight, qzeros, scales, group_size, pairs


def gemm_stack_to_gemv(qweight, qzeros, scales, interleave_halves=False):
    """Stacked GEMM-layout experts (qweight [E, K, N/8], qzeros [E, K/g, N/8], scales [E, K/g, N]) -> the same integers, zero
    points and scales in the GEMV layout, stacked (qweight [E, N, K/8], qzeros [E, N, ZW], scales [E, N, 8 ZW]); no
    floating-point arithmetic.  interleave_halves: output row 2 j = column j, row 2 j + 1 = column N/2 + j of every expert
    (Mixtral's w1|w3 concatenation, awq/models/mixtral.py:131-142, as (gate_j, up_j) row pairs: the form the row-streaming
    kernel's silu-pairs epilogue reads).  One expert at a time (the unpacked integers of one 4096 x 28672 matrix are 470 MB)."""
    E, K, NW = qweight.shape
    N, G = NW * 8, qzeros.shape[1]
    g = K // G
    zw = calculate_zeros_width(K, g)
    dev = qweight.device
    oq = torch.empty((E, N, K // 8), dtype=torch.int32, device=dev)
    oz = torch.empty((E, N, zw), dtype=torch.int32, device=dev)
    os_ = torch.zeros((E, N, zw * 8), dtype=torch.float16, device=dev)
    perm = None
    if interleave_halves:
        perm = torch.stack([torch.arange(N // 2, device=dev), torch.arange(N // 2, N, device=dev)], dim=1).reshape(-1)
    for e in range(E):
        w = _unpack_rows(qweight[e], AWQ_ORDER).t()          # [N, K]
        z = _unpack_rows(qzeros[e], AWQ_ORDER).t()           # [N, G]
        sc = scales[e].t()                                   # [N, G]
        if perm is not None:
            w, z, sc = w.index_select(0, perm), z.index_select(0, perm), sc.index_select(0, perm)
        oq[e] = pack_rows_int4(w.contiguous(), GEMV_ORDER)
        oz[e] = pack_zeros_nk(z.contiguous(), zw)
        os_[e, :, :G] = sc
        del w, z, sc
    return ExpertStackGemv(oq, oz, os_, g, interleave_halves)


def convert_linear(m, version):
    """Repack one WQLinear_* into another layout (returns `m` itself if it already has it)."""
    version = version.lower()
    if isinstance(m, _CLASSES[version]):
        return m
    w, z, s, bias = unpack_linear(m)
    return pack_linear(version, w, z, s, bias, m.in_features, m.out_features, m.group_size)


def convert_model(model, version):
    """Repack every WQLinear_* of `model` in place; returns the number of modules converted."""
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if isinstance(child, tuple(_CLASSES.values())) and not isinstance(child, _CLASSES[version.lower()]):
                setattr(parent, name, convert_linear(child, version))
                n += 1
    return n
