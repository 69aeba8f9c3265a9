"""Host-side (torch, any device) packers for the AWQ int4 buffers.

Vectorised replacements for the Python column loops of the reference packers
(awq/modules/linear/gemm.py:196-249, gemv.py:110-153); they produce bit-identical buffers
(pinned by tests/test_modules_cpu.py against reference-packed fixtures).  These run once,
offline, at quantisation time -- they are host logic, not the hot path.
"""
import torch

# nibble i of a GEMM-layout word holds logical column ORDER[i] (awq/modules/linear/gemm.py:222)
AWQ_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)


def pack_rows_int4(values, order=AWQ_ORDER):
    """values [R, C] integer tensor with entries 0..15, C % 8 == 0 -> int32 [R, C/8]; nibble i of
    word c holds values[:, 8c + order[i]]."""
    R, C = values.shape
    v = values.to(torch.int32).reshape(R, C // 8, 8)
    out = torch.zeros((R, C // 8), dtype=torch.int32, device=values.device)
    for i, o in enumerate(order):
        out |= v[:, :, o] << (4 * i)  # int32 wraps exactly like the reference's `|= col << shift`
    return out


def quantize_int_weights_kn(weight_nk, scales_gn, zeros_gn, group_size):
    """Integer weights [K, N] = round((W^T + z*s) / s) with the dtype promotion of
    awq/modules/linear/gemm.py:196-203 (fp16 scales, z*s in the dtype the caller passed)."""
    scale_zeros = zeros_gn * scales_gn
    s_half = scales_gn.clone().half()
    wt = weight_nk.t()
    num = wt + scale_zeros.repeat_interleave(group_size, dim=0)
    return torch.round(num / s_half.repeat_interleave(group_size, dim=0)).to(torch.int)


GEMV_ORDER = (0, 1, 2, 3, 4, 5, 6, 7)  # ordinal nibble order (awq/modules/linear/gemv.py:128)


def calculate_zeros_width(in_features, group_size=128, pack_num=8):
    """Words per qzeros row of the GEMV layouts (awq/modules/linear/gemv.py:12-24): ceil(K/g/8),
    rounded up to a multiple of 1 / 2 / 4 for group sizes >= 128 / 64 / 32."""
    if group_size >= 128:
        mult = 1
    elif group_size == 64:
        mult = 2
    elif group_size == 32:
        mult = 4
    else:
        raise NotImplementedError
    base = (in_features // group_size + pack_num - 1) // pack_num
    return (base + mult - 1) // mult * mult


def quantize_int_weights_nk(weight_nk, scales_ng, zeros_ng, padded_scales, group_size):
    """Integer weights [N, K] with the dtype promotion of awq/modules/linear/gemv.py:110-121
    (scale*zero in the caller's dtype, divided by the fp16 padded scales)."""
    scale_zeros = zeros_ng * scales_ng
    G = scales_ng.shape[1]
    num = weight_nk + scale_zeros.repeat_interleave(group_size, dim=1)
    return torch.round(num / padded_scales[:, :G].repeat_interleave(group_size, dim=1)).to(torch.int)


def pack_zeros_nk(zeros_ng, zeros_width):
    """[N, G] zero points -> int32 [N, ZW], nibble i of word c = zeros[:, 8c+i], zero past G
    (awq/modules/linear/gemv.py:135-152)."""
    N, G = zeros_ng.shape
    padded = torch.zeros((N, zeros_width * 8), dtype=torch.int32, device=zeros_ng.device)
    padded[:, :G] = zeros_ng.to(torch.int32)
    return pack_rows_int4(padded, GEMV_ORDER)


def pack_intweight_fast(intweight_nk):
    """[N, K] integers 0..15 -> int16 [N/4, K] of the GEMVFast layout (closed form of pack_intweight,
    awq/modules/linear/gemv_fast.py:26-65 with interleave=4, kstride=64):
    out[r, 64b + 16i + 8h + t] nibble j = w[4r + i, 64b + 32h + 8j + t]."""
    N, K = intweight_nk.shape
    assert N % 4 == 0 and K % 64 == 0
    w = intweight_nk.to(torch.int32).reshape(N // 4, 4, K // 64, 2, 4, 8)  # r, i, b, h, j, t
    w = w.permute(0, 2, 1, 3, 5, 4)                                       # r, b, i, h, t, j
    packed = w[..., 0] | (w[..., 1] << 4) | (w[..., 2] << 8) | (w[..., 3] << 12)
    return packed.reshape(N // 4, K).to(torch.int16)  # wraps like numpy astype("int16")
