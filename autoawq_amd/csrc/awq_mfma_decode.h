// awq_mfma_decode.h -- int4 -> MFMA operand helpers shared by the decode kernels (gemv_mfma.hip,
// gemv_chain.hip).  See gemv_mfma.hip's header for how a packed GEMM-layout word becomes a
// v_mfma_f32_16x16x32_f16 B fragment with one shift and one v_and_or per column pair.
#pragma once
#include "awq_device.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

AWQ_DEV rsrc_t mk_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

constexpr uint32_t OOB = 0x80000000u;  // lane offset beyond every descriptor: returns 0, no traffic

// fp16 pair of columns (2J, 2J+1) of a packed word with the nibbles left where they are: bits 0-3 /
// 16-19 under exponent 2^10 (0x6400: 1024 + w), bits 4-7 / 20-23 under exponent 2^6 (0x5400: 64 + w);
// nibbles 2, 3, 6, 7 come from ONE shared `q >> 8`.  5 VALU ops per packed word; the bias (1024 or 64)
// goes through the group factorisation: y += s * (acc - (bias_J + z) * sum_x), every product exact in fp32.
template <int J>
AWQ_DEV uint32_t pairb(uint32_t q, uint32_t q8) {
    if constexpr (J == 0) return and_or(q, 0x000F000Fu, 0x64006400u);
    else if constexpr (J == 1) return and_or(q, 0x00F000F0u, 0x54005400u);
    else if constexpr (J == 2) return and_or(q8, 0x000F000Fu, 0x64006400u);
    else return and_or(q8, 0x00F000F0u, 0x54005400u);
}

AWQ_DEV float4_t mfma16(u32x4v a, u32x4v b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0,
                                                  0, 0);
}
