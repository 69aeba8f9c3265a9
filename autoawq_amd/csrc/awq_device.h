// awq_device.h -- device-side helpers shared by the gfx950 AWQ kernels.
//
// Bit layout being decoded (reference packer
// awq/modules/linear/gemm.py:220-249, unpack awq/utils/packing_utils.py:4-43):
//   qweight[k, c] (int32) nibble i holds logical column 8c + ORDER[i], ORDER = [0,2,4,6,1,3,5,7].
// So nibbles (J, J+4) of a word are logical columns (2J, 2J+1): masking a word shifted right by
// 4J with 0x000F000F leaves exactly that column pair in the low/high 16-bit halves.  OR-ing
// 0x6400 into each half turns it into the fp16 number 1024+nibble; subtracting the fp16 number
// 1024+zero is exact, and one packed fp16 multiply by the scale pair then rounds once -- bit
// identical to the reference's int8 subtract followed by int8*fp16 -> fp16
// (awq/utils/packing_utils.py:98-100).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define AWQ_DEV __device__ __forceinline__

AWQ_DEV half2_t u2h2(uint32_t v) { return __builtin_bit_cast(half2_t, v); }
AWQ_DEV uint32_t h22u(half2_t v) { return __builtin_bit_cast(uint32_t, v); }

// (a & mask) | magic in ONE VALU op.  gfx950 VOP3 takes no literals, so with literal constants
// hipcc emits v_and_b32 + v_or_b32.  Passing the constants through an empty asm makes them opaque
// register values (mask in an SGPR, magic in a VGPR: constant-bus limit 1) and instruction
// selection then picks v_and_or_b32 itself.  The operation is NOT written in inline asm: hipcc
// pads no hazards for an asm statement's outputs (VALU write -> MFMA source needs wait states,
// cdna_hip_programming.md 5.7), and a decoded word feeds an MFMA right away.
AWQ_DEV uint32_t opaque_sgpr(uint32_t c) {
    asm("" : "+s"(c));
    return c;
}
AWQ_DEV uint32_t opaque_vgpr(uint32_t c) {
    asm("" : "+v"(c));
    return c;
}
AWQ_DEV uint32_t and_or(uint32_t a, uint32_t mask, uint32_t magic) {
    return (a & opaque_sgpr(mask)) | opaque_vgpr(magic);
}

// silu(g) * u in fp32, the arithmetic of awq_silu_and_mul (elementwise.hip): ONE definition for the separate launch and for
// every kernel that folds the activation into its staging or epilogue.  Contraction is off and every intermediate is pinned
// in a register (empty asm), so instruction selection cannot fuse across the steps differently from one kernel to the next
// (observed: 1 output in 10^4 -- the fp32 -> fp16 ties -- off by an fp16 ulp between two kernels with the same source expression).
AWQ_DEV float awq_silu_mul_f32(float g, float u) {
#pragma clang fp contract(off)
    float e = expf(-g);
    asm("" : "+v"(e));
    float d = 1.0f + e;
    asm("" : "+v"(d));
    float q = g / d;
    asm("" : "+v"(q));
    float r = q * u;  // rounded to fp32 HERE: left to the caller's fp16 conversion, the product can become one v_fma_mixlo_f16
    asm("" : "+v"(r));  // (a single rounding of the exact product), which differs from fp32-then-fp16 on exact ties
    return r;
}

// half2(1024 + column 2J, 1024 + column 2J+1) of one packed word
template <int J>
AWQ_DEV uint32_t awq_pair_magic(uint32_t q) {
    if constexpr (J == 0)
        return and_or(q, 0x000F000Fu, 0x64006400u);
    else
        return and_or(q >> (4 * J), 0x000F000Fu, 0x64006400u);
}

// fp16 pair (w - z) * s for columns (2J, 2J+1); zmagic = awq_pair_magic<J>(qzeros word)
template <int J>
AWQ_DEV half2_t awq_dq_pair(uint32_t q, half2_t zmagic, half2_t s) {
    half2_t d = u2h2(awq_pair_magic<J>(q)) - zmagic;  // exact: integers in [-15, 15]
    return d * s;                                     // one rounding
}

// 16-byte global load, optionally non-temporal (streamed-once weights)
template <bool NT>
AWQ_DEV u32x4 ld16(const void* p) {
    if constexpr (NT)
        return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else
        return *reinterpret_cast<const u32x4*>(p);
}

// Agent-scope (whole GPU, all 8 XCDs) relaxed accesses: lowered to sc1 loads / write-through
// stores, the cross-XCD-coherent forms (MI355X_MICROARCH.md, inter-workgroup visibility).
AWQ_DEV void st_agent_f32(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
AWQ_DEV float ld_agent_f32(const float* p) {
    uint32_t u = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(float, u);
}
