// awq_internal.h -- host-side launcher prototypes shared between the .hip translation units
// and capi.hip.  Not part of the public ABI (that is include/awq_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/awq_hip.h"

int awq_launch_dequant(const int32_t* qweight, const uint16_t* scales, const int32_t* qzeros, uint16_t* out,
                       int64_t K, int64_t N, int64_t g, hipStream_t stream);
int awq_launch_unpack(const int32_t* q, uint8_t* out, int64_t rows, int64_t words, hipStream_t stream);

struct AwqGemmArgs {
    const uint16_t* x;       // [M, K] fp16
    const int32_t* qweight;  // [K, N/8]
    const uint16_t* scales;  // [K/g, N]
    const int32_t* qzeros;   // [K/g, N/8]
    const uint16_t* bias;    // [N] or null
    uint16_t* y;             // [M, N]
    int M, K, N, g;
    int* counters;   // control words (last one = error flag); zero on entry, zero on exit
    float* partial;  // split-K slabs: tagged 8-byte granules [S-1, M, N] (or fp32 [S, M, N] in two-pass mode)
    size_t partial_floats;
    hipStream_t stream;
};

int awq_launch_gemm_naive(const AwqGemmArgs& a);
// nlog: log2 of column-lanes per wave (2..4), splitk >= 1, two_pass: separate reduce kernel
int awq_launch_gemv_valu(const AwqGemmArgs& a, int nlog, int splitk, bool two_pass, bool nt, int ablate = 0);
int awq_gemv_valu_default_split(int K, int N, int nlog);
int awq_launch_splitk_reduce(const AwqGemmArgs& a, int splitk);
// MFMA skinny GEMM, M <= 16; wpl = packed words per lane (2 or 4)
int awq_launch_gemm_skinny(const AwqGemmArgs& a, int wpl, int splitk, bool nt, void* trace = nullptr);
int awq_skinny_default_split(int K, int N, int wpl);
