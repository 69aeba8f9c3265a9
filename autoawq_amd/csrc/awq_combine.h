// awq_combine.h -- in-launch, deterministic split-K combine through tagged granules.
//
// Every (slab s, row m, column n) partial sum travels as ONE naturally aligned 8-byte granule
// {tag = 1, fp32 value} written by ONE write-through (agent-scope, sc1) store: the data is the
// flag (cdna_hip_programming.md Guideline 16, form R2), so producers need no fence, no drain and
// no ticket -- they store and exit.  The block with the LAST slab index of a column tile is the
// tile's reducer: after its own K-slice it polls the other slabs' granules with agent-scope
// (sc1, L1-bypassing) 8-byte loads until every tag reads 1, adds the values in fixed slab order
// (bitwise reproducible) and writes the fp16 result.  It then zeroes the granules it consumed,
// so the workspace is all-zero again when the kernel ends (the caller zeroes it once, at
// allocation).
//
// Progress: producers never wait on anything.  Reducers are the last blocks in dispatch order
// and there are at most `tiles` (< resident capacity) of them, so spinning reducers cannot starve
// producers; the spin is bounded and raises *err instead of hanging.
#pragma once
#include "awq_device.h"

typedef unsigned long long awq_granule_t;

AWQ_DEV void awq_publish(awq_granule_t* g, float v) {
    const awq_granule_t bits = (1ull << 32) | (awq_granule_t)__builtin_bit_cast(uint32_t, v);
    __hip_atomic_store(g, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum slabs 0..nslab-1 of one output element, in slab order; `stride` = granules between slabs.
// Returns false if the bounded spin expired (caller reports through *err).
template <int MAXS>
AWQ_DEV bool awq_collect(awq_granule_t* base, int64_t stride, int nslab, float& sum) {
    static_assert(MAXS <= 64, "slab count is capped at 64");
    for (unsigned spins = 0;; ++spins) {
        unsigned long long pending = 0;
        float s = 0.f;
        for (int s0 = 0; s0 < nslab; s0 += 8) {  // 8 independent 8-byte loads in flight
            awq_granule_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (s0 + u < nslab)
                           ? __hip_atomic_load(base + (int64_t)(s0 + u) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : (1ull << 32);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pending += (v[u] >> 32) != 1ull;
                s += __builtin_bit_cast(float, (uint32_t)v[u]);
            }
        }
        if (pending == 0) {
            sum = s;
            return true;
        }
        if (spins > (1u << 20)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

// Re-arm: write-through (sc1) zero stores, which also drop the line from this XCD's L2 so that no
// stale copy of a granule can be hit by a later launch's polling loads.
AWQ_DEV void awq_clear(awq_granule_t* base, int64_t stride, int nslab) {
    for (int s = 0; s < nslab; ++s)
        __hip_atomic_store(base + (int64_t)s * stride, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
