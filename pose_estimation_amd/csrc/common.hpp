// common.hpp — shared host/device helpers for libposehot (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/posehot.h"

#define PH_WAVE 64

// Every float decision in this library is an explicit sequence of binary32 fma/mul ops
// (DESIGN.md "Arithmetic contracts"); the library is built with -ffp-contract=off so the
// compiler adds no contraction of its own.  These two spell the intent at the call sites.
__device__ __forceinline__ float ph_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

static inline int ph_launch_status(void) {
    return hipGetLastError() == hipSuccess ? PH_OK : PH_ERR_LAUNCH;
}

static inline bool ph_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

static inline int ph_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t ph_round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Zero-fill of n 32-bit words as a KERNEL, never hipMemsetAsync: under stream capture (a caller's hipGraph, plan.py)
// the memset nodes of this ROCm did not take effect on replay — counters kept the previous replay's contents
// (benchmarks/dbg/graph_stage.py) — while kernel nodes replay as they were launched.
static __global__ __launch_bounds__(256) void ph_zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline void ph_zero_words(void* p, size_t n_words, hipStream_t st) {
    if (n_words == 0) return;
    size_t blocks = (n_words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ph_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, n_words);
}
