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
