// mfma_srcc.hip — what the C operand of the FIRST MFMA of a chain costs: chains of 8 v_mfma_f32_32x32x16_f16 (the L2 filter's
// shape: K = 128), two chains per wave, 3 waves per SIMD; the first MFMA of every chain takes C from
//   0: its own accumulator (one endless chain, the reference)   1: the inline constant 0   2: 16 other registers (presets)
// build: hipcc --offload-arch=gfx950 -O3 mfma_srcc.hip -o mfma_srcc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline f16x8 rnd(unsigned s) {
    f16x8 v;
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; v[i] = (_Float16)(((int)(s >> 9) % 2048 - 1024) * (1.0f / 1024.0f)); }
    return v;
}
template <int MODE, int CHAIN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a0 = rnd(threadIdx.x * 7 + 1), a1 = rnd(threadIdx.x * 13 + 5), b0 = rnd(threadIdx.x * 29 + 3), b1 = rnd(threadIdx.x * 31 + 11);
    f32x16 p;
    for (int i = 0; i < 16; ++i) p[i] = (float)(threadIdx.x + i) * 1e-3f;
    asm volatile("" : "+v"(p));
    f32x16 c0 = {}, c1 = {};
    float s = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a0), "+v"(p));            // opaque: nothing of the chain can be hoisted out of the loop
        if (MODE == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0); }
        if (MODE == 1) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, f32x16{}, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, f32x16{}, 0, 0, 0); }
        if (MODE == 2) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, p, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, p, 0, 0, 0); }
#pragma unroll
        for (int u = 1; u < CHAIN; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(u & 1 ? a1 : a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(u & 1 ? a1 : a0, b1, c1, 0, 0, 0);
        }
        asm volatile("" : "+v"(c0), "+v"(c1));
    }
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int CHAIN>
void run(float* out) {
    const int wps = 3, blocks = 256 * wps, iters = 1600000 / CHAIN;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, CHAIN>), dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, CHAIN>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 32768.0 * 2 * CHAIN * (double)iters * 1024 * wps;
    printf("chain of %d, first C = %s: %7.1f TFLOP/s (%.3f of 2516.6)\n", CHAIN, MODE == 0 ? "accumulator" : MODE == 1 ? "inline 0   " : "registers  ",
           flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2516.6e12);
}
int main() {
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
    for (int rep = 0; rep < 1; ++rep) {
        run<0, 8>(out); run<1, 8>(out); run<2, 8>(out);
        run<0, 4>(out); run<1, 4>(out); run<2, 4>(out);
        run<1, 1>(out); run<2, 1>(out);
    }
    return 0;
}
