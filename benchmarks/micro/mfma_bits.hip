// mfma_bits.hip — does the sustained (power-limited) rate of v_mfma_f32_32x32x16_f16 depend on how many mantissa bits of the
// fp16 operands are in use?  Pseudo-random operands with the low Z mantissa bits cleared, Z = 0, 2, 4, 6, 10; 3 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 mfma_bits.hip -o mfma_bits
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ inline f16x8 rnd(unsigned s, unsigned short mask) {
    f16x8 v;
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; v[i] = (_Float16)(((int)(s >> 9) % 2048 - 1024) * (1.0f / 1024.0f) + 1.0f); }
    u16x8 b = __builtin_bit_cast(u16x8, v);
    for (int i = 0; i < 8; ++i) b[i] &= mask;
    return __builtin_bit_cast(f16x8, b);
}
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned short mask) {
    f16x8 a0 = rnd(threadIdx.x * 7 + 1, mask), a1 = rnd(threadIdx.x * 13 + 5, mask), b0 = rnd(threadIdx.x * 29 + 3, mask), b1 = rnd(threadIdx.x * 31 + 11, mask);
    f32x16 c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
    const int wps = 3, blocks = 256 * wps, iters = 100000;
    const int zs[] = {0, 2, 4, 6, 10, 0, 4};
    for (int z : zs) {
        const unsigned short mask = (unsigned short)(0xFFFFu << z);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, mask);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mask);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = 32768.0 * 4 * (double)iters * 1024 * wps;
        printf("low %2d mantissa bits cleared: %7.1f TFLOP/s (%.3f of 2516.6)\n", z, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2516.6e12);
    }
    return 0;
}
