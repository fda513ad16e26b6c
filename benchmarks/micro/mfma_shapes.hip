// mfma_shapes.hip — sustained f16 MFMA rate of the two shapes on random-ish data (clock behaviour under load):
// v_mfma_f32_32x32x16_f16 (2 chains, 1 chain) vs v_mfma_f32_16x16x32_f16 (4 chains), 1-3 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f16x8 rnd(unsigned s) {
    f16x8 v;
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; v[i] = (_Float16)(((int)(s >> 9) % 2048 - 1024) * (1.0f / 1024.0f)); }
    return v;
}
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a0 = rnd(threadIdx.x * 7 + 1), a1 = rnd(threadIdx.x * 13 + 5), b0 = rnd(threadIdx.x * 29 + 3), b1 = rnd(threadIdx.x * 31 + 11);
    float s = 0;
    if (SHAPE == 0) {
        f32x16 c0 = {}, c1 = {};
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    } else if (SHAPE == 2) {                             // ONE chain: every MFMA waits for the one before it
        f32x16 c0 = {};
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c0, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += c0[i];
    } else {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c3, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE>
void run(float* out, int wps) {
    const int blocks = 256 * wps, iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_iter = SHAPE != 1 ? 2.0 * 32768 : 4.0 * 16384;     // per wave
    printf("%s waves/SIMD=%d: %7.1f TFLOP/s (%.3f of 2516.6)\n", SHAPE == 1 ? "16x16x32 x4 chains" : SHAPE == 2 ? "32x32x16 x1 chain " : "32x32x16 x2 chains", wps,
           flop_per_iter * 1024 * wps * iters / (ms * 1e-3) / 1e12, flop_per_iter * 1024 * wps * iters / (ms * 1e-3) / 2516.6e12);
}
int main() {
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
    for (int wps = 1; wps <= 3; ++wps) { run<0>(out, wps); run<2>(out, wps); run<1>(out, wps); }
    return 0;
}
