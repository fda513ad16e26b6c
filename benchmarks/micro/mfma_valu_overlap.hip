// mfma_valu_overlap.hip — can VALU instructions of the SAME wave issue while its MFMAs execute?
// Loop body: 2 independent v_mfma_f32_32x32x16_f16 chains (one MFMA each per iteration) followed by NV independent
// v_fma_f32 (or v_max3_f32) on other registers.  Time per iteration vs NV, at 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
    f32x16 c0 = {}, c1 = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float g = 1.0001f, h = 1e-7f;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 15]) : "v"(g), "v"(h));
            else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[n & 15]) : "v"(g), "v"(h));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND>
void run(float* out, int wps) {
    const int blocks = 256 * wps, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves, each iteration = 2 MFMAs (64 matrix-pipe clocks) + NV VALU
    const double ns_per_iter_per_wave = ms * 1e6 / iters / wps;
    printf("%s NV=%2d waves/SIMD=%d: %7.1f ns per (2 MFMA + NV VALU) per wave  -> %6.1f TFLOP/s f16 MFMA\n", KIND ? "max3" : "fma ", NV, wps,
           ns_per_iter_per_wave, 2.0 * 32768 * 1024 * wps * iters / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 4);
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 0>(out, wps); run<4, 0>(out, wps); run<8, 0>(out, wps); run<12, 0>(out, wps); run<16, 0>(out, wps);
        run<24, 0>(out, wps); run<32, 0>(out, wps); run<16, 1>(out, wps); run<32, 1>(out, wps);
    }
    return 0;
}
