// valu_rates.hip — measured issue rate of the VALU instructions the Hamming and scoring kernels are
// made of, on the whole chip (calibration for the "VALU roofline" used in DESIGN.md).
// build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed * 2654435761u + threadIdx.x, c = b ^ 0x5555;
    float f0 = a0 * 1e-9f, f1 = a1 * 1e-9f, f2 = a2 * 1e-9f, f3 = a3 * 1e-9f, f4 = a4 * 1e-9f, f5 = a5 * 1e-9f, f6 = a6 * 1e-9f, f7 = a7 * 1e-9f;
    float g = b * 1e-10f, hh = c * 1e-10f;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = {f1, f0}, p5 = {f3, f2}, p6 = {f5, f4}, p7 = {f7, f6}, pg = {g, hh};
    for (int i = 0; i < iters; ++i) {
#define I8(ins) asm volatile(ins(%0) ins(%1) ins(%2) ins(%3) ins(%4) ins(%5) ins(%6) ins(%7) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define F8(ins) asm volatile(ins(%0) ins(%1) ins(%2) ins(%3) ins(%4) ins(%5) ins(%6) ins(%7) \
        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(g), "v"(hh));
#define P8(ins) asm volatile(ins(%0) ins(%1) ins(%2) ins(%3) ins(%4) ins(%5) ins(%6) ins(%7) \
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pg));
#define XOR(r) "v_xor_b32 " #r ", " #r ", %8\n"
#define BCNT(r) "v_bcnt_u32_b32 " #r ", %8, " #r "\n"
#define MED3(r) "v_med3_u32 " #r ", " #r ", %8, %9\n"
#define MINU(r) "v_min_u32 " #r ", " #r ", %8\n"
#define LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 1, %8\n"
#define ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n"
#define FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define FMAC(r) "v_fmac_f32 " #r ", %8, %9\n"
#define MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %8, %8\n"
#define CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8\n"
        if (OP == 0) { REP8(I8(XOR)) }
        if (OP == 1) { REP8(I8(BCNT)) }
        if (OP == 2) { REP8(I8(MED3)) }
        if (OP == 3) { REP8(I8(MINU)) }
        if (OP == 4) { REP8(I8(LSHLADD)) }
        if (OP == 5) { REP8(I8(ADD3)) }
        if (OP == 6) { REP8(F8(FMA)) }
        if (OP == 7) { REP8(F8(FMAC)) }
        if (OP == 8) { REP8(F8(MUL)) }
        if (OP == 9) { REP8(P8(PKFMA)) }
        if (OP == 10) { asm volatile("" ::: "vcc"); REP8(F8(CMP)) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^
                                          __float_as_uint(p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y);
}

template <int OP>
void run(const char* name, uint32_t* out, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, 1u);
    hipEventRecord(a);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters * 64;           // wave-instructions
    const double per_simd_per_s = instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-16s waves/SIMD %d: %8.3f ms  %6.3f G wave-instr/s/SIMD  => %.2f cycles/instr @2.4GHz  (%.1f T lane-ops/s chip)\n",
           name, waves_per_simd, ms, per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s, instr * 64 / (ms * 1e-3) * 1e-12);
}

int main() {
    uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32", out, w); run<1>("v_bcnt_u32_b32", out, w); run<2>("v_med3_u32", out, w); run<3>("v_min_u32", out, w);
        run<4>("v_lshl_add_u32", out, w); run<5>("v_add3_u32", out, w); run<6>("v_fma_f32", out, w); run<7>("v_fmac_f32", out, w);
        run<8>("v_mul_f32", out, w); run<9>("v_pk_fma_f32", out, w); run<10>("v_cmp_lt_f32", out, w);
    }
    return 0;
}
