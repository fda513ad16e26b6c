// valu_banks.hip — does the VGPR bank of the source operands change the issue rate of v_fma_f32 / v_fmac_f32 on gfx950?
// Eight independent instructions per group on hard-coded registers: sources spread over the four banks (reg % 4) vs all
// sources in one bank.  build: hipcc --offload-arch=gfx950 -O3 valu_banks.hip -o valu_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    asm volatile("v_mov_b32 v40, 1.0\n v_mov_b32 v41, 0.5\n v_mov_b32 v42, 0.25\n v_mov_b32 v43, 0.125\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.25\n v_mov_b32 v47, 0.125\n"
                 "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 0.5\n v_mov_b32 v50, 0.25\n v_mov_b32 v51, 0.125\n v_mov_b32 v52, 1.0\n v_mov_b32 v53, 0.5\n v_mov_b32 v54, 0.25\n v_mov_b32 v55, 0.125\n"
                 "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n"
                 "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        if (OP == 0)        // v_fma, three sources in three different banks, dst in the fourth
            asm volatile("v_fma_f32 v56, v41, v42, v43\n v_fma_f32 v57, v42, v43, v44\n v_fma_f32 v58, v43, v44, v45\n v_fma_f32 v59, v44, v45, v46\n"
                         "v_fma_f32 v60, v45, v46, v47\n v_fma_f32 v61, v46, v47, v48\n v_fma_f32 v62, v47, v48, v49\n v_fma_f32 v63, v48, v49, v50\n" ::: CLOB);
        if (OP == 1)        // v_fma, three sources in ONE bank
            asm volatile("v_fma_f32 v56, v40, v44, v48\n v_fma_f32 v57, v41, v45, v49\n v_fma_f32 v58, v42, v46, v50\n v_fma_f32 v59, v43, v47, v51\n"
                         "v_fma_f32 v60, v44, v48, v52\n v_fma_f32 v61, v45, v49, v53\n v_fma_f32 v62, v46, v50, v54\n v_fma_f32 v63, v47, v51, v55\n" ::: CLOB);
        if (OP == 2)        // v_fma, two sources in one bank, third elsewhere
            asm volatile("v_fma_f32 v56, v40, v44, v49\n v_fma_f32 v57, v41, v45, v50\n v_fma_f32 v58, v42, v46, v51\n v_fma_f32 v59, v43, v47, v48\n"
                         "v_fma_f32 v60, v44, v48, v53\n v_fma_f32 v61, v45, v49, v54\n v_fma_f32 v62, v46, v50, v55\n v_fma_f32 v63, v47, v51, v52\n" ::: CLOB);
        if (OP == 3)        // v_fmac (dst is the third source), sources and dst in three different banks
            asm volatile("v_fmac_f32 v56, v41, v42\n v_fmac_f32 v57, v42, v43\n v_fmac_f32 v58, v43, v44\n v_fmac_f32 v59, v44, v45\n"
                         "v_fmac_f32 v60, v45, v46\n v_fmac_f32 v61, v46, v47\n v_fmac_f32 v62, v47, v48\n v_fmac_f32 v63, v48, v49\n" ::: CLOB);
        if (OP == 4)        // v_fmac, both sources and dst in ONE bank
            asm volatile("v_fmac_f32 v56, v40, v44\n v_fmac_f32 v57, v41, v45\n v_fmac_f32 v58, v42, v46\n v_fmac_f32 v59, v43, v47\n"
                         "v_fmac_f32 v60, v44, v48\n v_fmac_f32 v61, v45, v49\n v_fmac_f32 v62, v46, v50\n v_fmac_f32 v63, v47, v51\n" ::: CLOB);
        if (OP == 5)        // dependent pairs: every instruction reads the result of the one before it
            asm volatile("v_fmac_f32 v56, v41, v42\n v_fmac_f32 v57, v56, v43\n v_fmac_f32 v58, v57, v44\n v_fmac_f32 v59, v58, v45\n"
                         "v_fmac_f32 v60, v59, v46\n v_fmac_f32 v61, v60, v47\n v_fmac_f32 v62, v61, v48\n v_fmac_f32 v63, v62, v49\n" ::: CLOB);
        if (OP == 6)        // v_fmac with a squared source (same register twice)
            asm volatile("v_fmac_f32 v56, v41, v41\n v_fmac_f32 v57, v42, v42\n v_fmac_f32 v58, v43, v43\n v_fmac_f32 v59, v44, v44\n"
                         "v_fmac_f32 v60, v45, v45\n v_fmac_f32 v61, v46, v46\n v_fmac_f32 v62, v47, v47\n v_fmac_f32 v63, v48, v48\n" ::: CLOB);
      }
    }
    float s;
    asm volatile("v_add_f32 %0, v56, v57\n v_add_f32 %0, %0, v58\n v_add_f32 %0, %0, v59\n v_add_f32 %0, %0, v60\n v_add_f32 %0, %0, v61\n v_add_f32 %0, %0, v62\n v_add_f32 %0, %0, v63\n" : "=v"(s) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, float* out, int wps) {
    const int blocks = 256 * wps, iters = 10000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters * 64;
    printf("%-44s waves/SIMD %d: %.2f cycles/instr @2.4GHz\n", name, wps, 2.4e9 / (instr / (ms * 1e-3) / 1024.0));
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 4}) {
        run<0>("v_fma  sources in 3 banks", out, w); run<1>("v_fma  sources in 1 bank", out, w); run<2>("v_fma  two sources share a bank", out, w);
        run<3>("v_fmac sources + dst in 3 banks", out, w); run<4>("v_fmac sources + dst in 1 bank", out, w);
        run<5>("v_fmac dependent chain", out, w); run<6>("v_fmac squared source", out, w);
    }
    return 0;
}
