// bpermute_conflicts.hip — what SQ_LDS_BANK_CONFLICT counts for ds_bpermute_b32 (the instruction behind __shfl): four kernels
// that issue nothing on the LDS pipe but 64-lane bpermutes, with the source lane
//   0: identity   1: lane ^ 1   2: rotation by d inside groups of ten lanes (fivepoint_roots_kernel's exchange)   3: all lanes read lane 0
// Run under `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS`: if the counter reads the same number of
// cycles per instruction for the identity as for the rotation, it describes the instruction (two passes of 32 lanes through
// the crossbar), not a conflict of the access pattern.
// build: hipcc --offload-arch=gfx950 -O3 bpermute_conflicts.hip -o bpermute_conflicts
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    const int lane = threadIdx.x;
    float x = (float)lane * 0.5f + 1.0f, s = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 1; d < 10; ++d) {
            int src;
            if (MODE == 0) src = lane;
            else if (MODE == 1) src = lane ^ 1;
            else if (MODE == 2) { const int g = lane / 10, kk = lane - 10 * g; int q = kk + d; q = q >= 10 ? q - 10 : q; src = min(10 * g + q, 63); }
            else src = 0;
            const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(src * 4, __float_as_int(x)));
            s += y;
            x = x * 1.0001f + 1e-3f;
        }
    }
    out[blockIdx.x * 64 + lane] = s;
}

template <int MODE>
void run(float* out, const char* name) {
    const int blocks = 2048, iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("mode %d (%s): %.3f ms for %ld bpermutes (waves x instructions)\n", MODE, name, ms, (long)blocks * iters * 9);
}

int main() {
    float* out;
    hipMalloc(&out, 2048 * 64 * sizeof(float));
    run<0>(out, "identity");
    run<1>(out, "lane ^ 1");
    run<2>(out, "rotation inside groups of ten");
    run<3>(out, "broadcast of lane 0");
    hipDeviceSynchronize();
    return 0;
}
