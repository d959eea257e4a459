// Micro-benchmark (gfx950): VALU issue rate per CU at the occupancy of the pair kernels (one 1024-lane workgroup per CU =
// 4 wavefronts per SIMD; also 8 per SIMD), for dependent chains of plain fp32 FMAs with ILP 1 / 2 / 4, and with the compiler's
// packed forms.  Prints wave-instructions per second for the whole chip and cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void __launch_bounds__(1024) k_fma(float* out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 512 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
        for (int ilp = 1; ilp <= 4; ilp *= 2) {
            const int blocks = 256 * wgs_per_cu;
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (ilp == 1) hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
                if (ilp == 2) hipLaunchKernelGGL(k_fma<2>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
                if (ilp == 4) hipLaunchKernelGGL(k_fma<4>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double n_inst = (double)blocks * 16 /*waves*/ * iters * 16.0 * ilp;
            printf("%d waves/SIMD  ILP %d: %.0f G wave-instr/s   (%.2f cycles per instruction per SIMD at 2.4 GHz)\n", 4 * wgs_per_cu, ilp,
                   n_inst / (ms * 1e-3) / 1e9, 256.0 * 4 * 2.4e9 / (n_inst / (ms * 1e-3)));
        }
    return 0;
}
