// Micro-benchmark (gfx950): which lanes of a wave64 share an LDS service group, per instruction.  Two lanes (0 and j) are active and read /
// update the SAME bank column at DIFFERENT addresses: one extra LDS cycle per instruction if they are served together, none if not.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_groups lds_groups.hip ; run on the GPU box.  Prints, per instruction, the lanes j that
// conflict with lane 0 (cycles per instruction above the no-conflict base).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(int j, int same, int iters, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // lane 0: byte 0; lane j: same column one bank row (256 B) further -- or the same address (control: broadcast, no conflict)
    const unsigned addr = (lane == 0 || same) ? 0u : 256u * 4u;   // 1024 B apart: same bank for every banking (32 or 64 banks of 4 B)
    float acc = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    if (lane == 0 || lane == j) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) { f4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr + u * 2048)); acc += v.x; }
                if (MODE == 1) { f2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr + u * 2048)); acc += v.x; }
                if (MODE == 2) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr + u * 2048)); acc += v; }
                if (MODE == 3) { unsigned long long one = 1ull; asm volatile("ds_add_u64 %0, %1" :: "v"(addr + u * 2048), "v"(one)); }
                if (MODE == 4) { f4 v = {1.f, 1.f, 1.f, 1.f}; asm volatile("ds_write_b128 %0, %1" :: "v"(addr + u * 2048), "v"(v)); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main() {
    const int iters = 2000, blocks = 64;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
    const char* names[] = {"ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_add_u64", "ds_write_b128"};
    for (int mode = 0; mode < 5; ++mode) {
        std::vector<double> t(65);
        for (int j = 0; j <= 64; ++j) {           // j = 64: control (lane 0 and lane 1 on the same address)
            std::vector<long long> c(blocks);
            const int jj = j == 64 ? 1 : (j == 0 ? 1 : j), same = j == 64;
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 65536, 0, jj, same, iters, out, cyc); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 65536, 0, jj, same, iters, out, cyc); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 65536, 0, jj, same, iters, out, cyc); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 65536, 0, jj, same, iters, out, cyc); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 65536, 0, jj, same, iters, out, cyc); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : c) avg += v; avg /= blocks;
            t[j] = avg / (16.0 * iters * 8);      // cycles per wave-instruction per CU (16 waves issue concurrently)
        }
        const double base = t[64];
        printf("%-14s base (same address) %.2f cycles/instr; lanes whose different-address access costs more than base + 25%%:\n   ", names[mode], base);
        for (int j = 1; j < 64; ++j) if (t[j] > base * 1.25) printf(" %d", j);
        printf("\n    cycles by lane:");
        for (int j = 1; j < 64; ++j) printf(" %.1f", t[j]);
        printf("\n");
    }
    return 0;
}
