// Micro-benchmark (gfx950): throughput of LDS atomics by type, vs plain LDS read/write, per CU with 16 waves.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(1024) k(const int* __restrict__ idx, int n_idx, int iters, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = 9600;
    for (int i = threadIdx.x; i < N * 2; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        int a;
        st = st * 1664525u + 1013904223u;
        if (n_idx == 0) a = (int)((st >> 8) % 9600u);                                        // random
        else if (n_idx == 1) a = (int)(((st >> 8) % 150u) * 64u + (threadIdx.x & 63));      // conflict-free
        else a = (int)((__shfl((int)(st >> 8), threadIdx.x & ~7) & 0x7fffffff) % 9600);     // 8 lanes share an address
        const float v = 1.0f + it;
        if (MODE == 0) __hip_atomic_fetch_add(lds + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) __hip_atomic_fetch_add((int*)lds + a, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add((unsigned long long*)__builtin_assume_aligned(lds, 8) + a, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) { lds[a] += v; }                                   // non-atomic read-modify-write
        if (MODE == 4) acc += lds[a];                                     // read only
        if (MODE == 5) lds[a] = v;                                        // write only
        if (MODE == 6) __hip_atomic_fetch_max((int*)lds + a, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = acc + lds[threadIdx.x];
}

int main() {
    const int n_idx = 1 << 10, iters = 4000, blocks = 256;
    std::vector<int> h(n_idx);
    srand(1);
    int *d; float* out; long long* cyc;
    hipMalloc(&d, n_idx * 4); hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
    const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "lds rmw (non-atomic)", "ds_read_b32", "ds_write_b32", "ds_max_i32"};
    for (int pattern = 0; pattern < 3; ++pattern) {
        // 0: uniformly random over 9600 words; 1: conflict-free (lane-distinct banks); 2: 8 lanes share an address
        for (int i = 0; i < n_idx; ++i) {
            if (pattern == 0) h[i] = rand() % 9600;
            else if (pattern == 1) h[i] = ((rand() % 150) * 64 + (i % 64)) % 9600;
            else h[i] = ((i / 8) * 37) % 9600;
        }
        hipMemcpy(d, h.data(), n_idx * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 7; ++mode) {
            std::vector<long long> c(blocks);
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 9600 * 8, 0, d, pattern, iters, out, cyc); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 9600 * 8, 0, d, pattern, iters, out, cyc); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 9600 * 8, 0, d, pattern, iters, out, cyc); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 9600 * 8, 0, d, pattern, iters, out, cyc); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 9600 * 8, 0, d, pattern, iters, out, cyc); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), 9600 * 8, 0, d, pattern, iters, out, cyc); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(1024), 9600 * 8, 0, d, pattern, iters, out, cyc); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : c) avg += v; avg /= blocks;
            // 16 waves x iters wave-instructions per CU
            printf("pattern %d  %-22s  %.1f cycles per wave-instruction per CU (clock64 units)\n", pattern, names[mode], avg / (16.0 * iters));
        }
    }
    return 0;
}
