// Micro-benchmark (gfx950): what a DEPENDENT STEP of a small-batch force pass costs, by where the dependency is kept.
//   1. a chain of dependent kernel launches on one stream (tiny kernels: one element per lane, a two-load gather): wall time per
//      launch, host enqueue time per launch, and the same chain replayed from a hipGraph;
//   2. the same chain as PHASES of one resident workgroup, intermediates in global memory, __syncthreads between phases;
//   3. the same with the intermediates in LDS.
// Prints microseconds per step.  Used to size the fused per-element passes.
#include <hip/hip_runtime.h>
#include <chrono>
#include <thread>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// one "node": out[i] = f(in[idx[i]], in[idx[i] ^ 1])
__global__ void k_step(const float* __restrict__ in, const int* __restrict__ idx, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[i];
    out[i] = in[j] * 1.0001f + in[j ^ 1] * 0.5f;
}

__global__ void __launch_bounds__(1024) k_phases_global(float* a, float* b, const int* __restrict__ idx, int n, int n_phase) {
    float* in = a; float* out = b;
    for (int p = 0; p < n_phase; ++p) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int j = idx[i];
            out[i] = in[j] * 1.0001f + in[j ^ 1] * 0.5f;
        }
        __syncthreads();
        float* t = in; in = out; out = t;
    }
}

__global__ void __launch_bounds__(1024) k_phases_lds(float* a, const int* __restrict__ idx, int n, int n_phase) {
    extern __shared__ float sm[];
    float* in = sm; float* out = sm + n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) in[i] = a[i];
    __syncthreads();
    for (int p = 0; p < n_phase; ++p) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int j = idx[i];
            out[i] = in[j] * 1.0001f + in[j ^ 1] * 0.5f;
        }
        __syncthreads();
        float* t = in; in = out; out = t;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) a[i] = in[i];
}

// shader clock under a given load: one wavefront per workgroup runs a chain of dependent FMAs (4 cycles each) and times it with the
// constant 100 MHz wall clock
__global__ void k_clock(float* out, long long* ticks, int iters) {
    float x = threadIdx.x * 1e-3f;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    }
    const long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const int n = 1024, n_step = 200;
    std::vector<int> h_idx(n);
    for (int i = 0; i < n; ++i) h_idx[i] = (i * 37 + 11) % n;
    float *a, *b; int* idx;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&idx, n * 4));
    CHECK(hipMemset(a, 0, n * 4)); CHECK(hipMemset(b, 0, n * 4));
    CHECK(hipMemcpy(idx, h_idx.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int wg = 256; wg <= 1024; wg *= 4) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            CHECK(hipEventRecord(e0, st));
            for (int s = 0; s < n_step; ++s)
                hipLaunchKernelGGL(k_step, dim3(n / wg), dim3(wg), 0, st, (s & 1) ? b : a, idx, (s & 1) ? a : b, n);
            CHECK(hipEventRecord(e1, st));
            auto t1 = std::chrono::steady_clock::now();
            CHECK(hipEventSynchronize(e1));
            auto t2 = std::chrono::steady_clock::now();
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2)
                printf("launch chain  wg %4d: %.2f us per launch (events), host enqueue %.2f us per launch, host wall %.2f us per launch\n", wg,
                       ms * 1e3 / n_step, std::chrono::duration<double, std::micro>(t1 - t0).count() / n_step,
                       std::chrono::duration<double, std::micro>(t2 - t0).count() / n_step);
        }
    }
    {   // the same chain from a graph
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int s = 0; s < n_step; ++s)
            hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, st, (s & 1) ? b : a, idx, (s & 1) ? a : b, n);
        CHECK(hipStreamEndCapture(st, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(ge, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            auto t2 = std::chrono::steady_clock::now();
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 3)
                printf("graph replay  wg  256: %.2f us per node (events), host wall %.2f us per node\n", ms * 1e3 / n_step,
                       std::chrono::duration<double, std::micro>(t2 - t0).count() / n_step);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_phases_global, dim3(1), dim3(1024), 0, st, a, b, idx, n, n_step);
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("resident phases, global intermediates: %.3f us per phase\n", ms * 1e3 / n_step);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_phases_lds, dim3(1), dim3(1024), 2 * n * 4, st, a, idx, n, n_step);
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("resident phases, LDS intermediates:    %.3f us per phase\n", ms * 1e3 / n_step);
    }
    {
        float* o; long long* tk; CHECK(hipMalloc(&o, 1024 * 64 * 4)); CHECK(hipMalloc(&tk, 1024 * 8));
        const int iters = 20000;     // 1.28 M dependent FMAs = 5.12 M cycles
        for (int wgs : {1, 8, 64, 256, 1024}) {
            for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k_clock, dim3(wgs), dim3(64), 0, st, o, tk, iters); CHECK(hipStreamSynchronize(st)); }
            long long t; CHECK(hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost));
            printf("shader clock with %4d one-wave workgroups resident: %.0f MHz (dependent-FMA chain, 4 cycles each)\n", wgs, 4.0 * 64 * iters / (t * 10e-9) / 1e6);
        }
        // right after an idle gap
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, st, o, tk, 2000); CHECK(hipStreamSynchronize(st));
        long long t; CHECK(hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost));
        printf("shader clock of a 0.25 ms kernel after 200 ms idle: %.0f MHz\n", 4.0 * 64 * 2000 / (t * 10e-9) / 1e6);
    }
    printf("done\n");
    return 0;
}
