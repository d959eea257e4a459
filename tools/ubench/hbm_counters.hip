// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel below moves a KNOWN, dense byte count (1 GiB, four times the 256 MiB Infinity Cache) in one of the access shapes
// of the belief-propagation solve (kernels_rotamer.hip): the counter value per byte is the factor tools/hbm_traffic.py applies.
//   build: hipcc --offload-arch=gfx950 -O3 -o hbm_counters hbm_counters.hip
//   run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o fetch -- ./hbm_counters   (and once more with --pmc WRITE_SIZE)
// Kernel names carry the shape: the summary script keys on them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define BYTES (1ull << 30)
#define GRID 4096
#define BLOCK 256

// reads ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) read_16B_per_lane_coalesced(const f4* __restrict__ p, float* out) {   // the guide's reference shape
    const size_t n = BYTES / 16, stride = (size_t)GRID * BLOCK;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) { const f4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(BLOCK) read_8B_per_lane_coalesced(const f2* __restrict__ p, float* out) {
    const size_t n = BYTES / 8, stride = (size_t)GRID * BLOCK;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) { const f2 v = p[i]; acc += v.x + v.y; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(BLOCK) read_4B_per_lane_coalesced(const float* __restrict__ p, float* out) {
    const size_t n = BYTES / 4, stride = (size_t)GRID * BLOCK;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) acc += p[i];
    if (acc == 123.456f) out[0] = acc;
}
// the solve's pair-matrix rows (bp_load_matrix): 24-byte records, consecutive lanes = consecutive records, a record as three 8-byte loads
__global__ void __launch_bounds__(BLOCK) read_24B_records_as_3x8B(const f2* __restrict__ p, float* out) {
    const size_t n = BYTES / 24, stride = (size_t)GRID * BLOCK;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const f2 a = p[3 * i], b = p[3 * i + 1], c = p[3 * i + 2];
        acc += a.x + a.y + b.x + b.y + c.x + c.y;
    }
    if (acc == 123.456f) out[0] = acc;
}
// ... of which only the first 12 bytes are read (3-state partner: P[i][slot][0..2]; the other half of the record is fetched with its line)
__global__ void __launch_bounds__(BLOCK) read_12B_of_24B_records(const float* __restrict__ p, float* out) {
    const size_t n = BYTES / 24, stride = (size_t)GRID * BLOCK;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const f2 a = *(const f2*)(p + 6 * i); acc += a.x + a.y + p[6 * i + 2];
    }
    if (acc == 123.456f) out[0] = acc;
}
// writes -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) write_16B_per_lane_coalesced(f4* __restrict__ p) {
    const size_t n = BYTES / 16, stride = (size_t)GRID * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) { f4 v = {1.f, 2.f, 3.f, (float)i}; p[i] = v; }
}
__global__ void __launch_bounds__(BLOCK) write_4B_per_lane_coalesced(float* __restrict__ p) {
    const size_t n = BYTES / 4, stride = (size_t)GRID * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) p[i] = (float)i;
}
// the solve's marginals and matrix reset (bp_marginal_slot, retire_packed): 24-byte records, one 4-byte store per entry, lanes 24 bytes apart
__global__ void __launch_bounds__(BLOCK) write_24B_records_as_6x4B(float* __restrict__ p) {
    const size_t n = BYTES / 24, stride = (size_t)GRID * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int j = 0; j < 6; ++j) p[6 * i + j] = (float)(i + j);
    }
}
// message rows that spill to global memory: 16-byte stores, coalesced (bp_store_row) -- same as write_16B; and 8-byte stores
__global__ void __launch_bounds__(BLOCK) write_8B_per_lane_coalesced(f2* __restrict__ p) {
    const size_t n = BYTES / 8, stride = (size_t)GRID * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) { f2 v = {1.f, (float)i}; p[i] = v; }
}

int main() {
    void* buf; float* out;
    if (hipMalloc(&buf, BYTES + 4096) != hipSuccess || hipMalloc((void**)&out, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, BYTES);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_16B_per_lane_coalesced, dim3(GRID), dim3(BLOCK), 0, 0, (const f4*)buf, out);
        hipLaunchKernelGGL(read_8B_per_lane_coalesced, dim3(GRID), dim3(BLOCK), 0, 0, (const f2*)buf, out);
        hipLaunchKernelGGL(read_4B_per_lane_coalesced, dim3(GRID), dim3(BLOCK), 0, 0, (const float*)buf, out);
        hipLaunchKernelGGL(read_24B_records_as_3x8B, dim3(GRID), dim3(BLOCK), 0, 0, (const f2*)buf, out);
        hipLaunchKernelGGL(read_12B_of_24B_records, dim3(GRID), dim3(BLOCK), 0, 0, (const float*)buf, out);
        hipLaunchKernelGGL(write_16B_per_lane_coalesced, dim3(GRID), dim3(BLOCK), 0, 0, (f4*)buf);
        hipLaunchKernelGGL(write_8B_per_lane_coalesced, dim3(GRID), dim3(BLOCK), 0, 0, (f2*)buf);
        hipLaunchKernelGGL(write_4B_per_lane_coalesced, dim3(GRID), dim3(BLOCK), 0, 0, (float*)buf);
        hipLaunchKernelGGL(write_24B_records_as_6x4B, dim3(GRID), dim3(BLOCK), 0, 0, (float*)buf);
        (void)hipDeviceSynchronize();
    }
    printf("hbm_counters: 9 shapes x 3 launches of %llu bytes each\n", (unsigned long long)BYTES);
    return 0;
}
