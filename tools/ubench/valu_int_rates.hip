// Micro-benchmark (gfx950): issue cost of the integer instructions that the pair kernels' index arithmetic compiles to, against v_add_u32 and
// v_fma_f32: v_mul_lo_u32, v_mul_u32_u24, v_mad_u32_u24, v_mad_u64_u32, v_lshl_add_u32, v_cndmask_b32, v_cvt_i32_f32, v_rsq_f32, v_exp_f32.
// One 1024-lane workgroup per CU (4 wavefronts per SIMD, the pair kernels' occupancy), four independent chains per lane, written as
// instructions so that the compiler cannot substitute.  Prints cycles per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_int_rates valu_int_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN4(STR) \
    asm volatile(STR : "+v"(x0) : "v"(a), "v"(b)); asm volatile(STR : "+v"(x1) : "v"(a), "v"(b)); \
    asm volatile(STR : "+v"(x2) : "v"(a), "v"(b)); asm volatile(STR : "+v"(x3) : "v"(a), "v"(b));

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned* out, int iters, unsigned a, unsigned b) {
    unsigned x0 = threadIdx.x, x1 = threadIdx.x + 1, x2 = threadIdx.x + 2, x3 = threadIdx.x + 3;
    unsigned long long y0 = x0, y1 = x1, y2 = x2, y3 = x3; const unsigned long long ya = ((unsigned long long)a << 32) | b, yb = ((unsigned long long)b << 32) | a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) { CHAIN4("v_add_u32 %0, %0, %1") }
            if (OP == 1) { CHAIN4("v_mul_lo_u32 %0, %0, %1") }
            if (OP == 2) { CHAIN4("v_mul_u32_u24 %0, %0, %1") }
            if (OP == 3) { CHAIN4("v_mad_u32_u24 %0, %0, %1, %2") }
            if (OP == 4) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y0) : "v"(a), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y1) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y2) : "v"(a), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y3) : "v"(a), "v"(b) : "vcc");
            }
            if (OP == 5) { CHAIN4("v_lshl_add_u32 %0, %0, 2, %1") }
            if (OP == 6) { CHAIN4("v_fma_f32 %0, %0, %1, %2") }
            if (OP == 7) { CHAIN4("v_cvt_i32_f32 %0, %0") }
            if (OP == 8) { CHAIN4("v_rsq_f32 %0, %0") }
            if (OP == 9) { CHAIN4("v_exp_f32 %0, %0") }
            if (OP == 10) { CHAIN4("v_mov_b32 %0, %1") }
            if (OP == 11) { CHAIN4("v_fract_f32 %0, %0") }
            if (OP == 12) { CHAIN4("v_med3_f32 %0, %0, %1, %2") }
            if (OP == 13) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(ya), "v"(yb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(ya), "v"(yb));
                            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(ya), "v"(yb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(ya), "v"(yb)); }
            if (OP == 14) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y0) : "v"(ya)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y1) : "v"(ya));
                            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y2) : "v"(ya)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y3) : "v"(ya)); }
            if (OP == 15) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y0) : "v"(ya)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y1) : "v"(ya));
                            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y2) : "v"(ya)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y3) : "v"(ya)); }
            if (OP == 16) { CHAIN4("v_cndmask_b32 %0, %0, %1, vcc") }
            if (OP == 17) { CHAIN4("v_max_i32 %0, %0, %1") }
            if (OP == 18) { CHAIN4("v_cvt_f32_i32 %0, %0") }
            if (OP == 19) { CHAIN4("v_ashrrev_i32 %0, 1, %0") }
            if (OP == 20) { CHAIN4("v_and_b32 %0, %0, %1") }
            if (OP == 21) { CHAIN4("v_mul_f32 %0, %0, %1") }
            if (OP == 22) { CHAIN4("v_add3_u32 %0, %0, %1, %2") }
            if (OP == 23) { CHAIN4("v_cvt_rpi_i32_f32 %0, %0") }
            if (OP == 24) { asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(y0) : "v"(ya)); asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(y1) : "v"(ya));
                            asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(y2) : "v"(ya)); asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(y3) : "v"(ya)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)(y0 + y1 + y2 + y3);
}

int main() {
    unsigned* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_u64_u32", "v_lshl_add_u32", "v_fma_f32", "v_cvt_i32_f32",
                           "v_rsq_f32", "v_exp_f32", "v_mov_b32", "v_fract_f32", "v_med3_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cndmask_b32",
                           "v_max_i32", "v_cvt_f32_i32", "v_ashrrev_i32", "v_and_b32", "v_mul_f32", "v_add3_u32", "v_cvt_rpi_i32_f32", "v_pk_mov_b32"};
    double base = 0.;
    for (int op = 0; op < 25; ++op) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
#define L(N) case N: hipLaunchKernelGGL(k<N>, dim3(256), dim3(1024), 0, 0, out, iters, 3u, 5u); break;
            switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double n_inst = 256.0 * 16 * iters * 8.0 * 4;           // wave-instructions on the chip
        const double per = ms * 1e-3 / (n_inst / (256.0 * 4));        // seconds per wave-instruction per SIMD
        if (op == 0) base = per;
        printf("%-16s %.2f x v_add_u32   (%.2f cycles per wave-instruction per SIMD at 2.1 GHz)\n", names[op], per / base, per * 2.1e9);
    }
    return 0;
}
