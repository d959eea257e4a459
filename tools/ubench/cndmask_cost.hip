// Issue cost of v_cndmask_b32 forms on gfx950: tools/ubench/valu_int_rates.hip measured ~20 cycles per wave-instruction for the VOP2 form
// reading vcc in four independent chains.  Variants here: the e64 form on an SGPR pair, dependent vs independent chains, with a v_cmp
// writing the mask in the loop, and the arithmetic replacements (v_bfi_b32 on a lane mask held in a VGPR, v_max/v_min clamps).
//   hipcc --offload-arch=gfx950 -O3 -o cndmask_cost cndmask_cost.hip && ./cndmask_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned a, unsigned b) {
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, m = (threadIdx.x & 1) ? 0xffffffffu : 0u;
    unsigned long long sm = __ballot(threadIdx.x & 1);
    for (int it = 0; it < iters; ++it) {
        REP16(
            if (OP == 0) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(a));
                           asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(a)); }
            if (OP == 1) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(a) : ); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x1) : "v"(a));
                           asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x2) : "v"(a)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x3) : "v"(a)); }
            if (OP == 2) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "s"(sm));
                           asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "s"(sm)); }
            if (OP == 3) { asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x0) : "v"(b), "v"(a), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x1) : "v"(b), "v"(a), "s"(sm));
                           asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x2) : "v"(b), "v"(a), "s"(sm)); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x3) : "v"(b), "v"(a), "s"(sm)); }
            if (OP == 4) { asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(x0) : "v"(a), "v"(m)); asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(x1) : "v"(a), "v"(m));
                           asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(x2) : "v"(a), "v"(m)); asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(x3) : "v"(a), "v"(m)); }
            if (OP == 5) { asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(a) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x1) : "v"(a) : "vcc");
                           asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x2) : "v"(a) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x3) : "v"(a) : "vcc"); }
            if (OP == 6) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x0), "v"(a) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x1), "v"(a) : "vcc");
                           asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x2), "v"(a) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x3), "v"(a) : "vcc"); }
            if (OP == 7) { asm volatile("v_max_u32 %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(x1) : "v"(a));
                           asm volatile("v_max_u32 %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(x3) : "v"(a)); }
            if (OP == 8) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x0) : "v"(b), "v"(a)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x1) : "v"(b), "v"(a));
                           asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x2) : "v"(b), "v"(a)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x3) : "v"(b), "v"(a)); }
            if (OP == 9) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x0), "v"(a) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x1), "v"(a) : "vcc");
                           asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x2), "v"(a) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x3), "v"(a) : "vcc"); }
            if (OP == 10) { asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(sm) : "v"(x0), "v"(a)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "s"(sm));
                            asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(sm) : "v"(x1), "v"(a)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "s"(sm)); }
        )
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)sm;
}

template <int OP> static float run(unsigned* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * 4, 256>>>(d, 10, 3, 5);
    hipEventRecord(e0);
    k<OP><<<256 * 4, 256>>>(d, iters, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    unsigned* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    const int iters = 2000;
    const char* names[] = {"v_add_u32 (reference)", "v_cndmask_b32 vcc (loop-invariant vcc), dst = src0", "v_cndmask_b32_e64 sgpr pair, dst = src0",
                           "v_cndmask_b32_e64 sgpr pair, independent of dst", "v_bfi_b32 with a VGPR lane mask", "v_cmp_lt_u32 vcc + v_cndmask vcc (2 instr)",
                           "v_cmp_lt_u32 vcc", "v_max_u32", "v_cndmask_b32 vcc, independent of dst", "v_cmp_lt_f32 vcc", "v_cmp_e64 sgpr + v_cndmask_e64 (2 instr, 2 chains)"};
    const int per_rep[] = {4, 4, 4, 4, 4, 8, 4, 4, 4, 4, 4};
    for (int op = 0; op < 11; ++op) {
        float ms = 0;
        switch (op) {
#define L(i) case i: ms = run<i>(d, iters); break;
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10)
        }
        // 4 waves per SIMD (4 blocks of 4 waves on each of 256 CUs, one round)
        const double instr_per_simd = 4.0 * iters * 16 * per_rep[op];
        printf("%-56s %.2f cycles per wave-instruction per SIMD at 2.1 GHz (4 waves/SIMD)\n", names[op], ms * 1e-3 * 2.1e9 / instr_per_simd);
    }
    return 0;
}
