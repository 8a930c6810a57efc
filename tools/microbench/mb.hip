// Instruction-rate microbenchmarks for the integer multiply roofline of the verify kernels (gfx950).
// Measures wave-instruction issue cost (cycles per wave64 instruction per SIMD) of the candidate multiply
// primitives at 1, 2, 4 and 8 waves per SIMD, with U independent dependency chains per lane.
//   hipcc --offload-arch=gfx950 -O3 -o mb mb.hip && ./mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define ITER 4096

template <int OP, int U>
__global__ void k(uint32_t* out, uint32_t seed) {
    uint32_t a[U], b[U]; uint64_t acc[U]; double d[U], e[U];
    for (int u = 0; u < U; u++) { a[u] = seed + threadIdx.x * 7 + u; b[u] = seed * 3 + u + blockIdx.x; acc[u] = a[u]; d[u] = 1.0 + a[u] * 1e-9; e[u] = 1.0 + 1e-9 * u; }
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[u]) : "vcc");
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 5) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[u]), "+v"(b[u]) : "v"(b[u]) : "vcc");
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[u]) : "v"(e[u]));
            if (OP == 7) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(acc[u]), "+v"(b[u]) : "v"(a[u]), "v"(b[u]) : "vcc");
            if (OP == 9) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 11) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 12) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[u]) : "v"(b[u]));
            // round 3: is the lone-wave mad rate (10.5 cycles against 6.9 for v_mul_lo) a WAW stall on the carry-out SGPR pair?
            if (OP == 13) {
                if ((u & 3) == 0) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[u]) : "s20", "s21");
                if ((u & 3) == 1) asm volatile("v_mad_u64_u32 %0, s[22:23], %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[u]) : "s22", "s23");
                if ((u & 3) == 2) asm volatile("v_mad_u64_u32 %0, s[24:25], %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[u]) : "s24", "s25");
                if ((u & 3) == 3) asm volatile("v_mad_u64_u32 %0, s[26:27], %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[u]) : "s26", "s27");
            }
            if (OP == 14) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[u]) : "v"(a[u]), "v"(b[u]) : "s20", "s21");
            if (OP == 15) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[u]));
            if (OP == 16) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[u]) : "v"(acc[(u + 1) % U]));
            if (OP == 17) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 18) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[u]));
            if (OP == 19) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(b[u]));
            if (OP == 20) asm volatile("v_mad_u64_u32 %0, vcc, %1, s30, %0" : "+v"(acc[u]) : "v"(a[u]) : "vcc");
        }
    }
    uint32_t r = 0;
    for (int u = 0; u < U; u++) r ^= a[u] ^ (uint32_t)acc[u] ^ (uint32_t)(acc[u] >> 32) ^ (uint32_t)d[u] ^ b[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP, int U>
void run(const char* name, int instr_per_op, uint32_t* d_out, double clk_ghz) {
    printf("%-28s U=%d :", name, U);
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * 4 * wps;      // one 64-lane block per wave slot
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<OP, U>), dim3(blocks), dim3(64), 0, 0, d_out, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, U>), dim3(blocks), dim3(64), 0, 0, d_out, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr_per_simd = (double)ITER * U * instr_per_op * wps;
        double cyc = ms * 1e-3 * clk_ghz * 1e9 / wave_instr_per_simd;
        printf("  %dw/SIMD %.2f cyc/instr", wps, cyc);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double clk = p.clockRate * 1e-6;   // GHz (nominal max; chip may run lower)
    printf("device %s CUs %d clock %.3f GHz (cycles below assume this clock)\n", p.gcnArchName, p.multiProcessorCount, clk);
    uint32_t* d_out; hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4);
    run<0, 1>("v_mad_u64_u32 (dep chain)", 1, d_out, clk);
    run<0, 4>("v_mad_u64_u32", 1, d_out, clk);
    run<0, 8>("v_mad_u64_u32", 1, d_out, clk);
    run<8, 4>("v_mad_u64_u32 + v_addc", 2, d_out, clk);
    run<1, 4>("v_mul_lo_u32", 1, d_out, clk);
    run<2, 4>("v_mul_hi_u32", 1, d_out, clk);
    run<3, 4>("v_mad_u32_u24", 1, d_out, clk);
    run<9, 4>("v_mad_i32_i24", 1, d_out, clk);
    run<7, 4>("v_mul_hi_u32_u24", 1, d_out, clk);
    run<4, 1>("v_add_u32 (dep chain)", 1, d_out, clk);
    run<4, 4>("v_add_u32", 1, d_out, clk);
    run<5, 4>("v_add_co + v_addc", 2, d_out, clk);
    run<10, 4>("v_lshl_add_u32", 1, d_out, clk);
    run<6, 1>("v_fma_f64 (dep chain)", 1, d_out, clk);
    run<6, 4>("v_fma_f64", 1, d_out, clk);
    run<11, 4>("v_pk_mul_lo_u16", 1, d_out, clk);
    run<12, 4>("v_dot4_u32_u8", 1, d_out, clk);
    run<14, 8>("v_mad_u64 sdst fixed s[20:21]", 1, d_out, clk);
    run<13, 8>("v_mad_u64 sdst rotating x4", 1, d_out, clk);
    run<20, 8>("v_mad_u64 sgpr multiplier", 1, d_out, clk);
    run<15, 8>("v_lshrrev_b64", 1, d_out, clk);
    run<16, 8>("v_lshl_add_u64", 1, d_out, clk);
    run<17, 8>("v_alignbit_b32", 1, d_out, clk);
    run<18, 8>("v_and_b32 (VOP2 literal)", 1, d_out, clk);
    run<19, 8>("v_mov_b32_dpp", 1, d_out, clk);
    return 0;
}
