// Microbenchmark: VALU issue rate on gfx950 for the instruction kinds the integrator uses, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void k(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m};
    unsigned u0 = threadIdx.x + 1, u1 = u0 + 7, u2 = u0 + 11, u3 = u0 + 13;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (KIND == 3) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 4) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0 | 1));) }
        if (KIND == 5) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == 6) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 7) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_lshrrev_b32 %1, 3, %1\n v_add_u32 %2, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0 | 1));) }
        if (KIND == 8) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0 | 1));) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 + u1 + u2 + u3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 1024 * 4 * sizeof(float));
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {            // waves per SIMD: blocks of 256 threads (4 waves = 1 per SIMD), wps blocks per CU
        int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, 256>>>(out, iters, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double insts_per_wave = (double)iters * 64;   // 16 reps x 4 instr
        // s_memtime ticks at 100 MHz on gfx9? report both views
        printf("%-14s waves/SIMD %d: wall %.3f ms -> %.2f ns per wave-instr per SIMD ; memtime ticks/instr(one wave) %.3f\n", name, wps,
               ms, ms * 1e6 / (insts_per_wave * wps), (double)c / insts_per_wave);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_mul_f32"); run<2>("v_fma_f32"); run<1>("v_pk_mul_f32"); run<5>("v_pk_fma_f32"); run<3>("v_rcp_f32"); run<6>("v_sqrt_f32");
    run<4>("v_mul_lo_u32"); run<7>("int/logic mix"); run<8>("u24/mul_hi mix");
    return 0;
}
