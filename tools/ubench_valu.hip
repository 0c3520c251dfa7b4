// Microbenchmark: VALU issue rate on gfx950 for the instruction kinds the integrator uses, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip
// Each wave runs 8 independent dependency chains (4 would leave a lone wave latency-bound).  Per configuration it prints
//   - ns per wave-instruction per SIMD from the host's events around a >= 10 ms launch,
//   - the in-kernel clock: delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz) of one wave over its loop,
//   - CYCLES per wave-instruction per SIMD = ns per wave-instruction per SIMD x that clock,
// so that an issue rate in cycles does not depend on guessing the clock the chip holds under this load.  (The measured
// wave's own delta s_memtime is NOT the launch's duration once the grid needs more than one round of workgroups.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void k(float* out, int iters, long long* cyc) {  // cyc[0] = delta s_memtime, cyc[1] = delta s_memrealtime
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0}, pm = {m, m};
    unsigned u0 = threadIdx.x + 1, u1 = u0 + 7, u2 = u0 + 11, u3 = u0 + 13, u4 = u0 + 17, u5 = u0 + 19, u6 = u0 + 23, u7 = u0 + 29;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        if (KIND == 3) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 4) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(threadIdx.x | 1));) }
        if (KIND == 5) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));) }
        if (KIND == 6) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 7) { REP16(asm volatile("v_xor_b32 %0, %0, %8\n v_lshrrev_b32 %1, 3, %1\n v_add_u32 %2, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_xor_b32 %4, %4, %8\n v_lshrrev_b32 %5, 3, %5\n v_add_u32 %6, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(threadIdx.x | 1));) }
        if (KIND == 8) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(threadIdx.x | 1));) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                 p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cyc[0] = t1 - t0;
        cyc[1] = r1 - r0;
    }
}

template <int KIND>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 1024 * 4 * sizeof(float));
    hipMalloc(&cyc, 16);
    const double insts_per_iter = 16.0 * 8.0;  // 16 reps x 8 chains
    for (int wps : {1, 2, 4, 8}) {            // waves per SIMD: blocks of 256 threads (4 waves = 1 per SIMD), wps blocks per CU
        const int blocks = 256 * wps;
        const int iters = 60000 / wps;         // every configuration issues the same instructions per SIMD: >= 10 ms
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, 256>>>(out, iters, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        const double insts_per_wave = (double)iters * insts_per_iter;
        const double ghz = (double)c[0] / ((double)c[1] * 10.0);  // s_memrealtime ticks at 100 MHz = 10 ns
        const double ns = ms * 1e6 / (insts_per_wave * wps);
        printf("%-14s waves/SIMD %d: wall %7.3f ms -> %.3f ns per wave-instr per SIMD ; in-kernel clock %.3f GHz ; %.2f cycles per "
               "wave-instr per SIMD\n", name, wps, ms, ns, ghz, ns * ghz);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_mul_f32"); run<2>("v_fma_f32"); run<1>("v_pk_mul_f32"); run<5>("v_pk_fma_f32"); run<3>("v_rcp_f32"); run<6>("v_sqrt_f32");
    run<4>("v_mul_lo_u32"); run<7>("int/logic mix"); run<8>("u24/mul_hi mix");
    return 0;
}
