// exact_math_check.hip — exhaustive / randomised search for cheap instruction sequences that reproduce the correctly
// rounded f32 results of  1/x, sqrt(x), x/c (c = PI, 2*PI), 1/sqrt(x) (two roundings) and a/b  bit for bit.
// Reference on the device = hipcc's IEEE expansions (-fhip-fp32-correctly-rounded-divide-sqrt, the default).
// Prints, per candidate, the number of mismatching inputs and the first few of them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ bool same(float a, float b) { return as_u(a) == as_u(b) || (a != a && b != b); }

constexpr float kPi = 3.14159265358979323846264338327950288f;
constexpr float kTwoPi = 2.0f * kPi;

// ---------------- candidates
__device__ __forceinline__ float rcp1(float b) {  // one Newton step
    float y0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float rcp2(float b) {  // two steps
    float y0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y0, 1.0f);
    float y1 = __builtin_fmaf(e, y0, y0);
    float e1 = __builtin_fmaf(-b, y1, 1.0f);
    return __builtin_fmaf(e1, y1, y1);
}
__device__ __forceinline__ float sqrt_a(float x) {  // rsq seed + one residual correction
    float y = __builtin_amdgcn_rsqf(x);
    float s0 = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(r, h, s0);
}
__device__ __forceinline__ float sqrt_b(float x) {  // v_sqrt seed + correction with rsq as 1/(2s)
    float s0 = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    float r = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(r, h, s0);
}
template <int C>
__device__ __forceinline__ float div_const(float x) {
    const float c = C == 0 ? kPi : kTwoPi;
    const float ic = C == 0 ? (float)(1.0 / (double)kPi) : (float)(1.0 / (double)kTwoPi);
    float q = x * ic;
    float r = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(r, ic, q);
}
// 1/sqrt(x) with two roundings (glam normalize: v * (1.0 / sqrt(dot)))
__device__ __forceinline__ float rsqrt2_a(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float s0 = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-s0, s0, x);
    float s = __builtin_fmaf(r, h, s0);
    float e = __builtin_fmaf(-s, y, 1.0f);
    float y1 = __builtin_fmaf(e, y, y);
    float e1 = __builtin_fmaf(-s, y1, 1.0f);
    return __builtin_fmaf(e1, y1, y1);
}
__device__ __forceinline__ float rsqrt2_b(float x) {  // one reciprocal step only
    float y = __builtin_amdgcn_rsqf(x);
    float s0 = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-s0, s0, x);
    float s = __builtin_fmaf(r, h, s0);
    float e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// a/b with y = CR reciprocal: one / two Markstein steps
__device__ __forceinline__ float div1(float a, float b, float y) {
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float div2(float a, float b, float y) {
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    float q1 = __builtin_fmaf(r, y, q);
    float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}

struct Result {
    unsigned long long bad;
    uint32_t first[8];
};

// which: 0 rcp1, 1 rcp2, 2 sqrt_a, 3 sqrt_b, 4 div_pi, 5 div_2pi, 6 rsqrt2_a, 7 rsqrt2_b
__global__ void unary_kernel(int which, uint32_t lo, uint32_t hi_inclusive, Result* res) {
    const unsigned long long span = (unsigned long long)hi_inclusive - lo + 1ull;
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < span; i += stride) {
        const uint32_t u = lo + (uint32_t)i;
        const float x = as_f(u);
        float ref, got;
        switch (which) {
            case 0: ref = 1.0f / x; got = rcp1(x); break;
            case 1: ref = 1.0f / x; got = rcp2(x); break;
            case 2: ref = __builtin_sqrtf(x); got = sqrt_a(x); break;
            case 3: ref = __builtin_sqrtf(x); got = sqrt_b(x); break;
            case 4: ref = x / kPi; got = div_const<0>(x); break;
            case 5: ref = x / kTwoPi; got = div_const<1>(x); break;
            case 6: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt2_a(x); break;
            default: ref = 1.0f / __builtin_sqrtf(x); got = rsqrt2_b(x); break;
        }
        if (!same(ref, got)) {
            unsigned long long k = atomicAdd(&res->bad, 1ull);
            if (k < 8) res->first[k] = u;
        }
    }
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15; return x;
}
// random pairs with exponents of a and b in [emin, emax] (biased exponent), any sign; which: 0 div1(rcp1) 1 div1(rcp2) 2 div2(rcp1) 3 div2(rcp2)
__global__ void binary_kernel(int which, uint32_t seed, uint32_t emin, uint32_t emax, int b_all_ones, unsigned long long n, Result* res) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t k0 = mix32((uint32_t)i ^ seed), k1 = mix32(k0 + (uint32_t)(i >> 32) * 0x9e3779b9u + 0x85ebca6bu), k2 = mix32(k1 ^ 0xdeadbeefu);
        uint32_t ea = emin + k2 % (emax - emin + 1), eb = emin + (k2 >> 12) % (emax - emin + 1);
        uint32_t ua = (k0 & 0x807fffffu) | (ea << 23);
        uint32_t ub = (k1 & 0x807fffffu) | (eb << 23);
        if (b_all_ones) ub |= 0x007fffffu >> (k2 >> 28);  // long runs of ones at the top of the significand
        const float a = as_f(ua), b = as_f(ub);
        const float ref = a / b;
        const float y = (which & 1) ? rcp2(b) : rcp1(b);
        const float got = (which & 2) ? div2(a, b, y) : div1(a, b, y);
        if (!same(ref, got)) {
            unsigned long long k = atomicAdd(&res->bad, 1ull);
            if (k < 4) { res->first[2 * k] = ua; res->first[2 * k + 1] = ub; }
        }
    }
}

int main() {
    Result* d;
    CK(hipMalloc(&d, sizeof(Result)));
    const char* names[] = {"rcp1", "rcp2", "sqrt_rsqseed", "sqrt_sqrtseed", "div_pi", "div_2pi", "rsqrt2_2step", "rsqrt2_1step"};
    struct Range { const char* name; uint32_t lo, hi; };
    // positive normals with biased exponent in [64, 190] (2^-63 .. 2^63), all positive normals, everything
    Range ranges[] = {{"pos 2^-63..2^64", 64u << 23, (191u << 23) - 1}, {"pos normals", 1u << 23, (255u << 23) - 1}, {"pos incl. 0, denormals", 0u, (255u << 23) - 1}};
    for (int w = 0; w < 8; ++w)
        for (auto& r : ranges) {
            CK(hipMemset(d, 0, sizeof(Result)));
            hipLaunchKernelGGL(unary_kernel, dim3(4096), dim3(256), 0, 0, w, r.lo, r.hi, d);
            CK(hipDeviceSynchronize());
            Result h;
            CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
            printf("%-14s %-24s bad=%llu", names[w], r.name, h.bad);
            for (unsigned k = 0; k < (h.bad < 4 ? h.bad : 4); ++k) { float f; memcpy(&f, &h.first[k], 4); printf(" %08x(%g)", h.first[k], f); }
            printf("\n");
            fflush(stdout);
        }
    // negative inputs for the sign-symmetric ones (rcp, div const)
    for (int w : {0, 1, 4, 5}) {
        CK(hipMemset(d, 0, sizeof(Result)));
        hipLaunchKernelGGL(unary_kernel, dim3(4096), dim3(256), 0, 0, w, 0x80000000u | (64u << 23), 0x80000000u | ((191u << 23) - 1), d);
        CK(hipDeviceSynchronize());
        Result h;
        CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-14s %-24s bad=%llu\n", names[w], "neg 2^-63..2^64", h.bad);
    }
    const char* bn[] = {"div1(rcp1)", "div1(rcp2)", "div2(rcp1)", "div2(rcp2)"};
    for (int w = 0; w < 4; ++w)
        for (int ones = 0; ones < 2; ++ones)
            for (int wide = 0; wide < 2; ++wide) {
                CK(hipMemset(d, 0, sizeof(Result)));
                const uint32_t emin = wide ? 64 : 120, emax = wide ? 190 : 134;
                const unsigned long long n = 1ull << 36;
                hipLaunchKernelGGL(binary_kernel, dim3(8192), dim3(256), 0, 0, w, 12345u + w, emin, emax, ones, n, d);
                CK(hipDeviceSynchronize());
                Result h;
                CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
                printf("%-12s exps[%u,%u] b_ones=%d  n=2^36 bad=%llu", bn[w], emin, emax, ones, h.bad);
                for (unsigned k = 0; k < (h.bad < 2 ? h.bad : 2); ++k) printf(" a=%08x b=%08x", h.first[2 * k], h.first[2 * k + 1]);
                printf("\n");
                fflush(stdout);
            }
    return 0;
}
