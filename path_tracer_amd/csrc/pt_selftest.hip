// pt_selftest.hip — on-device proof that the short correctly-rounded sqrt / reciprocal / divide sequences of
// pt_device.h return the bits of hipcc's IEEE expansions on THIS GPU (v_rcp_f32 / v_rsq_f32 are only specified to
// 1 ulp, so the proof is re-run where the library runs): exhaustive over all 2^32 inputs for the one-operand forms,
// randomised (hash-generated, all exponents, special values mixed in) for the two- and three-operand forms and for
// the one-divide-per-ray scan against the literal scan.  Exposed as ir_selftest_exact_math.
#include "pt_kernels.h"

namespace pt {
namespace {

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ bool same(float a, float b) { return f32_bits(a) == f32_bits(b) || (a != a && b != b); }

// checks 0..5: every f32 bit pattern
__global__ __launch_bounds__(256) void selftest_unary(unsigned long long* bad) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    uint32_t miss[6] = {0, 0, 0, 0, 0, 0};
    for (; i < (1ull << 32); i += stride) {
        const uint32_t u = (uint32_t)i;
        const float x = as_f(u);
        const float ax = __builtin_fabsf(x);
        const bool mid = mid_range(ax);
        miss[0] += !same(sqrt_cr(x), __builtin_sqrtf(x));
        miss[1] += !same(rcp_sqrt_cr(x), 1.0f / __builtin_sqrtf(x));
        if (mid) miss[2] += !same(rcp_seq(x), 1.0f / x);
        if (mid || u == 0u) {
            miss[3] += !same(div_const_seq<0>(x), x / kPi);
            miss[4] += !same(div_const_seq<1>(x), x / kTwoPi);
        }
        {   // atan2_poly's quotient: the smaller magnitude over the larger one, every |x| against a hashed partner
            const uint32_t k = mix32(u);
            const float other = as_f((k & 0x007fffffu) | ((64u + (k >> 23) % 64u) << 23));  // [2^-63, 2)
            const bool steep = other > ax;
            const float mx = steep ? other : ax, mn = steep ? ax : other;
            if (in_pow2_range<64, 128>(mx) && (f32_bits(mn) == 0u || in_pow2_range<91, 128>(mn)))
                miss[5] += !same(div_seq(mn, mx, rcp_seq(mx)), mn / mx);
        }
    }
    for (int k = 0; k < 6; ++k)
        if (miss[k]) atomicAdd(&bad[k], (unsigned long long)miss[k]);
}

// a float with a uniformly random significand and a biased exponent drawn from [elo, ehi]; `special` picks, once in
// 64, a value from the edge set instead
__device__ float rand_float(uint32_t h, uint32_t h2, uint32_t elo, uint32_t ehi, bool allow_negative) {
    const uint32_t e = elo + (h2 >> 8) % (ehi - elo + 1u);
    uint32_t u = (h & 0x007fffffu) | (e << 23);
    const uint32_t sel = h2 & 63u;
    if (sel == 0u) u = 0u;                                  // +0
    if (sel == 1u) u = (e << 23);                           // power of two
    if (sel == 2u) u = (e << 23) | 0x007fffffu;             // all-ones significand
    if (sel == 3u) u = (e << 23) | (0x007fffffu >> (h & 15u));
    if (allow_negative && (h2 & 0x40u)) u |= 0x80000000u;
    return as_f(u);
}

// checks 6..8: div_seq pairs, scatter_weight, roulette_divide
__global__ __launch_bounds__(256) void selftest_random(unsigned long long n, uint32_t seed, unsigned long long* bad) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    uint32_t miss[3] = {0, 0, 0};
    for (; i < n; i += stride) {
        const uint32_t k0 = mix32((uint32_t)i ^ seed), k1 = mix32(k0 + (uint32_t)(i >> 32) * 0x9e3779b9u + 1u);
        const uint32_t k2 = mix32(k1 ^ 0x85ebca6bu), k3 = mix32(k2 + 0x27d4eb2fu), k4 = mix32(k3 ^ k0), k5 = mix32(k4 + 7u);
        {   // a / b inside div_seq's stated domain
            const float b = rand_float(k0, k1, 64, 190, true);
            const float a = rand_float(k2, k3, 27, 190, true);
            const float q = a / b;
            const float aq = __builtin_fabsf(q);
            const bool dom = mid_range(__builtin_fabsf(b)) && (f32_bits(a) == 0u || __builtin_fabsf(a) >= 7.8886090522101181e-31f) &&
                             (f32_bits(a) == 0u || (aq >= 7.8886090522101181e-31f && aq < 1.2676506002282294e+30f));
            if (dom) miss[0] += !same(div_seq(a, b, rcp_seq(b)), q);
        }
        {   // scatter_weight: brdf components as the host promises (+0 or [2^-30, 2^30)), cos anywhere incl. 0, tiny, huge
            const f3 brdf = mk3(rand_float(k0, k3, 97, 156, false), rand_float(k1, k4, 97, 156, false), rand_float(k2, k5, 97, 156, false));
            const float c = rand_float(k4, k2, (k5 & 1u) ? 100u : 1u, (k5 & 1u) ? 130u : 253u, false);
            const f3 f = scatter_weight(brdf, c, true), r = scatter_weight(brdf, c, false);
            miss[1] += !(same(f.x, r.x) && same(f.y, r.y) && same(f.z, r.z));
        }
        {   // roulette_divide: non-negative components anywhere (zeros, denormals, tiny, large), p = their maximum
            const uint32_t lo = (k3 & 1u) ? 90u : 0u, hi = (k3 & 1u) ? 130u : 254u;
            const f3 c = mk3(rand_float(k5, k0, lo, hi, false), rand_float(k4, k1, lo, hi, false), rand_float(k3, k2, lo, hi, false));
            const float p = __builtin_fmaxf(c.x, __builtin_fmaxf(c.y, c.z));
            const f3 f = roulette_divide(c, p, true), r = roulette_divide(c, p, false);
            miss[2] += !(same(f.x, r.x) && same(f.y, r.y) && same(f.z, r.z));
        }
    }
    for (int k = 0; k < 3; ++k)
        if (miss[k]) atomicAdd(&bad[6 + k], (unsigned long long)miss[k]);
}

// check 9: one-divide-per-ray scan == literal scan.  Each workgroup builds 8 spheres in LDS from hashes (with exact
// duplicates and 1-ulp-perturbed copies so that ties and near ties occur), each lane shoots rays at them.
__global__ __launch_bounds__(256) void selftest_trace(uint32_t rays_per_lane, uint32_t seed, unsigned long long* bad) {
    __shared__ float4 s_sph[8];
    if (threadIdx.x < 8) {
        const uint32_t j = threadIdx.x & 3u;  // spheres 4..7 repeat 0..3
        const uint32_t h = mix32(blockIdx.x * 8u + j + seed);
        float4 s;
        s.x = (float)(h & 1023u) * (1.0f / 128.0f) - 4.0f;
        s.y = (float)((h >> 10) & 1023u) * (1.0f / 128.0f) - 4.0f;
        s.z = (float)((h >> 20) & 1023u) * (1.0f / 128.0f) - 4.0f;
        float r = 0.25f + (float)(mix32(h) & 255u) * (1.0f / 256.0f);
        if (threadIdx.x >= 4 && (h & 1u)) r = as_f(f32_bits(r) + 1u + (h >> 30));  // near duplicate
        if (threadIdx.x >= 4 && (h & 2u)) s.x = as_f(f32_bits(s.x) ^ 1u);
        s.w = r * r;
        s_sph[threadIdx.x] = s;
    }
    __syncthreads();
    uint32_t miss = 0;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < rays_per_lane; ++k) {
        const uint32_t h0 = mix32(gid * 0x9e3779b9u + k + seed), h1 = mix32(h0 + 1u), h2 = mix32(h1 + 2u);
        const float4 tgt = s_sph[h0 & 7u];
        f3 o = mk3((float)(h0 >> 8 & 4095u) * (1.0f / 256.0f) - 8.0f, (float)(h1 >> 8 & 4095u) * (1.0f / 256.0f) - 8.0f,
                   (float)(h2 >> 8 & 4095u) * (1.0f / 256.0f) - 8.0f);
        f3 d = mk3(tgt.x, tgt.y, tgt.z) - o;
        const float jit = (float)(h2 & 255u) * (1.0f / 1024.0f);
        d = d + mk3(jit * (float)((int)(h0 >> 28) - 8), jit * (float)((int)(h1 >> 28) - 8), jit * (float)((int)(h2 >> 28) - 8)) * 0.125f;
        if ((h1 & 7u) == 0u) d = d * as_f(((h1 >> 3 & 255u) << 23) | 0x00400000u);  // any scale, incl. 0, denormal, inf
        if ((h1 & 63u) == 1u) { o = mk3(tgt.x, tgt.y, tgt.z); }                     // origin at a centre
        uint32_t c0 = 0, c1 = 0;
        const HitRec f = trace_linear<true>(s_sph, 8, o, d, c0), r = trace_linear_ieee<true>(s_sph, 8, o, d, c1);
        miss += !(f.idx == r.idx && same(f.t, r.t) && c0 == c1);
    }
    if (miss) atomicAdd(&bad[9], (unsigned long long)miss);
}

}  // namespace

hipError_t launch_selftest(unsigned long long* d_bad, uint32_t log2_random, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_unary, dim3(8192), dim3(256), 0, stream, d_bad);
    hipLaunchKernelGGL(selftest_random, dim3(8192), dim3(256), 0, stream, 1ull << log2_random, 0x5e1f7e57u, d_bad);
    const uint32_t rays_per_lane = (uint32_t)((1ull << log2_random) / (8192ull * 256ull));
    hipLaunchKernelGGL(selftest_trace, dim3(8192), dim3(256), 0, stream, rays_per_lane ? rays_per_lane : 1u, 0x7ace0001u, d_bad);
    return hipGetLastError();
}

}  // namespace pt
