// pt_device.h — device-side building blocks of the gfx950 integrator (wave64, FP32 VALU).
//
// Arithmetic contract: this translation unit is compiled with -ffp-contract=off and without fast-math,
// hipcc's default correctly rounded f32 divide/sqrt stays on.  Every expression below is written in the
// operation order of the reference (glam 0.30 scalar Vec3 semantics), so that results are bit-identical
// to an IEEE-754 CPU evaluation in the same order.  Citations: /root/reference/core-engine/src.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// __host__ too: tests/cpp/lbvh_check.hip runs the traversal code below on the CPU against the scan
#define PT_DEV __host__ __device__ __forceinline__

namespace pt {

constexpr float kPi = 3.14159265358979323846264338327950288f;  // std::f32::consts::PI
constexpr float kTwoPi = 2.0f * kPi;                            // Rust folds 2.0 * PI to one f32 constant
constexpr float kEps = 1.1920929e-07f;                          // f32::EPSILON
constexpr float kF32Max = 3.40282347e+38f;                      // f32::MAX

struct f3 {
    float x, y, z;
};
PT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
PT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PT_DEV float dot(f3 a, f3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
PT_DEV f3 cross(f3 a, f3 b) {  // glam vec3.rs
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

// ----------------------------------------------------------------------------------------------------
// Correctly rounded f32 sqrt / reciprocal / divide in few instructions.
//
// hipcc expands an IEEE f32 sqrt into ~20 and a divide into ~12 issue slots (range scaling, v_div_scale/fmas/fixup,
// +-1 ulp probing, hazard nops); those expansions were ~45 % of the integrator's VALU stream.  For operands in the
// mid range [2^-63, 2^64) the sequences below return the SAME correctly rounded bits:
//   sqrt_seq : y = rsq(x); s0 = x*y; s = fma(fma(-s0,s0,x), y/2, s0)           exhaustive over every f32 in range
//   rcp_seq  : y0 = rcp(b); y = fma(fma(-b,y0,1), y0, y0)                       exhaustive, both signs
//   x / c    : q = x*RN(1/c); q' = fma(fma(-q,c,x), RN(1/c), q), c = PI, 2PI    exhaustive, both signs and +0
//   div_seq  : y = RN(1/b); q = a*y; two Markstein steps q <- fma(fma(-b,q,a), y, q).  After the first step q is
//              within (1/2 + 2^-23) ulp of a/b, i.e. faithful, and a faithful quotient refined with the correctly
//              rounded reciprocal rounds correctly (Markstein 1990).  Needs the residuals exact and nothing
//              under/overflowing: b in the mid range, a == +0 or 2^-100 <= |a| < 2^64, 2^-100 <= |a/b| < 2^100.
// tools/exact_math_check.hip is the search harness; ir_selftest_exact_math re-proves every use below on the device
// the library runs on (tests/test_gpu_exact_math.py).  Callers branch per WAVE: the sequence is taken only if every
// active lane's operands are in range (one scalar branch on a ballot), otherwise the whole wave evaluates the plain
// IEEE expression, so results never depend on which path ran.  Host builds (tests/cpp) use the IEEE expression.
// ----------------------------------------------------------------------------------------------------
PT_DEV uint32_t f32_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
// lo <= x < hi for positive normal lo, hi given by their biased exponents: false for 0, negatives, denormals, inf,
// NaN.  One subtract + one unsigned compare.
template <uint32_t ELO, uint32_t EHI>
PT_DEV bool in_pow2_range(float x) { return (f32_bits(x) - (ELO << 23)) < ((EHI - ELO) << 23); }
PT_DEV bool mid_range(float x) { return in_pow2_range<64, 191>(x); }  // 2^-63 <= x < 2^64

#if defined(__HIP_DEVICE_COMPILE__)
#define PT_FAST_MATH 1
PT_DEV bool wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }
PT_DEV float sqrt_seq(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float h = 0.5f * y;
    const float r = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(r, h, s0);
}
PT_DEV float rcp_seq(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
PT_DEV float div_seq(float a, float b, float y) {  // y = rcp_seq(b)
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}
template <int TWO_PI>
PT_DEV float div_const_seq(float x) {  // x / PI or x / (2 PI); x == +0 or 2^-63 <= |x| < 2^64
    constexpr float c = TWO_PI ? kTwoPi : kPi;
    constexpr float ic = TWO_PI ? 0.15915494309189535f : 0.3183098861837907f;  // RN(1/c)
    const float q = x * ic;
    const float r = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(r, ic, q);
}
#else
#define PT_FAST_MATH 0
// host pass: the same names as plain IEEE expressions (kernel bodies are parsed for the host too)
PT_DEV bool wave_all(bool ok) { return ok; }
PT_DEV float sqrt_seq(float x) { return __builtin_sqrtf(x); }
PT_DEV float rcp_seq(float b) { return 1.0f / b; }
PT_DEV float div_seq(float a, float b, float) { return a / b; }
template <int TWO_PI>
PT_DEV float div_const_seq(float x) { return x / (TWO_PI ? kTwoPi : kPi); }
#endif

PT_DEV float sqrt_cr(float x) {  // == sqrtf(x) for every x
#if PT_FAST_MATH
    if (__builtin_expect(wave_all(mid_range(x)), 1)) return sqrt_seq(x);
#endif
    return __builtin_sqrtf(x);
}
PT_DEV float rcp_sqrt_cr(float x) {  // == 1.0f / sqrtf(x) (two roundings: glam length_recip) for every x
#if PT_FAST_MATH
    if (__builtin_expect(wave_all(mid_range(x)), 1)) return rcp_seq(sqrt_seq(x));  // sqrt lands in [2^-32, 2^32]
#endif
    return 1.0f / __builtin_sqrtf(x);
}
PT_DEV f3 normalize(f3 a) { return a * rcp_sqrt_cr(dot(a, a)); }  // glam: v * length_recip()

// ----------------------------------------------------------------------------------------------------
// Counter-based RNG (replaces rand::ThreadRng, sampler.rs:4-6,21-28).
// draw k of sample s of pixel p under a 64-bit seed:
//     pkey = mix32(p ^ seed_lo);  skey = mix32(pkey + s*0x9E3779B9 + seed_hi);  u = mix32(skey + k*0x85EBCA6B)
//     f    = (u >> 8) * 2^-24          (rand 0.9 StandardUniform for f32: 24 high bits, [0,1))
// mix32 is a bijective xorshift-multiply finaliser ("lowbias32" family), so within a pixel distinct s give
// distinct stream keys and within a stream distinct k give distinct outputs.
// ----------------------------------------------------------------------------------------------------
PT_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
PT_DEV uint32_t pixel_key(uint32_t seed_lo, uint32_t pixel) { return mix32(pixel ^ seed_lo); }
PT_DEV uint32_t sample_key(uint32_t seed_hi, uint32_t pkey, uint32_t sample) {
    return mix32(pkey + sample * 0x9e3779b9u + seed_hi);
}
PT_DEV float draw(uint32_t skey, uint32_t k) {
    return (float)(mix32(skey + k * 0x85ebca6bu) >> 8) * 5.9604644775390625e-08f;
}

// ----------------------------------------------------------------------------------------------------
// Polynomial transcendentals: plain + - * / (no FMA, no hardware sin/cos/rcp approximations), so the
// same source order evaluated on a CPU gives the same bits.  Classic single-precision minimax sets.
// ----------------------------------------------------------------------------------------------------
PT_DEV void sincos_poly(float x, float& s_out, float& c_out) {  // 0 <= x <= 2*pi
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    int j = (int)(FOPI * x);
    j = (j + 1) & ~1;
    float y = (float)j;
    float r = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
               0.5f * z + 1.0f;
    int q = (j >> 1) & 3;
    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? sp : cp;
    s = (q & 2) ? -s : s;
    c = ((q == 1) || (q == 2)) ? -c : c;
    s_out = s;
    c_out = c;
}
PT_DEV float asin_core(float a) {  // |a| <= 0.5
    float z = a * a;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
            1.6666752422e-1f) * z * a + a;
}
// acos: the three classic ranges share one asin_core evaluation (same operations per lane as the branchy form:
// 1 + x == 1 - |x| for x < 0), so a wave pays for one polynomial instead of three divergent ones.
PT_DEV float acos_poly(float x) {
    const float PIO2 = 1.5707963267948966f;
    const float ax = __builtin_fabsf(x);
    const bool big = ax > 0.5f;
    float arg = x;
    if (big) arg = sqrt_cr(0.5f * (1.0f - ax));
    const float p = asin_core(arg);
    const float two_p = 2.0f * p;
    return big ? ((x > 0.5f) ? two_p : kPi - two_p) : PIO2 - p;
}
// atan2 with ONE divide: q = min(|x|, |y|) / max(|x|, |y|) in [0, 1], atan(q) = q * P(q^2) with a degree-7 P fitted here
// (Chebyshev interpolation of atan(sqrt z) / sqrt z on [0, 1]; largest error of the f32 evaluation 1.7e-7 rad, the
// classic three-range reduction it replaces had two divides and the same error), then the octant: pi/2 - a when
// |y| > |x|, pi - a for x < 0, sign of y.  x == y == 0 gives 0, the sign of zero is ignored.  Plain * + - / in this
// order, so that a CPU restatement gives the same bits.
PT_DEV float atan2_poly(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const bool steep = ay > ax;
    const float mx = steep ? ay : ax, mn = steep ? ax : ay;
    float q;
#if PT_FAST_MATH
    // |x|, |y| of a (normalised) direction: mx in [2^-63, 2), mn == +0 or in [2^-36, 2) keeps the quotient in
    // [2^-37, 1]: div_seq's domain.  Anything else takes the IEEE divide, wave by wave.
    if (__builtin_expect(wave_all(in_pow2_range<64, 128>(mx) && (f32_bits(mn) == 0u || in_pow2_range<91, 128>(mn))), 1))
        q = div_seq(mn, mx, rcp_seq(mx));
    else
#endif
        q = mn / mx;
    const float z = q * q;
    float p = -0.004668773151934147f;
    p = p * z + 0.02416618913412094f;
    p = p * z - 0.0593671016395092f;
    p = p * z + 0.09906096756458282f;
    p = p * z - 0.14016585052013397f;
    p = p * z + 0.19969235360622406f;
    p = p * z - 0.33331960439682007f;
    p = p * z + 0.9999998807907104f;
    float a = p * q;
    if (mx == 0.0f) a = 0.0f;  // 0 / 0
    if (steep) a = 1.5707963267948966f - a;
    if (x < 0.0f) a = kPi - a;
    if (y < 0.0f) a = -a;
    return a;
}

// ----------------------------------------------------------------------------------------------------
// Camera block (pinhole_camera.rs:7-23 cached state; tan(fov/2) evaluated once on the host, :86)
// ----------------------------------------------------------------------------------------------------
struct CameraBlock {
    float xc[3], yc[3], zc[3];  // x/y/z axis columns of local_to_world (upper 3 rows)
    float pos[3];
    float tan_half_fov;
    float aspect;
};

struct Ray {
    f3 o, d;
};

// PinholeCamera::get_ray, pinhole_camera.rs:78-98
PT_DEV Ray pinhole_ray(const CameraBlock& cam, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
    float vx = ((float)x + 0.5f) / (float)w;
    float vy = ((float)y + 0.5f) / (float)h;
    vx = (vx * 2.0f - 1.0f) * cam.tan_half_fov;
    vy = (vy * 2.0f - 1.0f) * cam.tan_half_fov;
    vx *= cam.aspect;
    f3 X = mk3(cam.xc[0], cam.xc[1], cam.xc[2]);
    f3 Y = mk3(cam.yc[0], cam.yc[1], cam.yc[2]);
    f3 Z = mk3(cam.zc[0], cam.zc[1], cam.zc[2]);
    f3 res = X * vx;         // glam Mat4::transform_vector3
    res = Y * vy + res;
    res = Z * -1.0f + res;
    Ray r;
    r.o = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
    r.d = normalize(res);
    return r;
}

// ----------------------------------------------------------------------------------------------------
// Closest hit: linear scan over an LDS-staged SoA buffer.  s_sph[i] = (cx, cy, cz, r*r).
// integrator.rs:95-134: per sphere  oc = o - c; a = d.d; b = 2(d.oc); c = oc.oc - r*r;
// disc = b*b - (4a)*c; skip if disc < 0; t = (-b - sqrt(disc)) / (2a); accept iff t > 0 && t < best.
// a, 4a and 2a do not depend on the sphere and are hoisted (same values).
// ----------------------------------------------------------------------------------------------------
struct HitRec {
    float t;
    int idx;  // -1 = miss
};

// The scan exactly as written in the reference: one divide per candidate root.
template <bool STATS>
PT_DEV HitRec trace_linear_ieee(const float4* __restrict__ s_sph, uint32_t n, f3 o, f3 d, uint32_t& disc_pos) {
    float best = kF32Max;
    int idx = -1;
    const float a = dot(d, d);
    const float a4 = 4.0f * a;
    const float a2 = 2.0f * a;
    for (uint32_t i = 0; i < n; ++i) {
        const float4 s = s_sph[i];
        f3 oc = o - mk3(s.x, s.y, s.z);
        float b = 2.0f * dot(d, oc);
        float c = dot(oc, oc) - s.w;
        float disc = b * b - a4 * c;
        if (disc < 0.0f) continue;
        if (STATS) disc_pos++;
        float sq = sqrt_cr(disc);
        float t = (-b - sq) / a2;
        if (t > 0.0f && t < best) {
            idx = (int)i;
            best = t;
        }
    }
    HitRec h;
    h.t = best;
    h.idx = idx;
    return h;
}

// Same result with ONE divide per ray.  t_i = RN(num_i / 2a) with num_i = -b_i - sqrt(disc_i) and a common
// denominator 2a > 0, and x -> RN(x / 2a) is monotone, so the scan's winner (smallest t, lowest index among equal t)
// is the sphere with the smallest positive numerator (lowest index among equal numerators) — unless a second,
// different numerator is so close that both may round to the same t, or t leaves (2^-100, f32::MAX) (underflow to
// 0, a == 0, overflow, NaN).  Those rays (a wave-uniform branch, practically never taken) rerun the literal scan.
template <bool STATS>
PT_DEV HitRec trace_linear(const float4* __restrict__ s_sph, uint32_t n, f3 o, f3 d, uint32_t& disc_pos) {
    const float kInf = __builtin_inff();
    float nmin = kInf, nsecond = kInf;  // smallest / second smallest positive numerator (with multiplicity)
    int idx = -1;
    const float a = dot(d, d);
    const float a4 = 4.0f * a;
    const float a2 = 2.0f * a;
    uint32_t dp = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const float4 s = s_sph[i];
        f3 oc = o - mk3(s.x, s.y, s.z);
        float b = 2.0f * dot(d, oc);
        float c = dot(oc, oc) - s.w;
        float disc = b * b - a4 * c;
        if (disc < 0.0f) continue;
        if (STATS) dp++;
        // the root's numerator -b - sqrt(disc) can only be positive for b < 0 (sqrt >= 0; NaN and -0 compare false and
        // would be rejected by num > 0 anyway): rays leaving a sphere they stand on skip its sqrt
        if (!(b < 0.0f)) continue;
        const float num = -b - sqrt_cr(disc);
        if (num > 0.0f) {
            nsecond = __builtin_fminf(nsecond, __builtin_fmaxf(nmin, num));
            if (num < nmin) {
                nmin = num;
                idx = (int)i;
            }
        }
    }
    HitRec h;
    h.t = kF32Max;
    h.idx = -1;
    bool redo = false;
    if (idx >= 0) {
        const float t = nmin / a2;
        h.t = t;
        h.idx = idx;
        // 1 + 2^-21: two numerators further apart than 4 ulp cannot round to the same normal quotient
        redo = !(t > 7.8886090522101181e-31f && t < kF32Max) || (nsecond != nmin && !(nsecond > nmin * 1.000000476837158203125f));
    }
    if (__builtin_expect(!wave_all(!redo), 0)) {
        if (redo) {
            uint32_t ignored = 0;
            h = trace_linear_ieee<false>(s_sph, n, o, d, ignored);
        }
    }
    if (STATS) disc_pos += dp;
    return h;
}

// ----------------------------------------------------------------------------------------------------
// Closest hit through the LBVH (lbvh.h).  Ordered traversal: an inner node carries both children's boxes, the nearer
// hit child is entered first and the farther one is pushed on a small per-lane stack, so that `best` shrinks early
// and prunes the far field.  Leaf tests are the scan's arithmetic; the accept rule adds "equal t keeps the lower
// original index", which is what a scan in index order with a strict '<' returns, so the result does not depend on
// the visiting order.  Node tests are conservative: each box is inflated by pad = (max-norm distance from the ray origin
// to the farthest box corner) * 0.003 before the slab test.  Why that is enough (u = 2^-24, M = max(|oc|, r), a = d.d):
//   * oc = o - c is one rounding per component, i.e. the exact difference to a centre moved by <= u |oc|;
//   * b = 2 dot(d, oc) carries <= 6u |d||oc|, cc = dot(oc, oc) - r*r <= 4u M^2, a <= 3u a; so the computed discriminant
//     is within E <= 68 u a M^2 of D = 4a (r^2 - rho^2), rho = distance of the centre from the ray's line;
//   * an accepted sphere has a computed t with disc >= 0; the point P = o + t d then satisfies
//     |P - c|^2 = rho^2 + a (t - t0)^2 <= r^2 + (E + 2 sqrt(disc) |e_b| + e_b^2)/(4a) + (roundings of t and of 2a)
//                                   <= r^2 + 45 u M^2,   so   |P - c| <= r + sqrt(45 u) M = r + 1.64e-3 M;
//   * the box holds c +- r (built one ulp outwards), so with L = the max-norm distance from the origin to its farthest
//     corner: |oc| <= sqrt(3) L, r <= L, M <= sqrt(3) L, and the slack is <= 1.64e-3 * sqrt(3) L = 2.84e-3 L < 0.003 L with
//     room for the slab test's own roundings (subtraction, hardware reciprocal, product: < 5u = 3e-7 of a slab
//     distance, against a margin of 1.6e-4 L): P lies inside the padded box, hence tmin <= t <= tmax, tmax > 0, and
//     tmin <= t <= best whenever the sphere could still win.  No sphere the scan would accept with t <= best is pruned.
//   * a record stores the box as centre + half extent (f16, rounded up): a box that contains the builder's, so everything
//     above holds for it; the test takes L = max_k(|centre_k - o_k| + e_k), which is >= that box's far-corner distance,
//     and evaluates slab k as RN(RN(d_k i_k) -+ RN((e_k + pad) |i_k|)): three roundings and the reciprocal's ulp, < 4u of
//     (|d_k| + e_k + pad) |i_k| <= 1.003 L |i_k|, i.e. < 2.4e-7 L |i_k| against the margin of 1.6e-4 L |i_k|.
// (Until round 2 the pad was 2^-7 of the L1 distance: 10 % more node visits and 17 % more leaf tests on the 10k scene.)
// ----------------------------------------------------------------------------------------------------
// Record `index` of a table of fewer than 2^28 16-byte records (the host checks the sizes at upload): a 32-bit byte offset
// off the table's uniform base, i.e. one shift and a load with a scalar base where `base[index]` with a 64-bit index costs
// three or four vector instructions of address arithmetic.
PT_DEV float4 load16(const float4* __restrict__ base, uint32_t index) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (uint32_t)(index << 4));
#else
    return base[index];
#endif
}
// y * w + x for y, w < 2^24 (one full-rate v_mad_u32_u24)
PT_DEV uint32_t row_major(uint32_t y, uint32_t w, uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (y & 0xffffffu) * (w & 0xffffffu) + x;  // the masks tell the compiler both factors fit 24 bits
#else
    return y * w + x;
#endif
}

struct LbvhView {
    const float4* __restrict__ wide;          // 3 x float4 per record (lbvh.h WideNode): n_inner wide nodes, then climb records
    const float4* __restrict__ leaf_spheres;  // leaf order: (cx, cy, cz, r*r)
    const int* __restrict__ leaf_index;       // leaf order -> original sphere index
    const int2* __restrict__ sphere_start;    // original sphere index -> (leaf ref, climb record or kCursorDone), lbvh.h
    uint32_t n_inner;                         // records below it are wide nodes, records from it on are climb records
    uint32_t stack_depth;                     // entries a traversal may need (<= kMaxStack)
};

// Traversal state of one ray.  A traversal is cursor_begin + cursor_step until cur == kCursorDone: the integrator's
// LBVH kernel interleaves steps of different lanes' rays with shading (paths_lbvh_kernel), trace_lbvh runs one to
// the end.  The stack lives outside (LDS on the device: entry k of this lane at stack[k * stride]).
constexpr int kCursorDone = 0x7fffffff;
#if !defined(__HIP_DEVICE_COMPILE__)
inline float half_bits_to_float(uint32_t h) {  // host builds of the traversal (tests/cpp): f16 -> f32 by hand
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = (float)m * 5.9604644775390625e-08f;  // m * 2^-24
    else if (e == 31) v = m ? __builtin_nanf("") : __builtin_inff();
    else v = __builtin_ldexpf((float)(m | 1024u), (int)e - 25);
    return (h & 0x8000u) ? -v : v;
}
#endif
PT_DEV float box_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
struct LbvhCursor {
    f3 o, d;
    float a4, a2, ix, iy, iz;
    float best;
    int idx;       // winner so far as a LEAF-ORDER position (-1: none); cursor_hit_index() translates it at the end
    int cur;       // >= 0: record (inner node below n_inner, climb record from n_inner on), < 0: leaf -(first * 8 + count) - 1,
                   // kCursorDone: finished
    int up;        // where the traversal goes on when the stack runs empty: the next climb record, or kCursorDone
    uint32_t sp;   // stack entries in use
};
PT_DEV void cursor_begin(LbvhCursor& c, f3 o, f3 d) {
    c.o = o;
    c.d = d;
    const float a = dot(d, d);
    c.a4 = 4.0f * a;
    c.a2 = 2.0f * a;
    // +-inf for an axis-parallel ray.  On the device the hardware reciprocal (1 ulp): the slab test only has to be
    // conservative, and the pad's margin covers a few ulp of every slab distance (see above)
    c.ix = box_rcp(d.x);
    c.iy = box_rcp(d.y);
    c.iz = box_rcp(d.z);
    c.best = kF32Max;
    c.idx = -1;
    c.cur = 0;
    c.up = kCursorDone;
    c.sp = 0;
}
// A ray that leaves a sphere starts at that sphere's leaf and climbs (lbvh.h "Climb records"): the leaf first, then the
// sibling subtrees of its ancestors, two levels per record.  start = LbvhView::sphere_start[sphere].  Same (t, index) as
// a traversal from the root, whatever `start` is: it only decides the visiting order.
PT_DEV void cursor_begin_at(LbvhCursor& c, f3 o, f3 d, int2 start) {
    cursor_begin(c, o, d);
#ifndef IR_NO_CLIMB  // (tools/ A/B builds: every traversal from the root)
    c.cur = start.x;
    c.up = start.y;
#endif
}
#ifndef IR_BOX_PAD
#define IR_BOX_PAD 0.003f  // other values only in tools/ experiments (they void the parity guarantee)
#endif
constexpr float kBoxPad = IR_BOX_PAD;
// Box given as centre and half extent (lbvh.h: a record stores the half extent as f16, rounded up).  d = centre - o,
// L = max_k(|d_k| + e_k) >= the max-norm distance from the origin to the farthest corner, slab k = d_k i_k -+ (e_k + pad) |i_k|
// (no min / max per axis: the half width is never negative).  A 0 * inf product (d_k == 0 or e_k + pad == 0 on an axis the ray
// is parallel to) is NaN and leaves the axis unconstrained (fmin / fmax return the other operand); an axis-parallel ray that
// passes outside the slab is NOT pruned by this form (inf - inf), which costs such a ray visits, never a hit.  A NaN half
// extent (the missing uncle of a climb record) makes every comparison false: no hit.
PT_DEV bool padded_box_hit(const LbvhCursor& c, f3 centre, f3 e, float& tmin) {
    const float dx = centre.x - c.o.x, dy = centre.y - c.o.y, dz = centre.z - c.o.z;
    const float pad = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(dx) + e.x, __builtin_fabsf(dy) + e.y), __builtin_fabsf(dz) + e.z) * kBoxPad;
    const float tcx = dx * c.ix, tcy = dy * c.iy, tcz = dz * c.iz;
    const float hx = (e.x + pad) * __builtin_fabsf(c.ix), hy = (e.y + pad) * __builtin_fabsf(c.iy), hz = (e.z + pad) * __builtin_fabsf(c.iz);
    tmin = __builtin_fmaxf(__builtin_fmaxf(tcx - hx, tcy - hy), tcz - hz);
    const float tmax = __builtin_fminf(__builtin_fminf(tcx + hx, tcy + hy), tcz + hz);
    return (tmax >= __builtin_fmaxf(tmin, 0.0f)) && (tmin <= c.best);
}
PT_DEV float half_lo(float packed) {  // the f16 in the low / high half of a record word, as f32 (exact)
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(f32_bits(packed) & 0xffffu));
#else
    return half_bits_to_float(f32_bits(packed) & 0xffffu);
#endif
}
PT_DEV float half_hi(float packed) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(f32_bits(packed) >> 16));
#else
    return half_bits_to_float(f32_bits(packed) >> 16);
#endif
}
// An empty stack hands the traversal to the next climb record (kCursorDone when there is none: a traversal from the root,
// or a climb that has reached it).  c.up is not cleared here: the record it names overwrites it when it is stepped.
PT_DEV int cursor_pop(LbvhCursor& c, const int* stack, uint32_t stride) {
    if (c.sp == 0) return c.up;
    c.sp--;
    return stack[c.sp * stride];
}
// Precondition: c.cur >= 0 and != kCursorDone (a record).  Tests both boxes, enters the nearer hit one and pushes the
// farther; with none hit, pops.  Afterwards c.cur is a record, a leaf (< 0) or kCursorDone.
template <bool STATS>
PT_DEV void cursor_inner_step(const LbvhView& v, LbvhCursor& c, int* stack, uint32_t stride, uint32_t& node_visits) {
    const uint32_t at = 3u * (uint32_t)c.cur;  // 48-byte records (lbvh.h); fewer than 3 * 2^24 of them: load16's range
    const float4 n0 = load16(v.wide, at), n1 = load16(v.wide, at + 1u), n2 = load16(v.wide, at + 2u);
    if (STATS) node_visits += 2;  // two box tests
    float tl, tr;
    const bool hl = padded_box_hit(c, mk3(n0.x, n0.y, n0.z), mk3(half_lo(n2.x), half_hi(n2.x), half_lo(n2.y)), tl);
    const bool hr = padded_box_hit(c, mk3(n1.x, n1.y, n1.z), mk3(half_hi(n2.y), half_lo(n2.z), half_hi(n2.z)), tr);
    const int rl = __builtin_bit_cast(int, n0.w), rr = __builtin_bit_cast(int, n1.w);
    // a climb record: the same two box tests (sibling and uncle subtrees), and it names the record two levels up in the
    // word a wide node leaves zero (no record has index 0).  Deciding on the loaded word keeps this ONE 16-byte load: a
    // test of c.cur against n_inner made the compiler fetch the word with a fifth, conditional load, and the kernel is
    // bound by the L1 data path.
    const int next_up = __builtin_bit_cast(int, n2.w);
    if (next_up != 0) c.up = next_up;
    // Written with selects around ONE push and ONE pop: the four-way if/else cascade cost the divergent wave four nested
    // exec-mask regions (35 scalar instructions per step, a fifth of the step).
    const bool both = hl && hr, any = hl || hr;
    const bool right_first = hr && (!hl || tr < tl);  // enter the right child first: it is nearer, or the only one hit
    const int near = right_first ? rr : rl, far = right_first ? rl : rr;
    if (both) {
        stack[c.sp * stride] = far;
        c.sp++;
    }
    int next = near;
    if (!any) next = cursor_pop(c, stack, stride);
    c.cur = next;
}
// Precondition: c.cur < 0 (leaf).  Tests its <= kLeafMax spheres with the scan's arithmetic, then pops.
template <bool STATS>
PT_DEV void cursor_leaf_step(const LbvhView& v, LbvhCursor& c, const int* stack, uint32_t stride, uint32_t& disc_pos, uint32_t& leaf_tests) {
    const uint32_t leaf = (uint32_t)(-(c.cur + 1));
    const uint32_t first = leaf >> 3, count = leaf & 7u;
    const f3 o = c.o, d = c.d;
    for (uint32_t k = 0; k < count; ++k) {
        const float4 s = load16(v.leaf_spheres, first + k);
        if (STATS) leaf_tests++;
        f3 oc = o - mk3(s.x, s.y, s.z);
        float b = 2.0f * dot(d, oc);
        float cc = dot(oc, oc) - s.w;
        float disc = b * b - c.a4 * cc;
        if (disc < 0.0f) continue;
        if (STATS) disc_pos++;
        // t = (-b - sqrt(disc)) / 2a > 0 needs b < 0 (2a is >= 0, +inf or NaN; see trace_linear)
        if (!(b < 0.0f)) continue;
        float sq = sqrt_cr(disc);
        float t = (-b - sq) / c.a2;
        // The winner is kept as its leaf-order position: the original index (one more dependent load) is only needed to
        // break an exact tie, and once per ray at the end (cursor_hit_index).
        bool better = t < c.best;
        if (t == c.best && c.idx >= 0) better = v.leaf_index[first + k] < v.leaf_index[c.idx];
        if (t > 0.0f && better) {
            c.idx = (int)(first + k);
            c.best = t;
        }
    }
    c.cur = cursor_pop(c, stack, stride);
}
// Original sphere index of a finished traversal's winner (-1: the ray escaped)
PT_DEV int cursor_hit_index(const LbvhView& v, const LbvhCursor& c) { return c.idx >= 0 ? v.leaf_index[c.idx] : -1; }
// one step of a traversal run on its own: the inner node and, if that leads to a leaf, the leaf
template <bool STATS>
PT_DEV void cursor_step(const LbvhView& v, LbvhCursor& c, int* stack, uint32_t stride, uint32_t& disc_pos, uint32_t& node_visits,
                        uint32_t& leaf_tests) {
    if (c.cur >= 0) cursor_inner_step<STATS>(v, c, stack, stride, node_visits);
    if (c.cur < 0) cursor_leaf_step<STATS>(v, c, stack, stride, disc_pos, leaf_tests);
}

template <bool STATS>
PT_DEV HitRec trace_lbvh(const LbvhView& v, int* stack, uint32_t stride, f3 o, f3 d, uint32_t& disc_pos, uint32_t& node_visits,
                         uint32_t& leaf_tests) {
    LbvhCursor c;
    cursor_begin(c, o, d);
    while (c.cur != kCursorDone) cursor_step<STATS>(v, c, stack, stride, disc_pos, node_visits, leaf_tests);
    HitRec h;
    h.t = c.best;
    h.idx = cursor_hit_index(v, c);
    return h;
}

// integrator.rs:136-156 closest_hit: hp = (o - c) + t*d; n = normalize(hp); pos = hp + c
PT_DEV void hit_geometry(float4 sph, f3 o, f3 d, float t, f3& pos, f3& nrm) {
    f3 c = mk3(sph.x, sph.y, sph.z);
    f3 oc = o - c;
    f3 hp = oc + t * d;
    nrm = normalize(hp);
    pos = hp + c;
}

// sampler.rs:8-18,43-53
PT_DEV f3 sample_hemisphere(f3 n, float r1, float r2) {
    float phi = kTwoPi * r1;
    float r = sqrt_cr(r2);
    float sn, cs;
    sincos_poly(phi, sn, cs);
    f3 local = mk3(r * cs, r * sn, 1.0f - r);
    f3 up = (__builtin_fabsf(n.z) < 0.999f) ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    f3 tangent = normalize(cross(up, n));
    f3 bitangent = cross(n, tangent);
    return local.x * tangent + local.y * bitangent + local.z * n;
}

// ExrImage::sample, exr.rs:13-24 (nearest texel).  Texels are stored padded to float4.
PT_DEV uint32_t texel_index(float f, uint32_t n) {
    // Rust: f.floor() as usize % n  (saturating cast: NaN / negative -> 0)
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_u32_f32 is that cast: it truncates toward zero (= floor for f >= 0), gives 0 for NaN and for negatives and
    // saturates above; the min keeps the host form's ceiling.  i >= n happens for u == 1 exactly (i == n) or never, so the
    // remainder sits behind a wave-uniform branch instead of ten predicated instructions in every lookup.
    uint32_t i;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(i) : "v"(f));
    i = i < 4294967040u ? i : 4294967040u;
    if (__builtin_expect(!wave_all(i < n), 0))
        if (i >= n) i = i % n;
    return i;
#else
    f = __builtin_floorf(f);
    uint32_t i = (f > 0.0f) ? (uint32_t)__builtin_fminf(f, 4294967040.0f) : 0u;
    if (i >= n) i = i % n;
    return i;
#endif
}
PT_DEV void sky_uv(f3 dir, float& u, float& v) {
    f3 d = normalize(dir);
    float cy = d.y;
    cy = (cy < -1.0f) ? -1.0f : cy;
    cy = (cy > 1.0f) ? 1.0f : cy;
    float theta = acos_poly(cy);
    float phi = atan2_poly(d.z, d.x);
#if PT_FAST_MATH
    // phi + PI is +0 or in [2^-23, 6.29]; theta is +0 or in [2^-13, 3.15] (acos_poly): inside div_const_seq's domain
    u = div_const_seq<1>(phi + kPi);
    v = div_const_seq<0>(theta);
#else
    u = (phi + kPi) / kTwoPi;
    v = theta / kPi;
#endif
}
PT_DEV f3 sky_nearest(const float4* __restrict__ sky, uint32_t sw, uint32_t sh, f3 dir) {
    float u, v;
    sky_uv(dir, u, v);
    uint32_t x = texel_index(u * (float)sw, sw);
    uint32_t y = texel_index(v * (float)sh, sh);
    float4 t = load16(sky, row_major(y, sw, x));
    return mk3(t.x, t.y, t.z);
}

// Bilinear equirect lookup (IR_SKY_BILINEAR; the north-star variant, NOT what exr.rs:20-23 does).  Texel centres sit
// at (i + 0.5) / n; columns wrap around in longitude, rows clamp at the poles; lerp(a, b, t) = a + (b - a) * t,
// first along x on both rows, then along y.  Same u, v as the nearest lookup.
PT_DEV int bilinear_cell(float f, float n, float& frac) {
    const float g = f * n - 0.5f;
    const float fl = __builtin_floorf(g);
    frac = g - fl;
    return (fl >= -1.0f) ? (int)__builtin_fminf(fl, 1073741824.0f) : -1;  // NaN -> -1
}
// ix mod w in [0, w).  A lookup's cell index lies in [-1, w - 1] (u in [0, 1]), so the wave takes the one-add form
// whenever every lane's index is within one period and the signed remainder (twenty instructions) otherwise.
PT_DEV int wrap_column(int ix, int w) {
    if (__builtin_expect(wave_all(ix >= -w && ix < w), 1)) return ix < 0 ? ix + w : ix;
    int x0 = ix % w;
    if (x0 < 0) x0 += w;
    return x0;
}
PT_DEV f3 sky_bilinear(const float4* __restrict__ sky, uint32_t sw, uint32_t sh, f3 dir) {
    float u, v, tx, ty;
    sky_uv(dir, u, v);
    const int ix = bilinear_cell(u, (float)sw, tx), iy = bilinear_cell(v, (float)sh, ty);
    const int w = (int)sw, h = (int)sh;
    const int x0 = wrap_column(ix, w);
    const int x1 = (x0 + 1 == w) ? 0 : x0 + 1;
    const int y0 = iy < 0 ? 0 : (iy > h - 1 ? h - 1 : iy);
    const int y1 = iy + 1 < 0 ? 0 : (iy + 1 > h - 1 ? h - 1 : iy + 1);
    const uint32_t r0 = row_major((uint32_t)y0, sw, 0u), r1 = row_major((uint32_t)y1, sw, 0u);
    const float4 a = load16(sky, r0 + (uint32_t)x0), b = load16(sky, r0 + (uint32_t)x1);
    const float4 c = load16(sky, r1 + (uint32_t)x0), d = load16(sky, r1 + (uint32_t)x1);
    const f3 top = mk3(a.x, a.y, a.z) + (mk3(b.x, b.y, b.z) - mk3(a.x, a.y, a.z)) * tx;
    const f3 bot = mk3(c.x, c.y, c.z) + (mk3(d.x, d.y, d.z) - mk3(c.x, c.y, c.z)) * tx;
    return top + (bot - top) * ty;
}

// The same lookup through the footprint table (pt_kernels.h RenderArgs::skyquad): record (iy + 1) * w + x0 holds the four
// texels (y0,x0) (y0,x1) (y1,x0) (y1,x1) the lookup with that cell reads, 64 bytes = one memory sector.  A sky too large
// for the caches costs a lookup one HBM access this way instead of the 2.2 the two image rows cost (measured on an
// 8192 x 4096 sky: 56 -> 118 Gsamples/s on C5); same texels, same arithmetic, same bits.
PT_DEV f3 sky_bilinear_quad(const float4* __restrict__ quad, uint32_t sw, uint32_t sh, f3 dir) {
    float u, v, tx, ty;
    sky_uv(dir, u, v);
    const int ix = bilinear_cell(u, (float)sw, tx), iy = bilinear_cell(v, (float)sh, ty);
    const int w = (int)sw, h = (int)sh;
    const int x0 = wrap_column(ix, w);
    const int row = iy + 1 < 0 ? 0 : (iy + 1 > h ? h : iy + 1);  // iy in [-1, h - 1] for any finite v; NaN gives -1
    const uint32_t rec = 4u * row_major((uint32_t)row, sw, (uint32_t)x0);  // < 2^28: the table is built for w * (h + 1) < 2^26
    const float4 a = load16(quad, rec), b = load16(quad, rec + 1u), c = load16(quad, rec + 2u), d = load16(quad, rec + 3u);
    const f3 top = mk3(a.x, a.y, a.z) + (mk3(b.x, b.y, b.z) - mk3(a.x, a.y, a.z)) * tx;
    const f3 bot = mk3(c.x, c.y, c.z) + (mk3(d.x, d.y, d.z) - mk3(c.x, c.y, c.z)) * tx;
    return top + (bot - top) * ty;
}

// ----------------------------------------------------------------------------------------------------
// Sky importance sampling (IR_SKY_SAMPLING_MIS; extension, the reference samples the BSDF only).
// The reference's hemisphere sample (sampler.rs:43-53) is (r cos, r sin, 1 - r), r = sqrt(r2), NOT normalised; its
// direction has solid-angle density 1 / (pi (sin t + cos t)^3) around the normal (t = polar angle), which equals
// L^3 / pi for L = |(r, 1 - r)|, and its path weight brdf cos / pdf is the albedo whatever the direction.  So a bounce
// estimates albedo * integral(density(w) sky(w) visible(w)): a second strategy that draws w from the sky's own
// distribution estimates the same integral, and the two are combined with the balance heuristic.  Plain IEEE
// operations in a fixed order (the oracle restates them).
// ----------------------------------------------------------------------------------------------------
struct SkyTables {
    const float* marginal;  // h + 1
    const float* cond;      // h x (w + 1)
    const float* pdf;       // h x w, see sky_distribution.h
    const uint32_t* marginal_guide;  // h + 1
    const uint32_t* cond_guide;      // h x (w + 1)
};
constexpr uint32_t kSkyDrawBase = 0x40000000u;  // draw indices of the sky strategy: kSkyDrawBase + 2 * bounce + {0, 1}

// largest i in [0, n) with cdf[i] <= x (cdf[0] == 0 <= x < 1 == cdf[n])
PT_DEV uint32_t cdf_find(const float* __restrict__ cdf, uint32_t n, float x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}
// The same index, found in a range narrowed by the guide (sky_distribution.h): with k = floor(x * n) the answer lies in
// [guide[k - 1], guide[k + 2]] — one bucket of slack on either side covers the rounding of x * n and of k / n — which for
// all but the darkest parts of an image is two or three entries instead of n.
PT_DEV uint32_t cdf_find_guided(const float* __restrict__ cdf, const uint32_t* __restrict__ guide, uint32_t n, float x) {
    uint32_t k = (uint32_t)(x * (float)n);
    k = k < n ? k : n - 1;
    uint32_t lo = guide[k > 0 ? k - 1 : 0];
    const uint32_t k2 = k + 2 < n ? k + 2 : n;
    uint32_t hi = guide[k2] + 1;  // cdf[hi] > x (hi <= n, cdf[n] == 1)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}
PT_DEV float cdf_frac(float x, float c0, float c1) {
    const float f = (x - c0) / (c1 - c0);
    return f > 0.99999994f ? 0.99999994f : f;
}
PT_DEV float bsdf_density_dir(float cos_n) {  // from the unit direction's cosine to the normal
    float s2 = 1.0f - cos_n * cos_n;
    if (s2 < 0.0f) s2 = 0.0f;
    const float sc = sqrt_cr(s2) + cos_n;
    float t = sc * sc;
    t = t * sc;
    t = kPi * t;
    return 1.0f / t;
}
PT_DEV float bsdf_density_r(float r) {  // from the sampler's radius
    const float q = 1.0f - r;
    const float l2 = r * r + q * q;
    return (l2 * sqrt_cr(l2)) / kPi;
}
PT_DEV float sky_density(const SkyTables& t, size_t texel, float sin_polar) {
    return sin_polar > 0.0f ? t.pdf[texel] / sin_polar : 0.0f;
}
// Direction drawn from the sky distribution: texel (ix, iy), a point inside it, unit direction ws, sin of its polar angle.
PT_DEV void sky_sample(const SkyTables& t, uint32_t sw, uint32_t sh, float x1, float x2, uint32_t& ix, uint32_t& iy, f3& ws,
                       float& sin_polar) {
    iy = cdf_find_guided(t.marginal, t.marginal_guide, sh, x1);
    const float fy = cdf_frac(x1, t.marginal[iy], t.marginal[iy + 1]);
    const float* row = t.cond + (size_t)iy * (sw + 1);
    ix = cdf_find_guided(row, t.cond_guide + (size_t)iy * (sw + 1), sw, x2);
    const float fx = cdf_frac(x2, row[ix], row[ix + 1]);
    const float u = ((float)ix + fx) / (float)sw;
    const float v = ((float)iy + fy) / (float)sh;
    const float theta = v * kPi, phi = u * kTwoPi;
    float st, ct, sp, cp;
    sincos_poly(theta, st, ct);
    sincos_poly(phi, sp, cp);
    ws = mk3(-(st * cp), ct, -(st * sp));
    sin_polar = st;
}
// The lookup of sky_nearest / sky_bilinear plus what the balance weight of a BSDF-sampled ray needs: the nearest texel
// (the sky distribution's cell) and the sine of the polar angle.
PT_DEV f3 sky_lookup_mis(const float4* __restrict__ sky, uint32_t sw, uint32_t sh, int bilinear, f3 dir, size_t& texel,
                         float& sin_polar) {
    const f3 d = normalize(dir);
    float cy = d.y;
    cy = (cy < -1.0f) ? -1.0f : cy;
    cy = (cy > 1.0f) ? 1.0f : cy;
    float s2 = 1.0f - cy * cy;
    if (s2 < 0.0f) s2 = 0.0f;
    sin_polar = sqrt_cr(s2);
    float u, v;
    sky_uv(dir, u, v);
    const uint32_t x = texel_index(u * (float)sw, sw);
    const uint32_t y = texel_index(v * (float)sh, sh);
    texel = (size_t)y * sw + x;
    if (bilinear) return sky_bilinear(sky, sw, sh, dir);
    const float4 t = sky[texel];
    return mk3(t.x, t.y, t.z);
}

// integrator.rs:66-70: cos = max(wi.n, 0); pdf = cos / PI; contribution *= (brdf * cos) / pdf.
// `fast_materials` (host, per scene): every brdf component is +0 or in [2^-30, 2^30], so with cos in [2^-60, 2^60]
// pdf is in the mid range, brdf*cos is +0 or >= 2^-90 and the quotient (~ albedo) is +0 or in [2^-29, 2^32].
PT_DEV f3 scatter_weight(f3 brdf, float cos_theta, bool fast_materials) {
#if PT_FAST_MATH
    if (__builtin_expect(fast_materials && wave_all(in_pow2_range<67, 187>(cos_theta)), 1)) {
        const float pdf = div_const_seq<0>(cos_theta);
        const float y = rcp_seq(pdf);
        const f3 n = brdf * cos_theta;
        return mk3(div_seq(n.x, pdf, y), div_seq(n.y, pdf, y), div_seq(n.z, pdf, y));
    }
#endif
    const float pdf = cos_theta / kPi;
    return brdf * cos_theta / pdf;
}

// integrator.rs:72-78: contribution /= p with p = max component.  `fast_materials` also promises non-negative
// albedo, hence contribution components >= +0.  In range: p in [2^-32, 2^32) and every component +0 or >= 2^-68,
// so each quotient is +0 or in [2^-100, 1].  (bits - 1 maps +0 above every positive float: one unsigned min3.)
PT_DEV f3 roulette_divide(f3 c, float p, bool fast_materials) {
#if PT_FAST_MATH
    const uint32_t ux = f32_bits(c.x) - 1u, uy = f32_bits(c.y) - 1u, uz = f32_bits(c.z) - 1u;
    const uint32_t uxy = ux < uy ? ux : uy;
    const uint32_t lowest = uxy < uz ? uxy : uz;
    if (__builtin_expect(fast_materials && wave_all(in_pow2_range<95, 159>(p) && lowest >= (59u << 23) - 1u), 1)) {
        const float y = rcp_seq(p);
        return mk3(div_seq(c.x, p, y), div_seq(c.y, p, y), div_seq(c.z, p, y));
    }
#endif
    return c / p;
}

}  // namespace pt
