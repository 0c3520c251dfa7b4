// oracle.cpp — CPU restatement of the InsploRay (SUGAT-BAGDE/Path_Tracer) render path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (path_tracer_amd/, include/) may
// include, link, load or call this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker / the CPU baseline.
//
// What it restates (all citations relative to /root/reference/core-engine/src):
//   cameras/pinhole_camera.rs:59-61,63-69,78-98,110-129   camera state + get_ray
//   integrator.rs:32-93                                    compute_incomming_radience
//   integrator.rs:95-134                                   trace_ray (linear closest hit)
//   integrator.rs:136-164                                  closest_hit / ray_miss
//   sampler.rs:8-18,43-53                                  hemisphere sample + tangent frame
//   file_formats/exr.rs:13-24                              equirect nearest-texel lookup
//   accumulators/tile_buffer.rs:24-28, full_buffer.rs:49-65,96-125   accumulate / merge / resolve
//   utils.rs:4-9                                           ARGB pack
//   renderer.rs:95-162                                     64x64 tile loop, worker pool, merge
//
// The reference is Rust and cannot be built in this environment (no cargo/rustc, no glam /
// rand / exr crate sources).  Vector arithmetic therefore follows the published semantics of
// glam 0.30 (scalar Vec3: dot = x*x + y*y + z*z left to right, cross as in glam's vec3.rs,
// normalize = v * (1.0 / sqrt(dot)), Mat4 * Vec4 = ((X*x + Y*y) + Z*z) + W*w, no FMA),
// compiled here with -ffp-contract=off and without fast-math.
//
// PARITY PIN STATUS: pinned by the reference's own two camera known-answer tests
// (pinhole_camera.rs:138-160 fov, :162-239 Blender matrix + 5 vertices; see
// tests/test_oracle_camera.py).  Everything else (integrator, sampler, sky lookup,
// accumulator, tone map) has no reference-owned test or fixture => "parity unpinned" by the
// reference for those parts; they are cross-checked by closed-form cases in tests/ only.
//
// RNG: the reference uses rand::ThreadRng (OS seeded, not reproducible, sampler.rs:4-6,21-28),
// so no random sequence is canonical.  The oracle offers
//   rng_mode 0: the counter-based generator the HIP path uses (defined again here, on purpose
//               not shared with the product sources), f = (u32 >> 8) * 2^-24 like rand's
//               StandardUniform f32;
//   rng_mode 1: an unrelated sequential generator (PCG32) for independent-stream statistics.
// math_mode 0 uses libm (sinf/cosf/acosf/atan2f as Rust's f32 methods do);
// math_mode 1 uses the polynomial sincos/acos/atan2 the HIP kernels use (restated here), so
// that same-stream comparisons can be bit-exact.

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr float kPi = 3.14159265358979323846264338327950288f;  // std::f32::consts::PI
constexpr float kEps = 1.1920929e-07f;                         // f32::EPSILON
constexpr float kF32Max = 3.40282347e+38f;                     // f32::MAX

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// glam vec3.rs cross()
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// glam: normalize = self * self.length_recip(); length_recip = 1.0 / sqrt(dot(self,self))
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

// ---------------------------------------------------------------------------------------------
// math_mode 1: restatement of the device polynomial transcendentals (path_tracer_amd/csrc/pt_device.h, "Polynomial transcendentals").
// Plain f32 + - * / only, no FMA: bit-reproducible between x86 SSE and gfx950 VALU.
// sin/cos and acos use the classic single-precision Cephes minimax sets; atan2 is a one-divide form fitted in pt_device.h.
// ---------------------------------------------------------------------------------------------
inline void poly_sincos(float x, float* s_out, float* c_out) {
    // valid for 0 <= x < 8*pi/4 .. (we only ever pass phi in [0, 2*pi])
    const float FOPI = 1.27323954473516f;  // 4/pi
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    int j = (int)(FOPI * x);
    j = (j + 1) & ~1;  // round up to even octant
    float y = (float)j;
    float r = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
               0.5f * z + 1.0f;
    int q = (j >> 1) & 3;  // quadrant
    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? sp : cp;
    if (q & 2) s = -s;
    if ((q == 1) || (q == 2)) c = -c;
    *s_out = s;
    *c_out = c;
}
inline float poly_asin_core(float a) {  // |a| <= 0.5
    float z = a * a;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
            1.6666752422e-1f) * z * a + a;
}
inline float poly_acos(float x) {  // x in [-1,1]; NaN propagates
    const float PIO2 = 1.5707963267948966f;
    if (x > 0.5f) return 2.0f * poly_asin_core(std::sqrt(0.5f * (1.0f - x)));
    if (x < -0.5f) return kPi - 2.0f * poly_asin_core(std::sqrt(0.5f * (1.0f + x)));
    return PIO2 - poly_asin_core(x);
}
inline float poly_atan2(float y, float x) {
    // one divide: q = min(|x|, |y|) / max(|x|, |y|), atan(q) = q * P(q * q) with the degree-7 P of the device code
    // (pt_device.h atan2_poly: fitted there, 1.7e-7 rad), then the octant; the sign of zero is ignored (0, 0 -> 0)
    const float ax = std::fabs(x), ay = std::fabs(y);
    const bool steep = ay > ax;
    const float mx = steep ? ay : ax, mn = steep ? ax : ay;
    const float q = mn / mx;
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
    if (mx == 0.0f) a = 0.0f;
    if (steep) a = 1.5707963267948966f - a;
    if (x < 0.0f) a = kPi - a;
    if (y < 0.0f) a = -a;
    return a;
}

// ---------------------------------------------------------------------------------------------
// RNG
// ---------------------------------------------------------------------------------------------
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
inline float u32_to_unit(uint32_t u) { return (float)(u >> 8) * 5.9604644775390625e-08f; }  // 2^-24

struct Pcg32 {
    uint64_t state, inc;
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
    }
    void seed(uint64_t s, uint64_t seq) {
        state = 0;
        inc = (seq << 1u) | 1u;
        next();
        state += s;
        next();
    }
};

struct Sampler {
    int mode;             // 0 counter-based, 1 pcg32
    uint32_t sample_key;  // mode 0
    uint32_t draw;        // mode 0: index of the next draw consumed sequentially (mode 1 only counts)
    Pcg32 pcg;
    float at(uint32_t k) {  // mode 0 only: draw k of this (pixel,sample) stream
        return u32_to_unit(mix32(sample_key + k * 0x85ebca6bu));
    }
};

// stream key = f(seed, pixel, sample): see path_tracer_amd/csrc/pt_device.h ("Counter-based RNG") for the product twin
inline uint32_t pixel_key(uint64_t seed, uint32_t pixel) { return mix32(pixel ^ (uint32_t)(seed & 0xffffffffu)); }
inline uint32_t sample_key(uint64_t seed, uint32_t pkey, uint32_t sample) {
    return mix32(pkey + sample * 0x9e3779b9u + (uint32_t)(seed >> 32));
}

// ---------------------------------------------------------------------------------------------
// Scene / camera views
// ---------------------------------------------------------------------------------------------
struct Scene {
    uint32_t n_spheres;
    const float* xyzr;        // 4 floats per sphere
    const int32_t* material;  // per sphere
    uint32_t n_materials;
    const float* mat9;  // albedo3, roughness, metalic, emission3, emissive_power  (scene.rs:5-12)
    float sky[3];
    const float* skybox;  // rgb, row-major, row 0 first (exr.rs:38) or nullptr
    uint32_t sky_w, sky_h;
    const struct SkyDist* dist = nullptr;  // sky importance sampling tables (extension, see SkyDist) or nullptr
};

struct Camera {
    float m[16];  // local_to_world, column-major (glam Mat4)
    float position[3];
    float fov;
    float aspect;
    uint32_t w, h;
};

struct Params {
    uint32_t bounces, rr_start;
    uint64_t seed;
    int rng_mode, math_mode;
};

struct Ray {
    V3 o, d;
};

// pinhole_camera.rs:78-98
inline Ray get_ray(const Camera& cam, float tan_half_fov, uint32_t x, uint32_t y) {
    float vx = ((float)x + 0.5f) / (float)cam.w;
    float vy = ((float)y + 0.5f) / (float)cam.h;
    vx = (vx * 2.0f - 1.0f) * tan_half_fov;
    vy = (vy * 2.0f - 1.0f) * tan_half_fov;
    vx *= cam.aspect;
    V3 rd = v3(vx, vy, -1.0f);
    // glam Mat4::transform_vector3: res = X*x; res = Y*y + res; res = Z*z + res
    V3 X = v3(cam.m[0], cam.m[1], cam.m[2]), Y = v3(cam.m[4], cam.m[5], cam.m[6]), Z = v3(cam.m[8], cam.m[9], cam.m[10]);
    V3 res = X * rd.x;
    res = Y * rd.y + res;
    res = Z * rd.z + res;
    return Ray{v3(cam.position[0], cam.position[1], cam.position[2]), normalize(res)};
}

struct Hit {
    float t;
    int32_t index;  // -1 = miss
    V3 pos, normal;
};

struct Stats {
    uint64_t samples, traces, tests, disc_pos, hits;
};

// integrator.rs:95-134 + 136-164
inline Hit trace_ray(const Scene& sc, const Ray& ray, Stats* st) {
    float hit_distance = kF32Max;
    int32_t closest = -1;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const float* s = sc.xyzr + 4 * i;
        V3 origin = ray.o - v3(s[0], s[1], s[2]);
        float a = dot(ray.d, ray.d);
        float b = 2.0f * dot(ray.d, origin);
        float c = dot(origin, origin) - s[3] * s[3];
        float discriminant = b * b - 4.0f * a * c;
        if (st) st->tests++;
        if (discriminant < 0.0f) continue;
        if (st) st->disc_pos++;
        float sqrt_d = std::sqrt(discriminant);
        float closest_t = (-b - sqrt_d) / (2.0f * a);
        if (closest_t > 0.0f && closest_t < hit_distance) {
            closest = (int32_t)i;
            hit_distance = closest_t;
        }
    }
    Hit h;
    if (closest >= 0) {
        const float* s = sc.xyzr + 4 * closest;
        V3 center = v3(s[0], s[1], s[2]);
        V3 origin = ray.o - center;
        V3 hit_point = origin + hit_distance * ray.d;
        h.t = hit_distance;
        h.index = closest;
        h.normal = normalize(hit_point);
        h.pos = hit_point + center;
    } else {
        h.t = -1.0f;
        h.index = -1;
        h.pos = v3(0, 0, 0);
        h.normal = v3(0, 0, 0);
    }
    return h;
}

// exr.rs:13-24 (nearest texel).  Rust `as usize` saturates: negative/NaN -> 0.
inline size_t f32_to_usize(float f) {
    if (!(f > 0.0f)) return 0;  // NaN, negative, zero
    if (f >= 1.8446744073709552e19f) return SIZE_MAX;
    return (size_t)f;
}
// Extension switch (not in the reference, which is nearest-only): 1 = bilinear lookup, the build's IR_SKY_BILINEAR.
// Process-global because it is test infrastructure; set through orc_set_sky_filter before rendering.
static int g_sky_filter = 0;
// Samples per accumulation block of a multi-sample orc_render call (the build's ir_set_sample_block, default 128).
static uint32_t g_sample_block = 128;
inline int bilinear_cell(float f, float n, float* frac) {
    float g = f * n - 0.5f;
    float fl = std::floor(g);
    *frac = g - fl;
    return (fl >= -1.0f) ? (int)std::fmin(fl, 1073741824.0f) : -1;
}
inline V3 sky_lookup(const Scene& sc, V3 dir, int math_mode, size_t* texel = nullptr, float* cos_polar = nullptr) {
    V3 d = normalize(dir);
    float cy = d.y;
    if (cy < -1.0f) cy = -1.0f;  // f32::clamp: NaN stays NaN
    if (cy > 1.0f) cy = 1.0f;
    if (cos_polar) *cos_polar = cy;
    float theta = math_mode ? poly_acos(cy) : std::acos(cy);
    float phi = math_mode ? poly_atan2(d.z, d.x) : std::atan2(d.z, d.x);
    float u = (phi + kPi) / (2.0f * kPi);
    float v = theta / kPi;
    if (texel)
        *texel = (f32_to_usize(std::floor(v * (float)sc.sky_h)) % sc.sky_h) * sc.sky_w +
                 f32_to_usize(std::floor(u * (float)sc.sky_w)) % sc.sky_w;
    if (g_sky_filter == 1) {
        float tx, ty;
        int ix = bilinear_cell(u, (float)sc.sky_w, &tx), iy = bilinear_cell(v, (float)sc.sky_h, &ty);
        int w = (int)sc.sky_w, h = (int)sc.sky_h;
        int x0 = ix % w;
        if (x0 < 0) x0 += w;
        int x1 = (x0 + 1 == w) ? 0 : x0 + 1;
        int y0 = iy < 0 ? 0 : (iy > h - 1 ? h - 1 : iy);
        int y1 = iy + 1 < 0 ? 0 : (iy + 1 > h - 1 ? h - 1 : iy + 1);
        auto tex = [&](int x, int y) { const float* p = sc.skybox + 3 * ((size_t)y * sc.sky_w + x); return v3(p[0], p[1], p[2]); };
        V3 a = tex(x0, y0), b = tex(x1, y0), c = tex(x0, y1), e = tex(x1, y1);
        V3 top = a + (b - a) * tx;
        V3 bot = c + (e - c) * tx;
        return top + (bot - top) * ty;
    }
    size_t x = f32_to_usize(std::floor(u * (float)sc.sky_w)) % sc.sky_w;
    size_t y = f32_to_usize(std::floor(v * (float)sc.sky_h)) % sc.sky_h;
    const float* p = sc.skybox + 3 * (y * sc.sky_w + x);
    return v3(p[0], p[1], p[2]);
}

// ---------------------------------------------------------------------------------------------
// Sky importance sampling (extension; the reference samples the BSDF only: integrator.rs:64-70,84-89).  Restates the
// build's IR_SKY_SAMPLING_MIS (path_tracer_amd/csrc/sky_distribution.cpp + pt_device.h) for parity:
//  * texel weight = luminance (0.2126, 0.7152, 0.0722; non-positive or non-finite -> 0) * sin(pi (y + 0.5) / h) in double;
//    row-conditional and marginal CDFs rounded to f32 (uniform where a sum is not positive), pdf per texel from the f32
//    CDF differences: (dM * dC) * w h / (2 pi^2), to be divided by sin(theta) of the direction;
//  * at a hit whose bounce ray will be traced, a second direction is drawn from that distribution (draw indices
//    0x40000000 + 2 bounce + {0,1}), traced as a shadow ray, and weighted against the reference's own sampler with the
//    balance heuristic.  The reference's sampler (sampler.rs:43-53: (r cos, r sin, 1 - r), not normalised) has solid-angle
//    density 1 / (pi (sin t + cos t)^3) around the normal, which is L^3 / pi for L = |(r, 1 - r)|, and its weight
//    brdf cos / pdf is the albedo for every direction, so both strategies estimate albedo * integral(density * sky).
// ---------------------------------------------------------------------------------------------
struct SkyDist {
    uint32_t w = 0, h = 0;
    std::vector<float> marginal, cond, pdf;
};
static int g_sky_sampling = 0;

inline void build_sky_dist(const float* rgb, uint32_t w, uint32_t h, SkyDist& out) {
    const double pi = 3.14159265358979323846;
    out.w = w;
    out.h = h;
    out.marginal.assign((size_t)h + 1, 0.0f);
    out.cond.assign((size_t)h * (w + 1), 0.0f);
    out.pdf.assign((size_t)h * w, 0.0f);
    std::vector<double> prefix((size_t)h * (w + 1), 0.0), rows(h, 0.0);
    double total = 0.0;
    for (int pass = 0; pass < 2; ++pass) {  // pass 1: uniform in solid angle, only if the image has no positive energy
        total = 0.0;
        for (uint32_t y = 0; y < h; ++y) {
            const double sy = std::sin(pi * ((double)y + 0.5) / (double)h);
            double* pr = &prefix[(size_t)y * (w + 1)];
            double acc = 0.0;
            for (uint32_t x = 0; x < w; ++x) {
                const float* t = rgb + 3 * ((size_t)y * w + x);
                double lum = (0.2126 * (double)t[0] + 0.7152 * (double)t[1]) + 0.0722 * (double)t[2];
                if (!(lum > 0.0) || std::isinf(lum)) lum = 0.0;
                if (pass == 1) lum = 1.0;
                acc += lum * sy;
                pr[x + 1] = acc;
            }
            rows[y] = acc;
            total += acc;
        }
        if (total > 0.0 && !std::isinf(total)) break;
    }
    double macc = 0.0;
    for (uint32_t y = 0; y < h; ++y) {
        float* c = &out.cond[(size_t)y * (w + 1)];
        const double* pr = &prefix[(size_t)y * (w + 1)];
        for (uint32_t x = 0; x < w; ++x)
            c[x + 1] = rows[y] > 0.0 ? (float)(pr[x + 1] / rows[y]) : (float)((double)(x + 1) / (double)w);
        c[w] = 1.0f;
        macc += rows[y];
        out.marginal[y + 1] = (float)(macc / total);
    }
    out.marginal[h] = 1.0f;
    const double scale = (double)w * (double)h / (2.0 * pi * pi);
    for (uint32_t y = 0; y < h; ++y) {
        const double dm = (double)out.marginal[y + 1] - (double)out.marginal[y];
        const float* c = &out.cond[(size_t)y * (w + 1)];
        for (uint32_t x = 0; x < w; ++x)
            out.pdf[(size_t)y * w + x] = (float)((dm * ((double)c[x + 1] - (double)c[x])) * scale);
    }
}

// largest i in [0, n) with cdf[i] <= x (cdf[0] == 0, cdf[n] == 1 > x)
inline uint32_t cdf_find(const float* cdf, uint32_t n, float x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}
inline float cdf_frac(float x, float c0, float c1) {
    float f = (x - c0) / (c1 - c0);
    return f > 0.99999994f ? 0.99999994f : f;
}
// density of the reference's hemisphere sampler per solid angle, from the unit direction's cosine to the normal
inline float bsdf_density_dir(float c) {
    float s2 = 1.0f - c * c;
    if (s2 < 0.0f) s2 = 0.0f;
    float sc = std::sqrt(s2) + c;
    float t = sc * sc;
    t = t * sc;
    t = kPi * t;
    return 1.0f / t;
}
// the same density from the sampler's radius r = sqrt(r2)
inline float bsdf_density_r(float r) {
    float q = 1.0f - r;
    float l2 = r * r + q * q;
    return (l2 * std::sqrt(l2)) / kPi;
}
inline float sky_density(const SkyDist& ds, size_t texel, float sin_polar) {
    return sin_polar > 0.0f ? ds.pdf[texel] / sin_polar : 0.0f;
}

// sampler.rs:8-18,43-53
inline V3 sample_hemisphere(V3 normal, float r1, float r2, int math_mode) {
    float phi = 2.0f * kPi * r1;
    float r = std::sqrt(r2);
    float sn, cs;
    if (math_mode) {
        poly_sincos(phi, &sn, &cs);
    } else {
        cs = std::cos(phi);
        sn = std::sin(phi);
    }
    V3 local = v3(r * cs, r * sn, 1.0f - r);
    V3 up = (std::fabs(normal.z) < 0.999f) ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    V3 tangent = normalize(cross(up, normal));
    V3 bitangent = cross(normal, tangent);
    return local.x * tangent + local.y * bitangent + local.z * normal;
}

inline float next_f32(Sampler& s) {
    if (s.mode == 0) {
        return s.at(s.draw++);
    }
    s.draw++;
    return u32_to_unit(s.pcg.next());
}

// integrator.rs:32-93.  Returns rgb; alpha is the constant 1.0 (:92).
inline V3 radiance(const Scene& sc, const Camera& cam, float tan_half_fov, const Params& pr, uint32_t x, uint32_t y,
                   Sampler& sampler, Stats* st) {
    static const float default_mat[9] = {1.0f, 1.0f, 1.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // integrator.rs:8-14
    Ray ray = get_ray(cam, tan_half_fov, x, y);
    V3 light = v3(0, 0, 0);
    V3 contribution = v3(1, 1, 1);
    float ray_density = 0.0f;  // sky sampling: density of the sampler that produced ray.d (bounce > 0)
    if (st) st->samples++;
    for (uint32_t bounce = 0; bounce < pr.bounces; ++bounce) {
        if (sampler.mode == 0) sampler.draw = 3 * bounce;  // draw index k = 3*bounce + {0,1,2}
        if (st) st->traces++;
        Hit payload = trace_ray(sc, ray, st);
        if (payload.index >= 0) {
            if (st) st->hits++;
            int32_t mid = sc.material[payload.index];
            const float* m = (mid < 0 || (uint32_t)mid >= sc.n_materials) ? default_mat : sc.mat9 + 9 * mid;
            V3 albedo = v3(m[0], m[1], m[2]);
            V3 emission = v3(m[5], m[6], m[7]);
            float power = m[8];
            light = light + emission * power * contribution;
            V3 normal = payload.normal;
            float r1 = next_f32(sampler);
            float r2 = next_f32(sampler);
            V3 wi = sample_hemisphere(normal, r1, r2, pr.math_mode);
            float cos_theta = std::fmax(dot(wi, normal), 0.0f);
            float pdf = cos_theta / kPi;
            V3 brdf = albedo / kPi;
            contribution = contribution * (brdf * cos_theta / pdf);
            if (bounce >= pr.rr_start) {
                float p = std::fmax(contribution.x, std::fmax(contribution.y, contribution.z));
                if (next_f32(sampler) > p) break;
                contribution = contribution / p;
            }
            ray.o = payload.pos + payload.normal * kEps;
            ray.d = wi;
            if (sc.dist && bounce + 1 < pr.bounces) {
                const SkyDist& ds = *sc.dist;
                ray_density = bsdf_density_r(std::sqrt(r2));
                float x1, x2;
                if (sampler.mode == 0) {
                    x1 = sampler.at(0x40000000u + 2 * bounce);
                    x2 = sampler.at(0x40000000u + 2 * bounce + 1);
                } else {
                    x1 = next_f32(sampler);
                    x2 = next_f32(sampler);
                }
                const uint32_t iy = cdf_find(ds.marginal.data(), ds.h, x1);
                const float fy = cdf_frac(x1, ds.marginal[iy], ds.marginal[iy + 1]);
                const float* crow = ds.cond.data() + (size_t)iy * (ds.w + 1);
                const uint32_t ix = cdf_find(crow, ds.w, x2);
                const float fx = cdf_frac(x2, crow[ix], crow[ix + 1]);
                const float u = ((float)ix + fx) / (float)ds.w;
                const float v = ((float)iy + fy) / (float)ds.h;
                const float theta = v * kPi, phi = u * (2.0f * kPi);
                float st_, ct_, sp_, cp_;
                if (pr.math_mode) {
                    poly_sincos(theta, &st_, &ct_);
                    poly_sincos(phi, &sp_, &cp_);
                } else {
                    st_ = std::sin(theta); ct_ = std::cos(theta);
                    sp_ = std::sin(phi); cp_ = std::cos(phi);
                }
                const V3 ws = v3(-(st_ * cp_), ct_, -(st_ * sp_));
                const float c = dot(ws, normal);
                if (c > 0.0f && st_ > 0.0f) {
                    const size_t texel = (size_t)iy * ds.w + ix;
                    const float pe = sky_density(ds, texel, st_);
                    const float pa = bsdf_density_dir(c);
                    const float wgt = pa / (pe + pa);
                    if (st) st->traces++;
                    Hit shadow = trace_ray(sc, Ray{ray.o, ws}, st);
                    if (shadow.index < 0) {
                        V3 sky;
                        if (g_sky_filter == 1) {
                            sky = sky_lookup(sc, ws, pr.math_mode);
                        } else {
                            const float* t = sc.skybox + 3 * texel;
                            sky = v3(t[0], t[1], t[2]);
                        }
                        light = light + (sky * contribution) * wgt;
                    }
                }
            }
        } else {
            if (sc.dist && bounce > 0) {
                size_t texel;
                float cy;
                V3 sky = sky_lookup(sc, ray.d, pr.math_mode, &texel, &cy);
                float s2 = 1.0f - cy * cy;
                if (s2 < 0.0f) s2 = 0.0f;
                const float pe = sky_density(*sc.dist, texel, std::sqrt(s2));
                const float wgt = ray_density / (ray_density + pe);
                light = light + (sky * contribution) * wgt;
            } else {
                V3 sky = sc.skybox ? sky_lookup(sc, ray.d, pr.math_mode) : v3(sc.sky[0], sc.sky[1], sc.sky[2]);
                light = light + sky * contribution;
            }
            break;
        }
    }
    return light;
}

void init_sampler(Sampler& s, const Params& pr, uint32_t pixel, uint32_t sample) {
    s.mode = pr.rng_mode;
    s.draw = 0;
    if (pr.rng_mode == 0) {
        s.sample_key = sample_key(pr.seed, pixel_key(pr.seed, pixel), sample);
    } else {
        s.pcg.seed(pr.seed ^ (0x9e3779b97f4a7c15ULL * (uint64_t)(sample + 1)), (uint64_t)pixel);
    }
}

}  // namespace

// =============================================================================================
// C ABI for ctypes (tests) and bench.py's cpu_baseline leg
// =============================================================================================
extern "C" {

// pinhole_camera.rs:59-69,110-129 — host-side camera state from (position, euler rotation, focal, sensor, size)
// out_m16: column-major local_to_world; out_dirs: forward, up, right (3 floats each)
void orc_pinhole_compute(const float position[3], const float rotation[3], float focal_length, float sensor_size,
                         uint32_t width, uint32_t height, float out_m16[16], float* out_fov, float* out_aspect,
                         float out_dirs[9]) {
    float fov = 2.0f * std::atan(sensor_size / (2.0f * focal_length));
    float aspect = (float)width / (float)height;
    auto ident = [](float* m) {
        for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    };
    float rx[16], ry[16], rz[16], tr[16];
    ident(rx); ident(ry); ident(rz); ident(tr);
    float s, c;
    s = std::sin(rotation[0]); c = std::cos(rotation[0]);  // glam from_rotation_x
    rx[5] = c; rx[6] = s; rx[9] = -s; rx[10] = c;
    s = std::sin(rotation[1]); c = std::cos(rotation[1]);  // from_rotation_y
    ry[0] = c; ry[2] = -s; ry[8] = s; ry[10] = c;
    s = std::sin(rotation[2]); c = std::cos(rotation[2]);  // from_rotation_z
    rz[0] = c; rz[1] = s; rz[4] = -s; rz[5] = c;
    tr[12] = position[0]; tr[13] = position[1]; tr[14] = position[2];
    // glam Mat4 * Mat4: column j of the product = A.mul_vec4(B.col(j)) = ((A.x*b.x + A.y*b.y) + A.z*b.z) + A.w*b.w
    auto matmul = [](const float* A, const float* B, float* out) {
        float tmp[16];
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i)
                tmp[4 * j + i] = ((A[i] * B[4 * j] + A[4 + i] * B[4 * j + 1]) + A[8 + i] * B[4 * j + 2]) + A[12 + i] * B[4 * j + 3];
        std::memcpy(out, tmp, sizeof(tmp));
    };
    float rzy[16], rot[16];
    matmul(rz, ry, rzy);
    matmul(rzy, rx, rot);  // (Rz*Ry)*Rx
    matmul(tr, rot, out_m16);
    *out_fov = fov;
    *out_aspect = aspect;
    if (out_dirs) {
        // forward = transform_vector3((0,0,-1)); up = y_axis.xyz; right = x_axis.xyz
        V3 X = v3(out_m16[0], out_m16[1], out_m16[2]), Y = v3(out_m16[4], out_m16[5], out_m16[6]),
           Z = v3(out_m16[8], out_m16[9], out_m16[10]);
        V3 res = X * 0.0f;
        res = Y * 0.0f + res;
        res = Z * -1.0f + res;
        out_dirs[0] = res.x; out_dirs[1] = res.y; out_dirs[2] = res.z;
        out_dirs[3] = Y.x; out_dirs[4] = Y.y; out_dirs[5] = Y.z;
        out_dirs[6] = X.x; out_dirs[7] = X.y; out_dirs[8] = X.z;
    }
}

// glam Mat4::transform_point3: res = X*x; res = Y*y + res; res = Z*z + res; res = W + res
void orc_transform_point3(const float m[16], const float p[3], float out[3]) {
    for (int i = 0; i < 3; ++i) {
        float res = m[i] * p[0];
        res = m[4 + i] * p[1] + res;
        res = m[8 + i] * p[2] + res;
        res = m[12 + i] + res;
        out[i] = res;
    }
}

static Camera make_cam(const float m16[16], const float pos[3], float fov, float aspect, uint32_t w, uint32_t h) {
    Camera c;
    std::memcpy(c.m, m16, sizeof(c.m));
    std::memcpy(c.position, pos, sizeof(c.position));
    c.fov = fov;
    c.aspect = aspect;
    c.w = w;
    c.h = h;
    return c;
}

float orc_tan_half_fov(float fov) { return std::tan(fov / 2.0f); }  // pinhole_camera.rs:86

// Primary rays: out = w*h*6 floats (origin, direction)
void orc_primary_rays(const float m16[16], const float pos[3], float fov, float aspect, uint32_t w, uint32_t h,
                      float* out) {
    Camera cam = make_cam(m16, pos, fov, aspect, w, h);
    float t = std::tan(cam.fov / 2.0f);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            Ray r = get_ray(cam, t, x, y);
            float* o = out + 6 * ((size_t)y * w + x);
            o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
        }
}

// Primary closest-hit AOV: index per pixel (-1 = miss) and optional t
void orc_primary_hits(const float m16[16], const float pos[3], float fov, float aspect, uint32_t w, uint32_t h,
                      const float* xyzr, uint32_t n_spheres, int32_t* out_index, float* out_t) {
    Camera cam = make_cam(m16, pos, fov, aspect, w, h);
    float t = std::tan(cam.fov / 2.0f);
    Scene sc{};
    sc.n_spheres = n_spheres;
    sc.xyzr = xyzr;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            Ray r = get_ray(cam, t, x, y);
            Hit hit = trace_ray(sc, r, nullptr);
            out_index[(size_t)y * w + x] = hit.index;
            if (out_t) out_t[(size_t)y * w + x] = hit.t;
        }
}

// Generic single-ray closest hit (for closed-form tests): returns index, fills t/pos/normal
int32_t orc_trace(const float* xyzr, uint32_t n_spheres, const float o[3], const float d[3], float* out_t,
                  float out_pos[3], float out_n[3]) {
    Scene sc{};
    sc.n_spheres = n_spheres;
    sc.xyzr = xyzr;
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    Hit h = trace_ray(sc, r, nullptr);
    *out_t = h.t;
    out_pos[0] = h.pos.x; out_pos[1] = h.pos.y; out_pos[2] = h.pos.z;
    out_n[0] = h.normal.x; out_n[1] = h.normal.y; out_n[2] = h.normal.z;
    return h.index;
}

void orc_sample_hemisphere(const float n[3], float r1, float r2, int math_mode, float out[3]) {
    V3 w = sample_hemisphere(v3(n[0], n[1], n[2]), r1, r2, math_mode);
    out[0] = w.x; out[1] = w.y; out[2] = w.z;
}

void orc_set_sky_filter(int filter) { g_sky_filter = filter; }
void orc_set_sample_block(uint32_t n) { g_sample_block = n ? n : 1; }
// 0 = BSDF sampling only (the reference), 1 = sky importance sampling with MIS (the build's IR_SKY_SAMPLING_MIS)
void orc_set_sky_sampling(int mode) { g_sky_sampling = mode; }
// the tables themselves, for table-level parity tests: marginal h+1, cond h*(w+1), pdf h*w floats
void orc_sky_distribution(const float* skybox, uint32_t sw, uint32_t sh, float* marginal, float* cond, float* pdf) {
    SkyDist d;
    build_sky_dist(skybox, sw, sh, d);
    std::memcpy(marginal, d.marginal.data(), d.marginal.size() * sizeof(float));
    std::memcpy(cond, d.cond.data(), d.cond.size() * sizeof(float));
    std::memcpy(pdf, d.pdf.data(), d.pdf.size() * sizeof(float));
}
void orc_sky_lookup(const float* skybox, uint32_t sw, uint32_t sh, const float d[3], int math_mode, float out[3]) {
    Scene sc{};
    sc.skybox = skybox;
    sc.sky_w = sw;
    sc.sky_h = sh;
    V3 c = sky_lookup(sc, v3(d[0], d[1], d[2]), math_mode);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

// counter-based RNG known answers
float orc_rng_draw(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k) {
    Sampler s;
    s.mode = 0;
    s.sample_key = sample_key(seed, pixel_key(seed, pixel), sample);
    return s.at(k);
}
uint32_t orc_rng_mix32(uint32_t x) { return mix32(x); }

void orc_poly_sincos(float x, float* s, float* c) { poly_sincos(x, s, c); }
float orc_poly_acos(float x) { return poly_acos(x); }
float orc_poly_atan2(float y, float x) { return poly_atan2(y, x); }

// full_buffer.rs:49-65 + utils.rs:4-9
static inline uint32_t f32_to_u32_sat(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
void orc_resolve_argb(const float* sum_rgba, const uint32_t* counts, size_t n, uint32_t* out) {
    const float inv_gamma = 1.0f / 2.2f;
    for (size_t i = 0; i < n; ++i) {
        uint32_t samples = counts[i] > 1 ? counts[i] : 1;
        float ch[4];
        for (int k = 0; k < 4; ++k) {
            float a = sum_rgba[4 * i + k] / (float)samples;
            a = a / (a + 1.0f);
            a = std::pow(a, inv_gamma);
            // glam Vec4::clamp = max(min).min(max); SSE max/min return the second operand on NaN
            a = (a > 0.0f) ? a : 0.0f;
            a = (a < 1.0f) ? a : 1.0f;
            ch[k] = a;
        }
        out[i] = (f32_to_u32_sat(ch[3] * 255.0f) << 24) | (f32_to_u32_sat(ch[0] * 255.0f) << 16) |
                 (f32_to_u32_sat(ch[1] * 255.0f) << 8) | f32_to_u32_sat(ch[2] * 255.0f);
    }
}

// The render entry: renderer.rs:95-162 restated.  One reference render() call == 1 sample per pixel;
// `spp` consecutive calls are folded here (pixel-outer within a tile, per-pixel sum order unchanged:
// fb = (...((fb + c0) + c1)...) exactly as merge_tile after a zero-initialised tile accumulate).
// Sample indices are sample_base .. sample_base+spp-1.  Tiles are 64x64, clipped (renderer.rs:105-119);
// `threads` workers pull tiles from a shared queue (concurrency/threadpool.rs:14-40), results are
// merged into the caller's buffers by the calling thread (renderer.rs:152-159).
// sum_rgba (w*h*4) and counts (w*h) are accumulated into (caller zeroes them for accumulate=false).
// tile_rank/tile_world: only tiles with (tile_linear_index % tile_world) == tile_rank are rendered.
// stats_out (5 x u64: samples, traces, tests, disc>=0, hits) may be NULL.
// Returns wall seconds.
double orc_render(const float m16[16], const float pos[3], float fov, float aspect, uint32_t w, uint32_t h,
                  const float* xyzr, const int32_t* material_id, uint32_t n_spheres, const float* mat9,
                  uint32_t n_materials, const float sky_color[3], const float* skybox, uint32_t sky_w, uint32_t sky_h,
                  uint32_t bounces, uint32_t rr_start, uint64_t seed, int rng_mode, int math_mode, uint32_t spp,
                  uint32_t sample_base, uint32_t threads, uint32_t tile_rank, uint32_t tile_world, float* sum_rgba,
                  uint32_t* counts, uint64_t* stats_out) {
    auto t0 = std::chrono::steady_clock::now();
    Camera cam = make_cam(m16, pos, fov, aspect, w, h);
    const float tan_half = std::tan(cam.fov / 2.0f);
    Scene sc;
    sc.n_spheres = n_spheres;
    sc.xyzr = xyzr;
    sc.material = material_id;
    sc.n_materials = n_materials;
    sc.mat9 = mat9;
    std::memcpy(sc.sky, sky_color, sizeof(sc.sky));
    sc.skybox = skybox;
    sc.sky_w = sky_w;
    sc.sky_h = sky_h;
    Params pr{bounces, rr_start, seed, rng_mode, math_mode};
    SkyDist dist;
    if (g_sky_sampling == 1 && skybox) {
        build_sky_dist(skybox, sky_w, sky_h, dist);
        sc.dist = &dist;
    }

    const uint32_t T = 64;
    const uint32_t tx_n = (w + T - 1) / T, ty_n = (h + T - 1) / T;
    const uint32_t n_tiles = tx_n * ty_n;
    if (threads == 0) threads = 1;
    if (tile_world == 0) tile_world = 1;

    struct TileResult {
        uint32_t ox, oy, tw, th;
        std::vector<float> fb;
        std::vector<uint32_t> cnt;
    };
    std::vector<TileResult> results(n_tiles);
    std::atomic<uint32_t> next_tile{0};
    std::vector<Stats> tstats(threads, Stats{0, 0, 0, 0, 0});

    auto worker = [&](uint32_t tid) {
        Stats* st = stats_out ? &tstats[tid] : nullptr;
        for (;;) {
            uint32_t ti = next_tile.fetch_add(1);
            if (ti >= n_tiles) break;
            if (ti % tile_world != tile_rank) continue;
            TileResult& tr = results[ti];
            tr.ox = (ti % tx_n) * T;
            tr.oy = (ti / tx_n) * T;
            tr.tw = std::min(T, w - tr.ox);
            tr.th = std::min(T, h - tr.oy);
            tr.fb.assign((size_t)tr.tw * tr.th * 4, 0.0f);
            tr.cnt.assign((size_t)tr.tw * tr.th, 0u);
            for (uint32_t dy = 0; dy < tr.th; ++dy)
                for (uint32_t dx = 0; dx < tr.tw; ++dx) {
                    uint32_t x = tr.ox + dx, y = tr.oy + dy;
                    uint32_t pixel = y * w + x;
                    float* f = &tr.fb[4 * ((size_t)dy * tr.tw + dx)];
                    // The reference renders ONE sample per render() call: a zeroed TileAccumulator takes it
                    // (tile_buffer.rs:24-28) and is merged, framebuffer += tile (full_buffer.rs:107-125).  A call of
                    // spp samples is restated as consecutive blocks of g_sample_block samples, each block one such
                    // "call": summed from zero in sample order, then merged into the frame in block order.  On a
                    // zeroed frame and spp <= block this is exactly spp successive reference calls.
                    const float* g = sum_rgba + 4 * (size_t)pixel;
                    float acc[4] = {g[0], g[1], g[2], g[3]};
                    for (uint32_t s0 = 0; s0 < spp; s0 += g_sample_block) {
                        float tile[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        const uint32_t s1 = std::min(spp, s0 + g_sample_block);
                        for (uint32_t s = s0; s < s1; ++s) {
                            Sampler smp;
                            init_sampler(smp, pr, pixel, sample_base + s);
                            V3 c = radiance(sc, cam, tan_half, pr, x, y, smp, st);
                            tile[0] += c.x; tile[1] += c.y; tile[2] += c.z; tile[3] += 1.0f;
                        }
                        for (int k = 0; k < 4; ++k) acc[k] += tile[k];
                    }
                    f[0] = acc[0]; f[1] = acc[1]; f[2] = acc[2]; f[3] = acc[3];
                    tr.cnt[(size_t)dy * tr.tw + dx] = spp;
                }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t i = 1; i < threads; ++i) pool.emplace_back(worker, i);
    worker(0);
    for (auto& th : pool) th.join();
    // merge (renderer.rs:152-159 / full_buffer.rs:107-125).  The tile already holds fb+sum, so store.
    for (uint32_t ti = 0; ti < n_tiles; ++ti) {
        if (ti % tile_world != tile_rank) continue;
        TileResult& tr = results[ti];
        for (uint32_t ty = 0; ty < tr.th; ++ty)
            for (uint32_t tx = 0; tx < tr.tw; ++tx) {
                size_t gi = (size_t)(tr.oy + ty) * w + (tr.ox + tx);
                size_t li = (size_t)ty * tr.tw + tx;
                for (int k = 0; k < 4; ++k) sum_rgba[4 * gi + k] = tr.fb[4 * li + k];
                counts[gi] += tr.cnt[li];
            }
    }
    if (stats_out) {
        Stats tot{0, 0, 0, 0, 0};
        for (auto& s : tstats) {
            tot.samples += s.samples; tot.traces += s.traces; tot.tests += s.tests; tot.disc_pos += s.disc_pos; tot.hits += s.hits;
        }
        stats_out[0] = tot.samples; stats_out[1] = tot.traces; stats_out[2] = tot.tests; stats_out[3] = tot.disc_pos; stats_out[4] = tot.hits;
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Per-sample radiance for one pixel (debugging / closed-form tests): out = spp*3 floats
void orc_pixel_samples(const float m16[16], const float pos[3], float fov, float aspect, uint32_t w, uint32_t h,
                       const float* xyzr, const int32_t* material_id, uint32_t n_spheres, const float* mat9,
                       uint32_t n_materials, const float sky_color[3], const float* skybox, uint32_t sky_w,
                       uint32_t sky_h, uint32_t bounces, uint32_t rr_start, uint64_t seed, int rng_mode, int math_mode,
                       uint32_t x, uint32_t y, uint32_t spp, uint32_t sample_base, float* out) {
    Camera cam = make_cam(m16, pos, fov, aspect, w, h);
    const float tan_half = std::tan(cam.fov / 2.0f);
    Scene sc;
    sc.n_spheres = n_spheres; sc.xyzr = xyzr; sc.material = material_id; sc.n_materials = n_materials; sc.mat9 = mat9;
    std::memcpy(sc.sky, sky_color, sizeof(sc.sky));
    sc.skybox = skybox; sc.sky_w = sky_w; sc.sky_h = sky_h;
    Params pr{bounces, rr_start, seed, rng_mode, math_mode};
    SkyDist dist;
    if (g_sky_sampling == 1 && skybox) {
        build_sky_dist(skybox, sky_w, sky_h, dist);
        sc.dist = &dist;
    }
    for (uint32_t s = 0; s < spp; ++s) {
        Sampler smp;
        init_sampler(smp, pr, y * w + x, sample_base + s);
        V3 c = radiance(sc, cam, tan_half, pr, x, y, smp, nullptr);
        out[3 * s] = c.x; out[3 * s + 1] = c.y; out[3 * s + 2] = c.z;
    }
}

}  // extern "C"
