// host_asan_main.cpp — host-only harness for the host code a user's scene reaches besides the EXR reader: the LBVH
// builder (csrc/lbvh.cpp), the sky importance-sampling tables (csrc/sky_distribution.cpp) and the camera / sky-lookup
// helpers (csrc/host_math.cpp), under AddressSanitizer + UBSan (tests/test_host_sanitizers.py builds it with
// g++ -fsanitize=address,undefined; no GPU, no HIP).  It checks the structural invariants of what the builders return
// and touches every element, so that an out-of-bounds index in a builder is an ASan report or an assertion here.
// usage: host_asan_main   (prints one line per case; exit code 0 = all invariants hold)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/insploray_hip.h"
#include "../../path_tracer_amd/csrc/lbvh.h"
#include "../../path_tracer_amd/csrc/sky_distribution.h"

#define CHECK(c)                                                          \
    do {                                                                  \
        if (!(c)) {                                                       \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);     \
            return false;                                                 \
        }                                                                 \
    } while (0)

static int32_t word(const pt::WideNode& w, int i) {
    int32_t v;
    std::memcpy(&v, &w.v[i], 4);
    return v;
}

// every reference of every record points at something that exists; every sphere is in exactly one leaf; every sphere's
// traversal start is the leaf that holds it and that leaf's record
static bool check_tree(const std::vector<float>& sph, const char* name, pt::LbvhSplit split) {
    const uint32_t n = (uint32_t)(sph.size() / 4);
    pt::Lbvh t;
    pt::build_lbvh(sph.data(), n, t, split);
    if (n == 0) {
        CHECK(t.wide.empty() && t.leaf_index.empty());
        std::printf("%-12s n=0 ok\n", name);
        return true;
    }
    CHECK(t.n_inner >= 1 && t.wide.size() == 2 * (size_t)t.n_inner + n);
    CHECK(t.leaf_index.size() == n && t.leaf_spheres.size() == 4 * (size_t)n && t.sphere_start.size() == 2 * (size_t)n);
    CHECK(t.stack_depth >= 1 && t.stack_depth <= pt::kMaxStack);
    std::vector<int> seen(n, 0), in_leaf(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        CHECK(t.leaf_index[i] >= 0 && (uint32_t)t.leaf_index[i] < n);
        seen[(size_t)t.leaf_index[i]]++;
    }
    for (uint32_t i = 0; i < n; ++i) CHECK(seen[i] == 1);
    auto check_ref = [&](int32_t ref, bool count_leaf) {
        if (ref >= 0) return (uint32_t)ref < t.n_inner;
        const uint32_t leaf = (uint32_t)(-(ref + 1)), first = leaf >> 3, count = leaf & 7u;
        if (count > pt::kLeafMax || first + count > n) return false;
        if (count_leaf)
            for (uint32_t k = 0; k < count; ++k) in_leaf[first + k]++;
        return true;
    };
    for (uint32_t i = 0; i < t.n_inner; ++i) {
        CHECK(check_ref(word(t.wide[i], 3), true) && check_ref(word(t.wide[i], 7), true));
        CHECK(word(t.wide[i], 11) == 0);  // a wide node names no next record
    }
    for (uint32_t i = 0; i < n; ++i) CHECK(in_leaf[i] == 1);
    for (size_t r = t.n_inner; r < t.wide.size(); ++r) {
        const int32_t next = word(t.wide[r], 11);
        if (next == 0) continue;  // a slot no node owns
        CHECK(check_ref(word(t.wide[r], 3), false) && check_ref(word(t.wide[r], 7), false));
        CHECK(next == pt::kClimbDone || ((uint32_t)next > t.n_inner && (size_t)next < 2 * (size_t)t.n_inner));
    }
    for (uint32_t i = 0; i < n; ++i) {
        const int32_t ref = t.sphere_start[2 * (size_t)i], rec = t.sphere_start[2 * (size_t)i + 1];
        CHECK(ref < 0);
        const uint32_t leaf = (uint32_t)(-(ref + 1)), first = leaf >> 3, count = leaf & 7u;
        bool found = false;
        for (uint32_t k = 0; k < count; ++k) found = found || (uint32_t)t.leaf_index[first + k] == i;
        CHECK(found);
        CHECK(rec == pt::kClimbDone || (size_t)rec == 2 * (size_t)t.n_inner + first);
    }
    double sum = 0.0;
    for (const pt::WideNode& w : t.wide)
        for (int k = 0; k < 12; ++k) sum += (double)word(w, k);  // touch everything
    std::printf("%-12s n=%u inner=%u depth=%u ok (%g)\n", name, n, t.n_inner, t.max_depth, sum);
    return true;
}

static bool check_sky(uint32_t w, uint32_t h, int kind) {
    std::mt19937 rng(w * 131u + h * 7u + (uint32_t)kind);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<float> img((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = u(rng);
            if (kind == 1) v = 0.0f;                                      // no weight anywhere: uniform fallback
            if (kind == 2 && i % 7 == 0) v = k == 0 ? INFINITY : (k == 1 ? NAN : -1.0f);  // texels that must get weight 0
            if (kind == 3) v = i == (size_t)w * h / 2 ? 1.0e30f : 0.0f;    // one texel holds everything
            img[4 * i + k] = v;
        }
    pt::SkyDistribution d;
    pt::build_sky_distribution(img.data(), w, h, d);
    CHECK(d.marginal.size() == (size_t)h + 1 && d.cond.size() == (size_t)h * (w + 1) && d.pdf.size() == (size_t)w * h);
    CHECK(d.marginal_guide.size() == (size_t)h + 1 && d.cond_guide.size() == (size_t)h * (w + 1));
    CHECK(d.marginal[0] == 0.0f && d.marginal[h] == 1.0f);
    for (uint32_t y = 0; y < h; ++y) {
        CHECK(d.marginal[y] <= d.marginal[y + 1]);
        const float* row = &d.cond[(size_t)y * (w + 1)];
        CHECK(row[0] == 0.0f && row[w] == 1.0f);
        for (uint32_t x = 0; x < w; ++x) CHECK(row[x] <= row[x + 1] && d.pdf[(size_t)y * w + x] >= 0.0f);
        for (uint32_t k = 0; k <= w; ++k) CHECK(d.cond_guide[(size_t)y * (w + 1) + k] < w);
    }
    for (uint32_t k = 0; k <= h; ++k) CHECK(d.marginal_guide[k] < h);
    std::printf("sky %ux%u kind %d ok\n", w, h, kind);
    return true;
}

static bool check_host_math() {
    const float pos[3] = {-2.4027f, -2.5716f, 3.5259f}, rot[3] = {0.1975f, -0.7941f, -1.9074f};
    float m[16], fov, aspect, dirs[9];
    CHECK(ir_pinhole_compute(pos, rot, 35.0f, 55.0f, 640, 360, m, &fov, &aspect, dirs) == IR_OK);
    CHECK(std::fabs(fov - 2.0f * std::atan(55.0f / 70.0f)) < 1e-6f && std::fabs(aspect - 640.0f / 360.0f) < 1e-6f);
    float ray[6];
    for (uint32_t y : {0u, 179u, 359u})
        for (uint32_t x : {0u, 320u, 639u}) {
            CHECK(ir_pinhole_get_ray(m, pos, fov, aspect, 640, 360, x, y, ray) == IR_OK);
            const float len = std::sqrt(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
            CHECK(std::fabs(len - 1.0f) < 1e-5f);
        }
    std::vector<float> sky(3 * 8 * 4);
    for (size_t i = 0; i < sky.size(); ++i) sky[i] = (float)i;
    const float dirs6[][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {0, 0, 0}, {NAN, 1, 0}, {INFINITY, 0, 0}};
    for (const auto& d : dirs6) {
        float rgb[3];
        CHECK(ir_exr_sample(sky.data(), 8, 4, d, rgb) == IR_OK);
        CHECK(rgb[0] >= 0.0f && rgb[0] < (float)sky.size());  // a texel of the image, whatever the direction
    }
    std::printf("host math ok\n");
    return true;
}

int main() {
    bool ok = true;
    std::mt19937 rng(99);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    auto random_scene = [&](uint32_t n) {
        std::vector<float> s(4 * (size_t)n);
        for (uint32_t i = 0; i < n; ++i) {
            s[4 * i] = -40.0f + 80.0f * u(rng);
            s[4 * i + 1] = -2.0f + 8.0f * u(rng);
            s[4 * i + 2] = -80.0f * u(rng);
            s[4 * i + 3] = 0.1f + 0.5f * u(rng);
        }
        return s;
    };
    bool all = true;
    for (pt::LbvhSplit split : {pt::LbvhSplit::Morton, pt::LbvhSplit::Sah}) {
    auto check_tree = [&](const std::vector<float>& sph, const char* name) { return ::check_tree(sph, name, split); };
    std::printf(split == pt::LbvhSplit::Sah ? "-- surface-area splits\n" : "-- Morton splits\n");
    for (uint32_t n : {0u, 1u, 2u, 3u, 4u, 5u, 7u, 64u, 1000u, 20000u}) ok = check_tree(random_scene(n), "random") && ok;
    {  // all spheres at one point: equal Morton codes all the way down
        std::vector<float> s = random_scene(300);
        for (uint32_t i = 0; i < 300; ++i) s[4 * i] = s[4 * i + 1] = 0.0f, s[4 * i + 2] = -20.0f;
        ok = check_tree(s, "one point") && ok;
    }
    {  // few distinct centres, non-finite centres and radii, negative radius
        std::vector<float> s = random_scene(2000);
        for (uint32_t i = 0; i < 2000; ++i) s[4 * i] = (float)(i % 7), s[4 * i + 1] = (float)(i % 3);
        s[4 * 5] = INFINITY; s[4 * 17 + 1] = NAN; s[4 * 40 + 3] = NAN; s[4 * 41 + 3] = INFINITY; s[4 * 77 + 3] = -0.4f;
        s[4 * 90 + 3] = 1.0e30f;  // a box beyond what an f16 half extent holds
        ok = check_tree(s, "degenerate") && ok;
    }
    {  // a Morton chain deeper than the traversal stack: the builder must fall back to median splits
        std::vector<float> s(4 * 400, 0.0f);
        uint32_t k = 0;
        for (int j = 0; j < 10; ++j)
            for (int axis = 0; axis < 3; ++axis) s[4 * k++ + axis] = std::ldexp(1.0f, j) / 1024.0f;
        s[4 * k] = s[4 * k + 1] = s[4 * k + 2] = 1.0f;
        for (uint32_t i = 0; i < 400; ++i) s[4 * i + 3] = 0.3f;
        ok = check_tree(s, "deep chain") && ok;
    }
    all = all && ok;
    }
    ok = all;
    for (int kind = 0; kind < 4; ++kind)
        for (auto wh : {std::pair<uint32_t, uint32_t>{1, 1}, {2, 1}, {17, 9}, {256, 128}}) ok = check_sky(wh.first, wh.second, kind) && ok;
    ok = check_host_math() && ok;
    std::printf(ok ? "ALL OK\n" : "SOME FAILED\n");
    return ok ? 0 : 1;
}
