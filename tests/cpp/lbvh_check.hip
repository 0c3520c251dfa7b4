// lbvh_check.hip — host-side check that the LBVH closest hit (pt::trace_lbvh over pt::build_lbvh) returns the same
// (t, index) bits as the linear scan (pt::trace_linear, the restatement of integrator.rs:106-127) for many rays.
// Both functions are the product's own device code compiled for the host (same -ffp-contract=off arithmetic).
// usage: lbvh_check <spheres.bin: n x 4 f32> <rays.bin: m x 6 f32>   -> prints "ok <m> rays" / first mismatch
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../path_tracer_amd/csrc/lbvh.h"
#include "../../path_tracer_amd/csrc/pt_device.h"

static std::vector<float> read_f32(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<float> v(n / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::vector<float> sph = read_f32(argv[1]), rays = read_f32(argv[2]);
    const uint32_t n = (uint32_t)(sph.size() / 4);
    const size_t m = rays.size() / 6;
    pt::Lbvh t;
    const bool sah = argc > 3 && strcmp(argv[3], "sah") == 0;  // split rule of the host build (lbvh.h LbvhSplit)
    pt::build_lbvh(sph.data(), n, t, sah ? pt::LbvhSplit::Sah : pt::LbvhSplit::Morton);
    std::vector<float4> lin(n);
    for (uint32_t i = 0; i < n; ++i) lin[i] = make_float4(sph[4 * i], sph[4 * i + 1], sph[4 * i + 2], sph[4 * i + 3] * sph[4 * i + 3]);
    pt::LbvhView v;
    v.wide = (const float4*)t.wide.data();
    v.leaf_spheres = (const float4*)t.leaf_spheres.data();
    v.leaf_index = t.leaf_index.data();
    v.sphere_start = (const int2*)t.sphere_start.data();
    v.n_inner = t.n_inner;
    v.stack_depth = t.stack_depth;
    if (t.stack_depth > pt::kMaxStack) { printf("stack depth %u exceeds kMaxStack\n", t.stack_depth); return 1; }
    int stack[pt::kMaxStack];
    unsigned long long visits = 0, leaves = 0, hits = 0, climb_visits = 0;
    for (size_t r = 0; r < m; ++r) {
        const float* p = &rays[6 * r];
        pt::f3 o = pt::mk3(p[0], p[1], p[2]), d = pt::mk3(p[3], p[4], p[5]);
        uint32_t dp0 = 0, dp1 = 0, nv = 0, lt = 0;
        pt::HitRec a = pt::trace_linear<false>(lin.data(), n, o, d, dp0);
        pt::HitRec b = pt::trace_lbvh<true>(v, stack, 1u, o, d, dp1, nv, lt);
        visits += nv; leaves += lt; hits += a.idx >= 0;
        if (a.idx != b.idx || memcmp(&a.t, &b.t, 4) != 0) {
            printf("MISMATCH ray %zu: scan (%d, %.9g) lbvh (%d, %.9g)\n", r, a.idx, a.t, b.idx, b.t);
            return 1;
        }
        // The same ray as a climb (lbvh.h "Climb records"): from the leaf of the sphere nearest to its origin (what a bounce
        // ray does) and from an arbitrary sphere (the start only decides the visiting order: any start must do).
        uint32_t nearest = 0;
        float nearest_d = INFINITY;
        for (uint32_t i = 0; i < n; ++i) {
            const float dx = sph[4 * i] - p[0], dy = sph[4 * i + 1] - p[1], dz = sph[4 * i + 2] - p[2];
            const float dist = fabsf(sqrtf(dx * dx + dy * dy + dz * dz) - fabsf(sph[4 * i + 3]));
            if (dist < nearest_d) { nearest_d = dist; nearest = i; }
        }
        const uint32_t starts[2] = {nearest, (uint32_t)((r * 2654435761ull + 12345u) % n)};
        for (int k = 0; k < 2; ++k) {
            pt::LbvhCursor c;
            pt::cursor_begin_at(c, o, d, v.sphere_start[starts[k]]);
            uint32_t cv = 0, cl = 0, cd = 0;
            while (c.cur != pt::kCursorDone) pt::cursor_step<true>(v, c, stack, 1u, cd, cv, cl);
            const int ci = pt::cursor_hit_index(v, c);
            if (ci != a.idx || memcmp(&c.best, &a.t, 4) != 0) {
                printf("MISMATCH ray %zu climbing from sphere %u: scan (%d, %.9g) climb (%d, %.9g)\n", r, starts[k], a.idx, a.t, ci, c.best);
                return 1;
            }
            if (k == 0) climb_visits += cv;
        }
    }
    printf("ok %zu rays, %u spheres, %u nodes, depth %u, hits %llu, mean node visits %.1f (climbing from the nearest sphere %.1f), mean leaf tests %.1f\n",
           m, n, t.n_inner, t.max_depth, hits, (double)visits / m, (double)climb_visits / m, (double)leaves / m);
    return 0;
}
