// pt_kernels.hip — gfx950 (CDNA4, wave64) kernels of the InsploRay integrator.
//
// render_kernel: one lane owns one pixel and runs its samples in order (regeneration loop): when a path
// ends, the lane adds it to its running sum and restarts from the pixel's primary hit with the next sample
// index.  The primary ray has no jitter (pinhole_camera.rs:81-84), so ray generation and the first
// closest-hit are shared by all samples of the pixel; everything after it is per sample.  Per-pixel sums
// are therefore formed in exactly the order of the reference (one sample per render() call,
// tile_buffer.rs:24-28 -> full_buffer.rs:107-125), which makes them bit-comparable to the CPU oracle.
// The sphere SoA buffer is staged in LDS once per workgroup and read wave-uniformly (LDS broadcast).
#include "pt_kernels.h"

#include <cstdlib>

namespace pt {

namespace {

struct WorkCounters {
    uint32_t disc, nodes, leaves;
};

// closest hit by the selected structure; centre of sphere idx for closest_hit.  (The LBVH has its own integrator
// kernel below; only the primary-hits kernel reaches the tree through here.)
template <int ACCEL, bool STATS, class Args>
__device__ __forceinline__ HitRec trace_any(const Args& a, const float4* s_sph, f3 o, f3 d, WorkCounters& wc) {
    if constexpr (ACCEL == kAccelLbvh) {
        int* stack = reinterpret_cast<int*>(const_cast<float4*>(s_sph)) + threadIdx.x;  // dynamic LDS = the stacks
        return trace_lbvh<STATS>(a.bvh, stack, blockDim.x, o, d, wc.disc, wc.nodes, wc.leaves);
    } else {
        return trace_linear<STATS>(s_sph, a.n_spheres, o, d, wc.disc);
    }
}
template <int ACCEL, class Args>
__device__ __forceinline__ float4 sphere_at(const Args& a, const float4* s_sph, int idx) {
    if constexpr (ACCEL == kAccelLbvh) {
        return a.spheres[idx];
    } else {
        return s_sph[idx];
    }
}

// SKYMIS: sky importance sampling (IR_SKY_SAMPLING_MIS, pt_device.h): one more direction per traced bounce, drawn from
// the sky's distribution and traced as a shadow ray; both strategies weighted with the balance heuristic.
template <int ACCEL, bool STATS, bool SKYMIS>
__global__ __launch_bounds__(256) void render_kernel(const RenderArgs a) {
    extern __shared__ float4 s_sph[];
    if constexpr (ACCEL == kAccelLinear) stage_spheres(s_sph, a.spheres, a.n_spheres);

    // grid = sample blocks (slowest) x owned tiles x 16 workgroups per tile
    const uint32_t wg = blockIdx.x % a.wgs_per_layer, blk = blockIdx.x / a.wgs_per_layer;
    uint32_t x, y;
    if (!lane_pixel(wg, a.tiles_x, a.tile_rank, a.tile_world, a.w, a.h, x, y)) return;
    const uint32_t pixel = y * a.w + x;

    // per-lane work statistics (STATS only): traces, discriminant >= 0 tests, hits
    uint32_t n_traces = 0, n_disc = 0, n_hits = 0;
    WorkCounters wc0 = {0, 0, 0}, wc = {0, 0, 0};
    // lane-activity profile (STATS only): for the loop top and its three lane-dependent sections, how many times a wave
    // ran the section (counted on its first active lane) and how many lanes were active in it
    uint32_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto profile = [&](int section) {
        if constexpr (STATS) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
            prof[2 * section] += (uint32_t)(__builtin_ctzll(m) == (int)(threadIdx.x & 63u));
            prof[2 * section + 1] += 1u;
        }
    };

    // ---- primary ray + primary closest hit (shared by all samples of the pixel)
    Ray r0;
    if (a.rays) {
        const float* p = a.rays + 6 * (size_t)pixel;
        r0.o = mk3(p[0], p[1], p[2]);
        r0.d = mk3(p[3], p[4], p[5]);
    } else {
        r0 = pinhole_ray(a.cam, x, y, a.w, a.h);
    }
    // this workgroup's sample block: like one reference render() call it sums its samples from zero in sample order
    // (TileAccumulator, tile_buffer.rs:24-28); the block sums are then merged into the accumulator in block order
    // (Accumulator::merge_tile, full_buffer.rs:107-125), by merge_kernel or, for a single block, right here
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t first = blk * a.sample_block;
    const uint32_t spp = (a.spp - first < a.sample_block) ? a.spp - first : a.sample_block;
    const uint32_t sample_base = a.sample_base + first;
    const uint32_t B = a.bounces, R = a.rr_start;
    const bool fast = a.fast_materials != 0;

    if (B == 0) {
        // integrator.rs:47: the bounce loop never runs, every sample is (0,0,0,1)
        for (uint32_t s = 0; s < spp; ++s) {
            acc.x += 0.0f; acc.y += 0.0f; acc.z += 0.0f; acc.w += 1.0f;
        }
    } else {
        const HitRec h0 = trace_any<ACCEL, STATS>(a, s_sph, r0.o, r0.d, wc0);
        const uint32_t disc0 = wc0.disc;
        if (h0.idx < 0) {
            // every sample of this pixel is sky(d0) * (1,1,1) added to a zero light (integrator.rs:84-89)
            f3 light = mk3(0.0f, 0.0f, 0.0f) + sky_eval(a, r0.d) * mk3(1.0f, 1.0f, 1.0f);
            for (uint32_t s = 0; s < spp; ++s) {
                acc.x += light.x; acc.y += light.y; acc.z += light.z; acc.w += 1.0f;
            }
            if (STATS) {
                n_traces = spp;
                n_disc = disc0 * spp;
            }
        } else {
            f3 pos0, nrm0;
            {
                const float4 sp = sphere_at<ACCEL>(a, s_sph, h0.idx);
                hit_geometry(sp, r0.o, r0.d, h0.t, pos0, nrm0);
            }
            const int idx0 = h0.idx;
            const uint32_t pkey = pixel_key(a.seed_lo, pixel);

            uint32_t s = 0, b = 0;
            uint32_t skey = sample_key(a.seed_hi, pkey, sample_base);
            f3 pos = pos0, nrm = nrm0;
            int idx = idx0;
            f3 contrib = mk3(1.0f, 1.0f, 1.0f), light = mk3(0.0f, 0.0f, 0.0f);
            if (STATS) { n_traces = 1; n_disc = disc0; n_hits = 1; }
            const SkyTables sky_tables = {a.sky_marginal, a.sky_cond, a.sky_pdf, a.sky_marginal_guide, a.sky_cond_guide};

            while (s < spp) {
                profile(0);
                // ---- shade the hit of bounce b (integrator.rs:50-81)
                const float4 m_brdf = a.shade[2 * idx];
                const float4 m_emit = a.shade[2 * idx + 1];
                light = light + mk3(m_emit.x, m_emit.y, m_emit.z) * contrib;
                const float r1 = draw(skey, 3 * b);
                const float r2 = draw(skey, 3 * b + 1);
                const f3 wi = sample_hemisphere(nrm, r1, r2);
                const float cos_theta = __builtin_fmaxf(dot(wi, nrm), 0.0f);
                contrib = contrib * scatter_weight(mk3(m_brdf.x, m_brdf.y, m_brdf.z), cos_theta, fast);
                bool done = false;
                if (b >= R) {
                    profile(1);
                    const float p = __builtin_fmaxf(contrib.x, __builtin_fmaxf(contrib.y, contrib.z));
                    if (draw(skey, 3 * b + 2) > p) {
                        done = true;
                    } else {
                        contrib = roulette_divide(contrib, p, fast);
                    }
                }
                if (!done) {
                    const f3 o = pos + nrm * kEps;
                    const f3 d = wi;
                    ++b;
                    if (b < B) {
                        float ray_density = 0.0f;
                        if constexpr (SKYMIS) {
                            ray_density = bsdf_density_r(sqrt_cr(r2));
                            uint32_t ix, iy;
                            f3 ws;
                            float sin_polar;
                            sky_sample(sky_tables, a.sky_w, a.sky_h, draw(skey, kSkyDrawBase + 2 * (b - 1)),
                                       draw(skey, kSkyDrawBase + 2 * (b - 1) + 1), ix, iy, ws, sin_polar);
                            const float c = dot(ws, nrm);
                            if (c > 0.0f && sin_polar > 0.0f) {
                                const size_t texel = (size_t)iy * a.sky_w + ix;
                                const float pe = sky_density(sky_tables, texel, sin_polar);
                                const float pa = bsdf_density_dir(c);
                                const float wgt = pa / (pe + pa);
                                const uint32_t before = wc.disc;
                                const HitRec hs = trace_any<ACCEL, STATS>(a, s_sph, o, ws, wc);
                                if (STATS) { n_traces++; n_disc += wc.disc - before; }
                                if (hs.idx < 0) {
                                    f3 sk;
                                    if (a.sky_filter == 1) {
                                        sk = sky_bilinear(a.skybox, a.sky_w, a.sky_h, ws);
                                    } else {
                                        const float4 t = a.skybox[texel];
                                        sk = mk3(t.x, t.y, t.z);
                                    }
                                    light = light + (sk * contrib) * wgt;
                                }
                            }
                        }
                        const uint32_t before = wc.disc;
                        const HitRec h = trace_any<ACCEL, STATS>(a, s_sph, o, d, wc);
                        if (STATS) { n_traces++; n_disc += wc.disc - before; }
                        if (h.idx >= 0) {
                            profile(2);
                            if (STATS) n_hits++;
                            const float4 sp = sphere_at<ACCEL>(a, s_sph, h.idx);
                            hit_geometry(sp, o, d, h.t, pos, nrm);
                            idx = h.idx;
                        } else {
                            profile(3);
                            if constexpr (SKYMIS) {
                                size_t texel;
                                float sin_polar;
                                const f3 sk = sky_lookup_mis(a.skybox, a.sky_w, a.sky_h, a.sky_filter, d, texel, sin_polar);
                                const float pe = sky_density(sky_tables, texel, sin_polar);
                                light = light + (sk * contrib) * (ray_density / (ray_density + pe));
                            } else {
                                light = light + sky_eval(a, d) * contrib;
                            }
                            done = true;
                        }
                    } else {
                        done = true;  // bounce budget exhausted: no sky term (integrator.rs:47,92)
                    }
                }
                if (done) {
                    acc.x += light.x; acc.y += light.y; acc.z += light.z; acc.w += 1.0f;
                    ++s;
                    b = 0;
                    pos = pos0; nrm = nrm0; idx = idx0;
                    contrib = mk3(1.0f, 1.0f, 1.0f);
                    light = mk3(0.0f, 0.0f, 0.0f);
                    skey = sample_key(a.seed_hi, pkey, sample_base + s);
                    if (STATS && s < spp) { n_traces++; n_disc += disc0; n_hits++; }
                }
            }
        }
    }
    if (a.partials) {
        a.partials[(size_t)blk * ((size_t)a.w * a.h) + pixel] = acc;
    } else {
        float4 t = a.sum[pixel];
        t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
        a.sum[pixel] = t;
        a.counts[pixel] += spp;
    }

    if (STATS) {
        // algorithmic counts, as the reference would execute them (primary ray per sample)
        atomicAdd(&a.stats[0], (unsigned long long)spp);
        atomicAdd(&a.stats[1], (unsigned long long)n_traces);
        atomicAdd(&a.stats[2], (unsigned long long)n_traces * a.n_spheres);
        atomicAdd(&a.stats[3], (unsigned long long)n_disc);
        atomicAdd(&a.stats[4], (unsigned long long)n_hits);
        // executed acceleration-structure work (the shared primary trace is counted once per launch)
        atomicAdd(&a.stats[5], (unsigned long long)wc.nodes + wc0.nodes);
        atomicAdd(&a.stats[6], (unsigned long long)wc.leaves + wc0.leaves);
        for (int k = 0; k < 8; ++k)
            if (prof[k]) atomicAdd(&a.stats[7 + k], (unsigned long long)prof[k]);
    }
}

// ----------------------------------------------------------------------------------------------------
// LBVH integrator: same samples, same sums as render_kernel, different schedule.  With a tree, the length of a
// traversal varies wildly between the rays of one wave (measured on the 10k-sphere scene: 10 % of the lanes active
// when every lane waits for the slowest ray of each bounce).  Here a lane is either TRAVERSING (it advances its own
// ray by one node per step) or WAITING for the shading block (finish the previous ray: hit point or sky + sample
// accumulation + regeneration; then emission, bounce sampling, roulette and the next ray).  The wave runs traversal
// steps for whoever is traversing and runs the shading block only once enough lanes wait for it (or nobody is left
// traversing), so short rays do not sit out long ones: they go on to their next ray.  Every lane still runs its own
// pixel's samples in order, so per-pixel sums are unchanged.
// ----------------------------------------------------------------------------------------------------
constexpr uint32_t kShadeBatch = 32;   // run the shading block when this many lanes wait for it
constexpr uint32_t kStepsPerRound = 8; // traversal steps between two looks at the wave's state

// SKYMIS (sky importance sampling, see render_kernel): the shadow ray is one more traversal of the lane's state machine,
// run before the bounce ray, which waits in registers (direction, density, the strategy's weighted sky term).
template <bool STATS, bool SKYMIS>
__global__ __launch_bounds__(256, SKYMIS ? 4 : 6) void render_lbvh_kernel(const RenderArgs a) {
    extern __shared__ float4 s_dyn[];  // traversal stacks: entry k of thread t at [k * 256 + t]
    int* const stack = reinterpret_cast<int*>(s_dyn) + threadIdx.x;
    const uint32_t wg = blockIdx.x % a.wgs_per_layer, blk = blockIdx.x / a.wgs_per_layer;
    uint32_t x, y;
    if (!lane_pixel(wg, a.tiles_x, a.tile_rank, a.tile_world, a.w, a.h, x, y)) return;
    const uint32_t pixel = y * a.w + x;
    uint32_t n_traces = 0, n_disc = 0, n_hits = 0;
    WorkCounters wc0 = {0, 0, 0}, wc = {0, 0, 0};
    // lane-activity profile (STATS only; ir_get_lane_profile): rounds of the wave loop (lanes with work left), shading
    // block, inner-node steps, leaf steps — wave executions (counted on the first active lane) and active lanes
    uint32_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto profile = [&](int section) {
        if constexpr (STATS) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
            prof[2 * section] += (uint32_t)(__builtin_ctzll(m) == (int)(threadIdx.x & 63u));
            prof[2 * section + 1] += 1u;
        }
    };

    Ray r0;
    if (a.rays) {
        const float* p = a.rays + 6 * (size_t)pixel;
        r0.o = mk3(p[0], p[1], p[2]);
        r0.d = mk3(p[3], p[4], p[5]);
    } else {
        r0 = pinhole_ray(a.cam, x, y, a.w, a.h);
    }
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t first = blk * a.sample_block;
    const uint32_t spp = (a.spp - first < a.sample_block) ? a.spp - first : a.sample_block;
    const uint32_t sample_base = a.sample_base + first;
    const uint32_t B = a.bounces, R = a.rr_start;
    const bool fast = a.fast_materials != 0;

    uint32_t s = 0;  // samples finished by this lane; the lane has work while s < spp
    f3 pos0 = mk3(0, 0, 0), nrm0 = mk3(0, 0, 0);
    int idx0 = -1;
    uint32_t disc0 = 0;
    if (B == 0) {
        for (; s < spp; ++s) { acc.x += 0.0f; acc.y += 0.0f; acc.z += 0.0f; acc.w += 1.0f; }
    } else {
        const HitRec h0 = trace_lbvh<STATS>(a.bvh, stack, 256u, r0.o, r0.d, wc0.disc, wc0.nodes, wc0.leaves);
        disc0 = wc0.disc;
        if (h0.idx < 0) {
            f3 light = mk3(0.0f, 0.0f, 0.0f) + sky_eval(a, r0.d) * mk3(1.0f, 1.0f, 1.0f);
            for (; s < spp; ++s) { acc.x += light.x; acc.y += light.y; acc.z += light.z; acc.w += 1.0f; }
            if (STATS) { n_traces = spp; n_disc = disc0 * spp; }
        } else {
            hit_geometry(a.spheres[h0.idx], r0.o, r0.d, h0.t, pos0, nrm0);
            idx0 = h0.idx;
            if (STATS) { n_traces = 1; n_disc = disc0; n_hits = 1; }
        }
    }
    const uint32_t pkey = pixel_key(a.seed_lo, pixel);
    uint32_t b = 0;
    uint32_t skey = sample_key(a.seed_hi, pkey, sample_base);
    f3 pos = pos0, nrm = nrm0;
    int idx = idx0;
    f3 contrib = mk3(1.0f, 1.0f, 1.0f), light = mk3(0.0f, 0.0f, 0.0f);
    LbvhCursor cur;
    cursor_begin(cur, mk3(0, 0, 0), mk3(0, 0, 1));
    cur.cur = kCursorDone;
    bool traversing = false;  // cur holds an unfinished traversal
    bool landed = false;      // cur holds a finished traversal whose result has not been consumed yet
    // SKYMIS only: cur is the shadow ray of the sky strategy; the bounce ray (pos + nrm * kEps, wi_next) follows it
    bool shadow = false;
    f3 wi_next = mk3(0, 0, 0), sky_term = mk3(0, 0, 0);
    float ray_density = 0.0f;
    const SkyTables sky_tables = {a.sky_marginal, a.sky_cond, a.sky_pdf, a.sky_marginal_guide, a.sky_cond_guide};

    for (;;) {
        const bool waiting = !traversing && s < spp;
        const uint32_t n_wait = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(waiting));
        const uint32_t n_trav = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(traversing));
        if (n_wait == 0 && n_trav == 0) break;
        if (waiting || traversing) profile(0);
        if (n_wait >= kShadeBatch || n_trav == 0) {
            if (waiting) {
                profile(1);
                if (landed) {  // consume the finished traversal (integrator.rs:50-51, 82-90)
                    landed = false;
                    if (STATS) { n_traces++; }
                    if (SKYMIS && shadow) {
                        if (cur.idx < 0) light = light + sky_term;
                        shadow = false;
                        cursor_begin(cur, pos + nrm * kEps, wi_next);
                        traversing = true;
                    } else if (cur.idx >= 0) {
                        if (STATS) n_hits++;
                        hit_geometry(a.spheres[cur.idx], cur.o, cur.d, cur.best, pos, nrm);
                        idx = cur.idx;
                    } else {
                        if constexpr (SKYMIS) {
                            size_t texel;
                            float sin_polar;
                            const f3 sk = sky_lookup_mis(a.skybox, a.sky_w, a.sky_h, a.sky_filter, cur.d, texel, sin_polar);
                            const float pe = sky_density(sky_tables, texel, sin_polar);
                            light = light + (sk * contrib) * (ray_density / (ray_density + pe));
                        } else {
                            light = light + sky_eval(a, cur.d) * contrib;
                        }
                        idx = -1;  // sample finished
                    }
                }
                if (!SKYMIS || !traversing) for (;;) {
                    if (idx < 0) {  // finish the sample, regenerate at the primary hit
                        acc.x += light.x; acc.y += light.y; acc.z += light.z; acc.w += 1.0f;
                        ++s;
                        if (s >= spp) break;
                        b = 0;
                        pos = pos0; nrm = nrm0; idx = idx0;
                        contrib = mk3(1.0f, 1.0f, 1.0f);
                        light = mk3(0.0f, 0.0f, 0.0f);
                        skey = sample_key(a.seed_hi, pkey, sample_base + s);
                        if (STATS) { n_traces++; n_disc += disc0; n_hits++; }
                    }
                    // shade the hit of bounce b (integrator.rs:50-81)
                    const float4 m_brdf = a.shade[2 * idx];
                    const float4 m_emit = a.shade[2 * idx + 1];
                    light = light + mk3(m_emit.x, m_emit.y, m_emit.z) * contrib;
                    const float r1 = draw(skey, 3 * b);
                    const float r2 = draw(skey, 3 * b + 1);
                    const f3 wi = sample_hemisphere(nrm, r1, r2);
                    const float cos_theta = __builtin_fmaxf(dot(wi, nrm), 0.0f);
                    contrib = contrib * scatter_weight(mk3(m_brdf.x, m_brdf.y, m_brdf.z), cos_theta, fast);
                    bool done = false;
                    if (b >= R) {
                        const float p = __builtin_fmaxf(contrib.x, __builtin_fmaxf(contrib.y, contrib.z));
                        if (draw(skey, 3 * b + 2) > p) {
                            done = true;
                        } else {
                            contrib = roulette_divide(contrib, p, fast);
                        }
                    }
                    if (!done) {
                        ++b;
                        if (b < B) {
                            if constexpr (SKYMIS) {
                                ray_density = bsdf_density_r(sqrt_cr(r2));
                                uint32_t ix, iy;
                                f3 ws;
                                float sin_polar;
                                sky_sample(sky_tables, a.sky_w, a.sky_h, draw(skey, kSkyDrawBase + 2 * (b - 1)),
                                           draw(skey, kSkyDrawBase + 2 * (b - 1) + 1), ix, iy, ws, sin_polar);
                                const float c = dot(ws, nrm);
                                if (c > 0.0f && sin_polar > 0.0f) {
                                    const size_t texel = (size_t)iy * a.sky_w + ix;
                                    const float pe = sky_density(sky_tables, texel, sin_polar);
                                    const float pa = bsdf_density_dir(c);
                                    f3 sk;
                                    if (a.sky_filter == 1) {
                                        sk = sky_bilinear(a.skybox, a.sky_w, a.sky_h, ws);
                                    } else {
                                        const float4 t = a.skybox[texel];
                                        sk = mk3(t.x, t.y, t.z);
                                    }
                                    sky_term = (sk * contrib) * (pa / (pe + pa));
                                    wi_next = wi;
                                    shadow = true;
                                    cursor_begin(cur, pos + nrm * kEps, ws);
                                    traversing = true;
                                    break;
                                }
                            }
                            cursor_begin(cur, pos + nrm * kEps, wi);
                            traversing = true;
                            break;
                        }
                    }
                    idx = -1;  // roulette or bounce budget ended the path: no sky term (integrator.rs:47,74,92)
                }
            }
        }
        // Two node steps per leaf step: lanes that stay on inner nodes (most of them) descend two levels for every pass
        // of the divergent leaf code, which then finds more lanes waiting for it (+7 % on C3 over one-and-one; three
        // node steps or a majority vote between the two kinds of step measured no better).
        for (uint32_t k = 0; k < kStepsPerRound; ++k) {
            if (traversing) {
                if (cur.cur >= 0) { profile(2); cursor_inner_step<STATS>(a.bvh, cur, stack, 256u, wc.nodes); }
                if (cur.cur >= 0 && cur.cur != kCursorDone) { profile(2); cursor_inner_step<STATS>(a.bvh, cur, stack, 256u, wc.nodes); }
                if (cur.cur < 0) { profile(3); cursor_leaf_step<STATS>(a.bvh, cur, stack, 256u, wc.disc, wc.leaves); }
                if (cur.cur == kCursorDone) {
                    traversing = false;
                    landed = true;
                }
            }
        }
    }
    if (STATS) n_disc += wc.disc;

    if (a.partials) {
        a.partials[(size_t)blk * ((size_t)a.w * a.h) + pixel] = acc;
    } else {
        float4 t = a.sum[pixel];
        t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
        a.sum[pixel] = t;
        a.counts[pixel] += spp;
    }
    if (STATS) {
        atomicAdd(&a.stats[0], (unsigned long long)spp);
        atomicAdd(&a.stats[1], (unsigned long long)n_traces);
        atomicAdd(&a.stats[2], (unsigned long long)n_traces * a.n_spheres);
        atomicAdd(&a.stats[3], (unsigned long long)n_disc);
        atomicAdd(&a.stats[4], (unsigned long long)n_hits);
        atomicAdd(&a.stats[5], (unsigned long long)wc.nodes + wc0.nodes);
        atomicAdd(&a.stats[6], (unsigned long long)wc.leaves + wc0.leaves);
        for (int k = 0; k < 8; ++k)
            if (prof[k]) atomicAdd(&a.stats[7 + k], (unsigned long long)prof[k]);
    }
}

template <int ACCEL>
__global__ __launch_bounds__(256) void primary_hits_kernel(const HitsArgs a) {
    extern __shared__ float4 s_sph[];
    if constexpr (ACCEL == kAccelLinear) stage_spheres(s_sph, a.spheres, a.n_spheres);
    uint32_t x, y;
    const uint32_t tiles_x = (a.w + kTile - 1) / kTile;
    if (!lane_pixel(blockIdx.x, tiles_x, 0u, 1u, a.w, a.h, x, y)) return;
    const uint32_t pixel = y * a.w + x;
    Ray r0;
    if (a.rays) {
        const float* p = a.rays + 6 * (size_t)pixel;
        r0.o = mk3(p[0], p[1], p[2]);
        r0.d = mk3(p[3], p[4], p[5]);
    } else {
        r0 = pinhole_ray(a.cam, x, y, a.w, a.h);
    }
    WorkCounters wc = {0, 0, 0};
    const HitRec h = trace_any<ACCEL, false>(a, s_sph, r0.o, r0.d, wc);
    a.out_idx[pixel] = h.idx;
    if (a.out_t) a.out_t[pixel] = (h.idx >= 0) ? h.t : -1.0f;  // ray_miss: hit_distance = -1 (integrator.rs:158-164)
}

// Accumulator::merge_tile (full_buffer.rs:107-125) for the sample blocks of one launch: framebuffer += block sum, in
// block order; sample_counts += samples of the launch.
__global__ __launch_bounds__(256) void merge_kernel(const RenderArgs a) {
    uint32_t x, y;
    if (!lane_pixel(blockIdx.x, a.tiles_x, a.tile_rank, a.tile_world, a.w, a.h, x, y)) return;
    const uint32_t pixel = y * a.w + x;
    const size_t npix = (size_t)a.w * a.h;
    float4 t = a.sum[pixel];
    for (uint32_t b = 0; b < a.n_blocks; ++b) {
        const float4 p = a.partials[(size_t)b * npix + pixel];
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
    }
    a.sum[pixel] = t;
    a.counts[pixel] += a.spp;
}

// Accumulator::get_argb_pixel (full_buffer.rs:49-65) + convert_to_argb (utils.rs:4-9)
__device__ __forceinline__ uint32_t to_byte(float c) {
    // Rust `as u32`: saturating, NaN -> 0
    float v = c * 255.0f;
    return (v > 0.0f) ? (uint32_t)__builtin_fminf(v, 4294967040.0f) : 0u;
}
__device__ __forceinline__ float tone(float sum, float n) {
    float a = sum / n;
    a = a / (a + 1.0f);
    a = powf(a, 1.0f / 2.2f);
    a = (a > 0.0f) ? a : 0.0f;  // clamp(0,1); NaN -> 0 like SSE max/min
    a = (a < 1.0f) ? a : 1.0f;
    return a;
}
__global__ __launch_bounds__(256) void resolve_kernel(const float4* __restrict__ sum, const uint32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ argb, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float4 s = sum[i];
        const uint32_t c = counts[i];
        const float nf = (float)(c > 1u ? c : 1u);
        argb[i] = (to_byte(tone(s.w, nf)) << 24) | (to_byte(tone(s.x, nf)) << 16) | (to_byte(tone(s.y, nf)) << 8) |
                  to_byte(tone(s.z, nf));
    }
}

}  // namespace

hipError_t init_kernels() {
    hipError_t e;
    const int max_lds = (int)(kMaxLdsSpheres * sizeof(float4));
    e = hipFuncSetAttribute((const void*)render_kernel<kAccelLinear, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)render_kernel<kAccelLinear, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)render_kernel<kAccelLinear, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)render_kernel<kAccelLinear, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)primary_hits_kernel<kAccelLinear>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    return e;
}

hipError_t launch_render(const RenderArgs& a_in, int accel, bool stats, hipStream_t stream) {
    const uint32_t owned = owned_tiles(a_in.n_tiles, a_in.tile_rank, a_in.tile_world);
    if (owned == 0 || a_in.spp == 0) return hipSuccess;
    RenderArgs a = a_in;
    a.wgs_per_layer = owned * kBlocksPerTile;
    a.n_blocks = (a.spp + a.sample_block - 1) / a.sample_block;
    if (a.n_blocks == 1) a.partials = nullptr;  // single block: the kernel merges into the accumulator itself
    const dim3 grid(a.wgs_per_layer * a.n_blocks), block(256);
    if (accel == kAccelLbvh) {
        const size_t lds = (size_t)a.bvh.stack_depth * 256 * sizeof(int);
        const bool skymis = a.sky_marginal != nullptr;
        if (stats) {
            if (skymis) hipLaunchKernelGGL((render_lbvh_kernel<true, true>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((render_lbvh_kernel<true, false>), grid, block, lds, stream, a);
        } else {
            if (skymis) hipLaunchKernelGGL((render_lbvh_kernel<false, true>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((render_lbvh_kernel<false, false>), grid, block, lds, stream, a);
        }
    } else {
        size_t lds = (size_t)a.n_spheres * sizeof(float4);
        // occupancy experiments only (tools/): IR_DEBUG_LDS_PAD=<bytes> pads the workgroup's LDS allocation
        static const size_t pad = getenv("IR_DEBUG_LDS_PAD") ? (size_t)atol(getenv("IR_DEBUG_LDS_PAD")) : 0;
        if (lds + pad <= kMaxLdsSpheres * sizeof(float4)) lds += pad;
        const bool skymis = a.sky_marginal != nullptr;
        if (stats) {
            if (skymis) hipLaunchKernelGGL((render_kernel<kAccelLinear, true, true>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((render_kernel<kAccelLinear, true, false>), grid, block, lds, stream, a);
        } else {
            if (skymis) hipLaunchKernelGGL((render_kernel<kAccelLinear, false, true>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((render_kernel<kAccelLinear, false, false>), grid, block, lds, stream, a);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.n_blocks > 1) {
        hipLaunchKernelGGL(merge_kernel, dim3(a.wgs_per_layer), block, 0, stream, a);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_primary_hits(const HitsArgs& a, int accel, hipStream_t stream) {
    const uint32_t tiles = ((a.w + kTile - 1) / kTile) * ((a.h + kTile - 1) / kTile);
    if (tiles == 0) return hipSuccess;
    if (accel == kAccelLbvh) {
        hipLaunchKernelGGL(primary_hits_kernel<kAccelLbvh>, dim3(tiles * kBlocksPerTile), dim3(256),
                           (size_t)a.bvh.stack_depth * 256 * sizeof(int), stream, a);
    } else {
        hipLaunchKernelGGL(primary_hits_kernel<kAccelLinear>, dim3(tiles * kBlocksPerTile), dim3(256),
                           (size_t)a.n_spheres * sizeof(float4), stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_resolve(const float4* sum, const uint32_t* counts, uint32_t* argb, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, sum, counts, argb, n);
    return hipGetLastError();
}

}  // namespace pt
