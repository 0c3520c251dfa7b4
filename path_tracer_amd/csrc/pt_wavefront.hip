// pt_wavefront.hip — the integrator as a wavefront pipeline with HBM path queues (IR_SCHEDULE_WAVEFRONT).
//
// Same samples, same per-sample radiance, same sums as the regeneration kernels of pt_kernels.hip; what differs is
// where a path lives between bounces.  One kernel per stage, paths stream through queues in device memory:
//
//   wf_primary   one lane per pixel: primary ray + primary closest hit (shared by every sample of the pixel, there
//                is no jitter, pinhole_camera.rs:81-84) -> per-pixel record (hit point, normal, index | sky value)
//   wf_bounce    one lane per path, bounce d: emission, hemisphere sample, roulette (integrator.rs:50-81), closest
//                hit of the bounce ray; a path that ends (miss -> + sky, roulette, bounce budget) writes its radiance
//                into values[sample][pixel], a surviving path is appended to the next queue.  Bounce 0 needs no
//                queue: its paths are (sample, pixel) pairs read off the per-pixel record (pixels whose primary ray
//                misses write the sky value straight away)
//   wf_commit    one lane per pixel: adds the batch's values IN SAMPLE ORDER to the running block sum (the tile
//                accumulator, tile_buffer.rs:24-28) and, after the block's last batch, merges it into the frame
//                (full_buffer.rs:107-125)
//
// Live paths are compacted between bounces with a wave64 ballot + prefix count (mbcnt); the waves of a workgroup
// add up in LDS and the workgroup takes its slots with ONE atomic on the queue's counter.  Queue order depends on scheduling, results do not: every path writes its own values slot, and
// the sums are formed by wf_commit in sample order.  Queue counters never travel to the host: every stage is a
// fixed-size grid that strides over *count entries.
#include "pt_kernels.h"

namespace pt {
namespace {


struct PathRec {
    uint32_t pix, smp;
    int idx;
    f3 pos, nrm, contrib, light;
};

__device__ __forceinline__ PathRec load_path(const WfQueue& q, uint32_t i) {
    PathRec p;
    p.pix = q.pix[i];
    p.smp = q.smp[i];
    p.idx = q.idx[i];
    const size_t c = q.cap;
    p.pos = mk3(q.f[i], q.f[c + i], q.f[2 * c + i]);
    p.nrm = mk3(q.f[3 * c + i], q.f[4 * c + i], q.f[5 * c + i]);
    p.contrib = mk3(q.f[6 * c + i], q.f[7 * c + i], q.f[8 * c + i]);
    p.light = mk3(q.f[9 * c + i], q.f[10 * c + i], q.f[11 * c + i]);
    return p;
}

// Compacting append.  Every thread of the workgroup must call it (it contains barriers): wave ballot + mbcnt give the
// rank inside the wave, an LDS counter the wave's offset inside the workgroup, one global atomic the workgroup's
// offset in the queue.  A path is dropped only if the queue is full, which the host's batch sizing rules out.
__device__ __forceinline__ void append_path(const WfQueue& q, bool live, const PathRec& p, uint32_t* s_cnt) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(live);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (threadIdx.x == 0) s_cnt[0] = 0;
    __syncthreads();
    uint32_t wave_base = 0;
    if (live && rank == 0) wave_base = atomicAdd(&s_cnt[0], (uint32_t)__builtin_popcountll(mask));
    __syncthreads();
    if (threadIdx.x == 0) s_cnt[1] = s_cnt[0] ? atomicAdd(q.count, s_cnt[0]) : 0u;
    __syncthreads();
    if (mask == 0ull) return;
    wave_base = (uint32_t)__builtin_amdgcn_readlane((int)wave_base, (int)__builtin_ctzll(mask));
    if (!live) return;
    const uint32_t i = s_cnt[1] + wave_base + rank;
    if (i >= q.cap) return;  // cannot happen: a stage never emits more paths than it reads
    q.pix[i] = p.pix;
    q.smp[i] = p.smp;
    q.idx[i] = p.idx;
    const size_t c = q.cap;
    q.f[i] = p.pos.x; q.f[c + i] = p.pos.y; q.f[2 * c + i] = p.pos.z;
    q.f[3 * c + i] = p.nrm.x; q.f[4 * c + i] = p.nrm.y; q.f[5 * c + i] = p.nrm.z;
    q.f[6 * c + i] = p.contrib.x; q.f[7 * c + i] = p.contrib.y; q.f[8 * c + i] = p.contrib.z;
    q.f[9 * c + i] = p.light.x; q.f[10 * c + i] = p.light.y; q.f[11 * c + i] = p.light.z;
}

template <int ACCEL>
__device__ __forceinline__ HitRec wf_trace(const WfArgs& a, const float4* s_dyn, f3 o, f3 d) {
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    if constexpr (ACCEL == kAccelLbvh) {
        int* stack = reinterpret_cast<int*>(const_cast<float4*>(s_dyn)) + threadIdx.x;  // dynamic LDS = the stacks
        return trace_lbvh<false>(a.r.bvh, stack, blockDim.x, o, d, c0, c1, c2);
    } else {
        return trace_linear<false>(s_dyn, a.r.n_spheres, o, d, c0);
    }
}
template <int ACCEL>
__device__ __forceinline__ float4 wf_sphere(const WfArgs& a, const float4* s_dyn, int idx) {
    if constexpr (ACCEL == kAccelLbvh) {
        return a.r.spheres[idx];
    } else {
        return s_dyn[idx];
    }
}

template <int ACCEL>
__global__ __launch_bounds__(256) void wf_primary(const WfArgs a) {
    extern __shared__ float4 s_dyn[];
    if constexpr (ACCEL == kAccelLinear) stage_spheres(s_dyn, a.r.spheres, a.r.n_spheres);
    uint32_t x, y;
    if (!lane_pixel(blockIdx.x, a.r.tiles_x, a.r.tile_rank, a.r.tile_world, a.r.w, a.r.h, x, y)) return;
    const uint32_t pixel = y * a.r.w + x;
    Ray r0;
    if (a.r.rays) {
        const float* p = a.r.rays + 6 * (size_t)pixel;
        r0.o = mk3(p[0], p[1], p[2]);
        r0.d = mk3(p[3], p[4], p[5]);
    } else {
        r0 = pinhole_ray(a.r.cam, x, y, a.r.w, a.r.h);
    }
    const HitRec h0 = wf_trace<ACCEL>(a, s_dyn, r0.o, r0.d);
    if (h0.idx < 0) {
        // every sample of this pixel is sky(d0) * (1,1,1) added to a zero light (integrator.rs:84-89)
        const f3 l0 = mk3(0.0f, 0.0f, 0.0f) + sky_eval(a.r, r0.d) * mk3(1.0f, 1.0f, 1.0f);
        a.prim_pos[pixel] = make_float4(l0.x, l0.y, l0.z, __builtin_bit_cast(float, -1));
        a.prim_nrm[pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        f3 pos0, nrm0;
        hit_geometry(wf_sphere<ACCEL>(a, s_dyn, h0.idx), r0.o, r0.d, h0.t, pos0, nrm0);
        a.prim_pos[pixel] = make_float4(pos0.x, pos0.y, pos0.z, __builtin_bit_cast(float, h0.idx));
        a.prim_nrm[pixel] = make_float4(nrm0.x, nrm0.y, nrm0.z, 0.0f);
    }
}

// FIRST: bounce 0, paths = (sample j of the batch, owned pixel) read off the per-pixel record; else paths = a.qin.
// SKYMIS: sky importance sampling (pt_kernels.hip paths_kernel): the same shadow ray, draws and weights.
template <int ACCEL, bool FIRST, bool SKYMIS>
__global__ __launch_bounds__(256) void wf_bounce(const WfArgs a) {
    extern __shared__ float4 s_dyn[];
    __shared__ uint32_t s_cnt[2];
    if constexpr (ACCEL == kAccelLinear) stage_spheres(s_dyn, a.r.spheres, a.r.n_spheres);
    const size_t npix = (size_t)a.r.w * a.r.h;
    const uint32_t B = a.r.bounces, R = a.r.rr_start, b = a.depth;
    const bool fast = a.r.fast_materials != 0;
    const SkyTables sky_tables = {a.r.sky_marginal, a.r.sky_cond, a.r.sky_pdf, a.r.sky_marginal_guide, a.r.sky_cond_guide};
    const uint32_t owned = owned_tiles(a.r.n_tiles, a.r.tile_rank, a.r.tile_world);
    const uint32_t pixel_wgs = owned * kBlocksPerTile;  // 256-pixel groups of the owned tiles
    uint32_t count;
    if constexpr (FIRST) {
        count = pixel_wgs * 256u * a.batch_n;
    } else {
        count = *a.qin.count < a.qin.cap ? *a.qin.count : a.qin.cap;
    }
    const uint32_t stride = gridDim.x * blockDim.x;
    // every thread of the grid runs the same number of rounds (append_path has barriers and ballots)
    const uint32_t rounds = (count + stride - 1) / stride;
    for (uint32_t k = 0; k < rounds; ++k) {
        const uint32_t i = k * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool have = i < count;
        PathRec p;
        bool survives = false;
        if constexpr (FIRST) {
            if (have) {
                const uint32_t group = i / 256u;  // (sample, 256-pixel group): consecutive threads = one 16x16 block
                p.smp = group / pixel_wgs;
                uint32_t x, y;
                have = lane_pixel(group % pixel_wgs, a.r.tiles_x, a.r.tile_rank, a.r.tile_world, a.r.w, a.r.h, x, y);
                if (have) {
                    p.pix = y * a.r.w + x;
                    const float4 pp = a.prim_pos[p.pix], pn = a.prim_nrm[p.pix];
                    p.idx = __builtin_bit_cast(int, pp.w);
                    p.pos = mk3(pp.x, pp.y, pp.z);
                    p.nrm = mk3(pn.x, pn.y, pn.z);
                    p.contrib = mk3(1.0f, 1.0f, 1.0f);
                    p.light = mk3(0.0f, 0.0f, 0.0f);
                    if (p.idx < 0) {  // primary ray missed: the sample is the sky value stored in the record
                        a.values[(size_t)p.smp * npix + p.pix] = make_float4(pp.x, pp.y, pp.z, 0.0f);
                        have = false;
                    }
                }
            }
        } else {
            if (have) p = load_path(a.qin, i);
        }
        if (have) {
            const uint32_t skey = sample_key(a.r.seed_hi, pixel_key(a.r.seed_lo, p.pix), a.r.sample_base + a.batch_first + p.smp);
            // shade the hit of bounce b (integrator.rs:50-81)
            const float4 m_brdf = load16(a.r.shade, 2u * (uint32_t)p.idx);
            const float4 m_emit = load16(a.r.shade, 2u * (uint32_t)p.idx + 1u);
            p.light = p.light + mk3(m_emit.x, m_emit.y, m_emit.z) * p.contrib;
            const float r1 = draw(skey, 3 * b);
            const float r2 = draw(skey, 3 * b + 1);
            const f3 wi = sample_hemisphere(p.nrm, r1, r2);
            const float cos_theta = __builtin_fmaxf(dot(wi, p.nrm), 0.0f);
            p.contrib = p.contrib * scatter_weight(mk3(m_brdf.x, m_brdf.y, m_brdf.z), cos_theta, fast);
            bool done = false;
            if (b >= R) {
                const float pr = __builtin_fmaxf(p.contrib.x, __builtin_fmaxf(p.contrib.y, p.contrib.z));
                if (draw(skey, 3 * b + 2) > pr) {
                    done = true;
                } else {
                    p.contrib = roulette_divide(p.contrib, pr, fast);
                }
            }
            if (!done && b + 1 < B) {
                const f3 o = p.pos + p.nrm * kEps;
                float ray_density = 0.0f;
                if constexpr (SKYMIS) {  // the second strategy: a direction from the sky's distribution, as a shadow ray
                    ray_density = bsdf_density_r(sqrt_cr(r2));
                    uint32_t ix, iy;
                    f3 ws;
                    float sin_polar;
                    sky_sample(sky_tables, a.r.sky_w, a.r.sky_h, draw(skey, kSkyDrawBase + 2 * b), draw(skey, kSkyDrawBase + 2 * b + 1),
                               ix, iy, ws, sin_polar);
                    const float c = dot(ws, p.nrm);
                    if (c > 0.0f && sin_polar > 0.0f) {
                        const size_t texel = (size_t)iy * a.r.sky_w + ix;
                        const float pe = sky_density(sky_tables, texel, sin_polar);
                        const float pa = bsdf_density_dir(c);
                        const float wgt = pa / (pe + pa);
                        const HitRec hs = wf_trace<ACCEL>(a, s_dyn, o, ws);
                        if (hs.idx < 0) {
                            f3 sk;
                            if (a.r.sky_filter == 1) {
                                sk = sky_bilinear(a.r.skybox, a.r.sky_w, a.r.sky_h, ws);
                            } else {
                                const float4 tx = a.r.skybox[texel];
                                sk = mk3(tx.x, tx.y, tx.z);
                            }
                            p.light = p.light + (sk * p.contrib) * wgt;
                        }
                    }
                }
                const HitRec h = wf_trace<ACCEL>(a, s_dyn, o, wi);
                if (h.idx >= 0) {
                    hit_geometry(wf_sphere<ACCEL>(a, s_dyn, h.idx), o, wi, h.t, p.pos, p.nrm);
                    p.idx = h.idx;
                    survives = true;
                } else if constexpr (SKYMIS) {
                    size_t texel;
                    float sin_polar;
                    const f3 sk = sky_lookup_mis(a.r.skybox, a.r.sky_w, a.r.sky_h, a.r.sky_filter, wi, texel, sin_polar);
                    const float pe = sky_density(sky_tables, texel, sin_polar);
                    p.light = p.light + (sk * p.contrib) * (ray_density / (ray_density + pe));
                } else {
                    p.light = p.light + sky_eval(a.r, wi) * p.contrib;
                }
            }
            // roulette / bounce budget end the path without a sky term (integrator.rs:47,74,92)
            if (!survives) a.values[(size_t)p.smp * npix + p.pix] = make_float4(p.light.x, p.light.y, p.light.z, 0.0f);
        }
        append_path(a.qout, survives, p, s_cnt);
    }
}

// ----------------------------------------------------------------------------------------------------
// Bounce stage for the LBVH: persistent waves with dynamic path fetch.  Ray lengths vary wildly, so a lane that has
// finished its path takes the next one instead of waiting for the wave (paths are independent here: nothing ties a
// lane to a pixel).  A wave owns a chunk of kWfChunk consecutive input entries and a chunk of output slots at a time
// (one global atomic per chunk); inside a chunk, lanes are served with ballot + mbcnt.  Unused output slots of a
// chunk are marked empty (idx = -1) and skipped by the next stage.  Lane states as in paths_lbvh_kernel: at an inner
// node, at a leaf, or waiting for the shading block (finish a landed ray / shade a fresh path).
// ----------------------------------------------------------------------------------------------------
constexpr uint32_t kWfChunk = 256;

__device__ __forceinline__ void store_path(const WfQueue& q, uint32_t i, const PathRec& p) {
    q.pix[i] = p.pix;
    q.smp[i] = p.smp;
    q.idx[i] = p.idx;
    const size_t c = q.cap;
    q.f[i] = p.pos.x; q.f[c + i] = p.pos.y; q.f[2 * c + i] = p.pos.z;
    q.f[3 * c + i] = p.nrm.x; q.f[4 * c + i] = p.nrm.y; q.f[5 * c + i] = p.nrm.z;
    q.f[6 * c + i] = p.contrib.x; q.f[7 * c + i] = p.contrib.y; q.f[8 * c + i] = p.contrib.z;
    q.f[9 * c + i] = p.light.x; q.f[10 * c + i] = p.light.y; q.f[11 * c + i] = p.light.z;
}

// SKYMIS: the shadow ray of the sky strategy is one more traversal of the lane's state machine, run before the bounce
// ray, which waits in registers (as in paths_lbvh_kernel).
template <bool FIRST, bool SKYMIS>
__global__ __launch_bounds__(256, SKYMIS ? 4 : 6) void wf_bounce_tree(const WfArgs a) {
    extern __shared__ float4 s_dyn[];  // traversal stacks: entry k of thread t at [k * 256 + t]
    int* const stack = reinterpret_cast<int*>(s_dyn) + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const size_t npix = (size_t)a.r.w * a.r.h;
    const uint32_t B = a.r.bounces, R = a.r.rr_start, b = a.depth;
    const bool fast = a.r.fast_materials != 0;
    const SkyTables sky_tables = {a.r.sky_marginal, a.r.sky_cond, a.r.sky_pdf, a.r.sky_marginal_guide, a.r.sky_cond_guide};
    const uint32_t pixel_wgs = owned_tiles(a.r.n_tiles, a.r.tile_rank, a.r.tile_world) * kBlocksPerTile;
    uint32_t count;
    if constexpr (FIRST) {
        count = pixel_wgs * 256u * a.batch_n;
    } else {
        count = *a.qin.count < a.qin.cap ? *a.qin.count : a.qin.cap;
    }
    // wave-uniform cursors
    uint32_t in_next = 0, in_end = 0, out_next = 0, out_end = 0;
    bool input_done = false;

    PathRec p;
    p.pix = p.smp = 0; p.idx = -1;
    p.pos = p.nrm = p.contrib = p.light = mk3(0, 0, 0);
    LbvhCursor cur;
    cursor_begin(cur, mk3(0, 0, 0), mk3(0, 0, 1));
    cur.cur = kCursorDone;
    bool busy = false;        // the lane holds a path
    bool fresh = false;       // ... that has not been shaded yet
    bool traversing = false;  // ... whose bounce ray is still in the tree
    // SKYMIS only: cur is the shadow ray; the bounce ray (same origin, wi_next) follows it
    bool shadow = false;
    f3 wi_next = mk3(0, 0, 0), sky_term = mk3(0, 0, 0);
    float ray_density = 0.0f;

    for (;;) {
        // ---- hand input entries to the idle lanes
        while (!input_done) {
            const unsigned long long need = __builtin_amdgcn_ballot_w64(!busy);
            if (need == 0ull) break;
            if (in_next == in_end) {  // next chunk
                uint32_t start = 0;
                if (lane == 0) start = atomicAdd(a.in_cursor, kWfChunk);
                start = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
                if (start >= count) {
                    input_done = true;
                    break;
                }
                in_next = start;
                in_end = start + kWfChunk < count ? start + kWfChunk : count;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t avail = in_end - in_next, wanted = (uint32_t)__builtin_popcountll(need);
            if (!busy && rank < avail) {
                const uint32_t i = in_next + rank;
                bool ok;
                if constexpr (FIRST) {
                    const uint32_t group = i / 256u;
                    p.smp = group / pixel_wgs;
                    uint32_t x, y;
                    // lane_pixel uses threadIdx.x for the position inside the 16x16 block: entry i % 256 plays that part
                    const uint32_t t = i & 255u, wave = t >> 6, ln = t & 63u;
                    const uint32_t k = (group % pixel_wgs) / kBlocksPerTile, sub = (group % pixel_wgs) % kBlocksPerTile;
                    const uint32_t tile = a.r.tile_rank + k * a.r.tile_world;
                    x = (tile % a.r.tiles_x) * kTile + (sub & 3) * kBlockEdge + (wave & 1) * 8 + (ln & 7);
                    y = (tile / a.r.tiles_x) * kTile + (sub >> 2) * kBlockEdge + (wave >> 1) * 8 + (ln >> 3);
                    ok = x < a.r.w && y < a.r.h;
                    if (ok) {
                        p.pix = y * a.r.w + x;
                        const float4 pp = a.prim_pos[p.pix], pn = a.prim_nrm[p.pix];
                        p.idx = __builtin_bit_cast(int, pp.w);
                        p.pos = mk3(pp.x, pp.y, pp.z);
                        p.nrm = mk3(pn.x, pn.y, pn.z);
                        p.contrib = mk3(1.0f, 1.0f, 1.0f);
                        p.light = mk3(0.0f, 0.0f, 0.0f);
                        if (p.idx < 0) {  // primary ray missed: the sample is the sky value stored in the record
                            a.values[(size_t)p.smp * npix + p.pix] = make_float4(pp.x, pp.y, pp.z, 0.0f);
                            ok = false;
                        }
                    }
                } else {
                    p = load_path(a.qin, i);
                    ok = p.idx >= 0;  // empty slot of a partly filled chunk
                }
                busy = ok;
                fresh = ok;
            }
            in_next += wanted < avail ? wanted : avail;
        }
        const bool waiting = busy && !traversing;  // fresh path to shade, or landed ray to finish
        const bool at_inner = traversing && cur.cur >= 0, at_leaf = traversing && cur.cur < 0;
        const uint32_t n_wait = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(waiting));
        const uint32_t n_inner = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(at_inner));
        const uint32_t n_leaf = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(at_leaf));
        if (n_wait + n_inner + n_leaf == 0) {
            if (input_done) break;
            continue;
        }
        if (n_wait >= 32u || n_inner + n_leaf == 0) {
            bool survives = false;
            if (waiting) {
                if (!fresh) {  // a ray has landed (integrator.rs:50-51, 82-90)
                    if (SKYMIS && shadow) {
                        if (cur.idx < 0) p.light = p.light + sky_term;
                        shadow = false;
                        cursor_begin_at(cur, cur.o, wi_next, a.r.bvh.sphere_start[p.idx]);  // the bounce ray leaves the same point as the shadow ray
                        traversing = true;
                    } else {
                        if (cur.idx >= 0) {
                            p.idx = cursor_hit_index(a.r.bvh, cur);
                            hit_geometry(a.r.spheres[p.idx], cur.o, cur.d, cur.best, p.pos, p.nrm);
                            survives = true;
                        } else {
                            if constexpr (SKYMIS) {
                                size_t texel;
                                float sin_polar;
                                const f3 sk = sky_lookup_mis(a.r.skybox, a.r.sky_w, a.r.sky_h, a.r.sky_filter, cur.d, texel, sin_polar);
                                const float pe = sky_density(sky_tables, texel, sin_polar);
                                p.light = p.light + (sk * p.contrib) * (ray_density / (ray_density + pe));
                            } else {
                                p.light = p.light + sky_eval(a.r, cur.d) * p.contrib;
                            }
                            a.values[(size_t)p.smp * npix + p.pix] = make_float4(p.light.x, p.light.y, p.light.z, 0.0f);
                        }
                        busy = false;
                    }
                } else {  // shade the hit of bounce b (integrator.rs:50-81)
                    fresh = false;
                    const uint32_t skey = sample_key(a.r.seed_hi, pixel_key(a.r.seed_lo, p.pix), a.r.sample_base + a.batch_first + p.smp);
                    const float4 m_brdf = load16(a.r.shade, 2u * (uint32_t)p.idx);
                    const float4 m_emit = load16(a.r.shade, 2u * (uint32_t)p.idx + 1u);
                    const int2 start = a.r.bvh.sphere_start[p.idx];  // the next ray starts at this sphere's leaf and climbs (lbvh.h)
                    p.light = p.light + mk3(m_emit.x, m_emit.y, m_emit.z) * p.contrib;
                    const float r1 = draw(skey, 3 * b);
                    const float r2 = draw(skey, 3 * b + 1);
                    const f3 wi = sample_hemisphere(p.nrm, r1, r2);
                    const float cos_theta = __builtin_fmaxf(dot(wi, p.nrm), 0.0f);
                    p.contrib = p.contrib * scatter_weight(mk3(m_brdf.x, m_brdf.y, m_brdf.z), cos_theta, fast);
                    bool done = false;
                    if (b >= R) {
                        const float pr = __builtin_fmaxf(p.contrib.x, __builtin_fmaxf(p.contrib.y, p.contrib.z));
                        if (draw(skey, 3 * b + 2) > pr) {
                            done = true;
                        } else {
                            p.contrib = roulette_divide(p.contrib, pr, fast);
                        }
                    }
                    if (!done && b + 1 < B) {
                        const f3 o = p.pos + p.nrm * kEps;
                        bool shadow_first = false;
                        if constexpr (SKYMIS) {
                            ray_density = bsdf_density_r(sqrt_cr(r2));
                            uint32_t ix, iy;
                            f3 ws;
                            float sin_polar;
                            sky_sample(sky_tables, a.r.sky_w, a.r.sky_h, draw(skey, kSkyDrawBase + 2 * b), draw(skey, kSkyDrawBase + 2 * b + 1),
                                       ix, iy, ws, sin_polar);
                            const float c = dot(ws, p.nrm);
                            if (c > 0.0f && sin_polar > 0.0f) {
                                const size_t texel = (size_t)iy * a.r.sky_w + ix;
                                const float pe = sky_density(sky_tables, texel, sin_polar);
                                const float pa = bsdf_density_dir(c);
                                f3 sk;
                                if (a.r.sky_filter == 1) {
                                    sk = sky_bilinear(a.r.skybox, a.r.sky_w, a.r.sky_h, ws);
                                } else {
                                    const float4 tx = a.r.skybox[texel];
                                    sk = mk3(tx.x, tx.y, tx.z);
                                }
                                sky_term = (sk * p.contrib) * (pa / (pe + pa));
                                wi_next = wi;
                                shadow = true;
                                shadow_first = true;
                                cursor_begin_at(cur, o, ws, start);
                            }
                        }
                        if (!shadow_first) cursor_begin_at(cur, o, wi, start);
                        traversing = true;
                    } else {  // roulette / bounce budget: no sky term (integrator.rs:47,74,92)
                        a.values[(size_t)p.smp * npix + p.pix] = make_float4(p.light.x, p.light.y, p.light.z, 0.0f);
                        busy = false;
                    }
                }
            }
            // ---- compacted append of the survivors (whole wave takes part)
            const unsigned long long smask = __builtin_amdgcn_ballot_w64(survives);
            if (smask != 0ull) {
                const uint32_t n = (uint32_t)__builtin_popcountll(smask);
                if (out_end - out_next < n) {
                    if (out_next + lane < out_end) a.qout.idx[out_next + lane] = -1;  // close the old chunk (< 64 slots left)
                    uint32_t start = 0;
                    if (lane == 0) start = atomicAdd(a.qout.count, kWfChunk);
                    out_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
                    out_end = out_next + kWfChunk;
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(smask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smask, 0u));
                if (survives) {
                    if (out_next + rank < a.qout.cap) {
                        store_path(a.qout, out_next + rank, p);
                    } else {
                        atomicOr(a.overflow, 1u);  // cannot happen with queues sized by wavefront_queue_slots
                    }
                }
                out_next += n;
            }
        } else if (n_inner >= n_leaf) {
            for (uint32_t k = 0; k < 2; ++k) {
                if (traversing && cur.cur >= 0) {
                    uint32_t nv = 0;
                    cursor_inner_step<false>(a.r.bvh, cur, stack, 256u, nv);
                    if (cur.cur == kCursorDone) traversing = false;
                }
            }
        } else {
            if (at_leaf) {
                uint32_t c0 = 0, c1 = 0;
                cursor_leaf_step<false>(a.r.bvh, cur, stack, 256u, c0, c1);
                if (cur.cur == kCursorDone) traversing = false;
            }
        }
    }
    // close the last output chunk
    for (uint32_t i = out_next + lane; i < out_end; i += 64u)
        if (i < a.qout.cap) a.qout.idx[i] = -1;
}

// bounces == 0: the bounce loop never runs, every sample is (0,0,0) (integrator.rs:47,92)
__global__ __launch_bounds__(256) void wf_zero_values(const WfArgs a) {
    uint32_t x, y;
    if (!lane_pixel(blockIdx.x, a.r.tiles_x, a.r.tile_rank, a.r.tile_world, a.r.w, a.r.h, x, y)) return;
    const size_t npix = (size_t)a.r.w * a.r.h;
    for (uint32_t j = 0; j < a.batch_n; ++j) a.values[(size_t)j * npix + y * a.r.w + x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__global__ __launch_bounds__(256) void wf_commit(const WfArgs a) {
    uint32_t x, y;
    if (!lane_pixel(blockIdx.x, a.r.tiles_x, a.r.tile_rank, a.r.tile_world, a.r.w, a.r.h, x, y)) return;
    const uint32_t pixel = y * a.r.w + x;
    const size_t npix = (size_t)a.r.w * a.r.h;
    float4 t = a.first_batch ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : a.tileacc[pixel];
    for (uint32_t j = 0; j < a.batch_n; ++j) {
        const float4 v = a.values[(size_t)j * npix + pixel];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += 1.0f;
    }
    if (a.last_batch) {  // Accumulator::merge_tile: framebuffer += tile, sample_counts += samples of the block
        float4 f = a.r.sum[pixel];
        f.x += t.x; f.y += t.y; f.z += t.z; f.w += t.w;
        a.r.sum[pixel] = f;
        a.r.counts[pixel] += a.block_spp;
    } else {
        a.tileacc[pixel] = t;
    }
}

}  // namespace

// Queue slots that `paths` paths per batch can need: the persistent LBVH stage hands out output slots in chunks of
// kWfChunk per wave and abandons fewer than 64 slots when it closes a chunk (<= 1/3 extra), plus one open chunk per wave.
size_t wavefront_queue_slots(size_t paths, uint32_t bounce_wgs) {
    return paths + paths / 3 + (size_t)bounce_wgs * 4 * kWfChunk + kWfChunk;
}

hipError_t init_wavefront_kernels() {
    const int max_lds = (int)(kMaxLdsSpheres * sizeof(float4));
    hipError_t e = hipFuncSetAttribute((const void*)wf_primary<kAccelLinear>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    const void* scans[] = {(const void*)wf_bounce<kAccelLinear, true, false>, (const void*)wf_bounce<kAccelLinear, false, false>,
                           (const void*)wf_bounce<kAccelLinear, true, true>, (const void*)wf_bounce<kAccelLinear, false, true>};
    for (const void* f : scans) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Renders a.r.spp samples per owned pixel (sample indices a.r.sample_base ...) into a.r.sum / a.r.counts.
// `bounce_wgs` = workgroups of a bounce stage (a few per CU).
hipError_t launch_wavefront(const WfArgs& a_in, int accel, uint32_t per_batch, uint32_t bounce_wgs, hipStream_t stream) {
    WfArgs a = a_in;
    const uint32_t owned = owned_tiles(a.r.n_tiles, a.r.tile_rank, a.r.tile_world);
    if (owned == 0 || a.r.spp == 0) return hipSuccess;
    const dim3 block(256), pix_grid(owned * kBlocksPerTile);
    if (per_batch == 0 || wavefront_queue_slots((size_t)owned * kTile * kTile * per_batch, bounce_wgs) > a.qin.cap)
        return hipErrorInvalidValue;  // the host sizes the queues with wavefront_queue_slots
    const size_t lds = accel == kAccelLbvh ? (size_t)a.r.bvh.stack_depth * 256 * sizeof(int) : (size_t)a.r.n_spheres * sizeof(float4);
    const bool skymis = a.r.sky_marginal != nullptr;  // sky importance sampling tables present (pt_kernels.h)
    hipError_t e;

    if (accel == kAccelLbvh) {
        hipLaunchKernelGGL(wf_primary<kAccelLbvh>, pix_grid, block, lds, stream, a);
    } else {
        hipLaunchKernelGGL(wf_primary<kAccelLinear>, pix_grid, block, lds, stream, a);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;

    const uint32_t total = a.r.spp, sb = a.r.sample_block, base0 = a.r.sample_base;
    for (uint32_t blk_first = 0; blk_first < total; blk_first += sb) {
        const uint32_t blk_n = total - blk_first < sb ? total - blk_first : sb;
        a.block_spp = blk_n;
        for (uint32_t done = 0; done < blk_n; done += per_batch) {
            a.batch_n = blk_n - done < per_batch ? blk_n - done : per_batch;
            a.batch_first = blk_first + done;
            a.first_batch = done == 0;
            a.last_batch = done + a.batch_n >= blk_n;
            a.r.sample_base = base0;
            WfQueue q0 = a_in.qin, q1 = a_in.qout;
            if (a.r.bounces == 0) {
                hipLaunchKernelGGL(wf_zero_values, pix_grid, block, 0, stream, a);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            }
            for (uint32_t d = 0; d < a.r.bounces; ++d) {
                a.depth = d;
                a.qin = q0;
                a.qout = q1;
                if ((e = hipMemsetAsync(q1.count, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
                if (accel == kAccelLbvh) {
                    if ((e = hipMemsetAsync(a.in_cursor, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
                    if (skymis) {
                        if (d == 0) hipLaunchKernelGGL((wf_bounce_tree<true, true>), dim3(bounce_wgs), block, lds, stream, a);
                        else hipLaunchKernelGGL((wf_bounce_tree<false, true>), dim3(bounce_wgs), block, lds, stream, a);
                    } else {
                        if (d == 0) hipLaunchKernelGGL((wf_bounce_tree<true, false>), dim3(bounce_wgs), block, lds, stream, a);
                        else hipLaunchKernelGGL((wf_bounce_tree<false, false>), dim3(bounce_wgs), block, lds, stream, a);
                    }
                } else {
                    if (skymis) {
                        if (d == 0) hipLaunchKernelGGL((wf_bounce<kAccelLinear, true, true>), dim3(bounce_wgs), block, lds, stream, a);
                        else hipLaunchKernelGGL((wf_bounce<kAccelLinear, false, true>), dim3(bounce_wgs), block, lds, stream, a);
                    } else {
                        if (d == 0) hipLaunchKernelGGL((wf_bounce<kAccelLinear, true, false>), dim3(bounce_wgs), block, lds, stream, a);
                        else hipLaunchKernelGGL((wf_bounce<kAccelLinear, false, false>), dim3(bounce_wgs), block, lds, stream, a);
                    }
                }
                if ((e = hipGetLastError()) != hipSuccess) return e;
                WfQueue t = q0;
                q0 = q1;
                q1 = t;
            }
            hipLaunchKernelGGL(wf_commit, pix_grid, block, 0, stream, a);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace pt
