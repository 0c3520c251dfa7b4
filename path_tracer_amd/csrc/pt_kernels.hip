// pt_kernels.hip — gfx950 (CDNA4, wave64) kernels of the InsploRay integrator.
//
// primary_kernel -> paths_kernel | paths_lbvh_kernel -> merge_kernel ("Schedule" below): a lane runs TASKS, one sample
// block of one pixel each, its samples in order (regeneration loop): when a path ends, the lane adds it to the block sum
// and restarts from the pixel's primary hit with the next sample index.  The primary ray has no jitter
// (pinhole_camera.rs:81-84), so ray generation and the first closest-hit are shared by all samples of the pixel;
// everything after it is per sample.  Block sums are formed in exactly the order of the reference (one sample per
// render() call, tile_buffer.rs:24-28 -> full_buffer.rs:107-125), which makes them bit-comparable to the CPU oracle.
// The sphere SoA buffer is staged in LDS once per workgroup and read wave-uniformly (LDS broadcast).
// A launch of a single block per pixel (the frontend's 1-spp redraw) is one kernel instead: paths_kernel<.., DIRECT>, a
// lane per pixel through the same loop body.
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

// ----------------------------------------------------------------------------------------------------
// Schedule.  A render launch is three kernels:
//   primary_kernel : one lane per owned pixel.  Ray generation and the first closest hit have no random input
//                    (pinhole_camera.rs:81-84: no pixel jitter), so they are evaluated once per pixel and launch.  A pixel
//                    whose primary ray escapes is finished right here: every sample is the same sky value, its block
//                    sums are formed by the same sequential adds and written out.  A pixel that hits something gets a
//                    32-byte record (hit point, pixel, normal, sphere) in the compact list of path pixels;
//   paths_kernel / paths_lbvh_kernel : persistent waves.  A TASK is one sample block of one path pixel (the reference's
//                    TileAccumulator for that pixel, tile_buffer.rs:24-28: summed from zero in sample order).  A lane
//                    takes a task, runs its samples in order with path regeneration (when a path ends, the lane adds it
//                    to the block sum and restarts at the pixel's primary hit with the next sample index), writes the
//                    block sum and takes the next task — lanes never wait for the slowest pixel of their wave or of
//                    their workgroup, and a frame of any size fills the chip.  Waves draw tasks in chunks of 64 (the
//                    first chunk of a wave is static, the rest come from 16 queues: one atomic per 64 tasks) and hand
//                    them to their lanes by ballot + mbcnt;
//   merge_kernel   : block sums into the accumulator, in block order (full_buffer.rs:107-125).
// Which lane runs a task never enters the arithmetic: results equal the one-lane-per-pixel formulation bit for bit.
// ----------------------------------------------------------------------------------------------------
constexpr uint32_t kTaskChunk = 64;

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {  // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Wave-uniform task dispenser.  The task ids [0, n_tasks) are cut into kTaskQueues contiguous ranges with a cursor each
// (pt_kernels.h): atomics on ONE address are served one after the other chip-wide (measured: 3 .. 10 ns each, so the 8 192
// first requests of a launch alone kept its last wave waiting for up to 80 us, and 4-sample blocks ran at a third of the
// rate of 16-sample blocks).  A wave starts on the queue of its workgroup (blockIdx mod kTaskQueues), draws chunks of 64
// ids from it and moves on to the next queue when one is exhausted; after n_queues exhausted queues it is done.  (Reading
// all cursors with one load to skip exhausted queues was tried: an empty launch drops from 88 to 38 us, but C1 gets 25 %
// slower — the loads see stale cursors and send waves to queues that are gone; tools/ab_c1.sh.)
// `want`: this lane asks for a task.  Every lane of the wave must call it together.
enum { kTaskNone = 0, kTaskGot = 1, kTaskExhausted = -1 };  // kTaskNone: not asked, or ask again (the chunk was short)
struct TaskPool {
    uint32_t next = 0, end = 0;     // wave-uniform: unclaimed ids [next, end) of the wave's current chunk
    uint32_t queue = 0, tried = 0;  // current queue; queues this wave has found exhausted
    uint32_t dyn_base = 0;          // ids below it were dealt out statically (task_pool_begin); the queues hold the rest
};
// The first chunk of every wave is its own: wave w of the grid starts with the ids [64 w, 64 w + 64) without asking
// anybody (a launch's first 7 168 requests would otherwise queue up behind 16 addresses, and a 1-spp interactive frame,
// whose tasks number about as many as the resident lanes, needs no atomic at all).  The queues hold what is left.
__device__ __forceinline__ TaskPool task_pool_begin(uint32_t n_tasks, uint32_t n_queues) {
    TaskPool pool;
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t wave = blockIdx.x * waves_per_wg + (threadIdx.x >> 6);
    const uint64_t capacity = (uint64_t)gridDim.x * waves_per_wg * kTaskChunk;
    pool.dyn_base = capacity < n_tasks ? (uint32_t)capacity : n_tasks;
    const uint64_t first = (uint64_t)wave * kTaskChunk;
    if (first < pool.dyn_base) {
        pool.next = (uint32_t)first;
        pool.end = first + kTaskChunk < pool.dyn_base ? (uint32_t)first + kTaskChunk : pool.dyn_base;
    }
    pool.queue = blockIdx.x & (n_queues - 1u);
    return pool;
}
__device__ __forceinline__ int take_task(TaskPool& pool, uint32_t* counters, uint32_t* s_exhausted, uint32_t n_queues, uint32_t n_tasks, bool want,
                                         uint32_t& task) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
    if (m == 0ull) return kTaskNone;
    const uint32_t n = (uint32_t)__builtin_popcountll(m), avail = pool.end - pool.next;
    uint32_t fresh = 0, fresh_n = 0;
    if (n > avail) {  // wave-uniform: one more chunk (n <= 64 = kTaskChunk)
        const uint32_t n_dyn = n_tasks - pool.dyn_base;  // ids of the queues, numbered from dyn_base
        const uint32_t per = task_queue_span(n_dyn, n_queues);
        const int leader = __builtin_ctzll(m);
        while (pool.tried < n_queues) {
            // What the workgroup's other waves already found out saves the failing atomic: at the end of a launch every
            // wave probes every queue once, 7 168 atomics per address at ~12 ns each (86 us for a launch without tasks);
            // with the bits shared in LDS it is one probe per workgroup.
            const uint32_t known = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t*)s_exhausted);
            // (a queue without ids — a launch smaller than the queues — needs no probe either.  One word in memory for
            // what OTHER workgroups found was tried too: its atomics serialise like the cursors', C1 +50 % time.)
            if (((known >> pool.queue) & 1u) || pool.queue * per >= n_dyn) {
                pool.queue = (pool.queue + 1u) & (n_queues - 1u);
                pool.tried++;
                continue;
            }
            uint32_t v = 0;
            if ((int)(threadIdx.x & 63u) == leader) v = atomicAdd(counters + (1u + pool.queue) * kTaskCounterStride, kTaskChunk);
            v = (uint32_t)__builtin_amdgcn_readlane((int)v, leader);
            const uint32_t qbeg = pool.queue * per;  // (< 2^32: launch_render bounds n_tasks)
            if (qbeg < n_dyn) {
                const uint32_t qlen = (n_dyn - qbeg < per) ? n_dyn - qbeg : per;
                if (v < qlen) {
                    fresh = pool.dyn_base + qbeg + v;
                    fresh_n = (qlen - v < kTaskChunk) ? qlen - v : kTaskChunk;
                    break;
                }
            }
            if ((int)(threadIdx.x & 63u) == leader) atomicOr(s_exhausted, 1u << pool.queue);
            pool.queue = (pool.queue + 1u) & (n_queues - 1u);
            pool.tried++;
        }
    }
    const uint32_t rank = lane_rank(m);
    bool got = true;
    if (rank < avail) {
        task = pool.next + rank;
    } else {
        task = fresh + (rank - avail);
        got = rank - avail < fresh_n;
    }
    if (n > avail) {
        const uint32_t used = (n - avail < fresh_n) ? n - avail : fresh_n;
        pool.next = fresh + used;
        pool.end = fresh + fresh_n;
    } else {
        pool.next += n;
    }
    if (!want) return kTaskNone;
    if (got) return kTaskGot;
    return pool.tried >= n_queues ? kTaskExhausted : kTaskNone;
}

// A pixel whose primary ray escapes (direction d0; disc0 = discriminant >= 0 tests of that trace): the launch's samples
// of the pixel, merged into the accumulator.
template <bool STATS>
__device__ __forceinline__ void finish_sky_pixel(const RenderArgs& a, uint32_t pixel, f3 d0, uint32_t disc0) {
    // every sample of this pixel is sky(d0) * (1,1,1) added to a zero light (integrator.rs:84-89).  Its block sums are
    // the same sequential adds a path lane would make (one full block, and the shorter last one on the way), and they
    // are merged right here, in block order like merge_kernel does: a sky pixel never enters the task list.
    f3 light = mk3(0.0f, 0.0f, 0.0f);
    if (a.bounces != 0) light = light + sky_eval(a, d0) * mk3(1.0f, 1.0f, 1.0f);
    const uint32_t n_full = a.spp / a.sample_block, rem = a.spp % a.sample_block;
    const uint32_t len = n_full ? a.sample_block : rem;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), last = acc;
    for (uint32_t s = 0; s < len; ++s) {
        acc.x += light.x; acc.y += light.y; acc.z += light.z; acc.w += 1.0f;
        if (s + 1 == rem) last = acc;
    }
    float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t n = 0;
    if (!a.fresh) {
        t = a.sum[pixel];
        n = a.counts[pixel];
    }
    for (uint32_t blk = 0; blk < n_full; ++blk) { t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w; }
    if (rem) { t.x += last.x; t.y += last.y; t.z += last.z; t.w += last.w; }
    a.sum[pixel] = t;
    a.counts[pixel] = n + a.spp;
    if (STATS) {  // algorithmic counts, as the reference would execute them: the primary ray per sample
        atomicAdd(&a.stats[0], (unsigned long long)a.spp);
        if (a.bounces != 0) {
            atomicAdd(&a.stats[1], (unsigned long long)a.spp);
            atomicAdd(&a.stats[2], (unsigned long long)a.spp * a.n_spheres);
            atomicAdd(&a.stats[3], (unsigned long long)disc0 * a.spp);
        }
    }
}

// Primary pass.  One lane per owned pixel; a workgroup of WG threads covers WG / 256 neighbouring 16x16 blocks of a tile
// (grid = owned tiles x 16 x 256 / WG).  The brute-force instantiation runs 1 024 threads: the workgroup's one atomic on the
// list length is what a 1-spp frame waits for (1080p: 8 100 workgroups of 256 queue up for ~70 us behind that address).
template <int ACCEL, bool STATS, int WG = 256>
__global__ __launch_bounds__(WG) void primary_kernel(const RenderArgs a) {
    extern __shared__ float4 s_sph[];
    constexpr uint32_t kWaves = WG / 64;
    __shared__ uint32_t s_wave_hits[kWaves], s_list_base;
    if constexpr (ACCEL == kAccelLinear) stage_spheres(s_sph, a.spheres, a.n_spheres);
    if (a.task_counters_next && blockIdx.x == 0 && threadIdx.x <= kTaskQueues) a.task_counters_next[threadIdx.x * kTaskCounterStride] = 0u;
    uint32_t x, y;
    const bool valid = block_pixel(blockIdx.x * (WG / 256) + (threadIdx.x >> 8), threadIdx.x & 255u, a.tiles_x, a.tile_rank, a.tile_world,
                                   a.w, a.h, x, y);
    const uint32_t pixel = y * a.w + x;
    Ray r0;
    r0.o = r0.d = mk3(0.0f, 0.0f, 0.0f);
    WorkCounters wc0 = {0, 0, 0};
    HitRec h0;
    h0.t = kF32Max;
    h0.idx = -1;
    if (valid) {
        if (a.rays) {
            const float* p = a.rays + 6 * (size_t)pixel;
            r0.o = mk3(p[0], p[1], p[2]);
            r0.d = mk3(p[3], p[4], p[5]);
        } else {
            r0 = pinhole_ray(a.cam, x, y, a.w, a.h);
        }
        // integrator.rs:47: with bounces == 0 the loop never runs and every sample is (0, 0, 0, 1)
        if (a.bounces != 0) h0 = trace_any<ACCEL, STATS>(a, s_sph, r0.o, r0.d, wc0);
    }
    // Places in the list of path pixels: one atomic per workgroup (all of them hit one address: a wave each would make
    // the frame's 5 000 .. 130 000 waves queue up behind it)
    const bool hit = h0.idx >= 0;
    const unsigned long long hit_mask = __builtin_amdgcn_ballot_w64(hit);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) s_wave_hits[wave] = (uint32_t)__builtin_popcountll(hit_mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t k = 0; k < kWaves; ++k) total += s_wave_hits[k];
        s_list_base = total ? atomicAdd(&a.task_counters[0], total) : 0u;
    }
    __syncthreads();
    if (!valid) return;
    if (hit) {
        f3 pos0, nrm0;
        hit_geometry(sphere_at<ACCEL>(a, s_sph, h0.idx), r0.o, r0.d, h0.t, pos0, nrm0);
        uint32_t before = 0;
        for (uint32_t k = 0; k < wave; ++k) before += s_wave_hits[k];
        const size_t at = (size_t)s_list_base + before + lane_rank(hit_mask);  // this pixel's place in the list of path pixels
        a.prim[2 * at] = make_float4(pos0.x, pos0.y, pos0.z, __builtin_bit_cast(float, pixel));
        a.prim[2 * at + 1] = make_float4(nrm0.x, nrm0.y, nrm0.z, __builtin_bit_cast(float, h0.idx));
        a.prim_disc[at] = wc0.disc;
        if (a.fresh) {  // the tasks' sums are added to a frame that starts at zero
            a.sum[pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            a.counts[pixel] = 0u;
        }
    } else {
        finish_sky_pixel<STATS>(a, pixel, r0.d, wc0.disc);
    }
    if (STATS) {  // executed acceleration-structure work of the shared primary trace: once per launch
        atomicAdd(&a.stats[5], (unsigned long long)wc0.nodes);
        atomicAdd(&a.stats[6], (unsigned long long)wc0.leaves);
        if (hit) atomicAdd(&a.stats[15], 1ull);
        if (a.bounces != 0) atomicAdd(&a.stats[16], (unsigned long long)wc0.disc);
        atomicAdd(&a.stats[17], 1ull);
    }
}

// Diagnostics: the wave's start or end time (RenderArgs::wave_times), two scalar instructions and a store when enabled
__device__ __forceinline__ void stamp_wave(const RenderArgs& a, uint32_t which) {
    if (a.wave_times) {
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wave < a.wave_times_cap && (threadIdx.x & 63u) == 0u) a.wave_times[2 * (size_t)wave + which] = __builtin_amdgcn_s_memrealtime();
    }
}

// State of the task a lane works on: a few registers; what follows from (pixel, block) is recomputed where it is needed.
struct TaskState {
    uint32_t pixel, task, s;  // task = list entry x n_blocks + block (the block sum's slot); s = samples of the block done so far
    float4 acc;               // the block sum
};
struct PrimaryHit {  // a path pixel's record from primary_kernel
    f3 pos, nrm;
    int idx;
    uint32_t pixel;
};
__device__ __forceinline__ uint32_t task_spp(const RenderArgs& a, uint32_t blk) {
    const uint32_t first = blk * a.sample_block;
    return (a.spp - first < a.sample_block) ? a.spp - first : a.sample_block;
}
// Task id -> (list entry, block).  The ids of the FULL blocks come first, pixel by pixel (a pixel's blocks are neighbours:
// the lanes of a wave then share primary hits, which the tree kernel's loads like); the ids of the short last blocks
// (spp % sample_block samples, one per pixel) come after all of them.  With one queue, a launch therefore ends on short
// tasks and drains for the length of a short task instead of a full one (tools/wave_timeline.py: the last wave of a C3
// launch of four 64-sample blocks per pixel ended 3 ms after the median wave, 9 % of the launch).
// list_from_end: the list of path pixels is dealt out from its END.  The primary pass lists the frame from image row 0 on,
// which with the reference's camera above a ground plane is the near field; the rows that border the sky hold the far field,
// where a pixel sees many spheres and its paths bounce more (tools/cost_map.py, profiles/r3_cost_map.txt: C3 11 box tests per
// path sample in the first tile row, 40 .. 45 in the last six; C2 1.12 bounce rays per sample in the first, 1.3 .. 1.6 in the
// last).  Dealt out from the start, the expensive pixels' blocks are the last full blocks and the launch waits for them (C3:
// last wave 2.9 ms after the median; from the end: 1.1 ms; C2 +2.3 %, C3 +3.5 % same-box).  Dealing from both ends towards the middle, or round-robin over 16 / 64 bands of the
// list, gained nothing or lost (profiles/r3_same_box_ab.txt).  A heuristic: the general answer is an order by measured cost.
__device__ __forceinline__ void decode_task(const RenderArgs& a, uint32_t n_entries, uint32_t task, uint32_t& entry, uint32_t& blk) {
    const uint32_t n_full = a.spp / a.sample_block, full_tasks = n_entries * n_full;
    if (task < full_tasks) {
        entry = task / n_full;
        blk = task - entry * n_full;
    } else {
        entry = task - full_tasks;
        blk = n_full;
    }
    if (a.list_from_end) entry = n_entries - 1u - entry;
}
__device__ __forceinline__ PrimaryHit load_primary(const RenderArgs& a, uint32_t entry) {
    const float4 r0 = a.prim[2 * (size_t)entry], r1 = a.prim[2 * (size_t)entry + 1];
    PrimaryHit h;
    h.pos = mk3(r0.x, r0.y, r0.z);
    h.pixel = f32_bits(r0.w);
    h.nrm = mk3(r1.x, r1.y, r1.z);
    h.idx = __builtin_bit_cast(int, r1.w);
    return h;
}
// The block sum leaves the lane: the reference's merge of one TileAccumulator pixel (full_buffer.rs:107-125), deferred to
// merge_kernel when the launch has several blocks.
// DIRECT (paths_kernel without a primary pass): nobody zeroed the pixel of a frame that starts with this launch.
// LEAN: the caller's launch always has several blocks (launch_render).
template <bool DIRECT = false, bool LEAN = false>
__device__ __forceinline__ void end_task(const RenderArgs& a, const TaskState& t) {
    if (LEAN || (!DIRECT && a.partials)) {
        a.partials[t.task] = t.acc;
    } else {  // a single block (task id = list entry): the whole launch of this pixel
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t n = 0;
        if (!DIRECT || !a.fresh) {
            v = a.sum[t.pixel];
            n = a.counts[t.pixel];
        }
        v.x += t.acc.x; v.y += t.acc.y; v.z += t.acc.z; v.w += t.acc.w;
        a.sum[t.pixel] = v;
        a.counts[t.pixel] = n + a.spp;
    }
}

// SKYMIS: sky importance sampling (IR_SKY_SAMPLING_MIS, pt_device.h): one more direction per traced bounce, drawn from
// the sky's distribution and traced as a shadow ray; both strategies weighted with the balance heuristic.
// WG: threads per workgroup.  256 everywhere, except that a scene whose spheres fill most of a CU's LDS (thousands of
// spheres: one workgroup per CU) runs workgroups of 1024 threads, so that 16 waves share the one staged copy instead of 4.
// DIRECT: a launch with a single sample block per pixel (the frontend's redraw: prepare_pixels adds one sample) has one
// task per path pixel, and list and dispenser cost more than they balance (launch_render has the measurements: 1080p at
// 1 spp 70 us instead of 160).  The DIRECT instantiation is a lane per pixel: no primary pass, no list, no counters;
// grid = 16 x owned tiles workgroups of 256; a lane traces its pixel's primary ray, finishes a sky pixel on the spot
// (finish_sky_pixel) and otherwise runs the pixel's block as its only task, through the same loop body.
// RELAXED: the instantiation without the 7-waves-per-SIMD register budget.  The budget (69 VGPRs instead of 73: +1.3 % on
// C2, +2 % on C5) is paid with scalar registers kept in vector lanes, and the bilinear sky lookup is where they are read
// back: under the bilinear filter the launch runs 2.4 % faster without the budget on the cache-resident sky and 8 % faster
// on one in HBM, which is bound by L2 misses, not by issue (C5 under an 8192x4096 sky: 109 vs 118 Gsamples/s; same-box
// runs, profiles/r3_same_box_ab.txt).  launch_render picks this one for bilinear launches.
template <bool STATS, bool SKYMIS, int WG = 256, bool DIRECT = false, bool RELAXED = false>
__global__ __launch_bounds__(WG, (WG == 256 && !SKYMIS && !STATS && !RELAXED) ? 7 : 1) void paths_kernel(const RenderArgs a) {
    extern __shared__ float4 s_sph[];
    __shared__ uint32_t s_exhausted;  // bit q: a wave of this workgroup found task queue q exhausted
    stamp_wave(a, 0);
    if (threadIdx.x == 0) s_exhausted = 0u;
    stage_spheres(s_sph, a.spheres, a.n_spheres);  // (ends with the barrier that publishes s_exhausted too)
    const uint32_t n_entries = DIRECT ? 0u : a.task_counters[0];  // path pixels (primary_kernel)
    const uint32_t n_tasks = n_entries * a.n_blocks;              // x sample blocks
    TaskPool pool;
    if constexpr (!DIRECT) pool = task_pool_begin(n_tasks, a.n_queues);

    // per-lane work statistics (STATS only): samples, traces, discriminant >= 0 tests, hits
    uint32_t n_samples = 0, n_traces = 0, n_disc = 0, n_hits = 0;
    WorkCounters wc = {0, 0, 0};
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

    const uint32_t B = a.bounces, R = a.rr_start;
    // The plain instantiation (the one the benchmark launches run) is compiled for what those launches are: several sample
    // blocks per pixel, a shading table inside the short divide sequences' range, the nearest-texel sky.  launch_render
    // sends every other launch to the general instantiation (RELAXED); each case dropped here took registers with it.
    constexpr bool kLean = !STATS && !SKYMIS && WG == 256 && !DIRECT && !RELAXED;
    const bool fast = kLean || a.fast_materials != 0;
    const SkyTables sky_tables = {a.sky_marginal, a.sky_cond, a.sky_pdf, a.sky_marginal_guide, a.sky_cond_guide};
    TaskState t;
    t.pixel = t.task = t.s = 0;
    t.acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    PrimaryHit p0;  // of the task's pixel; three spheres leave the registers for it
    p0.pos = p0.nrm = mk3(0.0f, 0.0f, 0.0f);
    p0.idx = 0;
    p0.pixel = 0;
    uint32_t p0_disc = 0;  // tests with discriminant >= 0 of the primary trace (statistics only)
    uint32_t t_spp = 0, t_key = 0, t_base = 0;  // samples of the block, pixel key, first sample index of the block
    bool have = false, finished = false, start = false;
    // path state
    uint32_t b = 0, skey = 0;
    f3 pos = p0.pos, nrm = p0.nrm;
    int idx = 0;
    f3 contrib = mk3(1.0f, 1.0f, 1.0f), light = mk3(0.0f, 0.0f, 0.0f);

    if constexpr (DIRECT) {  // the lane's pixel is its one task (what primary_kernel does for the listed launches)
        static_assert(WG == 256, "a 16x16 block per workgroup");
        if (a.task_counters_next && blockIdx.x == 0 && threadIdx.x <= kTaskQueues) a.task_counters_next[threadIdx.x * kTaskCounterStride] = 0u;
        finished = true;
        uint32_t x, y;
        if (lane_pixel(blockIdx.x, a.tiles_x, a.tile_rank, a.tile_world, a.w, a.h, x, y)) {
            const uint32_t pixel = y * a.w + x;
            Ray r0;
            if (a.rays) {
                const float* p = a.rays + 6 * (size_t)pixel;
                r0.o = mk3(p[0], p[1], p[2]);
                r0.d = mk3(p[3], p[4], p[5]);
            } else {
                r0 = pinhole_ray(a.cam, x, y, a.w, a.h);
            }
            HitRec h0;
            h0.t = kF32Max;
            h0.idx = -1;
            uint32_t disc0 = 0;
            if (a.bounces != 0) h0 = trace_linear<STATS>(s_sph, a.n_spheres, r0.o, r0.d, disc0);
            if (STATS) {  // (as primary_kernel counts them)
                if (h0.idx >= 0) atomicAdd(&a.stats[15], 1ull);
                atomicAdd(&a.stats[16], (unsigned long long)disc0);
                atomicAdd(&a.stats[17], 1ull);
            }
            if (h0.idx >= 0) {
                hit_geometry(s_sph[h0.idx], r0.o, r0.d, h0.t, p0.pos, p0.nrm);
                p0.idx = h0.idx;
                p0.pixel = t.pixel = pixel;
                p0_disc = disc0;
                t_spp = a.spp;  // (launch_render: a single block)
                t_key = pixel_key(a.seed_lo, pixel);
                t_base = a.sample_base;
                have = true;
                start = true;
            } else {
                finish_sky_pixel<STATS>(a, pixel, r0.d, disc0);
            }
        }
    }

    for (;;) {
        // ---- lanes without a task take the next one
        uint32_t task = 0;
        const bool want = !have && !finished;
        int took = kTaskNone;
        if constexpr (!DIRECT) took = take_task(pool, a.task_counters, &s_exhausted, a.n_queues, n_tasks, want, task);
        // (a lane's three outcomes as one branch and one mask operation: a three-way branch costs the loop 40 register
        // copies per iteration)
        finished = finished || took == kTaskExhausted;
        if (took == kTaskGot) {
            uint32_t entry, blk;
            decode_task(a, n_entries, task, entry, blk);
            t.task = entry * a.n_blocks + blk;  // where the block sum goes: a pixel's blocks side by side (merge_kernel)
            p0 = load_primary(a, entry);
            if (STATS) p0_disc = a.prim_disc[entry];
            t.pixel = p0.pixel;
            t.s = 0;
            t.acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            t_spp = task_spp(a, blk);
            t_key = pixel_key(a.seed_lo, t.pixel);
            t_base = a.sample_base + blk * a.sample_block;
            have = true;
            start = true;
        }
        if (__builtin_amdgcn_ballot_w64(have || !finished) == 0ull) break;  // every lane ran out of tasks
        if (have) {
            profile(0);
            if (start) {  // a sample starts at the pixel's primary hit (one place for the first and the later ones)
                start = false;
                b = 0;
                skey = sample_key(a.seed_hi, t_key, t_base + t.s);
                pos = p0.pos; nrm = p0.nrm; idx = p0.idx;
                contrib = mk3(1.0f, 1.0f, 1.0f);
                light = mk3(0.0f, 0.0f, 0.0f);
                if (STATS) { n_traces++; n_disc += p0_disc; n_hits++; }
            }
            // ---- shade the hit of bounce b (integrator.rs:50-81)
            const float4 m_brdf = load16(a.shade, 2u * (uint32_t)idx);
            const float4 m_emit = load16(a.shade, 2u * (uint32_t)idx + 1u);
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
                            const HitRec hs = trace_linear<STATS>(s_sph, a.n_spheres, o, ws, wc.disc);
                            if (STATS) { n_traces++; n_disc += wc.disc - before; }
                            if (hs.idx < 0) {
                                f3 sk;
                                if (a.sky_filter == 1) {
                                    sk = sky_bilinear(a.skybox, a.sky_w, a.sky_h, ws);
                                } else {
                                    const float4 tx = a.skybox[texel];
                                    sk = mk3(tx.x, tx.y, tx.z);
                                }
                                light = light + (sk * contrib) * wgt;
                            }
                        }
                    }
                    const uint32_t before = wc.disc;
                    const HitRec h = trace_linear<STATS>(s_sph, a.n_spheres, o, d, wc.disc);
                    if (STATS) { n_traces++; n_disc += wc.disc - before; }
                    if (h.idx >= 0) {
                        profile(2);
                        if (STATS) n_hits++;
                        hit_geometry(s_sph[h.idx], o, d, h.t, pos, nrm);
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
                            // (the plain instantiation only ever runs nearest-texel launches: launch_render)
                            light = light + sky_eval<STATS || SKYMIS || WG != 256 || DIRECT || RELAXED>(a, d) * contrib;
                        }
                        done = true;
                    }
                } else {
                    done = true;  // bounce budget exhausted: no sky term (integrator.rs:47,92)
                }
            }
            if (done) {
                t.acc.x += light.x; t.acc.y += light.y; t.acc.z += light.z; t.acc.w += 1.0f;
                ++t.s;
                if (t.s == t_spp) {
                    end_task<DIRECT, kLean>(a, t);
                    if (STATS) n_samples += t_spp;
                    have = false;
                } else {  // regenerate at the primary hit with the next sample index
                    start = true;
                }
            }
        }
    }

    stamp_wave(a, 1);
    if (STATS) {
        // algorithmic counts, as the reference would execute them (primary ray per sample)
        atomicAdd(&a.stats[0], (unsigned long long)n_samples);
        atomicAdd(&a.stats[1], (unsigned long long)n_traces);
        atomicAdd(&a.stats[2], (unsigned long long)n_traces * a.n_spheres);
        atomicAdd(&a.stats[3], (unsigned long long)n_disc);
        atomicAdd(&a.stats[4], (unsigned long long)n_hits);
        for (int k = 0; k < 8; ++k)
            if (prof[k]) atomicAdd(&a.stats[7 + k], (unsigned long long)prof[k]);
    }
}

// ----------------------------------------------------------------------------------------------------
// LBVH integrator: same tasks, same sums.  With a tree, the length of a traversal varies wildly between the rays of one
// wave (measured on the 10k-sphere scene: 10 % of the lanes active when every lane waits for the slowest ray of each
// bounce).  Here a lane is either TRAVERSING (it advances its own ray by one node per step) or WAITING for the shading
// block (finish the previous ray: hit point or sky + sample accumulation + regeneration or the next task; then emission,
// bounce sampling, roulette and the next ray).  The wave runs traversal steps for whoever is traversing and runs the
// shading block only once enough lanes wait for it (or nobody is left traversing), so short rays do not sit out long
// ones: they go on to their next ray.
// ----------------------------------------------------------------------------------------------------
#ifndef IR_SHADE_BATCH
#define IR_SHADE_BATCH 32
#endif
#ifndef IR_STEPS_PER_ROUND
#define IR_STEPS_PER_ROUND 3
#endif
constexpr uint32_t kShadeBatch = IR_SHADE_BATCH;         // run the shading block when this many lanes wait for it
constexpr uint32_t kStepsPerRound = IR_STEPS_PER_ROUND;  // traversal steps between two looks at the wave's state

// SKYMIS (sky importance sampling, see paths_kernel): the shadow ray is one more traversal of the lane's state machine,
// run before the bounce ray, which waits in registers (direction, density, the strategy's weighted sky term).
// MAY_BILINEAR = false: the plain instantiation, compiled like paths_kernel's for what the benchmark launches are (several
// blocks per pixel, a shading table in the short divide sequences' range, no bilinear sky filter); launch_render sends
// every other launch to the general one.
template <bool STATS, bool SKYMIS, bool MAY_BILINEAR = true>
__global__ __launch_bounds__(256, SKYMIS ? 4 : 6) void paths_lbvh_kernel(const RenderArgs a) {
    extern __shared__ float4 s_dyn[];  // traversal stacks: entry k of thread t at [k * 256 + t]
    int* const stack = reinterpret_cast<int*>(s_dyn) + threadIdx.x;
    __shared__ uint32_t s_exhausted;  // bit q: a wave of this workgroup found task queue q exhausted
    stamp_wave(a, 0);
    if (threadIdx.x == 0) s_exhausted = 0u;
    __syncthreads();
    const uint32_t n_entries = a.task_counters[0];
    const uint32_t n_tasks = n_entries * a.n_blocks;
    TaskPool pool = task_pool_begin(n_tasks, a.n_queues);
    uint32_t n_samples = 0, n_traces = 0, n_disc = 0, n_hits = 0;
    WorkCounters wc = {0, 0, 0};
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
    const uint32_t B = a.bounces, R = a.rr_start;
    constexpr bool kLean = !MAY_BILINEAR;
    const bool fast = kLean || a.fast_materials != 0;
    const SkyTables sky_tables = {a.sky_marginal, a.sky_cond, a.sky_pdf, a.sky_marginal_guide, a.sky_cond_guide};

    // The task's state (pixel, block, samples done, block sum) is touched once per sample: it lives in LDS behind the
    // traversal stacks, not in 7 registers (80 VGPRs = 6 waves per SIMD is what this kernel lives on).  For the same
    // reason the shading point is not kept: the next ray's origin, pos + nrm * EPSILON (integrator.rs:80), goes straight
    // into the cursor.  (The pixel's primary-hit record IS kept, see p0 below: the registers are there since the 48-byte
    // records, and loads are what this kernel is short of.)
    uint32_t* const s_task = reinterpret_cast<uint32_t*>(s_dyn) + (size_t)a.bvh.stack_depth * 256 + threadIdx.x;
    auto task_word = [&](int k) -> uint32_t& { return s_task[k * 256]; };  // 0 list entry, 1 block, 2 samples done, 3..6 sum
    bool have = false, finished = false;
    uint32_t b = 0, skey = 0;
    f3 nrm = mk3(0.0f, 0.0f, 0.0f);
    int idx = 0;
    f3 contrib = mk3(1.0f, 1.0f, 1.0f), light = mk3(0.0f, 0.0f, 0.0f);
    LbvhCursor cur;
    // The primary hit of the task's pixel stays in registers for the task's samples: reading the record again for every
    // sample was two 16-byte loads through the L1 data path, which is what this kernel waits for (C3 +1.3 %).
    PrimaryHit p0;
    p0.pos = p0.nrm = mk3(0.0f, 0.0f, 0.0f);
    p0.idx = 0;
    p0.pixel = 0;
    auto restart = [&]() {  // a sample starts at the pixel's primary hit
        const uint32_t entry = task_word(0), blk = task_word(1), done = task_word(2);
        b = 0;
        cur.o = p0.pos + p0.nrm * kEps;
        nrm = p0.nrm;
        idx = p0.idx;
        contrib = mk3(1.0f, 1.0f, 1.0f);
        light = mk3(0.0f, 0.0f, 0.0f);
        skey = sample_key(a.seed_hi, pixel_key(a.seed_lo, p0.pixel), a.sample_base + blk * a.sample_block + done);
        if (STATS) { n_traces++; n_disc += a.prim_disc[entry]; n_hits++; }
    };
    cursor_begin(cur, mk3(0, 0, 0), mk3(0, 0, 1));
    cur.cur = kCursorDone;
    bool traversing = false;  // cur holds an unfinished traversal
    bool landed = false;      // cur holds a finished traversal whose result has not been consumed yet
    // SKYMIS only: cur is the shadow ray of the sky strategy; the bounce ray (pos + nrm * kEps, wi_next) follows it
    bool shadow = false;
    f3 wi_next = mk3(0, 0, 0), sky_term = mk3(0, 0, 0);
    float ray_density = 0.0f;

    for (;;) {
        // lanes that are not traversing want the shading block: to take a task, or to go on with the one they have
        const bool waiting = !traversing && !finished;
        const uint32_t n_wait = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(waiting));
        const uint32_t n_trav = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(traversing));
        if (n_wait == 0 && n_trav == 0) break;
        if (waiting || traversing) profile(0);
        if (n_wait >= kShadeBatch || n_trav == 0) {
            uint32_t task = 0;
            const bool want = waiting && !have;
            const int took = take_task(pool, a.task_counters, &s_exhausted, a.n_queues, n_tasks, want, task);
            finished = finished || took == kTaskExhausted;
            if (took == kTaskGot) {
                uint32_t entry, blk;
                decode_task(a, n_entries, task, entry, blk);
                task_word(0) = entry;
                task_word(1) = blk;
                task_word(2) = 0u;
                task_word(3) = task_word(4) = task_word(5) = task_word(6) = 0u;  // the block sum starts at +0
                have = true;
                p0 = load_primary(a, entry);
                restart();
            }
            if (waiting && have) {
                profile(1);
                if (landed) {  // consume the finished traversal (integrator.rs:50-51, 82-90)
                    landed = false;
                    if (STATS) { n_traces++; }
                    if (SKYMIS && shadow) {
                        if (cur.idx < 0) light = light + sky_term;
                        shadow = false;
                        cursor_begin_at(cur, cur.o, wi_next, a.bvh.sphere_start[idx]);  // the bounce ray leaves the same point as the shadow ray
                        traversing = true;
                    } else if (cur.idx >= 0) {
                        if (STATS) n_hits++;
                        f3 pos;
                        idx = cursor_hit_index(a.bvh, cur);
                        // the centre out of the leaf-ordered array the winning test just read (same three floats as
                        // a.spheres[idx]; hit_geometry takes nothing else): no load that waits for idx
                        hit_geometry(a.bvh.leaf_spheres[cur.idx], cur.o, cur.d, cur.best, pos, nrm);
                        cur.o = pos + nrm * kEps;
                    } else {
                        if constexpr (SKYMIS) {
                            size_t texel;
                            float sin_polar;
                            const f3 sk = sky_lookup_mis(a.skybox, a.sky_w, a.sky_h, a.sky_filter, cur.d, texel, sin_polar);
                            const float pe = sky_density(sky_tables, texel, sin_polar);
                            light = light + (sk * contrib) * (ray_density / (ray_density + pe));
                        } else {
                            light = light + sky_eval<MAY_BILINEAR>(a, cur.d) * contrib;
                        }
                        idx = -1;  // sample finished
                    }
                }
                if (!SKYMIS || !traversing) for (;;) {
                    if (idx < 0) {  // finish the sample; regenerate at the primary hit, or hand the block sum in
                        float4 acc = make_float4(__builtin_bit_cast(float, task_word(3)), __builtin_bit_cast(float, task_word(4)),
                                                 __builtin_bit_cast(float, task_word(5)), __builtin_bit_cast(float, task_word(6)));
                        acc.x += light.x; acc.y += light.y; acc.z += light.z; acc.w += 1.0f;
                        const uint32_t done_now = task_word(2) + 1u, blk = task_word(1);
                        if (done_now >= task_spp(a, blk)) {
                            TaskState t;
                            t.pixel = f32_bits(a.prim[2 * (size_t)task_word(0)].w);
                            t.task = task_word(0) * a.n_blocks + blk;
                            t.s = done_now;
                            t.acc = acc;
                            end_task<false, kLean>(a, t);
                            if (STATS) n_samples += done_now;
                            have = false;  // the next shading block brings the next task
                            break;
                        }
                        task_word(2) = done_now;
                        task_word(3) = f32_bits(acc.x); task_word(4) = f32_bits(acc.y);
                        task_word(5) = f32_bits(acc.z); task_word(6) = f32_bits(acc.w);
                        restart();
                    }
                    // shade the hit of bounce b (integrator.rs:50-81)
                    const float4 m_brdf = load16(a.shade, 2u * (uint32_t)idx);
                    const float4 m_emit = load16(a.shade, 2u * (uint32_t)idx + 1u);
                    // the next ray leaves sphere idx: its traversal starts at that sphere's leaf and climbs (lbvh.h)
                    const int2 start = a.bvh.sphere_start[idx];
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
                                        const float4 tx = a.skybox[texel];
                                        sk = mk3(tx.x, tx.y, tx.z);
                                    }
                                    sky_term = (sk * contrib) * (pa / (pe + pa));
                                    wi_next = wi;
                                    shadow = true;
                                    cursor_begin_at(cur, cur.o, ws, start);
                                    traversing = true;
                                    break;
                                }
                            }
                            cursor_begin_at(cur, cur.o, wi, start);
                            // the start leaf right here, with every lane of the shading block, instead of as the first of
                            // the wave's thinly populated leaf steps (C3 +1.8 %, profiles/r3_same_box_ab.txt)
                            if (cur.cur < 0) cursor_leaf_step<STATS>(a.bvh, cur, stack, 256u, wc.disc, wc.leaves);
                            traversing = cur.cur != kCursorDone;
                            landed = !traversing;
                            break;
                        }
                    }
                    idx = -1;  // roulette or bounce budget ended the path: no sky term (integrator.rs:47,74,92)
                }
            }
        }
        // Two node steps per leaf step: lanes that stay on inner nodes (most of them) descend two levels for every pass
        // of the divergent leaf code, which then finds more lanes waiting for it (+7 % on C3 over one-and-one; three
        // node steps or a majority vote between the two kinds of step measured no better).  Three such steps between two
        // looks at the wave's state (8 was right while lanes owned pixels, 4 with tasks; with traversals that start at
        // their leaf and are shorter, 3: +1.6 %, 2 the same, 6 -1 %).
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

    stamp_wave(a, 1);
    if (STATS) {
        atomicAdd(&a.stats[0], (unsigned long long)n_samples);
        atomicAdd(&a.stats[1], (unsigned long long)n_traces);
        atomicAdd(&a.stats[2], (unsigned long long)n_traces * a.n_spheres);
        atomicAdd(&a.stats[3], (unsigned long long)n_disc);
        atomicAdd(&a.stats[4], (unsigned long long)n_hits);
        atomicAdd(&a.stats[5], (unsigned long long)wc.nodes);
        atomicAdd(&a.stats[6], (unsigned long long)wc.leaves);
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
// block order; sample_counts += samples of the launch.  One lane per path pixel (list entry); sky pixels were merged by
// the primary pass.
__global__ __launch_bounds__(256) void merge_kernel(const RenderArgs a) {
    const uint32_t entry = blockIdx.x * 256u + threadIdx.x;
    if (entry >= a.task_counters[0]) return;
    const uint32_t pixel = f32_bits(a.prim[2 * (size_t)entry].w);
    const float4* p = a.partials + (size_t)entry * a.n_blocks;
    float4 t = a.sum[pixel];
    // eight loads in flight, then their adds in block order: one load per add made the launch wait n_blocks memory
    // latencies in a row (C1, 16 blocks of 4: 17 us)
    uint32_t b = 0;
    for (; b + 8 <= a.n_blocks; b += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[b + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { t.x += v[k].x; t.y += v[k].y; t.z += v[k].z; t.w += v[k].w; }
    }
    for (; b < a.n_blocks; ++b) {
        const float4 v = p[b];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
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
    // a^(1/2.2) through the hardware log2 / exp2 (1 ulp each): the result becomes a byte, at most 1 LSB from libm's powf
    // (the bar of tests/test_gpu_parity.py::test_resolve_argb_within_one_lsb); the library powf was 2/3 of this kernel
    // (powf(-inf, 1/2.2) is +inf — a mean of exactly -1 — where log2 would give NaN; every other negative is NaN in both)
    a = (a == -__builtin_inff()) ? __builtin_inff() : a;
    a = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(a) * (1.0f / 2.2f));
    a = (a > 0.0f) ? a : 0.0f;  // clamp(0,1); NaN -> 0 like SSE max/min
    a = (a < 1.0f) ? a : 1.0f;
    return a;
}
__global__ __launch_bounds__(256) void resolve_kernel(const float4* __restrict__ sum, const uint32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ argb, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // the weight channel of a rendered pixel is its sample count (every sample adds 1): the mean is exactly 1 and the
    // byte is the same for all of them; one powf per thread instead of one per pixel
    const uint32_t alpha_of_one = to_byte(tone(1.0f, 1.0f));
    for (; i < n; i += stride) {
        const float4 s = sum[i];
        const uint32_t c = counts[i];
        const float nf = (float)(c > 1u ? c : 1u);
        const uint32_t alpha = (s.w == nf) ? alpha_of_one : to_byte(tone(s.w, nf));
        argb[i] = (alpha << 24) | (to_byte(tone(s.x, nf)) << 16) | (to_byte(tone(s.y, nf)) << 8) | to_byte(tone(s.z, nf));
    }
}

// Footprint table of sky_bilinear (pt_device.h): record (row, x0), row = iy + 1 in [0, h], holds the texels of rows
// y0 = clamp(row - 1), y1 = clamp(row) and columns x0, x1 = x0 + 1 (wrapping).
__global__ __launch_bounds__(256) void sky_quad_kernel(const float4* __restrict__ sky, uint32_t w, uint32_t h, float4* __restrict__ quad) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)(h + 1) * w) return;
    const uint32_t row = (uint32_t)(i / w), x0 = (uint32_t)(i % w);
    const uint32_t x1 = (x0 + 1 == w) ? 0u : x0 + 1;
    const uint32_t y0 = row == 0 ? 0u : row - 1, y1 = row > h - 1 ? h - 1 : row;
    quad[4 * i + 0] = sky[(size_t)y0 * w + x0];
    quad[4 * i + 1] = sky[(size_t)y0 * w + x1];
    quad[4 * i + 2] = sky[(size_t)y1 * w + x0];
    quad[4 * i + 3] = sky[(size_t)y1 * w + x1];
}

}  // namespace

hipError_t launch_sky_quad(const float4* sky, uint32_t w, uint32_t h, float4* quad, hipStream_t stream) {
    const size_t n = (size_t)(h + 1) * w;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sky_quad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sky, w, h, quad);
    return hipGetLastError();
}

hipError_t init_kernels() {
    hipError_t e;
    const int max_lds = (int)(kMaxLdsSpheres * sizeof(float4));
    const void* big_lds[] = {(const void*)paths_kernel<false, false>,          (const void*)paths_kernel<true, false>,
                             (const void*)paths_kernel<false, true>,           (const void*)paths_kernel<true, true>,
                             (const void*)paths_kernel<false, false, 1024>,    (const void*)paths_kernel<true, false, 1024>,
                             (const void*)paths_kernel<false, false, 256, false, true>,
                             (const void*)paths_kernel<false, false, 256, true>, (const void*)paths_kernel<true, false, 256, true>,
                             (const void*)paths_kernel<false, true, 256, true>,  (const void*)paths_kernel<true, true, 256, true>,
                             (const void*)primary_kernel<kAccelLinear, false, 1024>, (const void*)primary_kernel<kAccelLinear, true, 1024>,
                             (const void*)primary_hits_kernel<kAccelLinear>};
    for (const void* f : big_lds) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

namespace {

constexpr size_t kBigWorkgroupFromLdsBytes = 40 * 1024;

// Workgroups of a persistent launch: what the device can hold at once (CUs x resident workgroups per CU for this kernel
// and LDS size), never more than the tasks can occupy.  A workgroup too many only queues and finds the task list empty.
hipError_t persistent_grid(const void* kernel, uint32_t threads, size_t lds, uint64_t max_tasks, uint32_t* out) {
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, lds);
    if (e != hipSuccess) return e;
    if (cus < 1) cus = 1;
    if (per_cu < 1) per_cu = 1;
    uint64_t wgs = (uint64_t)cus * (uint64_t)per_cu;
    const uint64_t useful = (max_tasks + threads - 1) / threads;
    if (wgs > useful) wgs = useful;
    *out = (uint32_t)(wgs < 1 ? 1 : wgs);
    return hipSuccess;
}

}  // namespace

hipError_t launch_render(const RenderArgs& a_in, int accel, bool stats, hipStream_t stream, hipEvent_t paths_begin, hipEvent_t paths_end) {
    const uint32_t owned = owned_tiles(a_in.n_tiles, a_in.tile_rank, a_in.tile_world);
    if (owned == 0 || a_in.spp == 0) return hipSuccess;
    RenderArgs a = a_in;
    a.wgs_per_layer = owned * kBlocksPerTile;
    a.n_blocks = (a.spp + a.sample_block - 1) / a.sample_block;
    if (a.n_blocks == 1) a.partials = nullptr;  // single block: the kernels merge into the accumulator themselves
    const dim3 block(256);
    hipError_t e = hipSuccess;
    if (!a.task_counters_next) {  // the caller does not alternate counter pairs: zero the one pair
        e = hipMemsetAsync(a.task_counters, 0, kTaskCounterPairStride * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    const bool skymis = a.sky_marginal != nullptr;
    // A launch of a single block per pixel (prepare_pixels, the frontend's redraw, is one of 1 spp): one kernel, a lane per
    // pixel (paths_kernel DIRECT).  Measured against list + tasks, which then has one task per path pixel (S1 scene,
    // tools/direct_crossover.py, kernel ms): 1080p 1 spp 0.070 vs 0.160, 16 spp 0.35 vs 0.37, 128 spp 2.09 vs 2.32;
    // 4K 1 spp 0.20 vs 0.32, 128 spp 6.50 vs 6.73; 800x450 0.032 vs 0.040, 128 spp 1.16 vs 1.14.
    static const bool direct_ok = !getenv("IR_DEBUG_DIRECT") || atoi(getenv("IR_DEBUG_DIRECT")) != 0;  // tools/ only
    // (not above 40 KB of spheres: there the listed launch shares one staged copy between 16 waves, see below)
    if (accel != kAccelLbvh && a.n_blocks == 1 && direct_ok && (size_t)a.n_spheres * sizeof(float4) <= kBigWorkgroupFromLdsBytes) {
        const void* direct = stats ? (skymis ? (const void*)paths_kernel<true, true, 256, true> : (const void*)paths_kernel<true, false, 256, true>)
                                   : (skymis ? (const void*)paths_kernel<false, true, 256, true> : (const void*)paths_kernel<false, false, 256, true>);
        void* params[] = {(void*)&a};
        if (paths_begin && (e = hipEventRecord(paths_begin, stream)) != hipSuccess) return e;
        e = hipLaunchKernel(direct, dim3(a.wgs_per_layer), block, params, (size_t)a.n_spheres * sizeof(float4), stream);
        if (e != hipSuccess) return e;
        if (paths_end && (e = hipEventRecord(paths_end, stream)) != hipSuccess) return e;
        return hipSuccess;
    }
    const uint64_t max_tasks = (uint64_t)a.wgs_per_layer * 256u * a.n_blocks;
    // Queues: one address serves 100 .. 300 M atomics/s chip-wide.  Long blocks on a big frame ask rarely (C2: 10 M/s)
    // and run 1.5 % faster on a single queue (measured on one box, tools/ab_bench.sh); short tasks (a 1-spp interactive
    // frame is a task per pixel: 4K at 1 spp took 1.0 ms behind one cursor) or few tasks per wave
    // (C1: every wave asks in the first microseconds, then every few iterations) need the requests spread out.
    const uint32_t task_samples = a.spp < a.sample_block ? a.spp : a.sample_block;  // (the last block may be shorter still)
    // (One queue is also what lets a launch end on its short last blocks, decode_task.)
    a.n_queues = (task_samples >= 24 && max_tasks >= (2ull << 20)) ? 1u : kTaskQueues;
    static const int forced_order = getenv("IR_DEBUG_TASK_ORDER") ? atoi(getenv("IR_DEBUG_TASK_ORDER")) : -1;  // tools/ only: 0 = from the start
    a.list_from_end = forced_order >= 0 ? (forced_order != 0) : 1u;
    static const int forced_queues = getenv("IR_DEBUG_TASK_QUEUES") ? atoi(getenv("IR_DEBUG_TASK_QUEUES")) : 0;  // tools/ only
    if (forced_queues >= 1 && forced_queues <= (int)kTaskQueues && (forced_queues & (forced_queues - 1)) == 0) a.n_queues = (uint32_t)forced_queues;
    const void* paths = nullptr;
    uint32_t path_threads = 256;
    size_t lds = 0;
    if (accel == kAccelLbvh) {
        lds = (size_t)a.bvh.stack_depth * 256 * sizeof(int);
        if (stats) hipLaunchKernelGGL((primary_kernel<kAccelLbvh, true>), dim3(a.wgs_per_layer), block, lds, stream, a);
        else hipLaunchKernelGGL((primary_kernel<kAccelLbvh, false>), dim3(a.wgs_per_layer), block, lds, stream, a);
        lds += 7 * 256 * sizeof(uint32_t);  // paths_lbvh_kernel keeps each lane's task state behind the stacks
        static const size_t tree_pad = getenv("IR_DEBUG_LDS_PAD") ? (size_t)atol(getenv("IR_DEBUG_LDS_PAD")) : 0;  // occupancy experiments (tools/)
        lds += tree_pad;
        paths = stats ? (skymis ? (const void*)paths_lbvh_kernel<true, true> : (const void*)paths_lbvh_kernel<true, false>)
                      : (skymis ? (const void*)paths_lbvh_kernel<false, true> : (const void*)paths_lbvh_kernel<false, false>);
        if (!stats && !skymis && !(a.skybox && a.sky_filter == 1) && a.n_blocks > 1 && a.fast_materials)
            paths = (const void*)paths_lbvh_kernel<false, false, false>;
    } else {
        lds = (size_t)a.n_spheres * sizeof(float4);
        // occupancy experiments only (tools/): IR_DEBUG_LDS_PAD=<bytes> pads the workgroup's LDS allocation
        static const size_t pad = getenv("IR_DEBUG_LDS_PAD") ? (size_t)atol(getenv("IR_DEBUG_LDS_PAD")) : 0;
        if (lds + pad <= kMaxLdsSpheres * sizeof(float4)) lds += pad;
        if (stats) hipLaunchKernelGGL((primary_kernel<kAccelLinear, true, 1024>), dim3(a.wgs_per_layer / 4), dim3(1024), lds, stream, a);
        else hipLaunchKernelGGL((primary_kernel<kAccelLinear, false, 1024>), dim3(a.wgs_per_layer / 4), dim3(1024), lds, stream, a);
        paths = stats ? (skymis ? (const void*)paths_kernel<true, true> : (const void*)paths_kernel<true, false>)
                      : (skymis ? (const void*)paths_kernel<false, true> : (const void*)paths_kernel<false, false>);
        if (!stats && !skymis && ((a.skybox && a.sky_filter == 1) || a.n_blocks == 1 || !a.fast_materials))
            paths = (const void*)paths_kernel<false, false, 256, false, true>;  // what the plain instantiation leaves out
        // more than 40 KB of spheres: fewer than four 256-thread workgroups fit a CU; let 16 waves share one staged copy
        if (lds > kBigWorkgroupFromLdsBytes && !skymis) {
            paths = stats ? (const void*)paths_kernel<true, false, 1024> : (const void*)paths_kernel<false, false, 1024>;
            path_threads = 1024;
        }
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    uint32_t grid = 1;
    e = persistent_grid(paths, path_threads, lds, max_tasks, &grid);
    if (e != hipSuccess) return e;
    void* params[] = {(void*)&a};
    if (paths_begin && (e = hipEventRecord(paths_begin, stream)) != hipSuccess) return e;
    e = hipLaunchKernel(paths, dim3(grid), dim3(path_threads), params, lds, stream);
    if (e != hipSuccess) return e;
    if (paths_end && (e = hipEventRecord(paths_end, stream)) != hipSuccess) return e;
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
