// pt_kernels.h — launch interface between the C-ABI host code (insploray_hip.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_device.h"

namespace pt {

constexpr uint32_t kTile = 64;        // reference tile edge, renderer.rs:105
constexpr uint32_t kBlockEdge = 16;   // one 256-thread workgroup renders 16x16 pixels = 4 waves of 8x8
constexpr uint32_t kBlocksPerTile = (kTile / kBlockEdge) * (kTile / kBlockEdge);
constexpr uint32_t kMaxLdsSpheres = 10000;  // 10000 * 16 B = 160000 B <= 160 KiB LDS of one CU

struct RenderArgs {
    CameraBlock cam;
    const float* rays;  // generic-camera primary rays (6 floats per pixel) or nullptr => pinhole
    uint32_t w, h;

    const float4* spheres;      // (cx, cy, cz, r)
    const float4* shade;        // per sphere: [2i] = (brdf.rgb, 0) with brdf = albedo / PI, [2i+1] = (emission*power, 0)
    uint32_t n_spheres;
    // host promise about the shading table (rebuild_shade): every brdf component is +0 or in [2^-30, 2^30]; lets
    // the kernel take the short exact divide sequences of pt_device.h (results are the same either way)
    uint32_t fast_materials;

    float sky[3];
    const float4* skybox;  // rgb padded to float4, row 0 first; nullptr => constant sky
    uint32_t sky_w, sky_h;
    int sky_filter;
    // Footprint table of the bilinear lookup, or nullptr: (sky_h + 1) x sky_w records of four texels, 64 bytes each
    // (pt_device.h sky_bilinear_quad).  The host builds it for skies that no cache holds (launch_sky_quad).
    const float4* skyquad;
    // sky importance sampling tables (sky_distribution.h); all nullptr => BSDF sampling only, as the reference
    const float* sky_marginal;
    const float* sky_cond;
    const float* sky_pdf;
    const uint32_t* sky_marginal_guide;
    const uint32_t* sky_cond_guide;

    uint32_t bounces, rr_start;
    uint32_t seed_lo, seed_hi;
    uint32_t spp, sample_base;

    // Samples [sample_base, sample_base + spp) of a launch are cut into blocks of sample_block samples; a task sums one
    // block of one path pixel from zero (in sample order), the block sums are merged in block order.
    // partials: block sums by task id (list entry x n_blocks + block): room for (owned pixels) x n_blocks float4 (unused
    // when the launch has a single block).  A lane's tasks are neighbours in this array, so are the stores of a wave.
    uint32_t sample_block;
    uint32_t n_blocks, wgs_per_layer;  // filled in by launch_render (wgs_per_layer = 16 x owned tiles)
    float4* partials;
    // 1: the accumulator starts at zero with this launch (renderer.rs:99-103); the primary pass writes instead of adding,
    // which saves the two fills of a non-accumulating render.  Only valid when the launch owns every pixel.
    uint32_t fresh;

    float4* sum;       // running radiance sum (rgb) + sample weight (a)
    uint32_t* counts;  // samples per pixel

    // Launch scratch of the primary pass and the task dispenser (pt_kernels.hip "Schedule"):
    float4* prim;             // list of path pixels (primary ray hits): [2e] = (hit point, pixel bits), [2e+1] = (normal, sphere index bits)
    uint32_t* prim_disc;      // per list entry: tests with discriminant >= 0 of the primary trace (statistics)
    // one set of counters: word 0 = list length; word (1 + q) * kTaskCounterStride = cursor of task queue q
    // (task = list entry * n_blocks + block; queue q holds the ids [q, q + 1) * task_queue_span(n_tasks, n_queues))
    uint32_t* task_counters;
    uint32_t list_from_end;  // 1: tasks walk the list of path pixels from its end (pt_kernels.hip decode_task; launch_render sets it)
    uint32_t n_queues;  // queues in use, a power of two <= kTaskQueues (launch_render: one when tasks are long and many)
    // The other set of counters, zeroed by this launch's primary pass for the next launch (the context alternates between
    // two sets, which saves a fill per launch), or nullptr when the caller zeroes task_counters itself.
    uint32_t* task_counters_next;

    // tile partition: this launch renders tiles t = tile_rank + k * tile_world, k = 0 .. n_owned_tiles-1
    uint32_t tiles_x, n_tiles, tile_rank, tile_world;

    LbvhView bvh;  // used by the LBVH instantiation only

    unsigned long long* stats;  // kNumStats counters or nullptr (only read by the STATS instantiation)
    // Diagnostics (tools/wave_timeline.py, IR_DEBUG_WAVE_TIMES): when not nullptr, every wave of the path kernel stores the
    // 100 MHz realtime counter at its start and at its end, [2 * wave] and [2 * wave + 1], for waves below wave_times_cap.
    unsigned long long* wave_times;
    uint32_t wave_times_cap;
};

struct HitsArgs {
    CameraBlock cam;
    const float* rays;
    uint32_t w, h;
    const float4* spheres;
    uint32_t n_spheres;
    LbvhView bvh;
    int32_t* out_idx;
    float* out_t;  // may be nullptr
};

// ---- helpers shared by the kernel translation units (pt_kernels.hip, pt_wavefront.hip)
// MAY_BILINEAR = false: the caller's launch never has the bilinear filter selected (launch_render sends those launches to
// another instantiation), so the kernel carries neither that code nor its uniform registers.
template <bool MAY_BILINEAR = true>
PT_DEV f3 sky_eval(const RenderArgs& a, f3 dir) {
    if (a.skybox) {
        if (MAY_BILINEAR && a.sky_filter == 1)  // IR_SKY_BILINEAR
            return a.skyquad ? sky_bilinear_quad(a.skyquad, a.sky_w, a.sky_h, dir) : sky_bilinear(a.skybox, a.sky_w, a.sky_h, dir);
        return sky_nearest(a.skybox, a.sky_w, a.sky_h, dir);
    }
    return mk3(a.sky[0], a.sky[1], a.sky[2]);
}
#if defined(__HIPCC__)
// (x, y) of a lane: thread `tid` (0..255) of workgroup `wg` of a 1D grid over the 16x16 blocks of the owned 64x64 tiles;
// waves are 8x8 pixels.
__device__ __forceinline__ bool block_pixel(uint32_t wg, uint32_t tid, uint32_t tiles_x, uint32_t tile_rank, uint32_t tile_world,
                                            uint32_t w, uint32_t h, uint32_t& x, uint32_t& y) {
    const uint32_t k = wg / kBlocksPerTile;
    const uint32_t sub = wg % kBlocksPerTile;
    const uint32_t tile = tile_rank + k * tile_world;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    x = tx * kTile + (sub & 3) * kBlockEdge + (wave & 1) * 8 + (lane & 7);
    y = ty * kTile + (sub >> 2) * kBlockEdge + (wave >> 1) * 8 + (lane >> 3);
    return x < w && y < h;
}
__device__ __forceinline__ bool lane_pixel(uint32_t wg, uint32_t tiles_x, uint32_t tile_rank, uint32_t tile_world, uint32_t w,
                                           uint32_t h, uint32_t& x, uint32_t& y) {  // a 256-thread workgroup per block
    return block_pixel(wg, threadIdx.x, tiles_x, tile_rank, tile_world, w, h, x, y);
}
__device__ __forceinline__ void stage_spheres(float4* s_sph, const float4* __restrict__ g, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        float4 s = g[i];
        s.w = s.w * s.w;  // radius * radius (integrator.rs:111), same product evaluated once
        s_sph[i] = s;
    }
    __syncthreads();
}
#endif
__host__ __device__ inline uint32_t owned_tiles(uint32_t n_tiles, uint32_t rank, uint32_t world) {
    return rank < n_tiles ? (n_tiles - rank + world - 1) / world : 0;
}

// ---- wavefront pipeline (pt_wavefront.hip)
struct WfQueue {         // SoA path records in device memory
    uint32_t* pix;       // pixel index
    uint32_t* smp;       // sample index inside the batch
    int32_t* idx;        // sphere the path sits on
    float* f;            // 12 planes of `cap` floats: position, normal, contribution, light (xyz each)
    uint32_t* count;     // entries in the queue (device counter)
    uint32_t cap;
};
struct WfArgs {
    RenderArgs r;
    float4* prim_pos;    // per pixel: (primary hit point, bits of the sphere index) or (sky value, -1)
    float4* prim_nrm;    // per pixel: primary hit normal
    float4* values;      // [sample of the batch][pixel]: radiance of each finished sample
    float4* tileacc;     // [pixel]: running sum of the current sample block
    WfQueue qin, qout;
    uint32_t* in_cursor;            // next unclaimed input entry (persistent LBVH bounce stage)
    uint32_t* overflow;             // set if a queue ever overflowed (sized so that it cannot)
    uint32_t batch_first, batch_n;  // samples [batch_first, batch_first + batch_n) of the call
    uint32_t block_spp;             // samples of the current block
    uint32_t first_batch, last_batch, depth;
};
hipError_t init_wavefront_kernels();
size_t wavefront_queue_slots(size_t paths, uint32_t bounce_wgs);
hipError_t launch_wavefront(const WfArgs& a, int accel, uint32_t per_batch, uint32_t bounce_wgs, hipStream_t stream);

// All launchers enqueue on `stream` and return the HIP status of the launch.
enum { kAccelLinear = 0, kAccelLbvh = 1 };
// samples, traces, tests, disc>=0, hits, bvh node visits, bvh leaf tests, 8 x lane profile; then what the launches really
// executed of the per-pixel work: [15] path pixels (primary ray hit), [16] disc >= 0 tests of the primary traces, [17] pixels —
// each once per pixel and launch (the counts before them charge the primary ray to every sample, as the reference runs it)
constexpr uint32_t kNumStats = 18;
// Task queues of the persistent path kernels (pt_kernels.hip take_task): counters 128 bytes apart, two sets of them.
#ifndef IR_TASK_QUEUES
#define IR_TASK_QUEUES 16  // (tools/ build variants with other counts)
#endif
constexpr uint32_t kTaskQueues = IR_TASK_QUEUES, kTaskCounterStride = 32;
constexpr uint32_t kTaskCounterPairStride = (1 + kTaskQueues) * kTaskCounterStride;  // words per set
__host__ __device__ inline uint32_t task_queue_span(uint32_t n_tasks, uint32_t n_queues) {  // ids per queue: whole chunks of 64
    return ((n_tasks + n_queues - 1) / n_queues + 63u) / 64u * 64u;
}
// scratch the launch needs besides `partials`: prim (32 B per pixel), prim_disc (4 B per pixel), task_counters
constexpr size_t kPrimBytesPerPixel = 32, kTaskBytesPerPixel = 4, kTaskCounterBytes = 2 * kTaskCounterPairStride * 4;
static_assert(kTaskCounterBytes % 128 == 0, "the no-alloc carve aligns on this");
// paths_begin / paths_end (may be null): events recorded right around the persistent path kernel, the dominant one
hipError_t launch_render(const RenderArgs& a, int accel, bool stats, hipStream_t stream, hipEvent_t paths_begin = nullptr,
                         hipEvent_t paths_end = nullptr);
hipError_t launch_primary_hits(const HitsArgs& a, int accel, hipStream_t stream);
hipError_t launch_resolve(const float4* sum, const uint32_t* counts, uint32_t* argb, size_t n, hipStream_t stream);
// footprint table of the bilinear sky lookup: quad must hold (h + 1) * w * 4 float4
hipError_t launch_sky_quad(const float4* sky, uint32_t w, uint32_t h, float4* quad, hipStream_t stream);
// exact-math self test (pt_selftest.hip): adds mismatch counts into d_bad[kNumSelftests] (zeroed by the caller)
constexpr uint32_t kNumSelftests = 10;
hipError_t launch_selftest(unsigned long long* d_bad, uint32_t log2_random, hipStream_t stream);
// one-time per-device kernel attribute setup (dynamic LDS above 64 KiB)
hipError_t init_kernels();

}  // namespace pt
