// insploray_hip.hip — host side of the C ABI declared in include/insploray_hip.h.
//
// Owns the device state that replaces the reference's CPU "device runtime" (the 4-thread pool and the two
// crossbeam channels of core-engine/src/concurrency/**): one HIP stream, the scene buffers, the accumulator
// (Accumulator.framebuffer / sample_counts, accumulators/full_buffer.rs:7-12) and the resolved ARGB frame.
// There is no CPU fallback anywhere in this file: without a HIP device ir_create fails.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/insploray_hip.h"
#include "ir_internal.h"
#include "lbvh.h"
#include "pt_lbvh_build.h"
#include "pt_kernels.h"
#include "sky_distribution.h"

namespace irx {

thread_local std::string g_global_error;

int fail(const ir_context* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        ctx->err = buf;
    } else {
        g_global_error = buf;
    }
    return code;
}

int ensure(ir_context* ctx, DeviceBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes && b.p) return IR_OK;
    if (b.p) {
        IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        IR_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, IR_ENOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    b.bytes = bytes;
    return IR_OK;
}

void release(DeviceBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

int use_device(const ir_context* ctx) {
    IR_HIP(ctx, hipSetDevice(ctx->device));
    return IR_OK;
}

int upload(ir_context* ctx, DeviceBuf& b, const void* src, size_t bytes) {
    IR_TRY(ensure(ctx, b, bytes));
    if (bytes) {
        IR_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return IR_OK;
}

}  // namespace irx

using namespace irx;

namespace {

// Scene -> device tables.  Material resolution follows integrator.rs:53-60 (material_id < 0 or out of range
// => DEFAULT_MATERIAL, :8-14); brdf = albedo / PI (:68) and emission_color * emissive_power (:62) do not depend
// on the path, so they are evaluated here once per sphere (same IEEE f32 operations as per hit).
int rebuild_shade(ir_context* ctx) {
    if (!ctx->shade_dirty) return IR_OK;
    const uint32_t n = ctx->n_spheres;
    const uint32_t m = (uint32_t)(ctx->h_materials.size() / 9);
    std::vector<float> t((size_t)n * 8 + 8, 0.0f);
    static const float def[9] = {1.0f, 1.0f, 1.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool fast = true;
    for (uint32_t i = 0; i < n; ++i) {
        const int32_t id = ctx->h_matid[i];
        const float* mt = (id < 0 || (uint32_t)id >= m) ? def : &ctx->h_materials[(size_t)id * 9];
        float* o = &t[(size_t)i * 8];
        o[0] = mt[0] / pt::kPi;
        o[1] = mt[1] / pt::kPi;
        o[2] = mt[2] / pt::kPi;
        for (int k = 0; k < 3; ++k) {
            uint32_t bits;
            memcpy(&bits, &o[k], 4);
            // +0, or 2^-30 <= brdf < 2^30 (rejects negatives, -0, denormals, inf, NaN): pt_device.h scatter_weight
            if (!(bits == 0u || (bits - (97u << 23)) < (60u << 23))) fast = false;
        }
        o[4] = mt[5] * mt[8];
        o[5] = mt[6] * mt[8];
        o[6] = mt[7] * mt[8];
    }
    IR_TRY(upload(ctx, ctx->d_shade, t.data(), t.size() * sizeof(float)));
    ctx->shade_dirty = false;
    ctx->fast_materials = fast;
    return IR_OK;
}

// Which closest-hit structure the next launch uses.  AUTO: the LDS scan up to kAutoLbvhFrom spheres (for a handful
// of spheres a tree only adds node tests), the LBVH above; more than kMaxLdsSpheres cannot be staged in LDS at all.
constexpr uint32_t kAutoLbvhFrom = 64;  // measured crossover, tools/accel_crossover.py
constexpr uint32_t kMaxSpheres = 1u << 24;  // leaf = first * 8 + count must stay a non-negative int32
int selected_accel(const ir_context* ctx) {
    if (ctx->n_spheres > pt::kMaxLdsSpheres) return pt::kAccelLbvh;
    if (ctx->accel == IR_ACCEL_LBVH) return ctx->n_spheres ? pt::kAccelLbvh : pt::kAccelLinear;
    if (ctx->accel == IR_ACCEL_LINEAR) return pt::kAccelLinear;
    return ctx->n_spheres >= kAutoLbvhFrom ? pt::kAccelLbvh : pt::kAccelLinear;
}

// Samples per accumulation block of a render when ir_set_sample_block selected IR_SAMPLE_BLOCK_AUTO: 128, halved until the
// launch has at least kAutoMinWorkgroups x 256 (pixel, block) tasks, about fifteen per resident lane, but not below 8
// samples (4 on a frame of fewer than kAutoSmallFrameWorkgroups x 256 pixels).  A launch lasts as long as its tasks keep
// every lane busy and then drains for about one more task; a task costs a switch.  Measured over 4 frame sizes x 7
// sample counts x 6 block sizes (tools/auto_block_sweep.py, profiles/r2_auto_block_sweep.txt): the fastest block has
// 26 000 .. 53 000 workgroups-worth of tasks everywhere (800x450x64: 4; 1080p x 64 / 128 / 256: 16 / 32 / 64; 4K: 64 .. 128),
// 4-sample blocks only pay on small frames (1080p x 32: 0.70 ms against 0.59 ms with 8), and a single block per pixel (the
// lane-per-pixel launch, pt_kernels.hip DIRECT) is what the halving ends in when the samples are few.  A pure function
// of the workgroups-worth of pixels this rank owns and spp, so that a caller (and the oracle in the tests) can reproduce
// the sum order.
constexpr uint32_t kAutoMinWorkgroups = 26000;
constexpr uint32_t kAutoSmallFrameWorkgroups = 4096;
// Round 3: a launch ends on the tasks of the LAST block of its pixels (pt_kernels.hip decode_task), so a block of 32 samples
// or more is then replaced by the size within an eighth of it whose last block (spp mod size) comes closest to a quarter
// block (the larger size on a tie; sizes that divide spp are skipped): 1080p x 1024 spp: 8 x 124 + 32; x 256 spp: 4 x 60 +
// 16.  The launch then drains for the length of a short task instead of a full one (tools/wave_timeline.py,
// tools/drain_ab.sh: C2 +1.5 %, C3 +3 %).  Shorter blocks are dealt out over 16 queues, which all end at the same time.
uint32_t auto_sample_block(uint32_t wgs_per_block, uint32_t spp) {
    const uint32_t min_block = wgs_per_block < kAutoSmallFrameWorkgroups ? 4u : 8u;
    uint32_t sb = 128;
    while (sb > min_block && (uint64_t)wgs_per_block * ((spp + sb - 1) / sb) < kAutoMinWorkgroups) sb /= 2;
    if (sb >= 32 && spp > sb) {
        uint32_t best = 0, best_d = 0xffffffffu;
        for (uint32_t c = sb + sb / 8; c >= sb - sb / 8; --c) {
            const uint32_t r = spp % c;
            if (r == 0) continue;
            const uint32_t d = 4 * r > c ? 4 * r - c : c - 4 * r;
            if (d < best_d) best_d = d, best = c;
        }
        if (best) sb = best;
    }
    return sb;
}

// Scratch of ir_render_noalloc: the launch scratch (primary-pass records, task list, the block sums of one launch: what
// ir_render keeps in hipMalloc'ed buffers).  One copy
// per GPU the library runs on, held until the process ends.  Host code never passes this NAME to a kernel or a memset:
// ir_create asks each device for ITS copy's address (hipGetSymbolAddress) and keeps it in g_noalloc_scratch[device].
constexpr size_t kNoAllocScratchBytes = 160ull << 20;
// per pixel: primary record 32 B + task list entry 4 B + two block sums of 16 B (a launch of at least two blocks)
constexpr size_t kNoAllocMaxPixels = (kNoAllocScratchBytes - 2 * pt::kTaskCounterBytes) / (pt::kPrimBytesPerPixel + pt::kTaskBytesPerPixel + 2 * sizeof(float4));
__device__ __align__(256) float4 g_noalloc_scratch_array[kNoAllocScratchBytes / sizeof(float4)];
constexpr int kMaxDevices = 64;
void* g_noalloc_scratch[kMaxDevices] = {nullptr};

int init_noalloc_scratch(int device, hipError_t* err) {  // with `device` current
    if (device < 0 || device >= kMaxDevices) return IR_OK;  // such a device simply has no no-alloc entry point
    if (g_noalloc_scratch[device]) return IR_OK;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch_array));
    if (e != hipSuccess || !p) {
        *err = e;
        return IR_EDEVICE;
    }
    g_noalloc_scratch[device] = p;
    return IR_OK;
}

// block sums of one launch (ir_render): at most this much scratch, see render_impl
constexpr size_t kPartialsBudgetBytes = 6ull << 30;

// AUTO builds on the device from this many spheres (measured crossover with the host build, tools/lbvh_build_times.py)
constexpr uint32_t kAutoDeviceBuildFrom = 2048;

int build_bvh_host(ir_context* ctx) {
    pt::Lbvh t;
    const bool sah = ctx->bvh_build == IR_BUILD_HOST_SAH;
    pt::build_lbvh(ctx->h_spheres.data(), ctx->n_spheres, t, sah ? pt::LbvhSplit::Sah : pt::LbvhSplit::Morton);
    IR_TRY(upload(ctx, ctx->d_bvh_nodes, t.wide.data(), t.wide.size() * sizeof(pt::WideNode)));
    IR_TRY(upload(ctx, ctx->d_bvh_spheres, t.leaf_spheres.data(), t.leaf_spheres.size() * sizeof(float)));
    IR_TRY(upload(ctx, ctx->d_bvh_index, t.leaf_index.data(), t.leaf_index.size() * sizeof(int32_t)));
    IR_TRY(upload(ctx, ctx->d_bvh_start, t.sphere_start.data(), t.sphere_start.size() * sizeof(int32_t)));
    ctx->bvh_nodes = t.n_inner;
    ctx->bvh_depth = t.max_depth;
    ctx->bvh_stack = t.stack_depth;
    ctx->bvh_built_by = sah ? IR_BUILD_HOST_SAH : IR_BUILD_HOST;
    return IR_OK;
}

// Same arrays as build_bvh_host, produced by pt_lbvh_build.hip from the spheres already resident on the device.
// *too_deep: the radix tree needs more than kMaxStack stack entries (the host builder then falls back to median splits).
int build_bvh_device(ir_context* ctx, bool* too_deep) {
    const uint32_t n = ctx->n_spheres;
    const bool refit = ctx->bvh_build == IR_BUILD_REFIT && ctx->bvh_topology_n == n;
    ctx->bvh_topology_n = 0;
    IR_TRY(ensure(ctx, ctx->d_bvh_scratch, pt::lbvh_build_scratch_bytes(n)));
    IR_TRY(ensure(ctx, ctx->d_bvh_spheres, (size_t)n * 16));
    IR_TRY(ensure(ctx, ctx->d_bvh_index, (size_t)n * 4));
    // room for the most records a tree of n spheres can have: n - 1 wide nodes, as many climb records of inner nodes, n of leaves
    const size_t records_bytes = (2 * (size_t)(n - 1) + n) * sizeof(pt::WideNode);
    IR_TRY(ensure(ctx, ctx->d_bvh_nodes, records_bytes));
    IR_TRY(ensure(ctx, ctx->d_bvh_start, (size_t)n * 8));
    pt::LbvhDeviceTree tree;
    const hipError_t e = pt::lbvh_build_device((const float4*)ctx->d_spheres.p, n, ctx->d_bvh_scratch.p, ctx->d_bvh_scratch.bytes,
                                               (float4*)ctx->d_bvh_spheres.p, (int*)ctx->d_bvh_index.p, &tree, refit, ctx->stream);
    if (e != hipSuccess)
        return fail(ctx, IR_EDEVICE, "device LBVH build of %u spheres, stage %s: %s", n, tree.failed_stage, hipGetErrorString(e));
    // (record slots no node owns stay zero, as in the host build)
    IR_HIP(ctx, hipMemsetAsync(ctx->d_bvh_nodes.p, 0, records_bytes, ctx->stream));
    IR_HIP(ctx, pt::lbvh_emit_device((const float4*)ctx->d_spheres.p, n, tree, (float4*)ctx->d_bvh_nodes.p, (int2*)ctx->d_bvh_start.p, ctx->stream));
    uint32_t info[pt::kLbvhInfoWords];
    IR_HIP(ctx, hipMemcpyAsync(info, tree.info, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *too_deep = info[6] + 1 > pt::kMaxStack;
    ctx->bvh_nodes = info[7];
    ctx->bvh_depth = info[6];
    ctx->bvh_stack = info[6] + 1;
    ctx->bvh_built_by = refit ? IR_BUILD_REFIT : IR_BUILD_DEVICE;
    if (!*too_deep) ctx->bvh_topology_n = n;
    return IR_OK;
}

int ensure_bvh(ir_context* ctx) {
    if (selected_accel(ctx) != pt::kAccelLbvh || ctx->bvh_valid) return IR_OK;
    const auto t0 = std::chrono::steady_clock::now();
    // fewer than 3 spheres are a single leaf (no radix tree to build)
    bool on_device = ctx->n_spheres >= 3 &&
                     (ctx->bvh_build == IR_BUILD_DEVICE || ctx->bvh_build == IR_BUILD_REFIT ||
                      (ctx->bvh_build == IR_BUILD_AUTO && ctx->n_spheres >= kAutoDeviceBuildFrom));
    if (on_device) {
        bool too_deep = false;
        IR_TRY(build_bvh_device(ctx, &too_deep));
        on_device = !too_deep;
    }
    if (!on_device) IR_TRY(build_bvh_host(ctx));
    ctx->bvh_build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ctx->bvh_valid = true;
    return IR_OK;
}

// Sky importance sampling tables, built on the host from the texels as they are on the device (ir_set_skybox keeps no
// host copy) the first time a render needs them.
int ensure_sky_tables(ir_context* ctx) {
    if (ctx->sky_tables_valid) return IR_OK;
    const uint32_t w = ctx->sky_w, h = ctx->sky_h;
    std::vector<float> texels((size_t)w * h * 4);
    IR_HIP(ctx, hipMemcpyAsync(texels.data(), ctx->d_skybox.p, texels.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pt::SkyDistribution d;
    pt::build_sky_distribution(texels.data(), w, h, d);
    IR_TRY(upload(ctx, ctx->d_sky_marginal, d.marginal.data(), d.marginal.size() * sizeof(float)));
    IR_TRY(upload(ctx, ctx->d_sky_cond, d.cond.data(), d.cond.size() * sizeof(float)));
    IR_TRY(upload(ctx, ctx->d_sky_pdf, d.pdf.data(), d.pdf.size() * sizeof(float)));
    IR_TRY(upload(ctx, ctx->d_sky_marginal_guide, d.marginal_guide.data(), d.marginal_guide.size() * sizeof(uint32_t)));
    IR_TRY(upload(ctx, ctx->d_sky_cond_guide, d.cond_guide.data(), d.cond_guide.size() * sizeof(uint32_t)));
    ctx->sky_tables_valid = true;
    return IR_OK;
}

// The bilinear lookup reads two image rows: 2.2 memory sectors per lookup on a sky no cache holds.  From this size on
// (float4 texels; the Infinity Cache is 256 MB) a footprint table four times the image gives every lookup its four texels
// in ONE 64-byte record (pt_device.h sky_bilinear_quad): HBM capacity (288 GB) spent on HBM accesses.
constexpr size_t kSkyQuadFromBytes = 128ull << 20;
bool wants_sky_quad(const ir_context* ctx) {
    static const int forced = getenv("IR_DEBUG_SKY_QUAD") ? atoi(getenv("IR_DEBUG_SKY_QUAD")) : -1;  // tools/ A/B runs only
    if (ctx->sky_filter != IR_SKY_BILINEAR || !ctx->d_skybox.p || !ctx->sky_w) return false;
    // the table's records are addressed with 32-bit byte offsets: beyond 2^26 records the two-row lookup stays
    if ((uint64_t)(ctx->sky_h + 1) * ctx->sky_w >= (1ull << 26)) return false;
    if (forced >= 0) return forced != 0;
    return (size_t)ctx->sky_w * ctx->sky_h * sizeof(float4) >= kSkyQuadFromBytes;
}
int ensure_sky_quad(ir_context* ctx) {
    if (!wants_sky_quad(ctx) || ctx->sky_quad_valid) return IR_OK;
    const size_t bytes = (size_t)(ctx->sky_h + 1) * ctx->sky_w * 4 * sizeof(float4);
    IR_TRY(ensure(ctx, ctx->d_skyquad, bytes));
    IR_HIP(ctx, pt::launch_sky_quad((const float4*)ctx->d_skybox.p, ctx->sky_w, ctx->sky_h, (float4*)ctx->d_skyquad.p, ctx->stream));
    ctx->sky_quad_valid = true;
    return IR_OK;
}

void fill_bvh(const ir_context* ctx, pt::LbvhView& v) {
    v.wide = (const float4*)ctx->d_bvh_nodes.p;
    v.leaf_spheres = (const float4*)ctx->d_bvh_spheres.p;
    v.leaf_index = (const int*)ctx->d_bvh_index.p;
    v.sphere_start = (const int2*)ctx->d_bvh_start.p;
    v.n_inner = ctx->bvh_nodes;
    v.stack_depth = ctx->bvh_stack;
}

void fill_camera(const ir_context* ctx, pt::CameraBlock& c) {
    for (int i = 0; i < 3; ++i) {
        c.xc[i] = ctx->cam_m[i];
        c.yc[i] = ctx->cam_m[4 + i];
        c.zc[i] = ctx->cam_m[8 + i];
        c.pos[i] = ctx->cam_pos[i];
    }
    c.tan_half_fov = std::tan(ctx->cam_fov / 2.0f);  // pinhole_camera.rs:86, f32 tan once per render
    c.aspect = ctx->cam_aspect;
}

int check_camera(ir_context* ctx, uint32_t w, uint32_t h, const float** rays) {
    *rays = nullptr;
    if (ctx->rays_w == w && ctx->rays_h == h && ctx->d_rays.p && w && h) {
        *rays = (const float*)ctx->d_rays.p;
        return IR_OK;
    }
    if (!ctx->have_pinhole) return fail(ctx, IR_ESTATE, "no camera: call ir_set_pinhole or ir_set_primary_rays (%ux%u) first", w, h);
    return IR_OK;
}

int set_size(ir_context* ctx, uint32_t w, uint32_t h) {
    if (ctx->ext_sum) {
        if (ctx->ext_w != w || ctx->ext_h != h)
            return fail(ctx, IR_EINVAL, "bound accumulator is %ux%u but render asked for %ux%u", ctx->ext_w, ctx->ext_h, w, h);
        ctx->width = w;
        ctx->height = h;
        return IR_OK;
    }
    if (ctx->width == w && ctx->height == h && ctx->d_sum.p) return IR_OK;
    // renderer.rs:84-88: a different resolution gets a fresh (zeroed) Accumulator
    const size_t n = (size_t)w * h;
    IR_TRY(ensure(ctx, ctx->d_sum, n * sizeof(float4)));
    IR_TRY(ensure(ctx, ctx->d_counts, n * sizeof(uint32_t)));
    IR_HIP(ctx, hipMemsetAsync(ctx->d_sum.p, 0, ctx->d_sum.bytes, ctx->stream));
    IR_HIP(ctx, hipMemsetAsync(ctx->d_counts.p, 0, ctx->d_counts.bytes, ctx->stream));
    ctx->width = w;
    ctx->height = h;
    ctx->samples_done = 0;
    return IR_OK;
}

}  // namespace

extern "C" {

uint32_t ir_abi_version(void) { return 3; }  // 3: round 3 (ir_download_accel_climb, ir_get_primary_stats, ir_hip_versions; 48-byte LBVH records)

int ir_hip_versions(int* out_compiled, int* out_runtime) {
    if (out_compiled) *out_compiled = HIP_VERSION;
    int rt = 0;
    const hipError_t e = hipRuntimeGetVersion(&rt);
    if (out_runtime) *out_runtime = e == hipSuccess ? rt : 0;
    return e == hipSuccess ? IR_OK : IR_EDEVICE;
}

const char* ir_last_global_error(void) { return g_global_error.c_str(); }

int ir_create(ir_context** out_ctx, int device) {
    if (!out_ctx) return fail(nullptr, IR_EINVAL, "ir_create: out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, IR_EDEVICE, "ir_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(nullptr, IR_EINVAL, "ir_create: device %d out of range [0,%d)", device, count);
    ir_context* ctx = new (std::nothrow) ir_context();
    if (!ctx) return fail(nullptr, IR_ENOMEM, "ir_create: out of host memory");
    ctx->device = device;
    auto bail = [&](const char* what, hipError_t err) {
        fail(nullptr, IR_EDEVICE, "ir_create: %s: %s", what, hipGetErrorString(err));
        ir_destroy(ctx);
        return IR_EDEVICE;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    ctx->stream = ctx->own_stream;
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    for (uint32_t k = 0; k < ir_context::kTimerRing; ++k) {
        if ((e = hipEventCreate(&ctx->ev_p0[k])) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreate(&ctx->ev_p1[k])) != hipSuccess) return bail("hipEventCreate", e);
    }
    if (init_noalloc_scratch(device, &e) != IR_OK) return bail("hipGetSymbolAddress(static scratch)", e);
    if ((e = pt::init_kernels()) != hipSuccess) return bail("kernel attribute setup", e);
    if ((e = pt::init_wavefront_kernels()) != hipSuccess) return bail("kernel attribute setup", e);
    *out_ctx = ctx;
    return IR_OK;
}

void ir_destroy(ir_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    release(ctx->d_spheres); release(ctx->d_shade); release(ctx->d_skybox); release(ctx->d_skyquad); release(ctx->d_wave_times); release(ctx->d_rays);
    if (ctx->h_argb) (void)hipHostFree(ctx->h_argb);
    release(ctx->d_sum); release(ctx->d_counts); release(ctx->d_argb); release(ctx->d_hit_idx); release(ctx->d_hit_t);
    release(ctx->d_stats); release(ctx->d_partials);
    release(ctx->d_prim); release(ctx->d_task_pixels); release(ctx->d_task_counters);
    release(ctx->d_wf_queue[0]); release(ctx->d_wf_queue[1]); release(ctx->d_wf_counts); release(ctx->d_wf_values);
    release(ctx->d_wf_prim); release(ctx->d_wf_tileacc);
    release(ctx->d_bvh_nodes); release(ctx->d_bvh_spheres); release(ctx->d_bvh_index); release(ctx->d_bvh_start); release(ctx->d_bvh_scratch);
    release(ctx->d_sky_marginal); release(ctx->d_sky_cond); release(ctx->d_sky_pdf);
    release(ctx->d_sky_marginal_guide); release(ctx->d_sky_cond_guide);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (uint32_t k = 0; k < ir_context::kTimerRing; ++k) {
        if (ctx->ev_p0[k]) (void)hipEventDestroy(ctx->ev_p0[k]);
        if (ctx->ev_p1[k]) (void)hipEventDestroy(ctx->ev_p1[k]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* ir_last_error(const ir_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ir_set_stream(ir_context* ctx, void* hip_stream) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return IR_OK;
}

int ir_set_spheres(ir_context* ctx, const float* spheres_xyzr, const int32_t* material_id, uint32_t n) {
    if (!ctx) return IR_EINVAL;
    if (n && (!spheres_xyzr || !material_id)) return fail(ctx, IR_EINVAL, "ir_set_spheres: NULL array with n=%u", n);
    if (n > kMaxSpheres) return fail(ctx, IR_ELIMIT, "ir_set_spheres: %u spheres exceed the limit of %u", n, kMaxSpheres);
    if (n > pt::kMaxLdsSpheres && ctx->accel == IR_ACCEL_LINEAR)
        return fail(ctx, IR_ELIMIT, "ir_set_spheres: %u spheres exceed the %u the linear scan stages in LDS (use IR_ACCEL_AUTO/LBVH)", n,
                    pt::kMaxLdsSpheres);
    IR_TRY(use_device(ctx));
    // The facade re-sends the scene on every render: what did not change keeps its device tables.  The shading table
    // depends on the material ids only, so moving spheres (the editor's common edit) keeps it; the LBVH depends on the
    // geometry only.
    const bool same_ids = n == ctx->n_spheres && (n == 0 || memcmp(ctx->h_matid.data(), material_id, (size_t)n * 4) == 0);
    const bool same_geometry = n == ctx->n_spheres && (n == 0 || memcmp(ctx->h_spheres.data(), spheres_xyzr, (size_t)n * 16) == 0);
    if (!same_geometry) {
        // The device copy goes first: a failed upload (ensure() may already have freed the old buffer) must not leave
        // host mirrors that make a retry with the same scene look unchanged.
        ctx->bvh_valid = false;
        const int rc = upload(ctx, ctx->d_spheres, spheres_xyzr, (size_t)n * 4 * sizeof(float));
        if (rc != IR_OK) {
            ctx->bvh_topology_n = 0;
            ctx->h_spheres.clear();
            ctx->h_matid.clear();
            ctx->n_spheres = 0;
            ctx->shade_dirty = true;
            return rc;
        }
        ctx->h_spheres.assign(spheres_xyzr, spheres_xyzr + (size_t)n * 4);
    }
    ctx->n_spheres = n;
    if (!same_ids) {
        ctx->h_matid.assign(material_id, material_id + n);
        ctx->shade_dirty = true;
    }
    return IR_OK;
}

int ir_set_materials(ir_context* ctx, const float* materials9, uint32_t n) {
    if (!ctx) return IR_EINVAL;
    if (n && !materials9) return fail(ctx, IR_EINVAL, "ir_set_materials: NULL array with n=%u", n);
    if (ctx->h_materials.size() == (size_t)n * 9 && (n == 0 || memcmp(ctx->h_materials.data(), materials9, (size_t)n * 36) == 0))
        return IR_OK;  // unchanged (re-sent on every render)
    ctx->h_materials.assign(materials9, materials9 + (size_t)n * 9);
    ctx->shade_dirty = true;
    return IR_OK;
}

int ir_set_sky_color(ir_context* ctx, const float rgb[3]) {
    if (!ctx) return IR_EINVAL;
    if (!rgb) return fail(ctx, IR_EINVAL, "ir_set_sky_color: NULL");
    memcpy(ctx->sky, rgb, sizeof(ctx->sky));
    return IR_OK;
}

int ir_set_skybox(ir_context* ctx, const float* rgb, uint32_t w, uint32_t h) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    ctx->sky_tables_valid = false;
    ctx->sky_quad_valid = false;
    if (!rgb) {
        IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        release(ctx->d_skybox);
        release(ctx->d_skyquad);
        ctx->sky_w = ctx->sky_h = 0;
        return IR_OK;
    }
    if (w == 0 || h == 0) return fail(ctx, IR_EINVAL, "ir_set_skybox: empty image %ux%u", w, h);
    // the kernels address texels with 32-bit byte offsets and 24-bit row arithmetic (pt_device.h load16 / row_major)
    if (w >= (1u << 24) || h >= (1u << 24) || (uint64_t)w * h >= (1ull << 28))
        return fail(ctx, IR_ELIMIT, "ir_set_skybox: %ux%u exceeds the supported 2^28 texels (4 GiB as float4)", w, h);
    const size_t n = (size_t)w * h;
    std::vector<float> padded(n * 4);
    for (size_t i = 0; i < n; ++i) {
        padded[4 * i + 0] = rgb[3 * i + 0];
        padded[4 * i + 1] = rgb[3 * i + 1];
        padded[4 * i + 2] = rgb[3 * i + 2];
        padded[4 * i + 3] = 0.0f;
    }
    IR_TRY(upload(ctx, ctx->d_skybox, padded.data(), padded.size() * sizeof(float)));
    ctx->sky_w = w;
    ctx->sky_h = h;
    return IR_OK;
}

int ir_set_sky_filter(ir_context* ctx, int filter) {
    if (!ctx) return IR_EINVAL;
    if (filter != IR_SKY_NEAREST && filter != IR_SKY_BILINEAR) return fail(ctx, IR_EINVAL, "ir_set_sky_filter: unknown filter %d", filter);
    ctx->sky_filter = filter;
    return IR_OK;
}

int ir_set_sky_sampling(ir_context* ctx, int mode) {
    if (!ctx) return IR_EINVAL;
    if (mode != IR_SKY_SAMPLING_BSDF && mode != IR_SKY_SAMPLING_MIS) return fail(ctx, IR_EINVAL, "ir_set_sky_sampling: unknown mode %d", mode);
    ctx->sky_sampling = mode;
    return IR_OK;
}

int ir_set_accel(ir_context* ctx, int accel) {
    if (!ctx) return IR_EINVAL;
    if (accel != IR_ACCEL_AUTO && accel != IR_ACCEL_LINEAR && accel != IR_ACCEL_LBVH)
        return fail(ctx, IR_EINVAL, "ir_set_accel: unknown mode %d", accel);
    if (accel == IR_ACCEL_LINEAR && ctx->n_spheres > pt::kMaxLdsSpheres)
        return fail(ctx, IR_ELIMIT, "ir_set_accel: the linear scan stages at most %u spheres in LDS, the scene has %u",
                    pt::kMaxLdsSpheres, ctx->n_spheres);
    ctx->accel = accel;
    return IR_OK;
}

int ir_set_accel_build(ir_context* ctx, int build) {
    if (!ctx) return IR_EINVAL;
    if (build != IR_BUILD_AUTO && build != IR_BUILD_HOST && build != IR_BUILD_DEVICE && build != IR_BUILD_REFIT && build != IR_BUILD_HOST_SAH)
        return fail(ctx, IR_EINVAL, "ir_set_accel_build: unknown mode %d", build);
    if (build != ctx->bvh_build) {
        ctx->bvh_valid = false;
        ctx->bvh_topology_n = 0;  // the first build in a new mode is a full one
    }
    ctx->bvh_build = build;
    return IR_OK;
}

int ir_set_pinhole(ir_context* ctx, const float local_to_world[16], const float position[3], float fov, float aspect_ratio) {
    if (!ctx) return IR_EINVAL;
    if (!local_to_world || !position) return fail(ctx, IR_EINVAL, "ir_set_pinhole: NULL");
    memcpy(ctx->cam_m, local_to_world, sizeof(ctx->cam_m));
    memcpy(ctx->cam_pos, position, sizeof(ctx->cam_pos));
    ctx->cam_fov = fov;
    ctx->cam_aspect = aspect_ratio;
    ctx->have_pinhole = true;
    // a pinhole camera supersedes previously uploaded generic rays
    ctx->rays_w = ctx->rays_h = 0;
    return IR_OK;
}

int ir_set_primary_rays(ir_context* ctx, const float* rays, uint32_t w, uint32_t h) {
    if (!ctx) return IR_EINVAL;
    if (!rays || w == 0 || h == 0) return fail(ctx, IR_EINVAL, "ir_set_primary_rays: NULL or empty (%ux%u)", w, h);
    IR_TRY(use_device(ctx));
    IR_TRY(upload(ctx, ctx->d_rays, rays, (size_t)w * h * 6 * sizeof(float)));
    ctx->rays_w = w;
    ctx->rays_h = h;
    return IR_OK;
}

int ir_set_integrator(ir_context* ctx, uint32_t bounces, uint32_t max_compulsory_bounces, uint64_t seed) {
    if (!ctx) return IR_EINVAL;
    if (bounces > 0x15555555u) return fail(ctx, IR_ELIMIT, "ir_set_integrator: bounces %u too large", bounces);
    ctx->bounces = bounces;
    ctx->rr_start = max_compulsory_bounces;
    ctx->seed = seed;
    return IR_OK;
}

int ir_set_schedule(ir_context* ctx, int schedule, uint32_t max_paths) {
    if (!ctx) return IR_EINVAL;
    if (schedule != IR_SCHEDULE_AUTO && schedule != IR_SCHEDULE_REGENERATION && schedule != IR_SCHEDULE_WAVEFRONT)
        return fail(ctx, IR_EINVAL, "ir_set_schedule: unknown schedule %d", schedule);
    if (max_paths != 0 && max_paths < 65536u) return fail(ctx, IR_EINVAL, "ir_set_schedule: max_paths %u below 65536", max_paths);
    ctx->schedule = schedule;
    if (max_paths) ctx->wf_max_paths = max_paths;
    return IR_OK;
}

int ir_set_sample_block(ir_context* ctx, uint32_t samples_per_block) {
    if (!ctx) return IR_EINVAL;
    if (samples_per_block > 4096)
        return fail(ctx, IR_EINVAL, "ir_set_sample_block: %u not in [1, 4096] and not IR_SAMPLE_BLOCK_AUTO (0)", samples_per_block);
    ctx->sample_block = samples_per_block;
    return IR_OK;
}

int ir_get_sample_block(const ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t* out_samples_per_block) {
    if (!ctx) return IR_EINVAL;
    if (!out_samples_per_block) return fail(ctx, IR_EINVAL, "ir_get_sample_block: NULL");
    const uint32_t tiles = ((width + pt::kTile - 1) / pt::kTile) * ((height + pt::kTile - 1) / pt::kTile);
    const bool by_tiles = ctx->part_mode == IR_PARTITION_TILES;
    const uint32_t wgs = pt::owned_tiles(tiles, by_tiles ? ctx->rank : 0, by_tiles ? ctx->world : 1) * pt::kBlocksPerTile;
    *out_samples_per_block = ctx->sample_block ? ctx->sample_block : auto_sample_block(wgs, spp);
    return IR_OK;
}

int ir_set_partition(ir_context* ctx, int mode, uint32_t rank, uint32_t world) {
    if (!ctx) return IR_EINVAL;
    if (mode != IR_PARTITION_NONE && mode != IR_PARTITION_TILES && mode != IR_PARTITION_SAMPLES)
        return fail(ctx, IR_EINVAL, "ir_set_partition: unknown mode %d", mode);
    if (world == 0 || rank >= world) return fail(ctx, IR_EINVAL, "ir_set_partition: rank %u not in [0,%u)", rank, world);
    ctx->part_mode = mode;
    ctx->rank = rank;
    ctx->world = world;
    return IR_OK;
}

}  // extern "C"

namespace {

// Device tables every launch needs.  May allocate and upload; after it succeeds, and until the scene, the sky or a
// structure setting changes, scene_is_prepared() holds and a render allocates nothing for the scene.
int prepare_scene(ir_context* ctx) {
    IR_TRY(rebuild_shade(ctx));
    IR_TRY(ensure(ctx, ctx->d_spheres, 16));
    IR_TRY(ensure_bvh(ctx));
    const bool skymis = ctx->sky_sampling == IR_SKY_SAMPLING_MIS && ctx->d_skybox.p && ctx->sky_w;
    if (skymis) IR_TRY(ensure_sky_tables(ctx));
    IR_TRY(ensure_sky_quad(ctx));
    return IR_OK;
}
bool scene_is_prepared(const ir_context* ctx) {
    if (ctx->shade_dirty || !ctx->d_shade.p || !ctx->d_spheres.p) return false;
    if (selected_accel(ctx) == pt::kAccelLbvh && !ctx->bvh_valid) return false;
    const bool skymis = ctx->sky_sampling == IR_SKY_SAMPLING_MIS && ctx->d_skybox.p && ctx->sky_w;
    if (wants_sky_quad(ctx) && !ctx->sky_quad_valid) return false;
    return !skymis || ctx->sky_tables_valid;
}

// The body of ir_render_async and ir_render_noalloc.  ext_* == nullptr: the context's film (resized, allocated);
// otherwise the caller's buffers and the static scratch, and nothing below may allocate.
int render_impl(ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, int accumulate, float4* ext_sum,
                uint32_t* ext_counts) {
    const bool noalloc = ext_sum != nullptr;
    const char* who = noalloc ? "ir_render_noalloc" : "ir_render";
    if ((uint64_t)width * height > 0x7fffffffull / 8)
        return fail(ctx, IR_ELIMIT, "%s: %ux%u exceeds the supported frame size", who, width, height);
    IR_TRY(use_device(ctx));
    ctx->events_recorded = false;
    ctx->paths_timed = false;
    ctx->last_launches = 0;
    ctx->stats_valid = false;
    ctx->output_combined = false;
    if (!noalloc) IR_TRY(set_size(ctx, width, height));
    if (width == 0 || height == 0) return IR_OK;  // RayTracer::new(0,0) then render(…,0,0,…) is a no-op loop
    const size_t n = (size_t)width * height;
    float4* const sum = noalloc ? ext_sum : film_sum(ctx);
    uint32_t* const counts = noalloc ? ext_counts : film_counts(ctx);
    if (noalloc) {
        // before anything is enqueued: a frame whose block sums do not fit the static scratch is refused outright
        if (n > kNoAllocMaxPixels)
            return fail(ctx, IR_ELIMIT, "%s: %ux%u pixels exceed the %zu the %zu MiB static scratch holds the launch scratch for", who,
                        width, height, kNoAllocMaxPixels, kNoAllocScratchBytes >> 20);
        if (ctx->schedule == IR_SCHEDULE_WAVEFRONT)
            return fail(ctx, IR_ESTATE, "%s: the wavefront schedule allocates its path queues; select IR_SCHEDULE_AUTO/REGENERATION", who);
        if (ctx->stats_enabled) return fail(ctx, IR_ESTATE, "%s: statistics need an allocated counter buffer; use ir_render", who);
        if (!scene_is_prepared(ctx))
            return fail(ctx, IR_ESTATE, "%s: the scene's device tables are stale; call ir_render (spp 0 is enough) once after "
                                        "changing the scene, the sky or the closest-hit structure", who);
    } else {
        IR_TRY(prepare_scene(ctx));
    }
    // renderer.rs:99-103: a non-accumulating render starts from a zero accumulator.  When the first launch of this call
    // visits every pixel anyway (task schedule, the whole frame owned), its primary pass writes the zeros: two fills less.
    const bool wavefront = ctx->schedule == IR_SCHEDULE_WAVEFRONT && !ctx->stats_enabled;
    const bool owns_all = !(ctx->part_mode == IR_PARTITION_TILES && ctx->world > 1);
    const bool fresh_in_kernel = !accumulate && spp != 0 && n != 0 && !wavefront && owns_all;
    // Everything that can still refuse the call is checked BEFORE the fills are skipped: an error return must leave a
    // zeroed film behind a non-accumulating call (as the fills always did), never old sums under samples_done == 0.
    const float* rays = nullptr;
    uint64_t base = accumulate ? ctx->samples_done : 0;
    if (ctx->part_mode == IR_PARTITION_SAMPLES) base = base * ctx->world + (uint64_t)ctx->rank * spp;
    int refused = IR_OK;
    if (spp != 0) {
        refused = check_camera(ctx, width, height, &rays);
        if (refused == IR_OK && base + spp > 0xffffffffull) refused = fail(ctx, IR_ELIMIT, "%s: sample index overflow", who);
    }
    if (!accumulate) {
        if (!fresh_in_kernel || refused != IR_OK) {
            IR_HIP(ctx, hipMemsetAsync(sum, 0, n * sizeof(float4), ctx->stream));
            IR_HIP(ctx, hipMemsetAsync(counts, 0, n * sizeof(uint32_t), ctx->stream));
        }
        ctx->samples_done = 0;
    }
    if (refused != IR_OK) return refused;
    if (spp == 0) return IR_OK;
    const int accel = selected_accel(ctx);
    // sky importance sampling needs a sky image; with a constant sky colour there is nothing to importance-sample
    const bool skymis = ctx->sky_sampling == IR_SKY_SAMPLING_MIS && ctx->d_skybox.p && ctx->sky_w;

    pt::RenderArgs a;
    memset(&a, 0, sizeof(a));
    fill_camera(ctx, a.cam);
    a.rays = rays;
    a.w = width;
    a.h = height;
    a.spheres = (const float4*)ctx->d_spheres.p;
    a.shade = (const float4*)ctx->d_shade.p;
    a.n_spheres = ctx->n_spheres;
    a.fast_materials = ctx->fast_materials ? 1u : 0u;
    memcpy(a.sky, ctx->sky, sizeof(a.sky));
    a.skybox = (const float4*)ctx->d_skybox.p;
    a.sky_w = ctx->sky_w;
    a.sky_h = ctx->sky_h;
    a.sky_filter = ctx->sky_filter;
    a.skyquad = (wants_sky_quad(ctx) && ctx->sky_quad_valid) ? (const float4*)ctx->d_skyquad.p : nullptr;
    if (skymis) {
        a.sky_marginal = (const float*)ctx->d_sky_marginal.p;
        a.sky_cond = (const float*)ctx->d_sky_cond.p;
        a.sky_pdf = (const float*)ctx->d_sky_pdf.p;
        a.sky_marginal_guide = (const uint32_t*)ctx->d_sky_marginal_guide.p;
        a.sky_cond_guide = (const uint32_t*)ctx->d_sky_cond_guide.p;
    }
    a.bounces = ctx->bounces;
    a.rr_start = ctx->rr_start;
    a.seed_lo = (uint32_t)(ctx->seed & 0xffffffffu);
    a.seed_hi = (uint32_t)(ctx->seed >> 32);
    a.sum = sum;
    a.counts = counts;
    a.tiles_x = (width + pt::kTile - 1) / pt::kTile;
    a.n_tiles = a.tiles_x * ((height + pt::kTile - 1) / pt::kTile);
    a.tile_rank = (ctx->part_mode == IR_PARTITION_TILES) ? ctx->rank : 0;
    a.tile_world = (ctx->part_mode == IR_PARTITION_TILES) ? ctx->world : 1;
    fill_bvh(ctx, a.bvh);

    if (ctx->stats_enabled) {
        IR_TRY(ensure(ctx, ctx->d_stats, pt::kNumStats * sizeof(unsigned long long)));
        IR_HIP(ctx, hipMemsetAsync(ctx->d_stats.p, 0, pt::kNumStats * sizeof(unsigned long long), ctx->stream));
        a.stats = (unsigned long long*)ctx->d_stats.p;
    }

    // sample indices of this call: [base, base + spp), checked above.  The sample block is resolved once, for both
    // schedules (IR_SAMPLE_BLOCK_AUTO is 0: the wavefront branch must never see it raw).
    const uint32_t owned_wgs = pt::owned_tiles(a.n_tiles, a.tile_rank, a.tile_world) * pt::kBlocksPerTile;
    const uint32_t sb = ctx->sample_block ? ctx->sample_block : auto_sample_block(owned_wgs, spp);
    ctx->last_sample_block = sb;

    if (ctx->schedule == IR_SCHEDULE_WAVEFRONT && !ctx->stats_enabled) {
        // queues sized for at least one sample of every owned pixel (rounded to whole 64x64 tiles)
        const uint32_t owned = pt::owned_tiles(a.n_tiles, a.tile_rank, a.tile_world);
        const size_t owned_pixels = (size_t)owned * pt::kTile * pt::kTile;
        size_t cap = ctx->wf_max_paths;
        if (cap < owned_pixels) cap = owned_pixels;
        size_t per_batch = owned_pixels ? cap / owned_pixels : 1;
        if (per_batch > sb) per_batch = sb;
        if (per_batch > spp) per_batch = spp;
        pt::WfArgs wa;
        memset(&wa, 0, sizeof(wa));
        const uint32_t bounce_wgs = 256 * 8;
        const size_t slots = pt::wavefront_queue_slots(per_batch * owned_pixels, bounce_wgs);
        if (slots > 0xfffffff0u) return fail(ctx, IR_ELIMIT, "ir_render: wavefront queue capacity overflow");
        for (int k = 0; k < 2; ++k) IR_TRY(ensure(ctx, ctx->d_wf_queue[k], slots * (3 + 12) * sizeof(uint32_t)));
        IR_TRY(ensure(ctx, ctx->d_wf_counts, 4 * sizeof(uint32_t)));
        IR_TRY(ensure(ctx, ctx->d_wf_values, per_batch * n * sizeof(float4)));
        IR_TRY(ensure(ctx, ctx->d_wf_prim, 2 * n * sizeof(float4)));
        IR_TRY(ensure(ctx, ctx->d_wf_tileacc, n * sizeof(float4)));
        auto queue = [&](int k) {
            pt::WfQueue q;
            uint32_t* qb = (uint32_t*)ctx->d_wf_queue[k].p;
            q.pix = qb;
            q.smp = qb + slots;
            q.idx = (int32_t*)(qb + 2 * slots);
            q.f = (float*)(qb + 3 * slots);
            q.count = (uint32_t*)ctx->d_wf_counts.p + k;
            q.cap = (uint32_t)slots;
            return q;
        };
        wa.r = a;
        wa.r.spp = spp;
        wa.r.sample_base = (uint32_t)base;
        wa.r.sample_block = sb;
        wa.prim_pos = (float4*)ctx->d_wf_prim.p;
        wa.prim_nrm = (float4*)ctx->d_wf_prim.p + n;
        wa.values = (float4*)ctx->d_wf_values.p;
        wa.tileacc = (float4*)ctx->d_wf_tileacc.p;
        wa.qin = queue(0);
        wa.qout = queue(1);
        wa.in_cursor = (uint32_t*)ctx->d_wf_counts.p + 2;
        wa.overflow = (uint32_t*)ctx->d_wf_counts.p + 3;
        IR_HIP(ctx, hipMemsetAsync(wa.overflow, 0, sizeof(uint32_t), ctx->stream));
        ctx->wf_check_overflow = true;
        IR_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        IR_HIP(ctx, pt::launch_wavefront(wa, accel, (uint32_t)per_batch, bounce_wgs, ctx->stream));
        IR_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        ctx->events_recorded = true;
        ctx->last_launches = 1;
        ctx->samples_done += spp;
        return IR_OK;
    }

    // bound the work of one launch (pixels x spheres x samples) so that a launch stays well under a second
    const double unit = (double)n * (double)(accel == pt::kAccelLbvh ? 64 : (ctx->n_spheres < 8 ? 8 : ctx->n_spheres));
    uint32_t chunk = (uint32_t)std::fmin(4096.0, std::fmax(1.0, 4.0e12 / unit));
    // whole sample blocks per launch, so that the block structure does not depend on the chunking
    // (rounded UP: a 4096-sample call in blocks of 127 is one launch of 32 blocks and the short last one, not two launches)
    chunk = (chunk + sb - 1) / sb * sb;
    // ... and bound the block sums of one launch (16 B per pixel and block): more launches rather than more memory.  A
    // small sample_block is what a caller sets to get the reference's per-sample sum order; it must not cost 100 GB.
    // The launch also needs the primary-pass records and the task list (36 B per pixel + counters).
    const size_t fixed_bytes = n * (pt::kPrimBytesPerPixel + pt::kTaskBytesPerPixel) + 2 * pt::kTaskCounterBytes;
    const size_t partials_budget = noalloc ? kNoAllocScratchBytes - fixed_bytes : kPartialsBudgetBytes;
    size_t budget_blocks = partials_budget / (n * sizeof(float4));
    if (budget_blocks < 2) budget_blocks = 2;  // (ir_render_noalloc refused such a frame above)
    if ((size_t)chunk / sb > budget_blocks) chunk = (uint32_t)budget_blocks * sb;
    const uint32_t max_blocks = ((spp < chunk ? spp : chunk) + sb - 1) / sb;
    if ((uint64_t)n * max_blocks > 0xffffffffull) return fail(ctx, IR_ELIMIT, "%s: too many tasks in one launch", who);
    a.sample_block = sb;
    if (noalloc) {
        // carve the static array: records | task list | counters | block sums
        char* base = (ctx->device < kMaxDevices) ? (char*)g_noalloc_scratch[ctx->device] : nullptr;
        if (!base) return fail(ctx, IR_ESTATE, "%s: no static scratch on device %d", who, ctx->device);
        a.prim = (float4*)base;
        size_t off = n * pt::kPrimBytesPerPixel;
        a.prim_disc = (uint32_t*)(base + off);
        off = (off + n * pt::kTaskBytesPerPixel + pt::kTaskCounterBytes - 1) / pt::kTaskCounterBytes * pt::kTaskCounterBytes;
        a.task_counters = (uint32_t*)(base + off);
        a.task_counters_next = nullptr;  // the array is shared by every context of the device: zeroed per launch
        off += pt::kTaskCounterBytes;
        a.partials = (float4*)(base + off);
    } else {
        if (max_blocks > 1) IR_TRY(ensure(ctx, ctx->d_partials, (size_t)max_blocks * n * sizeof(float4)));
        IR_TRY(ensure(ctx, ctx->d_prim, n * pt::kPrimBytesPerPixel));
        IR_TRY(ensure(ctx, ctx->d_task_pixels, n * pt::kTaskBytesPerPixel));
        if (!ctx->d_task_counters.p || !ctx->task_counters_clean) {
            // two pairs of counters used in turn: a launch's primary pass zeroes the pair of the next launch
            IR_TRY(ensure(ctx, ctx->d_task_counters, pt::kTaskCounterBytes));
            IR_HIP(ctx, hipMemsetAsync(ctx->d_task_counters.p, 0, pt::kTaskCounterBytes, ctx->stream));
            ctx->task_counters_clean = true;
            ctx->launch_parity = 0;
        }
        a.partials = (float4*)ctx->d_partials.p;
        a.prim = (float4*)ctx->d_prim.p;
        a.prim_disc = (uint32_t*)ctx->d_task_pixels.p;
    }
    IR_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    uint32_t done = 0;
    while (done < spp) {
        const uint32_t now = (spp - done < chunk) ? (spp - done) : chunk;
        a.spp = now;
        a.sample_base = (uint32_t)(base + done);
        a.fresh = (fresh_in_kernel && done == 0) ? 1u : 0u;
        if (!noalloc && owned_wgs != 0) {  // (a rank that owns no tile launches nothing: the pairs stay as they are)
            uint32_t* const pairs = (uint32_t*)ctx->d_task_counters.p;
            a.task_counters = pairs + ctx->launch_parity * pt::kTaskCounterPairStride;
            a.task_counters_next = pairs + (ctx->launch_parity ^ 1u) * pt::kTaskCounterPairStride;
            ctx->launch_parity ^= 1u;
        }
        // diagnostics for tools/wave_timeline.py: every wave of the path kernel stamps its start and end; the launch is
        // waited for and the stamps are written to the named file (one line per wave), launch after launch
        static const char* const wave_times_path = getenv("IR_DEBUG_WAVE_TIMES");
        constexpr uint32_t kWaveTimesCap = 1u << 16;
        a.wave_times = nullptr;
        a.wave_times_cap = 0;
        if (wave_times_path) {
            IR_TRY(ensure(ctx, ctx->d_wave_times, (size_t)kWaveTimesCap * 16));
            IR_HIP(ctx, hipMemsetAsync(ctx->d_wave_times.p, 0, (size_t)kWaveTimesCap * 16, ctx->stream));
            a.wave_times = (unsigned long long*)ctx->d_wave_times.p;
            a.wave_times_cap = kWaveTimesCap;
        }
        const uint32_t timer = (uint32_t)(ctx->launches_timed % ir_context::kTimerRing);
        const hipError_t le = pt::launch_render(a, accel, ctx->stats_enabled, ctx->stream, ctx->ev_p0[timer], ctx->ev_p1[timer]);
        if (wave_times_path && le == hipSuccess) {
            std::vector<unsigned long long> stamps((size_t)kWaveTimesCap * 2);
            IR_HIP(ctx, hipMemcpyAsync(stamps.data(), ctx->d_wave_times.p, stamps.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (FILE* f = fopen(wave_times_path, "w")) {
                for (uint32_t wv = 0; wv < kWaveTimesCap; ++wv)
                    if (stamps[2 * (size_t)wv]) fprintf(f, "%llu %llu\n", stamps[2 * (size_t)wv], stamps[2 * (size_t)wv + 1]);
                fclose(f);
            }
        }
        if (le != hipSuccess) {
            ctx->task_counters_clean = false;  // whatever ran may have left the counters anywhere
            return fail(ctx, IR_EDEVICE, "%s: launch failed: %s", who, hipGetErrorString(le));
        }
        if (owned_wgs != 0) {  // (a rank that owns no tile recorded nothing)
            ctx->paths_timed = true;
            ctx->launches_timed++;
        }
        done += now;
        ctx->last_launches++;
    }
    IR_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->events_recorded = true;
    ctx->samples_done += spp;
    if (ctx->stats_enabled) {
        ctx->stats_valid = true;
        ctx->stats_enabled = false;  // one render only
    }
    return IR_OK;
}

}  // namespace

extern "C" {

int ir_render_async(ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, int accumulate) {
    if (!ctx) return IR_EINVAL;
    return render_impl(ctx, width, height, spp, accumulate, nullptr, nullptr);
}

int ir_render_noalloc(ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, int accumulate, void* d_sum_rgba,
                      void* d_counts) {
    if (!ctx) return IR_EINVAL;
    if (!d_sum_rgba || !d_counts) return fail(ctx, IR_EINVAL, "ir_render_noalloc: NULL output buffer");
    if (((uintptr_t)d_sum_rgba & 15u) || ((uintptr_t)d_counts & 3u)) return fail(ctx, IR_EINVAL, "ir_render_noalloc: misaligned output buffer");
    return render_impl(ctx, width, height, spp, accumulate, (float4*)d_sum_rgba, (uint32_t*)d_counts);
}

int ir_synchronize(ir_context* ctx) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->wf_check_overflow) {
        ctx->wf_check_overflow = false;
        uint32_t flag = 0;
        IR_HIP(ctx, hipMemcpy(&flag, (uint32_t*)ctx->d_wf_counts.p + 3, sizeof(flag), hipMemcpyDeviceToHost));
        if (flag) return fail(ctx, IR_EDEVICE, "wavefront path queue overflowed: the frame is incomplete (internal sizing error)");
    }
    return IR_OK;
}

int ir_render(ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, int accumulate) {
    if (!ctx) return IR_EINVAL;
    const auto t0 = std::chrono::steady_clock::now();
    IR_TRY(ir_render_async(ctx, width, height, spp, accumulate));
    IR_TRY(ir_synchronize(ctx));
    ctx->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return IR_OK;
}

int ir_get_output_argb(ir_context* ctx, uint32_t* out, size_t n) {
    if (!ctx) return IR_EINVAL;
    const size_t have = (size_t)ctx->width * ctx->height;
    if (n != have) return fail(ctx, IR_EINVAL, "ir_get_output_argb: n=%zu but the frame has %zu pixels", n, have);
    if (n == 0) return IR_OK;
    if (!out) return fail(ctx, IR_EINVAL, "ir_get_output_argb: NULL");
    IR_TRY(use_device(ctx));
    IR_TRY(ensure(ctx, ctx->d_argb, n * sizeof(uint32_t)));
    const float4* sum;
    const uint32_t* counts;
    IR_TRY(output_film(ctx, &sum, &counts));
    IR_HIP(ctx, pt::launch_resolve(sum, counts, (uint32_t*)ctx->d_argb.p, n, ctx->stream));
    IR_HIP(ctx, hipMemcpyAsync(out, ctx->d_argb.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

int ir_map_output_argb(ir_context* ctx, const uint32_t** out, size_t* out_n) {
    if (!ctx) return IR_EINVAL;
    if (!out) return fail(ctx, IR_EINVAL, "ir_map_output_argb: NULL");
    const size_t n = (size_t)ctx->width * ctx->height;
    *out = nullptr;
    if (out_n) *out_n = n;
    if (n == 0) return IR_OK;
    IR_TRY(use_device(ctx));
    if (ctx->h_argb_bytes < n * sizeof(uint32_t)) {
        if (ctx->h_argb) (void)hipHostFree(ctx->h_argb);
        ctx->h_argb = nullptr;
        ctx->h_argb_bytes = 0;
        const hipError_t e = hipHostMalloc((void**)&ctx->h_argb, n * sizeof(uint32_t), hipHostMallocDefault);
        if (e != hipSuccess) {
            ctx->h_argb = nullptr;
            return fail(ctx, IR_ENOMEM, "hipHostMalloc(%zu bytes): %s", n * sizeof(uint32_t), hipGetErrorString(e));
        }
        ctx->h_argb_bytes = n * sizeof(uint32_t);
    }
    IR_TRY(ensure(ctx, ctx->d_argb, n * sizeof(uint32_t)));
    const float4* sum;
    const uint32_t* counts;
    IR_TRY(output_film(ctx, &sum, &counts));
    IR_HIP(ctx, pt::launch_resolve(sum, counts, (uint32_t*)ctx->d_argb.p, n, ctx->stream));
    IR_HIP(ctx, hipMemcpyAsync(ctx->h_argb, ctx->d_argb.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = ctx->h_argb;
    return IR_OK;
}

int ir_get_radiance(ir_context* ctx, float* sum_rgba, uint32_t* counts, size_t n) {
    if (!ctx) return IR_EINVAL;
    const size_t have = (size_t)ctx->width * ctx->height;
    if (n != have) return fail(ctx, IR_EINVAL, "ir_get_radiance: n=%zu but the frame has %zu pixels", n, have);
    if (n == 0) return IR_OK;
    IR_TRY(use_device(ctx));
    const float4* d_sum;
    const uint32_t* d_counts;
    IR_TRY(output_film(ctx, &d_sum, &d_counts));
    if (sum_rgba) IR_HIP(ctx, hipMemcpyAsync(sum_rgba, d_sum, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (counts) IR_HIP(ctx, hipMemcpyAsync(counts, d_counts, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

int ir_set_radiance(ir_context* ctx, const float* sum_rgba, const uint32_t* counts, size_t n) {
    if (!ctx) return IR_EINVAL;
    const size_t have = (size_t)ctx->width * ctx->height;
    if (n != have) return fail(ctx, IR_EINVAL, "ir_set_radiance: n=%zu but the frame has %zu pixels", n, have);
    if (n == 0) return IR_OK;
    IR_TRY(use_device(ctx));
    ctx->output_combined = false;
    if (sum_rgba) IR_HIP(ctx, hipMemcpyAsync(film_sum(ctx), sum_rgba, n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    if (counts) IR_HIP(ctx, hipMemcpyAsync(film_counts(ctx), counts, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

int ir_get_primary_hits(ir_context* ctx, int32_t* out_index, float* out_t, size_t n) {
    if (!ctx) return IR_EINVAL;
    uint32_t w = ctx->width, h = ctx->height;
    if ((w == 0 || h == 0) && ctx->rays_w) {
        w = ctx->rays_w;
        h = ctx->rays_h;
    }
    const size_t have = (size_t)w * h;
    if (n != have) return fail(ctx, IR_EINVAL, "ir_get_primary_hits: n=%zu but the frame has %zu pixels", n, have);
    if (n == 0) return IR_OK;
    if (!out_index) return fail(ctx, IR_EINVAL, "ir_get_primary_hits: NULL");
    IR_TRY(use_device(ctx));
    const float* rays = nullptr;
    IR_TRY(check_camera(ctx, w, h, &rays));
    IR_TRY(ensure(ctx, ctx->d_spheres, 16));
    IR_TRY(ensure_bvh(ctx));
    IR_TRY(ensure(ctx, ctx->d_hit_idx, n * sizeof(int32_t)));
    if (out_t) IR_TRY(ensure(ctx, ctx->d_hit_t, n * sizeof(float)));
    pt::HitsArgs a;
    memset(&a, 0, sizeof(a));
    fill_camera(ctx, a.cam);
    a.rays = rays;
    a.w = w;
    a.h = h;
    a.spheres = (const float4*)ctx->d_spheres.p;
    a.n_spheres = ctx->n_spheres;
    fill_bvh(ctx, a.bvh);
    a.out_idx = (int32_t*)ctx->d_hit_idx.p;
    a.out_t = out_t ? (float*)ctx->d_hit_t.p : nullptr;
    IR_HIP(ctx, pt::launch_primary_hits(a, selected_accel(ctx), ctx->stream));
    IR_HIP(ctx, hipMemcpyAsync(out_index, ctx->d_hit_idx.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (out_t) IR_HIP(ctx, hipMemcpyAsync(out_t, ctx->d_hit_t.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

int ir_get_size(const ir_context* ctx, uint32_t* width, uint32_t* height) {
    if (!ctx) return IR_EINVAL;
    if (width) *width = ctx->width;
    if (height) *height = ctx->height;
    return IR_OK;
}

double ir_last_render_seconds(const ir_context* ctx) { return ctx ? ctx->last_seconds : 0.0; }

double ir_last_kernel_ms(const ir_context* ctx, uint32_t* n_launches) {
    if (n_launches) *n_launches = ctx ? ctx->last_launches : 0;
    if (!ctx || !ctx->events_recorded) return 0.0;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0.0;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return 0.0;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return 0.0;
    return (double)ms;
}

double ir_last_paths_kernel_ms(const ir_context* ctx) {
    if (!ctx || !ctx->paths_timed || ctx->launches_timed == 0) return 0.0;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0.0;
    const uint32_t k = (uint32_t)((ctx->launches_timed - 1) % ir_context::kTimerRing);
    if (hipEventSynchronize(ctx->ev_p1[k]) != hipSuccess) return 0.0;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_p0[k], ctx->ev_p1[k]) != hipSuccess) return 0.0;
    return (double)ms;
}

int ir_get_paths_kernel_history(const ir_context* ctx, double* out_ms, uint32_t max_n, uint32_t* out_n) {
    if (!ctx) return IR_EINVAL;
    if (!out_n || (!out_ms && max_n)) return fail(ctx, IR_EINVAL, "ir_get_paths_kernel_history: NULL");
    uint64_t n = ctx->launches_timed < ir_context::kTimerRing ? ctx->launches_timed : ir_context::kTimerRing;
    if (n > max_n) n = max_n;
    *out_n = (uint32_t)n;
    if (n == 0) return IR_OK;
    IR_TRY(use_device(ctx));
    const uint64_t first = ctx->launches_timed - n;
    IR_HIP(ctx, hipEventSynchronize(ctx->ev_p1[(ctx->launches_timed - 1) % ir_context::kTimerRing]));
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t k = (uint32_t)((first + i) % ir_context::kTimerRing);
        float ms = 0.0f;
        IR_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_p0[k], ctx->ev_p1[k]));
        out_ms[i] = (double)ms;
    }
    return IR_OK;
}

uint32_t ir_samples_accumulated(const ir_context* ctx) { return ctx ? ctx->samples_done : 0; }

int ir_set_samples_accumulated(ir_context* ctx, uint32_t samples) {
    if (!ctx) return IR_EINVAL;
    ctx->samples_done = samples;
    return IR_OK;
}

void* ir_device_radiance(ir_context* ctx) { return ctx ? (void*)film_sum(ctx) : nullptr; }
void* ir_device_counts(ir_context* ctx) { return ctx ? (void*)film_counts(ctx) : nullptr; }

int ir_bind_accumulator(ir_context* ctx, void* d_sum_rgba, void* d_counts, uint32_t width, uint32_t height) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->output_combined = false;
    if (!d_sum_rgba && !d_counts) {
        ctx->ext_sum = ctx->ext_counts = nullptr;
        ctx->ext_w = ctx->ext_h = 0;
        ctx->width = ctx->height = 0;  // next render reallocates the context-owned film
        ctx->samples_done = 0;
        return IR_OK;
    }
    if (!d_sum_rgba || !d_counts) return fail(ctx, IR_EINVAL, "ir_bind_accumulator: both buffers are required");
    if (((uintptr_t)d_sum_rgba & 15u) || ((uintptr_t)d_counts & 3u)) return fail(ctx, IR_EINVAL, "ir_bind_accumulator: misaligned buffer");
    ctx->ext_sum = d_sum_rgba;
    ctx->ext_counts = d_counts;
    ctx->ext_w = width;
    ctx->ext_h = height;
    ctx->width = width;
    ctx->height = height;
    ctx->samples_done = 0;
    return IR_OK;
}

int ir_enable_stats(ir_context* ctx, int enable) {
    if (!ctx) return IR_EINVAL;
    ctx->stats_enabled = enable != 0;
    return IR_OK;
}

int ir_get_stats(ir_context* ctx, uint64_t out5[5]) {
    if (!ctx) return IR_EINVAL;
    if (!out5) return fail(ctx, IR_EINVAL, "ir_get_stats: NULL");
    if (!ctx->stats_valid) return fail(ctx, IR_ESTATE, "ir_get_stats: the last render did not collect statistics");
    IR_TRY(use_device(ctx));
    unsigned long long tmp[5];
    IR_HIP(ctx, hipMemcpyAsync(tmp, ctx->d_stats.p, sizeof(tmp), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 5; ++i) out5[i] = tmp[i];
    return IR_OK;
}

int ir_get_accel_stats(ir_context* ctx, uint64_t out2[2]) {
    if (!ctx) return IR_EINVAL;
    if (!out2) return fail(ctx, IR_EINVAL, "ir_get_accel_stats: NULL");
    if (!ctx->stats_valid) return fail(ctx, IR_ESTATE, "ir_get_accel_stats: the last render did not collect statistics");
    IR_TRY(use_device(ctx));
    unsigned long long tmp[pt::kNumStats];
    IR_HIP(ctx, hipMemcpyAsync(tmp, ctx->d_stats.p, sizeof(tmp), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out2[0] = tmp[5];
    out2[1] = tmp[6];
    return IR_OK;
}

int ir_get_primary_stats(ir_context* ctx, uint64_t out3[3]) {
    if (!ctx) return IR_EINVAL;
    if (!out3) return fail(ctx, IR_EINVAL, "ir_get_primary_stats: NULL");
    if (!ctx->stats_valid) return fail(ctx, IR_ESTATE, "ir_get_primary_stats: the last render did not collect statistics");
    IR_TRY(use_device(ctx));
    unsigned long long tmp[pt::kNumStats];
    IR_HIP(ctx, hipMemcpyAsync(tmp, ctx->d_stats.p, sizeof(tmp), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 3; ++i) out3[i] = tmp[15 + i];
    return IR_OK;
}

int ir_get_lane_profile(ir_context* ctx, uint64_t out8[8]) {
    if (!ctx) return IR_EINVAL;
    if (!out8) return fail(ctx, IR_EINVAL, "ir_get_lane_profile: NULL");
    if (!ctx->stats_valid) return fail(ctx, IR_ESTATE, "ir_get_lane_profile: the last render did not collect statistics");
    IR_TRY(use_device(ctx));
    unsigned long long tmp[pt::kNumStats];
    IR_HIP(ctx, hipMemcpyAsync(tmp, ctx->d_stats.p, sizeof(tmp), hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i) out8[i] = tmp[7 + i];
    return IR_OK;
}

int ir_get_accel_info(ir_context* ctx, int* out_accel, uint32_t* out_nodes, uint32_t* out_depth) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    IR_TRY(ensure_bvh(ctx));
    const int accel = selected_accel(ctx);
    if (out_accel) *out_accel = (accel == pt::kAccelLbvh) ? IR_ACCEL_LBVH : IR_ACCEL_LINEAR;
    if (out_nodes) *out_nodes = (accel == pt::kAccelLbvh) ? ctx->bvh_nodes : 0;
    if (out_depth) *out_depth = (accel == pt::kAccelLbvh) ? ctx->bvh_depth : 0;
    return IR_OK;
}

int ir_get_accel_build_info(ir_context* ctx, int* out_builder, double* out_seconds) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    IR_TRY(ensure_bvh(ctx));
    const bool lbvh = selected_accel(ctx) == pt::kAccelLbvh;
    if (out_builder) *out_builder = lbvh ? ctx->bvh_built_by : IR_BUILD_HOST;
    if (out_seconds) *out_seconds = lbvh ? ctx->bvh_build_seconds : 0.0;
    return IR_OK;
}

int ir_download_accel_climb(ir_context* ctx, void* climb_records, uint32_t n_records, int32_t* sphere_start, uint32_t n_spheres) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    IR_TRY(ensure_bvh(ctx));
    if (selected_accel(ctx) != pt::kAccelLbvh) return fail(ctx, IR_ESTATE, "ir_download_accel_climb: the scene is rendered with the linear scan");
    const uint32_t have = ctx->bvh_nodes + ctx->n_spheres;  // records of inner nodes (the root's slot unused) and of leaves, by first position
    if (n_records != have || n_spheres != ctx->n_spheres)
        return fail(ctx, IR_EINVAL, "ir_download_accel_climb: the LBVH has %u climb records over %u spheres, buffers are for %u / %u", have,
                    ctx->n_spheres, n_records, n_spheres);
    if (climb_records)
        IR_HIP(ctx, hipMemcpyAsync(climb_records, (const char*)ctx->d_bvh_nodes.p + (size_t)ctx->bvh_nodes * sizeof(pt::WideNode),
                                   (size_t)n_records * sizeof(pt::WideNode), hipMemcpyDeviceToHost, ctx->stream));
    if (sphere_start)
        IR_HIP(ctx, hipMemcpyAsync(sphere_start, ctx->d_bvh_start.p, (size_t)n_spheres * 8, hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

int ir_download_accel(ir_context* ctx, void* wide_nodes, uint32_t n_nodes, float* leaf_spheres, int32_t* leaf_index, uint32_t n_spheres) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    IR_TRY(ensure_bvh(ctx));
    if (selected_accel(ctx) != pt::kAccelLbvh) return fail(ctx, IR_ESTATE, "ir_download_accel: the scene is rendered with the linear scan");
    if (n_nodes != ctx->bvh_nodes || n_spheres != ctx->n_spheres)
        return fail(ctx, IR_EINVAL, "ir_download_accel: the LBVH has %u nodes over %u spheres, buffers are for %u / %u", ctx->bvh_nodes,
                    ctx->n_spheres, n_nodes, n_spheres);
    if (wide_nodes)
        IR_HIP(ctx, hipMemcpyAsync(wide_nodes, ctx->d_bvh_nodes.p, (size_t)n_nodes * sizeof(pt::WideNode), hipMemcpyDeviceToHost, ctx->stream));
    if (leaf_spheres)
        IR_HIP(ctx, hipMemcpyAsync(leaf_spheres, ctx->d_bvh_spheres.p, (size_t)n_spheres * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (leaf_index)
        IR_HIP(ctx, hipMemcpyAsync(leaf_index, ctx->d_bvh_index.p, (size_t)n_spheres * 4, hipMemcpyDeviceToHost, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

int ir_selftest_exact_math(ir_context* ctx, uint64_t* out_mismatches, uint32_t n_out, uint32_t log2_random_samples) {
    if (!ctx) return IR_EINVAL;
    if (!out_mismatches || n_out != pt::kNumSelftests)
        return fail(ctx, IR_EINVAL, "ir_selftest_exact_math: out must hold exactly %u counters", pt::kNumSelftests);
    if (log2_random_samples < 21 || log2_random_samples > 40)
        return fail(ctx, IR_EINVAL, "ir_selftest_exact_math: log2_random_samples %u not in [21, 40]", log2_random_samples);
    IR_TRY(use_device(ctx));
    DeviceBuf bad;
    IR_TRY(ensure(ctx, bad, pt::kNumSelftests * sizeof(unsigned long long)));
    int rc = IR_OK;
    unsigned long long tmp[pt::kNumSelftests];
    hipError_t e = hipMemsetAsync(bad.p, 0, sizeof(tmp), ctx->stream);
    if (e == hipSuccess) e = pt::launch_selftest((unsigned long long*)bad.p, log2_random_samples, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp, bad.p, sizeof(tmp), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, IR_EDEVICE, "ir_selftest_exact_math: %s", hipGetErrorString(e));
    release(bad);
    if (rc == IR_OK)
        for (uint32_t i = 0; i < pt::kNumSelftests; ++i) out_mismatches[i] = tmp[i];
    return rc;
}

void ir_free(void* p) { free(p); }

}  // extern "C"
