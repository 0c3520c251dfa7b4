// ir_internal.h — the context behind the C ABI (include/insploray_hip.h) and the helpers its translation units share
// (insploray_hip.hip: scene, camera, render, film; insploray_comm.hip: the multi-GPU exchange step).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/insploray_hip.h"

namespace irx {

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Comm;  // RCCL communicator + staging of the exchange step (insploray_comm.hip)

}  // namespace irx

struct ir_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;    // around all launches of a render call
    // around the path kernel of each launch: a ring of event pairs, so that a caller that enqueues many renders without
    // waiting (bench.py's timed region) can still read every launch's duration afterwards (ir_get_paths_kernel_history)
    static constexpr uint32_t kTimerRing = 64;
    hipEvent_t ev_p0[kTimerRing] = {}, ev_p1[kTimerRing] = {};
    uint64_t launches_timed = 0;  // launches whose path kernel was bracketed since the context was created
    bool paths_timed = false;     // the last render launched a path kernel
    mutable std::string err;

    // scene (host copies kept so that the per-sphere shading table can be rebuilt when either side changes)
    std::vector<float> h_spheres;   // n x 4
    std::vector<int32_t> h_matid;   // n
    std::vector<float> h_materials; // m x 9
    bool shade_dirty = true;
    bool fast_materials = false;
    uint32_t n_spheres = 0;
    irx::DeviceBuf d_spheres, d_shade;
    // LBVH (built before the first launch that needs it, on the host or on the device: ir_set_accel_build)
    irx::DeviceBuf d_bvh_nodes, d_bvh_spheres, d_bvh_index, d_bvh_start, d_bvh_scratch;  // d_bvh_nodes: wide nodes + climb records (lbvh.h)
    uint32_t bvh_nodes = 0, bvh_depth = 0, bvh_stack = 0;
    bool bvh_valid = false;
    int bvh_build = IR_BUILD_AUTO;
    int bvh_built_by = IR_BUILD_HOST;
    uint32_t bvh_topology_n = 0;  // sphere count whose device-built radix tree is still in d_bvh_scratch (0 = none)
    double bvh_build_seconds = 0.0;
    float sky[3] = {0.0f, 0.0f, 0.0f};
    irx::DeviceBuf d_skybox;
    uint32_t sky_w = 0, sky_h = 0;
    int sky_filter = IR_SKY_NEAREST;
    int sky_sampling = IR_SKY_SAMPLING_BSDF;
    irx::DeviceBuf d_sky_marginal, d_sky_cond, d_sky_pdf, d_sky_marginal_guide, d_sky_cond_guide;
    bool sky_tables_valid = false;
    irx::DeviceBuf d_wave_times;   // diagnostics (IR_DEBUG_WAVE_TIMES): start / end time of the path kernel's waves
    irx::DeviceBuf d_skyquad;      // footprint table of the bilinear lookup (pt_kernels.h skyquad), built for large skies
    bool sky_quad_valid = false;
    int accel = IR_ACCEL_AUTO;

    // camera
    bool have_pinhole = false;
    float cam_m[16] = {0};
    float cam_pos[3] = {0};
    float cam_fov = 0.0f, cam_aspect = 1.0f;
    irx::DeviceBuf d_rays;
    uint32_t rays_w = 0, rays_h = 0;

    // integrator (reference defaults, renderer.rs:38-41)
    uint32_t bounces = 5, rr_start = 2;
    uint32_t sample_block = 128;  // IR_SAMPLE_BLOCK_AUTO (0): chosen per render from the launch grid
    irx::DeviceBuf d_partials;
    bool task_counters_clean = false;  // both counter pairs of d_task_counters are zero / being kept zero by the launches
    uint32_t launch_parity = 0;        // which pair the next launch uses
    irx::DeviceBuf d_prim, d_task_pixels, d_task_counters;  // launch scratch of the primary pass / task dispenser
    // wavefront schedule: path queues and per-sample values (allocated on first use)
    int schedule = IR_SCHEDULE_AUTO;
    bool wf_check_overflow = false;
    uint32_t wf_max_paths = 128u << 20;
    irx::DeviceBuf d_wf_queue[2], d_wf_counts, d_wf_values, d_wf_prim, d_wf_tileacc;
    uint64_t seed = 0x0123456789ABCDEFull;

    // partition
    int part_mode = IR_PARTITION_NONE;
    uint32_t rank = 0, world = 1;
    irx::Comm* comm = nullptr;     // ir_comm_init
    bool output_combined = false;  // the last ir_combine finished after the last render: readbacks show the combined film

    // film
    uint32_t width = 0, height = 0;
    irx::DeviceBuf d_sum, d_counts, d_argb, d_hit_idx, d_hit_t;
    uint32_t* h_argb = nullptr;  // pinned host frame handed out by ir_map_output_argb
    size_t h_argb_bytes = 0;
    void* ext_sum = nullptr;
    void* ext_counts = nullptr;
    uint32_t ext_w = 0, ext_h = 0;
    uint32_t samples_done = 0;

    // stats
    bool stats_enabled = false;
    irx::DeviceBuf d_stats;
    bool stats_valid = false;

    // timing
    double last_seconds = 0.0;
    bool events_recorded = false;
    uint32_t last_launches = 0;
    uint32_t last_sample_block = 0;
};

namespace irx {

// records the message on the context (or, without one, for ir_last_global_error) and returns `code`
int fail(const ir_context* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

#define IR_HIP(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return irx::fail(ctx, IR_EDEVICE, "%s: %s", #call, hipGetErrorString(e_));   \
    } while (0)

#define IR_TRY(expr)                  \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != IR_OK) return rc_; \
    } while (0)

int ensure(ir_context* ctx, DeviceBuf& b, size_t bytes);  // grow-only device buffer
void release(DeviceBuf& b);
int use_device(const ir_context* ctx);
int upload(ir_context* ctx, DeviceBuf& b, const void* src, size_t bytes);  // blocking, through the context's stream

inline float4* film_sum(ir_context* ctx) { return (float4*)(ctx->ext_sum ? ctx->ext_sum : ctx->d_sum.p); }
inline uint32_t* film_counts(ir_context* ctx) { return (uint32_t*)(ctx->ext_counts ? ctx->ext_counts : ctx->d_counts.p); }

// exchange step (insploray_comm.hip)
void comm_release(ir_context* ctx);                 // ir_destroy
// Film a readback shows: the combined film on the root after ir_combine, else the context's accumulator.  Makes the
// context's stream wait for an exchange still in flight.
int output_film(ir_context* ctx, const float4** sum, const uint32_t** counts);

}  // namespace irx
