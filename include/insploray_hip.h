/* insploray_hip.h — C ABI of the MI355X (gfx950) rendering backend for InsploRay's core-engine.
 *
 * This is the drop-in boundary: exactly the entry points a `insploray-hip-sys` FFI crate would bind so
 * that `insploray::renderer::RayTracer` (reference core-engine/src/renderer.rs:13-173) keeps its public
 * surface while the per-pixel integrator (integrator.rs:32-165), ray generation
 * (cameras/pinhole_camera.rs:78-98), sky lookup (file_formats/exr.rs:13-24), accumulation
 * (accumulators/) and resolve (full_buffer.rs:49-65, utils.rs:4-9) run as HIP kernels.
 * All citations are relative to /root/reference/core-engine/src.
 *
 * Conventions
 *   - plain pointers and sizes only; every host pointer argument is borrowed for the duration of the
 *     call and copied before return;
 *   - every function returns 0 on success, a negative IR_E* code on failure, never aborts;
 *     ir_last_error(ctx) gives the message of the last failure on that context;
 *   - a context is bound to ONE GPU (one process per GPU; multi-GPU = one context per rank, partial
 *     radiance buffers combined on a root rank by ir_combine over RCCL, see ir_set_partition / ir_comm_init;
 *     a caller with its own transport can use ir_bind_accumulator or ir_pack_owned_tiles instead);
 *   - a context is single-caller (not re-entrant), like RayTracer which is `&mut self` everywhere;
 *   - there is no CPU fallback: without a HIP device ir_create fails.
 */
#ifndef INSPLORAY_HIP_H
#define INSPLORAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ir_context ir_context;

enum {
    IR_OK = 0,
    IR_EINVAL = -1,   /* bad argument (null pointer, zero size, index out of range)            */
    IR_EDEVICE = -2,  /* HIP runtime error (message in ir_last_error)                           */
    IR_ESTATE = -3,   /* call order problem (e.g. render before a camera was set)               */
    IR_ENOMEM = -4,   /* device or host allocation failed                                       */
    IR_EIO = -5,      /* file could not be read / not a supported EXR                           */
    IR_ELIMIT = -6    /* size beyond what this build supports                                   */
};

enum { IR_SKY_NEAREST = 0, IR_SKY_BILINEAR = 1 };  /* exr.rs:20-23 is NEAREST */
enum { IR_SKY_SAMPLING_BSDF = 0, IR_SKY_SAMPLING_MIS = 1 };  /* integrator.rs:64-70,84-89 is BSDF */
enum { IR_PARTITION_NONE = 0, IR_PARTITION_TILES = 1, IR_PARTITION_SAMPLES = 2 };
enum { IR_ACCEL_AUTO = 0, IR_ACCEL_LINEAR = 1, IR_ACCEL_LBVH = 2 };
enum { IR_BUILD_AUTO = 0, IR_BUILD_HOST = 1, IR_BUILD_DEVICE = 2, IR_BUILD_REFIT = 3, IR_BUILD_HOST_SAH = 4 };
enum { IR_SCHEDULE_AUTO = 0, IR_SCHEDULE_REGENERATION = 1, IR_SCHEDULE_WAVEFRONT = 2 };
enum { IR_SAMPLE_BLOCK_AUTO = 0 };  /* ir_set_sample_block */
#define IR_COMM_ID_BYTES 128        /* sizeof(ncclUniqueId), ir_comm_unique_id / ir_comm_init */

/* ABI version of this header (bumped on incompatible change). */
uint32_t ir_abi_version(void);
/* HIP_VERSION this library was compiled against and hipRuntimeGetVersion() of the runtime copy the process ended up with
 * (major * 10000000 + minor * 100000 + patch).  A process that also holds PyTorch-ROCm runs on the runtime torch bundles
 * (INTEGRATION.md section 3); the Python mirror warns when the two differ in their major version.  No counterpart in the
 * reference (it has no device runtime). */
int ir_hip_versions(int* out_compiled, int* out_runtime);

/* -------------------------------------------------------------------------------------------------
 * Context.  Replaces RayTracer::new / Drop (renderer.rs:29-59, 182-191): where the reference spawns a
 * 4-thread pool, this creates the device state (stream, events, buffers) on HIP device `device`.
 * ------------------------------------------------------------------------------------------------- */
int ir_create(ir_context** out_ctx, int device);
void ir_destroy(ir_context* ctx);
const char* ir_last_error(const ir_context* ctx);
/* Message of the last failure of a call that had no usable context (ir_create). Thread-local. */
const char* ir_last_global_error(void);

/* Use an existing HIP stream (e.g. torch's current stream) for all launches and copies of this context.
 * `hip_stream` is a hipStream_t passed as void*; NULL restores the context's own stream. */
int ir_set_stream(ir_context* ctx, void* hip_stream);

/* -------------------------------------------------------------------------------------------------
 * Scene (scene.rs:5-41).  The reference Scene is a caller-owned AoS the renderer read-locks per render;
 * the façade flattens it and calls these before ir_render (there is no dirty flag in the reference API).
 *   spheres_xyzr : n x 4 floats  (Sphere.position.xyz, Sphere.radius)          scene.rs:27-31
 *   material_id  : n x int32     (Sphere.material_id; <0 or >= n_materials => DEFAULT_MATERIAL,
 *                                 integrator.rs:8-14,53-60)
 *   materials9   : n x 9 floats  (albedo.xyz, roughness, metalic, emission_color.xyz, emissive_power)
 *                                 scene.rs:5-12
 * ------------------------------------------------------------------------------------------------- */
int ir_set_spheres(ir_context* ctx, const float* spheres_xyzr, const int32_t* material_id, uint32_t n);
int ir_set_materials(ir_context* ctx, const float* materials9, uint32_t n);
/* Scene.default_sky_color (scene.rs:38), used when there is no skybox (integrator.rs:84-87). */
int ir_set_sky_color(ir_context* ctx, const float rgb[3]);
/* Scene.skybox = Some(ExrImage{pixels_buffer,width,height}) (exr.rs:6-10): rgb is w*h*3 floats, row-major,
 * row 0 first (exr.rs:38).  rgb == NULL removes the skybox (Scene.skybox = None). */
int ir_set_skybox(ir_context* ctx, const float* rgb, uint32_t w, uint32_t h);
/* IR_SKY_NEAREST (default) reproduces ExrImage::sample (exr.rs:13-24).  IR_SKY_BILINEAR is an extension the
 * reference does not have: same u, v; texel centres at (i + 0.5) / n, longitude wraps, latitude clamps,
 * a + (b - a) * t along x then y. */
int ir_set_sky_filter(ir_context* ctx, int filter);
/* IR_SKY_SAMPLING_BSDF (default) is the reference's estimator: a bounce direction from its hemisphere sampler
 * (sampler.rs:43-53), the sky met only when that ray escapes (integrator.rs:84-89).  IR_SKY_SAMPLING_MIS is an
 * extension for sun-dominated skies (workload C5): at every hit whose bounce ray is traced, a second direction is drawn
 * from the sky image's own distribution (luminance x solid angle), traced as a shadow ray, and the two strategies are
 * combined with the balance heuristic.  Same expectation as the reference's estimator, far lower
 * variance; costs one more trace per bounce.  Needs a skybox (with a constant sky colour it changes nothing).  Both
 * schedules and both closest-hit structures implement it with the same draws and sums. */
int ir_set_sky_sampling(ir_context* ctx, int mode);
/* Closest-hit structure: IR_ACCEL_LINEAR = the reference's linear scan (integrator.rs:106-127) over an
 * LDS-staged SoA buffer (at most 10 000 spheres); IR_ACCEL_LBVH = linear BVH over the spheres whose (t, index)
 * equals the scan's bit for bit (leaf tests are the scan's arithmetic, node tests are conservative, ties keep the
 * lowest index); IR_ACCEL_AUTO (default) = scan below 64 spheres (the measured crossover), LBVH from there. */
int ir_set_accel(ir_context* ctx, int accel);
/* What the next launch will use for the current scene: *out_accel = IR_ACCEL_LINEAR or IR_ACCEL_LBVH, LBVH node
 * count and depth (0 for the scan).  Builds the LBVH if it is selected and stale.  Any pointer may be NULL. */
int ir_get_accel_info(ir_context* ctx, int* out_accel, uint32_t* out_nodes, uint32_t* out_depth);
/* Where the LBVH is built when the scene changed (the reference's editor, app.rs:59-116, edits spheres between
 * frames; the reference itself has nothing to rebuild).  IR_BUILD_HOST: Morton sort and top-down splits on the CPU,
 * arrays uploaded.  IR_BUILD_DEVICE: the same tree built by kernels from the spheres already on the device (radix
 * sort, Karras radix tree, bottom-up boxes), byte-identical arrays; a tree too deep for the traversal stack falls
 * back to the host's median-split build.  IR_BUILD_AUTO (default): device from 2048 spheres.  IR_BUILD_REFIT: device
 * build when the sphere count changed (or first), otherwise the previous tree is kept and only its boxes follow the
 * moved spheres (no sort, no tree construction): for edits that move spheres a little, frame after frame.  Hits stay
 * the scan's bit for bit with any valid tree; traversal gets slower the further the spheres stray from where the tree
 * was built, so select IR_BUILD_DEVICE again (or change the count) to rebuild.  IR_BUILD_HOST_SAH: host build that
 * splits by a binned surface-area heuristic instead of Morton codes: several times the build time (10 000 spheres: ~6 ms)
 * for a tree that traverses faster (1.03x on the 10 000 uniformly scattered spheres of BASELINE config 3, 1.10x on 100 000,
 * 1.13x / 1.31x on 10 000 / 100 000 clustered spheres, profiles/r3_sah_vs_morton.txt); for scenes that stay as they are for
 * many launches.  Same hits, same bits. */
int ir_set_accel_build(ir_context* ctx, int build);
/* Builder that produced the current LBVH (IR_BUILD_HOST / IR_BUILD_DEVICE / IR_BUILD_REFIT / IR_BUILD_HOST_SAH) and the wall time of that build in
 * seconds, uploads and synchronisation included.  Builds the LBVH if it is selected and stale. */
int ir_get_accel_build_info(ir_context* ctx, int* out_builder, double* out_seconds);
/* Copies the LBVH arrays to the host (for inspection and tests): n_nodes x 48-byte wide nodes (layout in
 * path_tracer_amd/csrc/lbvh.h), n_spheres x 4 floats (cx, cy, cz, r*r) and n_spheres original indices in leaf
 * order.  n_nodes / n_spheres must equal ir_get_accel_info's node count / the scene's sphere count; any pointer
 * may be NULL.  IR_ESTATE when the scene is rendered with the linear scan. */
int ir_download_accel(ir_context* ctx, void* wide_nodes, uint32_t n_nodes, float* leaf_spheres, int32_t* leaf_index,
                      uint32_t n_spheres);
/* The rest of the LBVH (for inspection and tests): the climb records that let a bounce ray start its traversal at the
 * leaf of the sphere it leaves (layout in path_tracer_amd/csrc/lbvh.h "Climb records": n_records = node count +
 * n_spheres entries of 48 bytes, those of the inner nodes first, then those of the leaves by first leaf position) and,
 * per sphere in original order, two int32: the sphere's leaf reference and that leaf's record (0x7fffffff: none).  The
 * reference has no acceleration structure (integrator.rs:106-127 is the scan these arrays prune).  Either pointer may be
 * NULL.  IR_ESTATE when the scene is rendered with the linear scan. */
int ir_download_accel_climb(ir_context* ctx, void* climb_records, uint32_t n_records, int32_t* sphere_start, uint32_t n_spheres);

/* -------------------------------------------------------------------------------------------------
 * Camera.  `dyn Camera` (cameras/mod.rs:7-17) is host code; PinholeCamera is the fast path.
 * ------------------------------------------------------------------------------------------------- */
/* PinholeCamera cached state (pinhole_camera.rs:7-23) as computed by ir_pinhole_compute:
 * local_to_world column-major (glam Mat4), position, fov (radians), aspect_ratio. */
int ir_set_pinhole(ir_context* ctx, const float local_to_world[16], const float position[3], float fov,
                   float aspect_ratio);
/* Any other Camera implementor: the façade evaluates get_ray(x,y) (cameras/mod.rs:8) on the host for every
 * pixel and uploads them: rays = w*h*6 floats (origin.xyz, direction.xyz), pixel (x,y) at y*w+x. */
int ir_set_primary_rays(ir_context* ctx, const float* rays, uint32_t w, uint32_t h);
/* Host helper restating PinholeCamera::{compute_fov, compute_transformation_matrix,
 * compute_camera_directions, set_image_resolutions} (pinhole_camera.rs:59-69,110-129).
 * out_dirs9 = forward.xyz, up.xyz, right.xyz (may be NULL).  Needs no context and no GPU. */
int ir_pinhole_compute(const float position[3], const float rotation_xyz[3], float focal_length,
                       float sensor_size, uint32_t width, uint32_t height, float out_local_to_world[16],
                       float* out_fov, float* out_aspect_ratio, float out_dirs9[9]);
/* Host helper restating PinholeCamera::get_ray (pinhole_camera.rs:78-98) for one pixel: out6 = origin, dir. */
int ir_pinhole_get_ray(const float local_to_world[16], const float position[3], float fov, float aspect_ratio,
                       uint32_t width, uint32_t height, uint32_t x, uint32_t y, float out6[6]);

/* -------------------------------------------------------------------------------------------------
 * Integrator parameters (integrator.rs:17-20; reference defaults bounces=5, max_compulsory_bounces=2,
 * renderer.rs:38-41).  `seed` keys the counter-based RNG that replaces rand::ThreadRng (sampler.rs:4-6):
 * draw k of sample s of pixel p is a pure function of (seed, p, s, k).
 * ------------------------------------------------------------------------------------------------- */
int ir_set_integrator(ir_context* ctx, uint32_t bounces, uint32_t max_compulsory_bounces, uint64_t seed);

/* How one ir_render call of spp samples is accumulated (default 128).  The reference adds ONE sample per render()
 * call: a zeroed TileAccumulator takes the sample (tile_buffer.rs:24-28) and is merged into the frame,
 * framebuffer += tile (full_buffer.rs:107-125).  A call with spp > 1 is cut into consecutive blocks of
 * samples_per_block samples; each block is summed from zero in sample order (the tile accumulator) and the block
 * sums are merged into the accumulator in block order.  Blocks are independent until the merge, which is what lets
 * one call fill the GPU.  On a zeroed accumulator a call with spp <= samples_per_block is bit-identical to spp
 * successive reference-style 1-sample calls.
 * IR_SAMPLE_BLOCK_AUTO (0): 128, halved until the launch has at least 26000 x 256 (pixel, block) tasks, not below 8 (4 on
 * frames of fewer than 4096 x 256 pixels): a launch of few long blocks lasts as long as its longest block runs on one lane
 * (the reference's own headless case, 800x450 at 64 spp: 4-sample blocks render it in 0.37 of the time of one 64-sample
 * block per pixel; 1080p at 128 spp: 32-sample blocks, 0.8 of the time).  A block of 32 samples or more is then replaced by
 * the size within an eighth of it whose last block, spp mod size, comes closest to a quarter block (sizes dividing spp
 * skipped, the larger size on a tie; 1080p at 1024 spp: 8 blocks of 124 and one of 32): a launch ends on its last blocks
 * and drains for the length of a short one.  A pure
 * function of the pixels this rank owns — 4096 per 64x64 tile — and spp; ir_get_sample_block returns what a render of
 * that shape would use, so that a caller (or a checker) can reproduce the order of the sums. */
int ir_set_sample_block(ir_context* ctx, uint32_t samples_per_block);
int ir_get_sample_block(const ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t* out_samples_per_block);

/* How the integrator is scheduled on the GPU (results are identical bit for bit, only speed differs):
 *   IR_SCHEDULE_REGENERATION : one kernel; a lane owns a pixel and runs its samples in order, restarting at the
 *                              primary hit when a path ends; path state never leaves the registers;
 *   IR_SCHEDULE_WAVEFRONT    : one kernel per stage (primary, generate, one per bounce, commit); paths stream
 *                              through queues in device memory and are compacted between bounces with a wave64
 *                              ballot/prefix count; max_paths bounds the queue memory (60 B per path and queue,
 *                              16 B per path for the per-sample values; 0 keeps the default of 128 Mi paths = ~22 GB);
 *   IR_SCHEDULE_AUTO         : the faster one for the scene as measured on MI355X.
 * Renders that collect statistics (ir_enable_stats) always use REGENERATION. */
int ir_set_schedule(ir_context* ctx, int schedule, uint32_t max_paths);

/* Multi-GPU work split for this context (rank in [0, world)).
 *   IR_PARTITION_TILES  : 64x64 tiles (renderer.rs:105-119) dealt round-robin; this rank renders tile t iff
 *                         t % world == rank and leaves every other pixel's sum/count untouched.
 *   IR_PARTITION_SAMPLES: every rank renders all pixels; ir_render(spp) renders the sample indices
 *                         [base + rank*spp, base + (rank+1)*spp) where base = world * (samples already
 *                         accumulated per rank), i.e. spp is PER RANK.
 * Partial buffers are combined on a root rank by ir_combine (below), or by the caller (a reduce of
 * ir_device_radiance / ir_device_counts, or a gather of ir_pack_owned_tiles payloads). */
int ir_set_partition(ir_context* ctx, int mode, uint32_t rank, uint32_t world);

/* -------------------------------------------------------------------------------------------------
 * Exchange step.  Replaces the merge loop of RayTracer::render (renderer.rs:152-159), where the main thread receives
 * each worker's TileAccumulator over a channel and adds it into the frame (full_buffer.rs:107-125): here a worker is a
 * GPU, the channel is RCCL over xGMI, and the "main thread" is the root rank.  One process per GPU, one context each.
 *   ir_comm_unique_id : rank 0 creates the rendezvous id (ncclGetUniqueId) and hands the IR_COMM_ID_BYTES to every
 *                       rank by any means the host has (a pipe, a file, MPI, torch.distributed ...);
 *   ir_comm_init      : collective; every rank joins with its rank and the world size (ncclCommInitRank on the
 *                       context's device).  RCCL is loaded at run time (librccl.so.1); IR_EDEVICE if it is absent;
 *   ir_combine        : collective, blocking.  IR_PARTITION_SAMPLES: one grouped ncclReduce(sum) of the float4 sums and
 *                       the u32 counts.  IR_PARTITION_TILES: tiles are disjoint, so merging is a gather: every rank
 *                       sends its owned tiles as one packed payload (20 B per pixel, 1/world of the frame), the root
 *                       unpacks them; bit-identical to a one-GPU render.  The combined film lives in a buffer of its own
 *                       on the root; every rank's own accumulator is left as it was, so progressive rendering can go on.
 *                       On the root, ir_get_output_argb / ir_map_output_argb / ir_get_radiance show the combined film
 *                       until that context's next render;
 *   ir_combine_async  : the same, enqueued: the exchange reads a snapshot of the accumulator taken on the render stream
 *   ir_combine_wait     and runs on a stream of its own, so the next ir_render overlaps it; ir_combine_wait blocks until
 *                       the last exchange has finished (and reports an asynchronous RCCL error, if any);
 *   ir_last_combine_ms: device time of the last exchange (its own stream, start to end), waiting for peers included.
 * ------------------------------------------------------------------------------------------------- */
int ir_comm_unique_id(uint8_t out_id[IR_COMM_ID_BYTES]);
int ir_comm_init(ir_context* ctx, const uint8_t id[IR_COMM_ID_BYTES], uint32_t rank, uint32_t world);
int ir_comm_destroy(ir_context* ctx);
int ir_combine(ir_context* ctx, uint32_t root);
int ir_combine_async(ir_context* ctx, uint32_t root);
int ir_combine_wait(ir_context* ctx);
double ir_last_combine_ms(const ir_context* ctx);
/* The transport-independent halves of the tile gather, for hosts that move bytes themselves.  A payload holds the
 * 64x64 tiles rank r owns under IR_PARTITION_TILES (tile t iff t % world == r, in increasing t), each tile row-major,
 * pixels outside the frame zero: all float4 sums first, then all u32 counts; ir_tile_payload_bytes gives its size.
 * ir_pack_owned_tiles packs this context's accumulator (its own rank / world; one rank owning everything without a
 * tile partition) into device memory; ir_unpack_tiles writes rank r's payload into this context's accumulator. */
size_t ir_tile_payload_bytes(uint32_t width, uint32_t height, uint32_t rank, uint32_t world);
int ir_pack_owned_tiles(ir_context* ctx, void* d_payload, size_t bytes);
int ir_unpack_tiles(ir_context* ctx, const void* d_payload, size_t bytes, uint32_t rank, uint32_t world);

/* -------------------------------------------------------------------------------------------------
 * Render.  Replaces RayTracer::render(scene, width, height, acc) (renderer.rs:95-162):
 *   - resolution change reallocates and zeroes the accumulator (renderer.rs:80-93);
 *   - accumulate == 0 zeroes it first (renderer.rs:99-103) == render_updated; accumulate != 0 == prepare_pixels;
 *   - spp samples per pixel are added (the reference adds exactly 1 per call; spp = 1 reproduces it) in sample
 *     blocks, see ir_set_sample_block;
 *   - blocks until the kernels have finished (the reference render() is synchronous).
 * ------------------------------------------------------------------------------------------------- */
int ir_render(ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, int accumulate);
/* Same, but returns after enqueueing on the context's stream (for callers that own the stream). */
int ir_render_async(ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, int accumulate);
int ir_synchronize(ir_context* ctx);
/* ir_render_async into the CALLER's device buffers, allocating nothing.  No counterpart in the reference: it is for
 * callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc).  It is NOT a graph-capture entry
 * point: every launch records timing events (ir_last_kernel_ms, ir_get_paths_kernel_history), and a recording of the call
 * into a HIP graph has not been made to work (round 3 tried; a replay faulted) — enqueue it on a stream.
 * d_sum_rgba = width*height float4, d_counts = width*height u32, on the context's device; accumulate == 0 zeroes them
 * first and restarts the sample sequence at 0, otherwise the samples continue the context's sequence
 * (ir_samples_accumulated) as ir_render's would.  The context's own accumulator and size are not touched.  The block
 * launch scratch (primary-hit records, task queue counters, the block sums of a launch, ir_set_sample_block) is a static
 * 160 MiB device array instead of allocated buffers, which
 *   - limits the frame to 2 467 109 pixels (68 B per pixel: the records and two block sums must fit; larger frames:
 *     IR_ELIMIT; 1920x1080 fits, 3840x2160 does not), launches are cut so that their block sums fit (results do not
 *     depend on the cut);
 *   - is shared by every context on that GPU and keeps the last call's contents (the task counters are zeroed by every
 *     launch, every other entry a launch reads was written by that launch): calls that overlap on one GPU race —
 *     serialise them.
 * The scene's device tables must be current: after changing the scene, the sky or the closest-hit structure call
 * ir_render once (spp 0 is enough), else IR_ESTATE.  IR_SCHEDULE_WAVEFRONT and ir_enable_stats allocate: IR_ESTATE.
 * Returns after enqueueing on the context's stream; ir_synchronize waits. */
int ir_render_noalloc(ir_context* ctx, uint32_t width, uint32_t height, uint32_t spp, int accumulate, void* d_sum_rgba,
                      void* d_counts);

/* RayTracer::get_output (renderer.rs:164-169): resolve (mean, Reinhard, gamma 1/2.2, clamp, 0xAARRGGBB) on the
 * device and copy n = width*height u32 pixels to `out`.  Row 0 is the bottom row, as in the reference. */
int ir_get_output_argb(ir_context* ctx, uint32_t* out, size_t n);
/* The same frame as a borrowed slice, which is what RayTracer::get_output returns (`&[u32]`, renderer.rs:164-169):
 * *out points at width*height pixels in page-locked host memory owned by the context (the device copies straight into
 * it, no staging: 0.25 instead of 0.5 ms per 1080p frame), valid until the next ir_map_output_argb, resize or
 * ir_destroy.  *out = NULL and *out_n = 0 for an empty frame; out_n may be NULL. */
int ir_map_output_argb(ir_context* ctx, const uint32_t** out, size_t* out_n);
/* Raw accumulator (Accumulator.framebuffer / sample_counts, full_buffer.rs:7-12): n x 4 floats, n x u32. */
int ir_get_radiance(ir_context* ctx, float* sum_rgba, uint32_t* counts, size_t n);
/* Overwrite the accumulator (checkpoint restore / tests).  Either pointer may be NULL (left unchanged). */
int ir_set_radiance(ir_context* ctx, const float* sum_rgba, const uint32_t* counts, size_t n);
/* Closest-hit sphere index of each pixel's primary ray (-1 = miss) for the current camera, scene and the
 * size of the last render (or of ir_set_primary_rays); runs the same ray-generation and intersection code
 * as the integrator.  out_t may be NULL. */
int ir_get_primary_hits(ir_context* ctx, int32_t* out_index, float* out_t, size_t n);

/* RayTracer::get_current_size (renderer.rs:65-67) and get_last_render_time (:171-173). */
int ir_get_size(const ir_context* ctx, uint32_t* width, uint32_t* height);
double ir_last_render_seconds(const ir_context* ctx);
/* Device time of the integrator kernel launches of the last render (HIP events on the launch stream), ms,
 * and how many launches that was. */
double ir_last_kernel_ms(const ir_context* ctx, uint32_t* n_launches);
/* Device time of the path kernel alone (the persistent kernel that runs every sample block of every pixel whose primary
 * ray hits something; the dominant kernel) in the LAST launch of the last render, ms; 0 when that render launched none. */
double ir_last_paths_kernel_ms(const ir_context* ctx);
/* The same for the last launches of this context, oldest first: up to 64 are kept (a ring of event pairs), so a caller
 * that enqueues renders with ir_render_async and waits once can still average over all of them.  *out_n = how many were
 * written (<= max_n).  Waits for the last launch. */
int ir_get_paths_kernel_history(const ir_context* ctx, double* out_ms, uint32_t max_n, uint32_t* out_n);
/* Samples per pixel accumulated by this context since the last reset. */
uint32_t ir_samples_accumulated(const ir_context* ctx);
/* Checkpoint restore: after ir_set_radiance with a stored accumulator, tell the context how many samples per pixel
 * it holds so that the next accumulate != 0 render continues the counter-based sample sequence behind them. */
int ir_set_samples_accumulated(ir_context* ctx, uint32_t samples);

/* Device pointers of the accumulator (float4 sum per pixel, u32 count per pixel) for RCCL collectives issued
 * by the caller (torch.distributed).  Valid until the next resolution change / ir_bind_accumulator. */
void* ir_device_radiance(ir_context* ctx);
void* ir_device_counts(ir_context* ctx);
/* Render into caller-owned device memory instead (e.g. torch tensors): sum = n x float4, counts = n x u32
 * for n = width*height of the next renders.  NULL, NULL returns to context-owned buffers. */
int ir_bind_accumulator(ir_context* ctx, void* d_sum_rgba, void* d_counts, uint32_t width, uint32_t height);

/* Work statistics of the NEXT render only (costs a little): algorithmic counts per §8(d) of SURVEY.md.
 * After the render, ir_get_stats fills 5 x u64: samples, trace_ray calls, sphere tests, tests with
 * discriminant >= 0, hits.  Counts are "as the reference would execute", i.e. per sample including the
 * primary ray even where the kernel shares it between samples of a pixel. */
int ir_enable_stats(ir_context* ctx, int enable);
int ir_get_stats(ir_context* ctx, uint64_t out5[5]);
/* Executed acceleration-structure work of the same render: LBVH node visits and leaf sphere tests (both 0 for the
 * linear scan).  The primary trace a pixel's samples share is counted once per launch here. */
int ir_get_accel_stats(ir_context* ctx, uint64_t out2[2]);
/* Lane activity of the same render, as 4 pairs (times a wave executed a section, lanes active in it); active lanes /
 * (64 x wave executions) is the section's lane utilisation.  Linear-scan integrator: top of the path loop, Russian
 * roulette (bounce >= max_compulsory_bounces), hit geometry, sky lookup + sample commit.  LBVH integrator: rounds of the
 * wave loop (lanes with work left), shading block, inner-node steps, leaf steps. */
int ir_get_lane_profile(ir_context* ctx, uint64_t out8[8]);
/* What the launches of the same render executed of the per-pixel work: path pixels (the primary ray hit something),
 * discriminant >= 0 tests of the primary traces, pixels — each counted once per pixel and launch.  ir_get_stats charges the
 * primary ray to every sample, as the reference runs it (integrator.rs:40-51); the kernels evaluate it once per
 * pixel and launch because it has no random input (pinhole_camera.rs:81-84), and finish a sky pixel with four adds per
 * sample.  bench.py forms the executed flop count (roofline.frac_executed) from the two. */
int ir_get_primary_stats(ir_context* ctx, uint64_t out3[3]);

/* Self test of the arithmetic contract (no counterpart in the reference).  The kernels replace hipcc's long IEEE
 * f32 sqrt / divide expansions by short sequences that return the same correctly rounded bits; this runs, on the
 * context's GPU, the exhaustive (all 2^32 inputs) and randomised (2^log2_random_samples cases) comparisons against
 * the IEEE expansions and fills 10 mismatch counters, all of which must be 0:
 *   [0] sqrt  [1] 1/sqrt (two roundings)  [2] 1/x  [3] x/PI  [4] x/2PI  [5] min/max of atan2's one divide
 *   [6] a/b pairs  [7] (brdf*cos)/(cos/PI) weights  [8] contribution/p  [9] one-divide closest-hit scan vs literal scan */
int ir_selftest_exact_math(ir_context* ctx, uint64_t* out_mismatches, uint32_t n_out, uint32_t log2_random_samples);

/* -------------------------------------------------------------------------------------------------
 * EXR (file_formats/exr.rs:26-49: ExrImage::load_exr_image == exr::read_first_rgba_layer_from_file).
 * From-scratch reader: single/multi-part, scanline and tiled (level 0) files, HALF/FLOAT/UINT channels,
 * NONE/RLE/ZIPS/ZIP/PIZ/PXR24/B44/B44A compression (the lossy DWAA/DWAB are reported as unsupported: IR_EIO); the first layer
 * that has R, G and B; alpha dropped; pixel (x,y) of the data window at y*width+x, row 0 = top of the file.
 * ir_exr_load allocates *out_rgb (w*h*3 floats) with malloc; release with ir_free.  Needs no GPU.
 * ir_exr_last_error() gives the message of the last failed EXR call on this thread.
 * ------------------------------------------------------------------------------------------------- */
int ir_exr_load(const char* path, float** out_rgb, uint32_t* out_w, uint32_t* out_h);
const char* ir_exr_last_error(void);
/* Writer (fixtures, "save image" — a reference TODO, README.md:104): RGB scanline EXR; compression 0 = NONE,
 * 1 = RLE, 2 = ZIPS, 3 = ZIP, 4 = PIZ, 5 = PXR24 (rounds FLOAT samples to 24 bits), 6 = B44, 7 = B44A (lossy, HALF samples
 * only: 4x4 blocks in 14 bytes; FLOAT samples pass through unchanged); pixel_type 1 = HALF, 2 = FLOAT. */
int ir_exr_save(const char* path, const float* rgb, uint32_t w, uint32_t h, int compression, int pixel_type);
/* The resolved frame as an 8-bit RGB PNG (the other half of "save image"): argb as ir_get_output_argb /
 * ir_map_output_argb return it (0xAARRGGBB, row 0 = bottom row, renderer.rs:164-169), written top-down; alpha is
 * dropped (it is the constant 0xBA).  Errors through ir_exr_last_error.  Needs no GPU. */
int ir_png_save(const char* path, const uint32_t* argb, uint32_t w, uint32_t h);
void ir_free(void* p);
/* Host restatement of ExrImage::sample (exr.rs:13-24) so the façade keeps that public method. */
int ir_exr_sample(const float* rgb, uint32_t w, uint32_t h, const float dir[3], float out_rgb[3]);

#ifdef __cplusplus
}
#endif
#endif /* INSPLORAY_HIP_H */
