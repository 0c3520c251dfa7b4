//! Raw bindings: one declaration per function of include/insploray_hip.h.
//! NOT COMPILED in this repository's build environment (no Rust toolchain); kept in sync by hand with the header
//! and with path_tracer_amd/_capi.py (whose prototype table IS tested against the header).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

#[repr(C)]
pub struct ir_context { _private: [u8; 0] }

pub const IR_OK: c_int = 0;
pub const IR_PARTITION_NONE: c_int = 0;
pub const IR_PARTITION_TILES: c_int = 1;
pub const IR_PARTITION_SAMPLES: c_int = 2;
pub const IR_SAMPLE_BLOCK_AUTO: u32 = 0;
pub const IR_COMM_ID_BYTES: usize = 128;

extern "C" {
    pub fn ir_abi_version() -> u32;
    pub fn ir_hip_versions(out_compiled: *mut c_int, out_runtime: *mut c_int) -> c_int;
    pub fn ir_create(out_ctx: *mut *mut ir_context, device: c_int) -> c_int;
    pub fn ir_destroy(ctx: *mut ir_context);
    pub fn ir_last_error(ctx: *const ir_context) -> *const c_char;
    pub fn ir_last_global_error() -> *const c_char;
    pub fn ir_set_stream(ctx: *mut ir_context, hip_stream: *mut c_void) -> c_int;
    pub fn ir_set_spheres(ctx: *mut ir_context, spheres_xyzr: *const c_float, material_id: *const i32, n: u32) -> c_int;
    pub fn ir_set_materials(ctx: *mut ir_context, materials9: *const c_float, n: u32) -> c_int;
    pub fn ir_set_sky_color(ctx: *mut ir_context, rgb: *const c_float) -> c_int;
    pub fn ir_set_skybox(ctx: *mut ir_context, rgb: *const c_float, w: u32, h: u32) -> c_int;
    pub fn ir_set_sky_filter(ctx: *mut ir_context, filter: c_int) -> c_int;
    pub fn ir_map_output_argb(ctx: *mut ir_context, out: *mut *const u32, out_n: *mut usize) -> c_int;
    pub fn ir_set_sky_sampling(ctx: *mut ir_context, mode: c_int) -> c_int;
    pub fn ir_set_accel(ctx: *mut ir_context, accel: c_int) -> c_int;
    pub fn ir_get_accel_info(ctx: *mut ir_context, out_accel: *mut c_int, out_nodes: *mut u32, out_depth: *mut u32) -> c_int;
    pub fn ir_set_accel_build(ctx: *mut ir_context, build: c_int) -> c_int;
    pub fn ir_get_accel_build_info(ctx: *mut ir_context, out_builder: *mut c_int, out_seconds: *mut f64) -> c_int;
    pub fn ir_download_accel(ctx: *mut ir_context, wide_nodes: *mut c_void, n_nodes: u32, leaf_spheres: *mut f32, leaf_index: *mut i32, n_spheres: u32) -> c_int;
    pub fn ir_download_accel_climb(ctx: *mut ir_context, climb_records: *mut c_void, n_records: u32, sphere_start: *mut i32, n_spheres: u32) -> c_int;
    pub fn ir_set_pinhole(ctx: *mut ir_context, local_to_world: *const c_float, position: *const c_float, fov: c_float, aspect_ratio: c_float) -> c_int;
    pub fn ir_set_primary_rays(ctx: *mut ir_context, rays: *const c_float, w: u32, h: u32) -> c_int;
    pub fn ir_pinhole_compute(position: *const c_float, rotation_xyz: *const c_float, focal_length: c_float, sensor_size: c_float,
                              width: u32, height: u32, out_local_to_world: *mut c_float, out_fov: *mut c_float,
                              out_aspect_ratio: *mut c_float, out_dirs9: *mut c_float) -> c_int;
    pub fn ir_pinhole_get_ray(local_to_world: *const c_float, position: *const c_float, fov: c_float, aspect_ratio: c_float,
                              width: u32, height: u32, x: u32, y: u32, out6: *mut c_float) -> c_int;
    pub fn ir_set_integrator(ctx: *mut ir_context, bounces: u32, max_compulsory_bounces: u32, seed: u64) -> c_int;
    pub fn ir_set_schedule(ctx: *mut ir_context, schedule: c_int, max_paths: u32) -> c_int;
    pub fn ir_set_sample_block(ctx: *mut ir_context, samples_per_block: u32) -> c_int;
    pub fn ir_get_sample_block(ctx: *mut ir_context, width: u32, height: u32, spp: u32, out_block: *mut u32) -> c_int;
    pub fn ir_set_partition(ctx: *mut ir_context, mode: c_int, rank: u32, world: u32) -> c_int;
    // exchange step (RCCL inside the library; renderer.rs:152-159 is the single-process merge it scales out)
    pub fn ir_comm_unique_id(out_id: *mut u8) -> c_int;
    pub fn ir_comm_init(ctx: *mut ir_context, unique_id: *const u8, rank: u32, world: u32) -> c_int;
    pub fn ir_comm_destroy(ctx: *mut ir_context) -> c_int;
    pub fn ir_combine(ctx: *mut ir_context, root: u32) -> c_int;
    pub fn ir_combine_async(ctx: *mut ir_context, root: u32) -> c_int;
    pub fn ir_combine_wait(ctx: *mut ir_context) -> c_int;
    pub fn ir_last_combine_ms(ctx: *const ir_context) -> c_double;
    pub fn ir_tile_payload_bytes(width: u32, height: u32, rank: u32, world: u32) -> usize;
    pub fn ir_pack_owned_tiles(ctx: *mut ir_context, d_out: *mut c_void, bytes: usize) -> c_int;
    pub fn ir_unpack_tiles(ctx: *mut ir_context, d_in: *const c_void, bytes: usize, rank: u32, world: u32) -> c_int;
    pub fn ir_render(ctx: *mut ir_context, width: u32, height: u32, spp: u32, accumulate: c_int) -> c_int;
    pub fn ir_render_async(ctx: *mut ir_context, width: u32, height: u32, spp: u32, accumulate: c_int) -> c_int;
    pub fn ir_synchronize(ctx: *mut ir_context) -> c_int;
    pub fn ir_render_noalloc(ctx: *mut ir_context, width: u32, height: u32, spp: u32, accumulate: c_int,
                             d_sum_rgba: *mut c_void, d_counts: *mut c_void) -> c_int;
    pub fn ir_get_output_argb(ctx: *mut ir_context, out: *mut u32, n: usize) -> c_int;
    pub fn ir_get_radiance(ctx: *mut ir_context, sum_rgba: *mut c_float, counts: *mut u32, n: usize) -> c_int;
    pub fn ir_set_radiance(ctx: *mut ir_context, sum_rgba: *const c_float, counts: *const u32, n: usize) -> c_int;
    pub fn ir_get_primary_hits(ctx: *mut ir_context, out_index: *mut i32, out_t: *mut c_float, n: usize) -> c_int;
    pub fn ir_get_size(ctx: *const ir_context, width: *mut u32, height: *mut u32) -> c_int;
    pub fn ir_last_render_seconds(ctx: *const ir_context) -> c_double;
    pub fn ir_last_kernel_ms(ctx: *const ir_context, n_launches: *mut u32) -> c_double;
    pub fn ir_last_paths_kernel_ms(ctx: *const ir_context) -> c_double;
    pub fn ir_get_paths_kernel_history(ctx: *const ir_context, out_ms: *mut c_double, max_n: u32, out_n: *mut u32) -> c_int;
    pub fn ir_samples_accumulated(ctx: *const ir_context) -> u32;
    pub fn ir_set_samples_accumulated(ctx: *mut ir_context, samples: u32) -> c_int;
    pub fn ir_device_radiance(ctx: *mut ir_context) -> *mut c_void;
    pub fn ir_device_counts(ctx: *mut ir_context) -> *mut c_void;
    pub fn ir_bind_accumulator(ctx: *mut ir_context, d_sum_rgba: *mut c_void, d_counts: *mut c_void, width: u32, height: u32) -> c_int;
    pub fn ir_enable_stats(ctx: *mut ir_context, enable: c_int) -> c_int;
    pub fn ir_get_stats(ctx: *mut ir_context, out5: *mut u64) -> c_int;
    pub fn ir_get_accel_stats(ctx: *mut ir_context, out2: *mut u64) -> c_int;
    pub fn ir_get_lane_profile(ctx: *mut ir_context, out8: *mut u64) -> c_int;
    pub fn ir_get_primary_stats(ctx: *mut ir_context, out3: *mut u64) -> c_int;
    pub fn ir_selftest_exact_math(ctx: *mut ir_context, out_mismatches: *mut u64, n_out: u32, log2_random_samples: u32) -> c_int;
    pub fn ir_exr_load(path: *const c_char, out_rgb: *mut *mut c_float, out_w: *mut u32, out_h: *mut u32) -> c_int;
    pub fn ir_exr_last_error() -> *const c_char;
    pub fn ir_exr_save(path: *const c_char, rgb: *const c_float, w: u32, h: u32, compression: c_int, pixel_type: c_int) -> c_int;
    pub fn ir_png_save(path: *const c_char, argb: *const u32, w: u32, h: u32) -> c_int;
    pub fn ir_free(p: *mut c_void);
    pub fn ir_exr_sample(rgb: *const c_float, w: u32, h: u32, dir: *const c_float, out_rgb: *mut c_float) -> c_int;
}
