//! `insploray` façade: the public surface of the reference crate (core-engine/src/lib.rs:8-16: `file_formats`, `cameras`,
//! `renderer`, `scene`, `Vec2`, `Vec3`) with `RayTracer::render` / `get_output` running on an MI355X through
//! libinsploray_hip (include/insploray_hip.h, bound by the `insploray-hip-sys` crate).
//!
//! NOT COMPILED in this repository's build environment (no Rust toolchain there).  What is checked instead:
//! tests/test_rust_sources.py (every `sys::` call names a declared extern, every extern is in the tested ctypes table,
//! build.rs lists the Makefile's sources and flags) and the compiled twins of this file, include/insploray.hpp
//! (tests/cpp/facade_test.cpp) and path_tracer_amd/*.py.
//!
//! Errors: where the reference panics (`unwrap`, index out of range) this panics with the library's message.
pub use glam::{Vec2, Vec3};

use insploray_hip_sys as sys;
use std::ffi::CStr;

pub(crate) mod ray {
    use glam::Vec3;
    /// ray.rs:3-6
    #[derive(Debug, Clone, Copy, Default)]
    pub struct Ray {
        pub origin: Vec3,
        pub direction: Vec3,
    }
}
use ray::Ray;

/// `rc != IR_OK` → panic with ir_last_error(ctx); the reference's render path has no error returns either.
fn check(ctx: *mut sys::ir_context, rc: i32, what: &str) {
    if rc != sys::IR_OK {
        let msg = unsafe {
            let p = if ctx.is_null() { sys::ir_last_global_error() } else { sys::ir_last_error(ctx) };
            if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
        };
        panic!("{what}: error {rc}: {msg}");
    }
}

pub mod cameras {
    use super::Ray;
    use glam::Vec3;
    use std::sync::{Arc, RwLock};

    /// What the device needs to generate a pinhole camera's rays itself (ir_set_pinhole).
    #[derive(Debug, Clone, Copy)]
    pub struct PinholeParams {
        pub local_to_world: [f32; 16], // column-major, glam's Mat4::to_cols_array
        pub position: [f32; 3],
        pub fov: f32,
        pub aspect_ratio: f32,
    }

    /// cameras/mod.rs:7-15, plus one method with a default body: a camera that is a pinhole camera hands over its
    /// cached state and the device generates the rays; every other implementor is sampled on the host per pixel.
    pub trait Camera {
        fn get_ray(&self, x: u32, y: u32) -> Ray;
        fn set_position(&mut self, position: Vec3);
        fn set_rotation(&mut self, rotation: Vec3);
        fn set_image_resolutions(&mut self, image_resolution: [u32; 2]);

        fn compute_transformation_matrix(&mut self);
        fn on_update(&mut self);

        fn as_pinhole(&self) -> Option<PinholeParams> {
            None
        }
    }

    pub type SharedCamera = Arc<RwLock<dyn Camera + Send + Sync>>; // cameras/mod.rs:17

    pub mod pinhole_camera {
        pub use super::Camera;
        use super::PinholeParams;
        use crate::{sys, Ray};
        use glam::{Mat4, Vec3};

        /// pinhole_camera.rs:7-23.  The cached state is computed by ir_pinhole_compute, the same host code that the
        /// tests pin against the reference's two camera tests (tests/test_oracle_camera.py).
        #[derive(Debug, Default)]
        pub struct PinholeCamera {
            pub position: Vec3,
            pub rotation: Vec3, // Euler x, y, z in radians

            pub image_size: [u32; 2],
            pub focal_length: f32,
            pub sensor_size: f32,

            aspect_ratio: f32,
            local_to_world: Mat4,
            pub forward: Vec3,
            pub up: Vec3,
            pub right: Vec3,

            pub fov: f32,
        }

        impl PinholeCamera {
            /// pinhole_camera.rs:26-46
            pub fn new(position: Vec3, rotation: Vec3, focal_length: f32, sensor_size: f32, image_size: [u32; 2]) -> Self {
                let mut camera = Self { position, rotation, focal_length, sensor_size, ..Default::default() };
                camera.set_image_resolutions(image_size);
                camera.on_update();
                camera
            }

            /// pinhole_camera.rs:48-56: only the field of view follows; the matrix does not depend on it
            pub fn set_focal_length(&mut self, focal_length: f32) {
                self.focal_length = focal_length;
                self.recompute();
            }

            pub fn set_sensor_size(&mut self, sensor_size: f32) {
                self.sensor_size = sensor_size;
                self.recompute();
            }

            pub fn get_camera_to_world_matrix(&self) -> Mat4 {
                self.local_to_world
            }

            /// fov, local_to_world, aspect_ratio and the three directions in one call (pinhole_camera.rs:59-69,110-129)
            fn recompute(&mut self) {
                let (mut m, mut fov, mut aspect, mut dirs) = ([0.0f32; 16], 0.0f32, 0.0f32, [0.0f32; 9]);
                let rc = unsafe {
                    sys::ir_pinhole_compute(
                        self.position.to_array().as_ptr(),
                        self.rotation.to_array().as_ptr(),
                        self.focal_length,
                        self.sensor_size,
                        self.image_size[0],
                        self.image_size[1],
                        m.as_mut_ptr(),
                        &mut fov,
                        &mut aspect,
                        dirs.as_mut_ptr(),
                    )
                };
                assert_eq!(rc, sys::IR_OK, "ir_pinhole_compute");
                self.local_to_world = Mat4::from_cols_array(&m);
                self.fov = fov;
                self.aspect_ratio = aspect;
                self.forward = Vec3::from_slice(&dirs[0..3]);
                self.up = Vec3::from_slice(&dirs[3..6]);
                self.right = Vec3::from_slice(&dirs[6..9]);
            }
        }

        impl Camera for PinholeCamera {
            /// pinhole_camera.rs:78-98 (the host restatement; the renderer does not call it for this camera)
            fn get_ray(&self, x: u32, y: u32) -> Ray {
                let mut o = [0.0f32; 6];
                let rc = unsafe {
                    sys::ir_pinhole_get_ray(
                        self.local_to_world.to_cols_array().as_ptr(),
                        self.position.to_array().as_ptr(),
                        self.fov,
                        self.aspect_ratio,
                        self.image_size[0],
                        self.image_size[1],
                        x,
                        y,
                        o.as_mut_ptr(),
                    )
                };
                assert_eq!(rc, sys::IR_OK, "ir_pinhole_get_ray");
                Ray { origin: Vec3::from_slice(&o[0..3]), direction: Vec3::from_slice(&o[3..6]) }
            }

            fn set_position(&mut self, position: Vec3) {
                self.position = position;
                self.on_update();
            }

            fn set_rotation(&mut self, rotation: Vec3) {
                self.rotation = rotation;
                self.on_update();
            }

            /// pinhole_camera.rs:110-113: the aspect ratio only; nothing else is recomputed
            fn set_image_resolutions(&mut self, image_resolution: [u32; 2]) {
                self.image_size = image_resolution;
                self.aspect_ratio = image_resolution[0] as f32 / image_resolution[1] as f32;
            }

            fn compute_transformation_matrix(&mut self) {
                self.recompute();
            }

            fn on_update(&mut self) {
                self.recompute();
            }

            fn as_pinhole(&self) -> Option<PinholeParams> {
                Some(PinholeParams {
                    local_to_world: self.local_to_world.to_cols_array(),
                    position: self.position.to_array(),
                    fov: self.fov,
                    aspect_ratio: self.aspect_ratio,
                })
            }
        }
    }
    pub use pinhole_camera::PinholeCamera;
}

pub mod file_formats {
    use crate::sys;
    use glam::Vec3;
    use std::ffi::{CStr, CString};

    /// file_formats/exr.rs:6-10
    #[derive(Default)]
    pub struct ExrImage {
        pub pixels_buffer: Vec<Vec3>,
        pub width: usize,
        pub height: usize,
    }

    impl ExrImage {
        /// file_formats/exr.rs:13-24 on the host (the renderer samples the uploaded copy on the device).
        /// Panics on an empty image, as the reference's `% self.width` does.
        pub fn sample(&self, dir: Vec3) -> Vec3 {
            let mut out = [0.0f32; 3];
            let rc = unsafe {
                sys::ir_exr_sample(
                    self.pixels_buffer.as_ptr() as *const f32, // glam::Vec3 is three packed f32
                    self.width as u32,
                    self.height as u32,
                    dir.to_array().as_ptr(),
                    out.as_mut_ptr(),
                )
            };
            assert_eq!(rc, sys::IR_OK, "ExrImage::sample on an empty image");
            Vec3::from_array(out)
        }

        /// file_formats/exr.rs:26-49: the first RGBA layer, alpha dropped.  The reference returns the `exr` crate's
        /// error; this one carries the library's message.
        pub fn load_exr_image(path: &str) -> Result<ExrImage, String> {
            let c_path = CString::new(path).map_err(|e| e.to_string())?;
            let (mut p, mut w, mut h) = (std::ptr::null_mut::<f32>(), 0u32, 0u32);
            let rc = unsafe { sys::ir_exr_load(c_path.as_ptr(), &mut p, &mut w, &mut h) };
            if rc != sys::IR_OK {
                let msg = unsafe { CStr::from_ptr(sys::ir_exr_last_error()) };
                return Err(msg.to_string_lossy().into_owned());
            }
            let n = w as usize * h as usize;
            let rgb = unsafe { std::slice::from_raw_parts(p, 3 * n) };
            let pixels_buffer = rgb.chunks_exact(3).map(Vec3::from_slice).collect();
            unsafe { sys::ir_free(p as *mut std::ffi::c_void) };
            Ok(ExrImage { pixels_buffer, width: w as usize, height: h as usize })
        }
    }
}

pub mod scene {
    use crate::file_formats::ExrImage;
    use glam::Vec3;

    /// scene.rs:5-25 (the spelling is the reference's)
    pub struct Matrial {
        pub albedo: Vec3,
        pub roughness: f32,
        pub metalic: f32,

        pub emission_color: Vec3,
        pub emissive_power: f32,
    }

    impl Default for Matrial {
        fn default() -> Self {
            Self { albedo: Vec3::ONE, roughness: 0.5, metalic: 0.0, emission_color: Vec3::ZERO, emissive_power: 0.0 }
        }
    }

    /// scene.rs:27-31
    pub struct Sphere {
        pub position: Vec3,
        pub radius: f32,
        pub material_id: i32,
    }

    /// scene.rs:34-41
    #[derive(Default)]
    pub struct Scene {
        pub spheres: Vec<Sphere>,
        pub materials: Vec<Matrial>,
        pub default_sky_color: Vec3,

        pub skybox: Option<ExrImage>,
    }

    impl Scene {
        /// scene.rs:44-93: the default sky box if it loads, a magenta unit-diameter sphere on a blue ground sphere
        pub fn get_example_scene() -> Self {
            let skybox = match ExrImage::load_exr_image("./assets/env/default_skybox_1.exr") {
                Ok(img) => Some(img),
                Err(e) => {
                    if let Ok(dir) = std::env::current_dir() {
                        println!("Current working directory: {}", dir.display());
                    }
                    eprintln!("Failed loading EXR: {}", e);
                    None
                }
            };
            let mut scene = Self { default_sky_color: Vec3::new(0.6, 0.7, 0.9), skybox, ..Default::default() };
            // (albedo, centre, radius), material i belongs to sphere i
            let objects = [
                (Vec3::new(1.0, 0.0, 1.0), Vec3::new(0.0, 0.0, 0.0), 0.5),
                (Vec3::new(0.2, 0.3, 1.0), Vec3::new(0.0, -100.5, 0.0), 100.0),
            ];
            for (i, (albedo, position, radius)) in objects.into_iter().enumerate() {
                scene.materials.push(Matrial { albedo, ..Default::default() });
                scene.spheres.push(Sphere { position, radius, material_id: i as i32 });
            }
            scene
        }
    }
}

pub mod renderer {
    use crate::cameras::{PinholeCamera, SharedCamera};
    use crate::scene::Scene;
    use crate::{check, sys};
    use glam::Vec3;
    use std::sync::{Arc, RwLock};
    use std::time::Duration;

    /// What was last uploaded as the sky box: address, shape and a strided probe of the pixels, so that a scene whose
    /// sky did not change costs no 100-MB upload per frame and an in-place edit is still seen (the Python mirror,
    /// path_tracer_amd/renderer.py, uses the same token).
    #[derive(PartialEq, Clone, Copy)]
    struct SkyToken {
        data: usize,
        width: usize,
        height: usize,
        probe: u64,
    }

    /// renderer.rs:13-26.  The integrator, accumulator and thread pool of the reference live behind `ctx`.
    pub struct RayTracer {
        ctx: *mut sys::ir_context,
        last_render_time: Duration,
        sky: Option<SkyToken>,

        pub active_camera: SharedCamera,
        /// Integrator { bounces, max_compulsory_bounces } (integrator.rs:17-20; renderer.rs:38-41 sets 5 and 2)
        pub bounces: u32,
        pub max_compulsory_bounces: u32,
        /// The reference seeds each worker from the OS; a render here is a function of this seed.
        pub seed: u64,
        /// Samples per pixel per `render` call; 1 is the reference.
        pub samples_per_call: u32,
    }

    // The context is used by one thread at a time (every method takes &mut self or only reads plain fields).
    unsafe impl Send for RayTracer {}

    impl RayTracer {
        /// renderer.rs:29-59, on device 0
        pub fn new(width: u32, height: u32) -> Self {
            Self::new_on_device(width, height, 0)
        }

        /// One context per GPU; a process per GPU is the supported layout.
        pub fn new_on_device(width: u32, height: u32, device: i32) -> Self {
            let mut ctx = std::ptr::null_mut();
            let rc = unsafe { sys::ir_create(&mut ctx, device) };
            check(std::ptr::null_mut(), rc, "ir_create");
            let camera = PinholeCamera::new(Vec3::new(0.0, 0.0, 2.0), Vec3::ZERO, 35.0, 55.0, [width, height]);
            Self {
                ctx,
                last_render_time: Duration::from_secs(0),
                sky: None,
                active_camera: Arc::new(RwLock::new(camera)),
                bounces: 5,
                max_compulsory_bounces: 2,
                seed: 0x0123_4567_89AB_CDEF,
                samples_per_call: 1,
            }
        }

        /// renderer.rs:61-63
        pub fn set_active_camera(&mut self, camera: SharedCamera) {
            self.active_camera = camera;
        }

        /// renderer.rs:65-67: the size of the last render, [0, 0] before the first
        pub fn get_current_size(&self) -> [u32; 2] {
            let (mut w, mut h) = (0u32, 0u32);
            unsafe { sys::ir_get_size(self.ctx, &mut w, &mut h) };
            [w, h]
        }

        /// renderer.rs:70-72
        #[inline]
        pub fn prepare_pixels(&mut self, scene: &Arc<RwLock<Scene>>, width: u32, height: u32) {
            self.render(scene, width, height, true);
        }

        /// renderer.rs:75-77
        #[inline]
        pub fn render_updated(&mut self, scene: &Arc<RwLock<Scene>>, width: u32, height: u32) {
            self.render(scene, width, height, false);
        }

        /// renderer.rs:95-162.  `acc == false` starts the accumulator again, a new size does too (renderer.rs:80-103);
        /// the tile loop, the workers and the merge are one ir_render call.
        pub fn render(&mut self, scene: &Arc<RwLock<Scene>>, width: u32, height: u32, acc: bool) {
            {
                let mut cam = self.active_camera.write().unwrap(); // renderer.rs:90-92
                cam.set_image_resolutions([width, height]);
            }
            {
                let scene = scene.read().unwrap(); // renderer.rs:122
                self.upload_scene(&scene);
            }
            self.upload_camera(width, height);
            let ctx = self.ctx;
            check(ctx, unsafe { sys::ir_set_integrator(ctx, self.bounces, self.max_compulsory_bounces, self.seed) }, "ir_set_integrator");
            check(ctx, unsafe { sys::ir_render(ctx, width, height, self.samples_per_call, acc as i32) }, "ir_render");
            self.last_render_time = Duration::from_secs_f64(unsafe { sys::ir_last_render_seconds(ctx) });
        }

        /// renderer.rs:164-169: mean, Reinhard, gamma, 0xAARRGGBB, row 0 at the bottom.  The slice is page-locked
        /// memory owned by the context and valid until the next call; the borrow on `&mut self` says the same.
        pub fn get_output(&mut self) -> &[u32] {
            let (mut p, mut n) = (std::ptr::null::<u32>(), 0usize);
            check(self.ctx, unsafe { sys::ir_map_output_argb(self.ctx, &mut p, &mut n) }, "ir_map_output_argb");
            if p.is_null() || n == 0 {
                return &[];
            }
            unsafe { std::slice::from_raw_parts(p, n) }
        }

        /// renderer.rs:171-173
        pub fn get_last_render_time(&self) -> Duration {
            self.last_render_time
        }

        // ---- additions (one C-ABI call each; none changes what the methods above return)

        /// Samples accumulated per pixel so far (Accumulator.sample_counts, full_buffer.rs:7-12)
        pub fn samples_accumulated(&self) -> u32 {
            unsafe { sys::ir_samples_accumulated(self.ctx) }
        }

        /// IR_PARTITION_TILES / IR_PARTITION_SAMPLES: this process renders its share of the frame
        pub fn set_partition(&mut self, mode: i32, rank: u32, world: u32) {
            check(self.ctx, unsafe { sys::ir_set_partition(self.ctx, mode, rank, world) }, "ir_set_partition");
        }

        /// Joins the RCCL communicator of `world` processes; `id` is rank 0's `comm_unique_id()` sent over any channel.
        pub fn comm_init(&mut self, id: &[u8; sys::IR_COMM_ID_BYTES], rank: u32, world: u32) {
            check(self.ctx, unsafe { sys::ir_comm_init(self.ctx, id.as_ptr(), rank, world) }, "ir_comm_init");
        }

        pub fn comm_unique_id() -> [u8; sys::IR_COMM_ID_BYTES] {
            let mut id = [0u8; sys::IR_COMM_ID_BYTES];
            check(std::ptr::null_mut(), unsafe { sys::ir_comm_unique_id(id.as_mut_ptr()) }, "ir_comm_unique_id");
            id
        }

        /// The merge of renderer.rs:152-159 across processes: after it `get_output` on `root` shows the whole frame.
        pub fn combine(&mut self, root: u32) {
            check(self.ctx, unsafe { sys::ir_combine(self.ctx, root) }, "ir_combine");
        }

        pub fn raw(&mut self) -> *mut sys::ir_context {
            self.ctx
        }

        // ---- private

        /// Scene (AoS, scene.rs:27-41) → the flat arrays of ir_set_spheres / ir_set_materials
        fn upload_scene(&mut self, scene: &Scene) {
            let ctx = self.ctx;
            let mut xyzr = Vec::with_capacity(4 * scene.spheres.len());
            let mut material_id = Vec::with_capacity(scene.spheres.len());
            for s in &scene.spheres {
                xyzr.extend_from_slice(&[s.position.x, s.position.y, s.position.z, s.radius]);
                material_id.push(s.material_id);
            }
            let mut materials9 = Vec::with_capacity(9 * scene.materials.len());
            for m in &scene.materials {
                materials9.extend_from_slice(&[
                    m.albedo.x, m.albedo.y, m.albedo.z, m.roughness, m.metalic,
                    m.emission_color.x, m.emission_color.y, m.emission_color.z, m.emissive_power,
                ]);
            }
            check(ctx, unsafe { sys::ir_set_materials(ctx, materials9.as_ptr(), scene.materials.len() as u32) }, "ir_set_materials");
            check(ctx, unsafe { sys::ir_set_spheres(ctx, xyzr.as_ptr(), material_id.as_ptr(), scene.spheres.len() as u32) }, "ir_set_spheres");
            check(ctx, unsafe { sys::ir_set_sky_color(ctx, scene.default_sky_color.to_array().as_ptr()) }, "ir_set_sky_color");

            let token = scene.skybox.as_ref().map(|img| {
                let floats = unsafe { std::slice::from_raw_parts(img.pixels_buffer.as_ptr() as *const f32, 3 * img.pixels_buffer.len()) };
                let stride = (floats.len() / 4096).max(1);
                let probe = floats.iter().step_by(stride).fold(0xcbf2_9ce4_8422_2325u64, |h, f| (h ^ f.to_bits() as u64).wrapping_mul(0x0000_0100_0000_01b3));
                SkyToken { data: floats.as_ptr() as usize, width: img.width, height: img.height, probe }
            });
            if token != self.sky {
                match (&scene.skybox, token) {
                    (Some(img), Some(_)) => {
                        assert_eq!(img.pixels_buffer.len(), img.width * img.height, "ExrImage: pixels_buffer does not match width * height");
                        let rc = unsafe { sys::ir_set_skybox(ctx, img.pixels_buffer.as_ptr() as *const f32, img.width as u32, img.height as u32) };
                        check(ctx, rc, "ir_set_skybox");
                    }
                    _ => check(ctx, unsafe { sys::ir_set_skybox(ctx, std::ptr::null(), 0, 0) }, "ir_set_skybox"),
                }
                self.sky = token;
            }
        }

        /// as_pinhole() → ir_set_pinhole, the device generates the rays (pinhole_camera.rs:78-98); any other camera:
        /// get_ray for every pixel on the host (integrator.rs:33 asks the camera per sample) → ir_set_primary_rays
        fn upload_camera(&mut self, width: u32, height: u32) {
            let ctx = self.ctx;
            let cam = self.active_camera.read().unwrap();
            if let Some(p) = cam.as_pinhole() {
                let rc = unsafe { sys::ir_set_pinhole(ctx, p.local_to_world.as_ptr(), p.position.as_ptr(), p.fov, p.aspect_ratio) };
                check(ctx, rc, "ir_set_pinhole");
                return;
            }
            let n = width as usize * height as usize;
            if n == 0 {
                return;
            }
            let mut rays = Vec::with_capacity(6 * n);
            for y in 0..height {
                for x in 0..width {
                    let r = cam.get_ray(x, y);
                    rays.extend_from_slice(&[r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z]);
                }
            }
            check(ctx, unsafe { sys::ir_set_primary_rays(ctx, rays.as_ptr(), width, height) }, "ir_set_primary_rays");
        }
    }

    impl Default for RayTracer {
        /// renderer.rs:176-180
        fn default() -> Self {
            Self::new(0, 0)
        }
    }

    impl Drop for RayTracer {
        /// renderer.rs:182-191 drops the pool and its channel; here the context and everything it holds on the GPU
        fn drop(&mut self) {
            unsafe { sys::ir_destroy(self.ctx) }
        }
    }
}

#[cfg(test)]
mod tests {
    //! The reference's two camera tests (pinhole_camera.rs:138-239) against this crate's PinholeCamera; they need the
    //! shared library but no GPU.  tests/test_oracle_camera.py runs the same vectors through the same C functions.
    use super::cameras::{Camera, PinholeCamera};
    use glam::{Mat4, Vec3};

    #[test]
    fn fov_calculation() {
        let camera = PinholeCamera::new(Vec3::ZERO, Vec3::ZERO, 35.0, 55.0, [1920, 1080]);
        let expected = 2.0 * (55.0f32 / (2.0 * 35.0)).atan();
        assert!((camera.fov - expected).abs() < f32::EPSILON, "expected {expected}, got {}", camera.fov);
    }

    #[test]
    fn transformation_matrix_validation_for_camera() {
        let camera = PinholeCamera::new(Vec3::new(-2.4027, -2.5716, 3.5259), Vec3::new(0.1975, -0.7941, -1.9074), 55.0, 35.0, [1920, 1080]);
        #[rustfmt::skip]
        let blender_rows = [
            -0.2315,  0.9717, 0.0458, -2.4027,
            -0.6616, -0.1918, 0.7249, -2.5716,
             0.7132,  0.1375, 0.6873,  3.5259,
             0.0,     0.0,    0.0,     1.0,
        ];
        let expected = Mat4::from_cols_array(&blender_rows).transpose();
        assert!(expected.abs_diff_eq(camera.get_camera_to_world_matrix(), 1e-4));
        let ray = camera.get_ray(960, 540);
        assert!((ray.direction.length() - 1.0).abs() < 1e-6);
    }
}
