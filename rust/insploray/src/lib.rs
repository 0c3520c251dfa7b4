//! `insploray` façade: same module paths and signatures as the reference crate (core-engine/src/lib.rs:8-16), the
//! body of `RayTracer::render` replaced by calls into libinsploray_hip.  NOT COMPILED here (no Rust toolchain in
//! this repository's build environment); the tested twins are include/insploray.hpp and path_tracer_amd/*.py.
pub use glam::{Vec2, Vec3};

pub mod ray { use glam::Vec3; pub struct Ray { pub origin: Vec3, pub direction: Vec3 } }

pub mod cameras {
    use super::ray::Ray;
    use glam::Vec3;
    use std::sync::{Arc, RwLock};
    /// cameras/mod.rs:7-15 plus one backwards-compatible addition: implementors that are pinhole cameras expose their
    /// cached state so the device generates rays itself; everyone else is sampled on the host per pixel.
    pub trait Camera {
        fn get_ray(&self, x: u32, y: u32) -> Ray;
        fn set_position(&mut self, position: Vec3);
        fn set_rotation(&mut self, rotation: Vec3);
        fn set_image_resolutions(&mut self, image_resolution: [u32; 2]);
        fn compute_transformation_matrix(&mut self);
        fn on_update(&mut self);
        fn as_pinhole(&self) -> Option<PinholeParams> { None }
    }
    pub struct PinholeParams { pub local_to_world: [f32; 16], pub position: [f32; 3], pub fov: f32, pub aspect_ratio: f32 }
    pub type SharedCamera = Arc<RwLock<dyn Camera + Send + Sync>>;
    // PinholeCamera: fields and methods as pinhole_camera.rs:7-130, state computed by ir_pinhole_compute.
}

pub mod renderer {
    use super::cameras::SharedCamera;
    use insploray_hip_sys as sys;
    use std::{sync::{Arc, RwLock}, time::Duration};
    pub struct RayTracer { ctx: *mut sys::ir_context, frame_buffer: Vec<u32>, last_render_time: Duration, pub active_camera: SharedCamera }
    impl RayTracer {
        // new / set_active_camera / get_current_size / prepare_pixels / render_updated / get_last_render_time as renderer.rs:29-77,171-173
        pub fn render(&mut self, scene: &Arc<RwLock<super::scene::Scene>>, width: u32, height: u32, acc: bool) {
            let scene = scene.read().unwrap();                       // renderer.rs:122
            let mut cam = self.active_camera.write().unwrap();       // renderer.rs:90-92
            cam.set_image_resolutions([width, height]);
            // flatten scene (AoS -> xyzr / material_id / materials9), upload camera:
            //   as_pinhole() => ir_set_pinhole, else get_ray for every pixel => ir_set_primary_rays
            // then: unsafe { sys::ir_render(self.ctx, width, height, 1, acc as i32) }; panic on != IR_OK like the unwraps
            let _ = (&scene, acc);
        }
        pub fn get_output(&mut self) -> &[u32] {
            // let (mut p, mut n) = (std::ptr::null(), 0usize);
            // unsafe { sys::ir_map_output_argb(self.ctx, &mut p, &mut n); std::slice::from_raw_parts(p, n) }
            // (page-locked memory owned by the context, valid until the next call: the borrow on &mut self says the same)
            &self.frame_buffer
        }
    }
    impl Drop for RayTracer { fn drop(&mut self) { unsafe { sys::ir_destroy(self.ctx) } } }
}

pub mod scene {
    use glam::Vec3;
    pub struct Matrial { pub albedo: Vec3, pub roughness: f32, pub metalic: f32, pub emission_color: Vec3, pub emissive_power: f32 }
    pub struct Sphere { pub position: Vec3, pub radius: f32, pub material_id: i32 }
    #[derive(Default)]
    pub struct Scene { pub spheres: Vec<Sphere>, pub materials: Vec<Matrial>, pub default_sky_color: Vec3, pub skybox: Option<super::file_formats::ExrImage> }
}

pub mod file_formats {
    use glam::Vec3;
    #[derive(Default)]
    pub struct ExrImage { pub pixels_buffer: Vec<Vec3>, pub width: usize, pub height: usize }
    // sample => ir_exr_sample ; load_exr_image => ir_exr_load (Err(String) from ir_exr_last_error)
}
