// insploray.hpp — header-only C++ façade over the C ABI (insploray_hip.h) with the names and meaning of the reference
// crate's public surface (core-engine/src/lib.rs:8-16).  The reference is Rust; no Rust toolchain exists in this
// repository's build environment, so this is the compiled-language host layer (the Rust sources are in rust/).
//
//   insploray::renderer::RayTracer      renderer.rs:13-191
//   insploray::scene::{Scene,Sphere,Matrial}   scene.rs:5-93
//   insploray::cameras::{Camera,PinholeCamera,SharedCamera}   cameras/mod.rs:7-17, pinhole_camera.rs:7-130
//   insploray::file_formats::ExrImage   file_formats/exr.rs:6-49
//
// Where the reference panics (`unwrap`) or returns Err this throws insploray::Error.
#pragma once
#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "insploray_hip.h"

namespace insploray {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct Vec3 {
    float x = 0, y = 0, z = 0;
};

struct Ray {  // ray.rs:3-6
    Vec3 origin, direction;
};

namespace file_formats {
struct ExrImage {  // exr.rs:6-10
    std::vector<Vec3> pixels_buffer;
    size_t width = 0, height = 0;

    Vec3 sample(Vec3 dir) const {  // exr.rs:13-24
        Vec3 out;
        int rc = ir_exr_sample(&pixels_buffer.data()->x, (uint32_t)width, (uint32_t)height, &dir.x, &out.x);
        if (rc != IR_OK) throw Error(rc, "ExrImage::sample on an empty image");
        return out;
    }
    static ExrImage load_exr_image(const std::string& path) {  // exr.rs:26-49
        float* p = nullptr;
        uint32_t w = 0, h = 0;
        int rc = ir_exr_load(path.c_str(), &p, &w, &h);
        if (rc != IR_OK) throw Error(rc, ir_exr_last_error());
        ExrImage img;
        img.width = w;
        img.height = h;
        img.pixels_buffer.resize((size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; ++i) img.pixels_buffer[i] = Vec3{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
        ir_free(p);
        return img;
    }
};
}  // namespace file_formats

namespace scene {
struct Matrial {  // scene.rs:5-25 (sic)
    Vec3 albedo{1, 1, 1};
    float roughness = 0.5f, metalic = 0.0f;
    Vec3 emission_color{0, 0, 0};
    float emissive_power = 0.0f;
};
struct Sphere {  // scene.rs:27-31
    Vec3 position;
    float radius = 1.0f;
    int32_t material_id = -1;
};
struct Scene {  // scene.rs:34-41
    std::vector<Sphere> spheres;
    std::vector<Matrial> materials;
    Vec3 default_sky_color;
    std::optional<file_formats::ExrImage> skybox;
    uint64_t skybox_version = 0;  // bump after editing skybox in place so that the renderer re-uploads it

    static Scene get_example_scene() {  // scene.rs:44-93
        Scene s;
        try {
            s.skybox = file_formats::ExrImage::load_exr_image("./assets/env/default_skybox_1.exr");
        } catch (const Error& e) {
            std::fprintf(stderr, "Failed loading EXR: %s\n", e.what());
        }
        s.default_sky_color = Vec3{0.6f, 0.7f, 0.9f};
        Matrial m0;
        m0.albedo = Vec3{1.0f, 0.0f, 1.0f};
        s.materials.push_back(m0);
        s.spheres.push_back(Sphere{Vec3{0, 0, 0}, 0.5f, 0});
        Matrial m1;
        m1.albedo = Vec3{0.2f, 0.3f, 1.0f};
        s.materials.push_back(m1);
        s.spheres.push_back(Sphere{Vec3{0, -100.5f, 0}, 100.0f, 1});
        return s;
    }
};
}  // namespace scene

namespace cameras {
struct PinholeParams {
    std::array<float, 16> local_to_world;
    Vec3 position;
    float fov, aspect_ratio;
};
struct Camera {  // cameras/mod.rs:7-15
    virtual ~Camera() = default;
    virtual Ray get_ray(uint32_t x, uint32_t y) const = 0;
    virtual void set_position(Vec3 position) = 0;
    virtual void set_rotation(Vec3 rotation) = 0;
    virtual void set_image_resolutions(std::array<uint32_t, 2> image_resolution) = 0;
    virtual void compute_transformation_matrix() = 0;
    virtual void on_update() = 0;
    // addition: pinhole-like cameras hand their cached state to the device; others are sampled on the host
    virtual std::optional<PinholeParams> as_pinhole() const { return std::nullopt; }
};
using SharedCamera = std::shared_ptr<Camera>;  // Arc<RwLock<dyn Camera + Send + Sync>>, cameras/mod.rs:17

struct PinholeCamera : Camera {  // pinhole_camera.rs:7-23
    Vec3 position, rotation;
    std::array<uint32_t, 2> image_size{0, 0};
    float focal_length = 0, sensor_size = 0;
    Vec3 forward, up, right;
    float fov = 0;

    PinholeCamera(Vec3 pos, Vec3 rot, float focal, float sensor, std::array<uint32_t, 2> size)
        : position(pos), rotation(rot), focal_length(focal), sensor_size(sensor) {
        set_image_resolutions(size);
        on_update();
    }
    void set_focal_length(float f) { focal_length = f; on_update(); }
    void set_sensor_size(float s) { sensor_size = s; on_update(); }
    std::array<float, 16> get_camera_to_world_matrix() const { return local_to_world_; }

    Ray get_ray(uint32_t x, uint32_t y) const override {
        float o[6];
        ir_pinhole_get_ray(local_to_world_.data(), &position.x, fov, aspect_ratio_, image_size[0], image_size[1], x, y, o);
        return Ray{Vec3{o[0], o[1], o[2]}, Vec3{o[3], o[4], o[5]}};
    }
    void set_position(Vec3 p) override { position = p; on_update(); }
    void set_rotation(Vec3 r) override { rotation = r; on_update(); }
    void set_image_resolutions(std::array<uint32_t, 2> r) override {
        image_size = r;
        aspect_ratio_ = (float)r[0] / (float)r[1];
    }
    void compute_transformation_matrix() override { on_update(); }
    void on_update() override {
        float dirs[9];
        ir_pinhole_compute(&position.x, &rotation.x, focal_length, sensor_size, image_size[0], image_size[1],
                           local_to_world_.data(), &fov, &aspect_ratio_, dirs);
        forward = Vec3{dirs[0], dirs[1], dirs[2]};
        up = Vec3{dirs[3], dirs[4], dirs[5]};
        right = Vec3{dirs[6], dirs[7], dirs[8]};
    }
    std::optional<PinholeParams> as_pinhole() const override {
        return PinholeParams{local_to_world_, position, fov, aspect_ratio_};
    }

private:
    float aspect_ratio_ = 0;
    std::array<float, 16> local_to_world_{};
};
}  // namespace cameras

namespace renderer {
class RayTracer {  // renderer.rs:13-26
public:
    cameras::SharedCamera active_camera;

    RayTracer(uint32_t width, uint32_t height, int device = 0) {  // renderer.rs:29-59
        int rc = ir_create(&ctx_, device);
        if (rc != IR_OK) throw Error(rc, ir_last_global_error());
        active_camera = std::make_shared<cameras::PinholeCamera>(Vec3{0, 0, 2}, Vec3{0, 0, 0}, 35.0f, 55.0f,
                                                                 std::array<uint32_t, 2>{width, height});
    }
    ~RayTracer() { ir_destroy(ctx_); }  // renderer.rs:182-191
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    void set_active_camera(cameras::SharedCamera camera) { active_camera = std::move(camera); }  // :61-63
    std::array<uint32_t, 2> get_current_size() const {                                              // :65-67
        uint32_t w = 0, h = 0;
        ir_get_size(ctx_, &w, &h);
        return {w, h};
    }
    void prepare_pixels(const scene::Scene& s, uint32_t w, uint32_t h) { render(s, w, h, true); }   // :70-72
    void render_updated(const scene::Scene& s, uint32_t w, uint32_t h) { render(s, w, h, false); }  // :75-77

    // renderer.rs:95-162; spp is an extension (1 == the reference)
    void render(const scene::Scene& s, uint32_t width, uint32_t height, bool acc, uint32_t spp = 1) {
        active_camera->set_image_resolutions({width, height});
        upload_scene(s);
        upload_camera(width, height);
        check(ir_set_integrator(ctx_, bounces, max_compulsory_bounces, seed));
        check(ir_render(ctx_, width, height, spp, acc ? 1 : 0));
    }
    const std::vector<uint32_t>& get_output() {  // renderer.rs:164-169
        auto sz = get_current_size();
        size_t n = (size_t)sz[0] * sz[1];
        if (frame_buffer_.size() != n) frame_buffer_.assign(n, 0xFF000000u);
        if (n) check(ir_get_output_argb(ctx_, frame_buffer_.data(), n));
        return frame_buffer_;
    }
    std::chrono::duration<double> get_last_render_time() const {  // renderer.rs:171-173
        return std::chrono::duration<double>(ir_last_render_seconds(ctx_));
    }

    // Integrator{bounces, max_compulsory_bounces} (integrator.rs:17-20); private constants in the reference (renderer.rs:38-41)
    uint32_t bounces = 5, max_compulsory_bounces = 2;
    uint64_t seed = 0x0123456789ABCDEFull;
    ir_context* raw() { return ctx_; }

    // ---- extensions the reference does not have (each one call into the C ABI)
    void set_sky_filter(int filter) { check(ir_set_sky_filter(ctx_, filter)); }              // IR_SKY_NEAREST == exr.rs:20-23
    void set_accel(int accel) { check(ir_set_accel(ctx_, accel)); }                          // IR_ACCEL_AUTO / LINEAR / LBVH
    void set_accel_build(int build) { check(ir_set_accel_build(ctx_, build)); }              // IR_BUILD_AUTO / HOST / DEVICE / REFIT
    void set_sky_sampling(int mode) { check(ir_set_sky_sampling(ctx_, mode)); }              // IR_SKY_SAMPLING_BSDF == integrator.rs:64-70,84-89
    void set_schedule(int schedule, uint32_t max_paths = 0) { check(ir_set_schedule(ctx_, schedule, max_paths)); }
    // get_output without the copy: the borrowed slice of renderer.rs:164-169, valid until the next call / resize
    std::pair<const uint32_t*, size_t> map_output() {
        const uint32_t* p = nullptr;
        size_t n = 0;
        check(ir_map_output_argb(ctx_, &p, &n));
        return {p, n};
    }
    void set_sample_block(uint32_t samples) { check(ir_set_sample_block(ctx_, samples)); }
    void set_partition(int mode, uint32_t rank, uint32_t world) { check(ir_set_partition(ctx_, mode, rank, world)); }
    uint32_t samples_accumulated() const { return ir_samples_accumulated(ctx_); }
    // Accumulator.framebuffer / sample_counts (full_buffer.rs:7-12): n x 4 floats, n x u32
    void get_radiance(std::vector<float>& sum_rgba, std::vector<uint32_t>& counts) {
        auto sz = get_current_size();
        size_t n = (size_t)sz[0] * sz[1];
        sum_rgba.assign(n * 4, 0.0f);
        counts.assign(n, 0u);
        if (n) check(ir_get_radiance(ctx_, sum_rgba.data(), counts.data(), n));
    }
    std::vector<int32_t> get_primary_hits() {
        auto sz = get_current_size();
        std::vector<int32_t> idx((size_t)sz[0] * sz[1]);
        if (!idx.empty()) check(ir_get_primary_hits(ctx_, idx.data(), nullptr, idx.size()));
        return idx;
    }

private:
    void check(int rc) const {
        if (rc != IR_OK) throw Error(rc, ir_last_error(ctx_));
    }
    void upload_scene(const scene::Scene& s) {
        std::vector<float> xyzr(s.spheres.size() * 4), mats(s.materials.size() * 9);
        std::vector<int32_t> mid(s.spheres.size());
        for (size_t i = 0; i < s.spheres.size(); ++i) {
            const auto& sp = s.spheres[i];
            xyzr[4 * i] = sp.position.x; xyzr[4 * i + 1] = sp.position.y; xyzr[4 * i + 2] = sp.position.z; xyzr[4 * i + 3] = sp.radius;
            mid[i] = sp.material_id;
        }
        for (size_t i = 0; i < s.materials.size(); ++i) {
            const auto& m = s.materials[i];
            float* o = &mats[9 * i];
            o[0] = m.albedo.x; o[1] = m.albedo.y; o[2] = m.albedo.z; o[3] = m.roughness; o[4] = m.metalic;
            o[5] = m.emission_color.x; o[6] = m.emission_color.y; o[7] = m.emission_color.z; o[8] = m.emissive_power;
        }
        check(ir_set_materials(ctx_, mats.data(), (uint32_t)s.materials.size()));
        check(ir_set_spheres(ctx_, xyzr.data(), mid.data(), (uint32_t)s.spheres.size()));
        check(ir_set_sky_color(ctx_, &s.default_sky_color.x));
        if (!s.skybox) {
            if (have_skybox_) check(ir_set_skybox(ctx_, nullptr, 0, 0));
            have_skybox_ = false;
        } else if (!have_skybox_ || sky_data_ != s.skybox->pixels_buffer.data() || sky_version_ != s.skybox_version) {
            check(ir_set_skybox(ctx_, &s.skybox->pixels_buffer.data()->x, (uint32_t)s.skybox->width, (uint32_t)s.skybox->height));
            have_skybox_ = true;
            sky_data_ = s.skybox->pixels_buffer.data();
            sky_version_ = s.skybox_version;
        }
    }
    void upload_camera(uint32_t w, uint32_t h) {
        if (auto p = active_camera->as_pinhole()) {
            check(ir_set_pinhole(ctx_, p->local_to_world.data(), &p->position.x, p->fov, p->aspect_ratio));
            return;
        }
        if ((size_t)w * h == 0) return;
        std::vector<float> rays((size_t)w * h * 6);
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                Ray r = active_camera->get_ray(x, y);
                float* o = &rays[6 * ((size_t)y * w + x)];
                o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
                o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
            }
        check(ir_set_primary_rays(ctx_, rays.data(), w, h));
    }

    ir_context* ctx_ = nullptr;
    std::vector<uint32_t> frame_buffer_;
    bool have_skybox_ = false;
    const Vec3* sky_data_ = nullptr;
    uint64_t sky_version_ = 0;
};
}  // namespace renderer

}  // namespace insploray
