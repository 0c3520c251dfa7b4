// Host test of include/insploray.hpp: camera known answer (pinhole_camera.rs:162-239), example scene, EXR round trip,
// and — if a GPU is present — one tiny render (argv[1] == "gpu").  Exit code 0 == pass.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "insploray.hpp"

using namespace insploray;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    cameras::PinholeCamera cam(Vec3{-2.4027f, -2.5716f, 3.5259f}, Vec3{0.1975f, -0.7941f, -1.9074f}, 55.0f, 35.0f, {1920, 1080});
    const float expect_rows[12] = {-0.2315f, 0.9717f, 0.0458f, -2.4027f, -0.6616f, -0.1918f, 0.7249f, -2.5716f,
                                   0.7132f, 0.1375f, 0.6873f, 3.5259f};
    auto m = cam.get_camera_to_world_matrix();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) CHECK(std::fabs(m[4 * c + r] - expect_rows[4 * r + c]) < 1e-4f);
    cameras::PinholeCamera c2(Vec3{}, Vec3{}, 35.0f, 55.0f, {1920, 1080});
    CHECK(std::fabs(c2.fov - 2.0f * std::atan(55.0f / 70.0f)) < 1.2e-7f);
    CHECK(c2.as_pinhole().has_value());
    Ray r = c2.get_ray(960, 540);
    CHECK(std::fabs(r.direction.z + 1.0f) < 1e-3f);

    scene::Scene s = scene::Scene::get_example_scene();  // no asset here: falls back to the constant sky
    CHECK(s.spheres.size() == 2 && s.materials.size() == 2 && !s.skybox.has_value());
    CHECK(s.spheres[1].radius == 100.0f && s.default_sky_color.z == 0.9f);

    // EXR writer -> reader -> ExrImage::sample
    float img[2 * 4 * 3];
    for (int i = 0; i < 24; ++i) img[i] = (float)i;
    CHECK(ir_exr_save("/tmp/_facade_test.exr", img, 4, 2, 3, 2) == IR_OK);
    auto e = file_formats::ExrImage::load_exr_image("/tmp/_facade_test.exr");
    CHECK(e.width == 4 && e.height == 2 && e.pixels_buffer[7].z == 23.0f);
    Vec3 up = e.sample(Vec3{0, 1, 0});
    CHECK(up.x >= 0.0f && up.x < 12.0f);  // row 0
    bool threw = false;
    try { file_formats::ExrImage::load_exr_image("/nonexistent.exr"); } catch (const Error&) { threw = true; }
    CHECK(threw);

    if (argc > 1 && !std::strcmp(argv[1], "gpu")) {
        renderer::RayTracer rt(0, 0);
        rt.set_active_camera(std::make_shared<cameras::PinholeCamera>(Vec3{0, 0, 2}, Vec3{}, 35.0f, 55.0f, std::array<uint32_t, 2>{64, 36}));
        rt.render_updated(s, 64, 36);
        for (int i = 0; i < 3; ++i) rt.prepare_pixels(s, 64, 36);
        const auto& px = rt.get_output();
        CHECK(px.size() == 64 * 36 && (px[0] >> 24) == 0xBA);
        CHECK(rt.get_current_size()[0] == 64);
        // extensions: 4 samples in one call on a fresh frame == the four reference-style calls above (bitwise)
        std::vector<float> four_calls, one_call;
        std::vector<uint32_t> counts;
        rt.get_radiance(four_calls, counts);
        CHECK(rt.samples_accumulated() == 4 && counts[100] == 4);
        rt.set_sample_block(128);
        rt.set_sky_filter(IR_SKY_NEAREST);
        rt.set_accel(IR_ACCEL_AUTO);
        rt.render(s, 64, 36, false, 4);
        rt.get_radiance(one_call, counts);
        CHECK(std::memcmp(four_calls.data(), one_call.data(), one_call.size() * sizeof(float)) == 0);
        auto hits = rt.get_primary_hits();
        CHECK(hits.size() == 64 * 36 && hits[18 * 64 + 32] == 0 && hits[0] == 1 && hits[35 * 64] == -1);
        std::printf("gpu render ok, %f s\n", rt.get_last_render_time().count());
    } else {
        threw = false;
        try { renderer::RayTracer rt(8, 8); } catch (const Error& err) { threw = std::strstr(err.what(), "no HIP device") != nullptr; }
        // with a GPU present construction succeeds; both outcomes are fine here
    }
    std::printf("facade ok\n");
    return 0;
}
