// host_math.cpp — host-only parts of the C ABI that need no GPU: the PinholeCamera cached state
// (reference core-engine/src/cameras/pinhole_camera.rs:59-69,78-129) and ExrImage::sample
// (file_formats/exr.rs:13-24).  The reference computes these on the CPU as well; they are here so that the
// façade crates (Rust / C++ / Python) share one implementation with glam's f32 operation order:
//   Mat4::from_rotation_{x,y,z}, Mat4::from_translation, Mat4 * Mat4 (column by column,
//   ((X*x + Y*y) + Z*z) + W*w), transform_vector3 (X*x, then Y*y + res, then Z*z + res).
// Compiled with -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/insploray_hip.h"

namespace {

struct Col4 {
    float v[4];
};

struct Mat4 {  // column-major like glam::Mat4
    Col4 c[4];
    static Mat4 identity() {
        Mat4 m;
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) m.c[j].v[i] = (i == j) ? 1.0f : 0.0f;
        return m;
    }
    Col4 mul_vec4(const Col4& r) const {
        Col4 o;
        for (int i = 0; i < 4; ++i) {
            float res = c[0].v[i] * r.v[0];
            res = res + c[1].v[i] * r.v[1];
            res = res + c[2].v[i] * r.v[2];
            res = res + c[3].v[i] * r.v[3];
            o.v[i] = res;
        }
        return o;
    }
    Mat4 operator*(const Mat4& r) const {
        Mat4 o;
        for (int j = 0; j < 4; ++j) o.c[j] = mul_vec4(r.c[j]);
        return o;
    }
    void transform_vector3(const float in[3], float out[3]) const {
        for (int i = 0; i < 3; ++i) {
            float res = c[0].v[i] * in[0];
            res = c[1].v[i] * in[1] + res;
            res = c[2].v[i] * in[2] + res;
            out[i] = res;
        }
    }
};

Mat4 rotation_x(float a) {
    const float s = std::sin(a), c = std::cos(a);
    Mat4 m = Mat4::identity();
    m.c[1] = Col4{{0.0f, c, s, 0.0f}};
    m.c[2] = Col4{{0.0f, -s, c, 0.0f}};
    return m;
}
Mat4 rotation_y(float a) {
    const float s = std::sin(a), c = std::cos(a);
    Mat4 m = Mat4::identity();
    m.c[0] = Col4{{c, 0.0f, -s, 0.0f}};
    m.c[2] = Col4{{s, 0.0f, c, 0.0f}};
    return m;
}
Mat4 rotation_z(float a) {
    const float s = std::sin(a), c = std::cos(a);
    Mat4 m = Mat4::identity();
    m.c[0] = Col4{{c, s, 0.0f, 0.0f}};
    m.c[1] = Col4{{-s, c, 0.0f, 0.0f}};
    return m;
}
Mat4 translation(const float t[3]) {
    Mat4 m = Mat4::identity();
    m.c[3] = Col4{{t[0], t[1], t[2], 1.0f}};
    return m;
}

}  // namespace

extern "C" {

int ir_pinhole_compute(const float position[3], const float rotation_xyz[3], float focal_length, float sensor_size,
                       uint32_t width, uint32_t height, float out_local_to_world[16], float* out_fov,
                       float* out_aspect_ratio, float out_dirs9[9]) {
    if (!position || !rotation_xyz || !out_local_to_world || !out_fov || !out_aspect_ratio) return IR_EINVAL;
    // compute_fov, pinhole_camera.rs:59-61
    *out_fov = 2.0f * std::atan(sensor_size / (2.0f * focal_length));
    // set_image_resolutions, :110-113
    *out_aspect_ratio = (float)width / (float)height;
    // compute_transformation_matrix, :115-123
    const Mat4 rot = rotation_z(rotation_xyz[2]) * rotation_y(rotation_xyz[1]) * rotation_x(rotation_xyz[0]);
    const Mat4 m = translation(position) * rot;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) out_local_to_world[4 * j + i] = m.c[j].v[i];
    if (out_dirs9) {
        // compute_camera_directions, :63-69
        const float minus_z[3] = {0.0f, 0.0f, -1.0f};
        m.transform_vector3(minus_z, out_dirs9);                 // forward
        for (int i = 0; i < 3; ++i) out_dirs9[3 + i] = m.c[1].v[i];  // up = y_axis.xyz
        for (int i = 0; i < 3; ++i) out_dirs9[6 + i] = m.c[0].v[i];  // right = x_axis.xyz
    }
    return IR_OK;
}

int ir_pinhole_get_ray(const float local_to_world[16], const float position[3], float fov, float aspect_ratio,
                       uint32_t width, uint32_t height, uint32_t x, uint32_t y, float out6[6]) {
    if (!local_to_world || !position || !out6) return IR_EINVAL;
    // pinhole_camera.rs:78-98
    float vx = ((float)x + 0.5f) / (float)width;
    float vy = ((float)y + 0.5f) / (float)height;
    const float t = std::tan(fov / 2.0f);
    vx = (vx * 2.0f - 1.0f) * t;
    vy = (vy * 2.0f - 1.0f) * t;
    vx *= aspect_ratio;
    Mat4 m;
    std::memcpy(&m, local_to_world, sizeof(m));
    const float rd[3] = {vx, vy, -1.0f};
    float d[3];
    m.transform_vector3(rd, d);
    const float inv = 1.0f / std::sqrt((d[0] * d[0]) + (d[1] * d[1]) + (d[2] * d[2]));
    for (int i = 0; i < 3; ++i) {
        out6[i] = position[i];
        out6[3 + i] = d[i] * inv;
    }
    return IR_OK;
}

int ir_exr_sample(const float* rgb, uint32_t w, uint32_t h, const float dir[3], float out_rgb[3]) {
    if (!rgb || !dir || !out_rgb || w == 0 || h == 0) return IR_EINVAL;
    // exr.rs:13-24
    const float pi = 3.14159265358979323846264338327950288f;
    const float inv = 1.0f / std::sqrt((dir[0] * dir[0]) + (dir[1] * dir[1]) + (dir[2] * dir[2]));
    const float dx = dir[0] * inv, dy = dir[1] * inv, dz = dir[2] * inv;
    float cy = dy;
    if (cy < -1.0f) cy = -1.0f;
    if (cy > 1.0f) cy = 1.0f;
    const float theta = std::acos(cy);
    const float phi = std::atan2(dz, dx);
    const float u = (phi + pi) / (2.0f * pi);
    const float v = theta / pi;
    auto to_index = [](float f, uint32_t n) -> size_t {
        f = std::floor(f);
        size_t i = (f > 0.0f) ? (f >= 1.8446744073709552e19f ? SIZE_MAX : (size_t)f) : 0;  // `as usize` saturates
        return i % n;
    };
    const size_t x = to_index(u * (float)w, w), y = to_index(v * (float)h, h);
    const float* p = rgb + 3 * (y * (size_t)w + x);
    out_rgb[0] = p[0];
    out_rgb[1] = p[1];
    out_rgb[2] = p[2];
    return IR_OK;
}

}  // extern "C"
