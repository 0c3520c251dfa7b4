#include "sky_distribution.h"

#include <cmath>

namespace pt {

namespace {

constexpr double kPiD = 3.14159265358979323846;

double texel_weight(const float* t, bool uniform) {
    if (uniform) return 1.0;
    const double lum = (0.2126 * (double)t[0] + 0.7152 * (double)t[1]) + 0.0722 * (double)t[2];
    return (lum > 0.0 && !std::isinf(lum)) ? lum : 0.0;
}

// guide[k] = largest i in [0, n) with cdf[i] <= (float)k / (float)n, for k = 0 .. n (cdf[0] == 0, non-decreasing)
void build_guide(const float* cdf, uint32_t n, uint32_t* guide) {
    uint32_t i = 0;
    for (uint32_t k = 0; k <= n; ++k) {
        const float x = (float)k / (float)n;
        while (i + 1 < n && cdf[i + 1] <= x) ++i;
        guide[k] = i;
    }
}

}  // namespace

void build_sky_distribution(const float* rgba, uint32_t w, uint32_t h, SkyDistribution& out) {
    const size_t stride = (size_t)w + 1;
    std::vector<double> running(stride * h, 0.0);  // per row: running[x + 1] = weight of columns 0..x
    std::vector<double> row_weight(h, 0.0);
    double total = 0.0;
    bool uniform = false;
    for (;;) {
        total = 0.0;
        for (uint32_t y = 0; y < h; ++y) {
            const double ring = std::sin(kPiD * ((double)y + 0.5) / (double)h);
            double* run = &running[stride * y];
            double sum = 0.0;
            for (uint32_t x = 0; x < w; ++x) {
                sum += texel_weight(rgba + 4 * ((size_t)y * w + x), uniform) * ring;
                run[x + 1] = sum;
            }
            row_weight[y] = sum;
            total += sum;
        }
        if (uniform || (total > 0.0 && !std::isinf(total))) break;
        uniform = true;
    }

    out.marginal.assign((size_t)h + 1, 0.0f);
    out.cond.assign(stride * h, 0.0f);
    out.pdf.assign((size_t)w * h, 0.0f);
    double rows_so_far = 0.0;
    for (uint32_t y = 0; y < h; ++y) {
        float* c = &out.cond[stride * y];
        const double* run = &running[stride * y];
        const bool empty = !(row_weight[y] > 0.0);
        for (uint32_t x = 1; x <= w; ++x) c[x] = empty ? (float)((double)x / (double)w) : (float)(run[x] / row_weight[y]);
        c[w] = 1.0f;
        rows_so_far += row_weight[y];
        out.marginal[y + 1] = (float)(rows_so_far / total);
    }
    out.marginal[h] = 1.0f;
    // what the f32 CDFs actually select, so that the density used in the weights is the density sampled
    const double per_area = (double)w * (double)h / (2.0 * kPiD * kPiD);
    for (uint32_t y = 0; y < h; ++y) {
        const double p_row = (double)out.marginal[y + 1] - (double)out.marginal[y];
        const float* c = &out.cond[stride * y];
        for (uint32_t x = 0; x < w; ++x) out.pdf[(size_t)y * w + x] = (float)((p_row * ((double)c[x + 1] - (double)c[x])) * per_area);
    }
    out.marginal_guide.assign((size_t)h + 1, 0u);
    build_guide(out.marginal.data(), h, out.marginal_guide.data());
    out.cond_guide.assign(stride * h, 0u);
    for (uint32_t y = 0; y < h; ++y) build_guide(&out.cond[stride * y], w, &out.cond_guide[stride * y]);
}

}  // namespace pt
