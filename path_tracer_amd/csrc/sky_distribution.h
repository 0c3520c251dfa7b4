// sky_distribution.h — tables for importance sampling the equirectangular sky (IR_SKY_SAMPLING_MIS).
//
// Extension: the reference samples the BSDF only (integrator.rs:64-70) and meets the sky by chance (:84-89); with a
// sun-dominated sky (workload C5) almost all of the variance is in that chance.  See pt_device.h "Sky importance
// sampling" for the estimator and DESIGN.md §4 for why it keeps the reference's expectation.
#pragma once
#include <stdint.h>

#include <vector>

namespace pt {

struct SkyDistribution {
    std::vector<float> marginal;  // h + 1: CDF over rows, [0] = 0, [h] = 1
    std::vector<float> cond;      // h x (w + 1): CDF over the columns of each row
    std::vector<float> pdf;       // h x w: probability of the texel x w h / (2 pi^2); / sin(theta) = density per solid angle
    // Search guides (pt_device.h cdf_find_guided): entry k of an n-entry CDF's guide = largest i in [0, n) with
    // cdf[i] <= (float)k / (float)n, k = 0 .. n.  They narrow the binary search to a few entries; the index found is
    // the one the plain search finds.
    std::vector<uint32_t> marginal_guide;  // h + 1
    std::vector<uint32_t> cond_guide;      // h x (w + 1)
};

// rgba: w x h texels padded to 4 floats (the device layout of ir_set_skybox), row 0 = zenith.
// Texel weight = luminance x sin(polar angle of the row centre), accumulated in double; texels whose luminance is not
// positive and finite get weight 0; an image without any weight is sampled uniformly in solid angle.
void build_sky_distribution(const float* rgba, uint32_t w, uint32_t h, SkyDistribution& out);

}  // namespace pt
