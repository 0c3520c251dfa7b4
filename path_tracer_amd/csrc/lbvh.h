// lbvh.h — linear BVH over spheres (host build, device traversal in pt_device.h / pt_kernels.hip).
//
// The reference has no acceleration structure: its closest hit is the linear scan of integrator.rs:106-127.  The LBVH
// only PRUNES that scan: leaf tests use the scan's arithmetic, the accept rule (t > 0, smaller t wins, equal t keeps
// the lower ORIGINAL index) reproduces "first strictly smaller in index order", and node tests are conservative
// (boxes are inflated at traversal time by a distance-proportional pad that dominates every rounding error of the
// sphere test), so the resulting (t, index) equals the scan's.
#pragma once
#include <stdint.h>

#include <vector>

namespace pt {

constexpr uint32_t kLeafMax = 4;

// 32 bytes.  Pre-order layout: the first child of an inner node is node+1; `skip` is the next node in pre-order after
// this node's subtree (== node count at the end).  leaf < 0 => inner node; else leaf = first * 8 + count into the
// leaf-ordered sphere arrays.
struct BvhNode {
    float lo[3];
    int32_t skip;
    float hi[3];
    int32_t leaf;
};

struct Lbvh {
    std::vector<BvhNode> nodes;
    std::vector<float> leaf_spheres;   // 4 floats per sphere in leaf order: cx, cy, cz, r*r
    std::vector<int32_t> leaf_index;   // original sphere index per leaf-order slot
    uint32_t max_depth = 0;
};

// spheres_xyzr: n x 4 floats (cx, cy, cz, r)
void build_lbvh(const float* spheres_xyzr, uint32_t n, Lbvh& out);

}  // namespace pt
