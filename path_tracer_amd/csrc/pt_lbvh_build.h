// pt_lbvh_build.h — device LBVH builder (pt_lbvh_build.hip): same tree and arrays as lbvh.cpp's host build.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace pt {

// Pointers into the caller's scratch buffer, valid until the next build.
struct LbvhDeviceTree {
    void* info = nullptr;    // uint32[8]: [6] = depth of the deepest node, [7] = wide (inner) node count
    void* order = nullptr;   // sorted position -> original sphere index
    void* nodes = nullptr;   // int4 per radix-tree node: first, last, split, parent
    void* boxes = nullptr;
    const char* failed_stage = "";  // set when lbvh_build_device returns an error
};

constexpr int kLbvhInfoWords = 8;

size_t lbvh_build_scratch_bytes(uint32_t n);

// Radix tree, boxes and the leaf-ordered arrays (n x float4 and n x int) of n >= 3 spheres (cx, cy, cz, r).
// refit: keep the sorted order and the tree of the previous call with the same n and the same scratch buffer, recompute
// only what depends on the spheres' coordinates (boxes, leaf arrays).  The result is a valid tree of the moved spheres
// (every box bounds its subtree), not the tree a fresh build would make.
hipError_t lbvh_build_device(const float4* d_spheres, uint32_t n, void* scratch, size_t scratch_bytes, float4* d_leaf_spheres,
                             int* d_leaf_index, LbvhDeviceTree* tree, bool refit, hipStream_t stream);

// Wide nodes in pre-order, then the climb records (lbvh.h), into d_wide (room for 2 (n - 1) + n records of 48 bytes, zeroed
// by the caller: slots no node owns stay zero); the traversal start of every sphere into d_sphere_start (n x int2, by
// original index); fills info[6] and info[7].
hipError_t lbvh_emit_device(const float4* d_spheres, uint32_t n, const LbvhDeviceTree& tree, float4* d_wide, int2* d_sphere_start,
                            hipStream_t stream);

}  // namespace pt
