// lbvh.h — linear BVH over spheres (host build, device traversal in pt_device.h / pt_kernels.hip).
//
// The reference has no acceleration structure: its closest hit is the linear scan of integrator.rs:106-127.  The LBVH
// only PRUNES that scan: leaf tests use the scan's arithmetic, the accept rule (t > 0, smaller t wins, equal t keeps
// the lower ORIGINAL index) reproduces "first strictly smaller in index order", and node tests are conservative
// (boxes are inflated at traversal time by a distance-proportional pad that dominates every rounding error of the
// sphere test, derivation in pt_device.h), so the resulting (t, index) equals the scan's.
#pragma once
#include <stdint.h>

#include <vector>

namespace pt {

constexpr uint32_t kLeafMax = 2;

// 32 bytes.  Pre-order layout: the first child of an inner node is node+1; `skip` is the next node in pre-order after
// this node's subtree (== node count at the end).  leaf < 0 => inner node; else leaf = first * 8 + count into the
// leaf-ordered sphere arrays.
struct BvhNode {
    float lo[3];
    int32_t skip;
    float hi[3];
    int32_t leaf;
};

// Traversal layout ("wide" inner nodes: both children's boxes in the parent, 64 bytes), built from the pre-order tree:
//   [0] = (left.lo.xyz,  left ref)   [1] = (left.hi.xyz,  right ref)   [2] = (right.lo.xyz, 0)   [3] = (right.hi.xyz, 0)
// ref >= 0: index of an inner node; ref < 0: leaf, -(ref + 1) = first * 8 + count into the leaf-ordered arrays
// (count may be 0: the padding child of a tree that is a single leaf).  Node 0 is the root.  One fetch tests both
// children, the nearer one is entered first and the farther one is pushed on a per-lane stack of at most
// `stack_depth` entries (<= kMaxStack, enforced by the builder).
struct WideNode {
    float v[16];
};
constexpr uint32_t kMaxStack = 32;

// Climb records ("start at the leaf the ray stands on"): a bounce ray leaves a sphere whose leaf is known, and every
// node on the way from the root to that leaf contains the origin, so a top-down traversal spends most of its steps
// walking down that path (two box tests per level, one of them a foregone conclusion).  A climb starts at the leaf and
// walks UP instead: at each ancestor only the sibling subtree is left to look at, and one record holds TWO levels of
// siblings, so the walk covers two levels per step with the same two box tests an inner node takes.  Records have the
// wide-node layout and live in the same array (`wide`), after the n_inner wide nodes:
//   record of inner node i (i >= 1)        at n_inner + i           (slot n_inner + 0, the root's, is unused: zeros)
//   record of the leaf starting at first   at 2 * n_inner + first   (positions that start no leaf: zeros)
//   [0] = (sibling.lo, sibling ref)  [1] = (sibling.hi, uncle ref)  [2] = (uncle.lo, next record)  [3] = (uncle.hi, 0)
// sibling = the other child of the node's parent, uncle = the other child of its grandparent (a NaN box that no ray
// hits, ref -1, when the parent is the root), next = record of the grandparent, or kClimbDone when the grandparent is
// the root or missing.  Visiting order never enters the result (lbvh.h top), and every leaf is either the start leaf
// or inside exactly one sibling subtree of the start leaf's ancestors, so a climb tests what a top-down traversal tests.
// sphere_start[original index] = (leaf ref, record of that leaf or kClimbDone).
constexpr int32_t kClimbDone = 0x7fffffff;  // == pt::kCursorDone (pt_device.h)

struct Lbvh {
    std::vector<WideNode> wide;       // n_inner wide nodes, then the climb records (2 * n_inner + n entries in all)
    uint32_t n_inner = 0;
    std::vector<int32_t> sphere_start;  // 2 per sphere, by original index
    uint32_t stack_depth = 0;
    std::vector<BvhNode> nodes;
    std::vector<float> leaf_spheres;   // 4 floats per sphere in leaf order: cx, cy, cz, r*r
    std::vector<int32_t> leaf_index;   // original sphere index per leaf-order slot
    uint32_t max_depth = 0;
};

// spheres_xyzr: n x 4 floats (cx, cy, cz, r)
void build_lbvh(const float* spheres_xyzr, uint32_t n, Lbvh& out);

}  // namespace pt
