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

// Traversal layout ("wide" inner nodes: both children's boxes in the parent), built from the pre-order tree.  A record is
// 48 bytes, three 16-byte loads: the kernel that walks it is bound by the L1 data path, which moves 16 bytes per lane
// and load whether the lane is active or not, so a box is stored as its centre (f32) and its half extent
// rounded UP to f16 — a box that contains the builder's box; node tests only have to be conservative (top of file):
//   [0] = (left centre xyz,  left ref)   [1] = (right centre xyz, right ref)
//   [2] = (half extents as six f16: left x | left y << 16, left z | right x << 16, right y | right z << 16; next record)
// centre = 0.5 * (lo + hi), half extent = max(hi - centre, centre - lo) moved one ulp up, then to f16 rounding up (above
// 65504: +inf, a box every ray hits); a box with a non-finite centre becomes (0, +inf).
// ref >= 0: index of an inner node; ref < 0: leaf, -(ref + 1) = first * 8 + count into the leaf-ordered arrays
// (count may be 0: the padding child of a tree that is a single leaf).  Node 0 is the root.  One fetch tests both
// children, the nearer one is entered first and the farther one is pushed on a per-lane stack of at most
// `stack_depth` entries (<= kMaxStack, enforced by the builder).  "next record" is 0 in a wide node.
struct WideNode {
    float v[12];
};
#if defined(__HIP__)
#define LBVH_HD __host__ __device__ inline
#else
#define LBVH_HD inline
#endif
// smallest f16 >= x for x >= +0 (NaN -> f16 NaN); integer arithmetic only, so host and device builders agree bit for bit
LBVH_HD uint32_t half_round_up(float x) {
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    if (u > 0x7f800000u) return 0x7e00u;   // NaN (negative inputs do not occur)
    if (u > 0x477fe000u) return 0x7c00u;   // above 65504: +inf
    if (u < 0x38800000u) {                 // below 2^-14: f16 subnormals are multiples of 2^-24
        const float s = x * 16777216.0f;   // exact
        uint32_t m = (uint32_t)s;
        if ((float)m < s) m++;
        return m;                          // (1024 = 0x0400 is 2^-14, the smallest normal)
    }
    const uint32_t mant = u & 0x7fffffu;
    uint32_t h = (((u >> 23) - 112u) << 10) | (mant >> 13);
    if (mant & 0x1fffu) h++;               // a carry out of the mantissa moves to the next exponent, as it should
    return h;
}
// centre and f16 half extent (as its bits) of one axis of a box
LBVH_HD void centre_extent(float lo, float hi, float* centre, uint32_t* half_extent_f16) {
    float c = 0.5f * (lo + hi);
    float e = (hi - c > c - lo) ? hi - c : c - lo;
    uint32_t cu, eu;
    __builtin_memcpy(&cu, &c, 4);
    if ((cu & 0x7f800000u) == 0x7f800000u || !(e == e)) {  // centre inf or NaN (an infinite box), or an extent that is NaN
        c = 0.0f;
        eu = 0x7f800000u;
    } else {
        __builtin_memcpy(&eu, &e, 4);
        if (eu < 0x7f800000u) eu++;  // one ulp up: the subtractions above round to nearest
    }
    float e_up;
    __builtin_memcpy(&e_up, &eu, 4);
    *centre = c;
    *half_extent_f16 = half_round_up(e_up);
}
// boxes A and B with their refs and the record's last word -> the 12 words of a record.  b_lo == nullptr: no box B (a
// climb record below the root has no uncle): centre 0, half extent NaN, which no ray hits (every comparison of the
// box test is false).
LBVH_HD void pack_record(const float* a_lo, const float* a_hi, int32_t a_ref, const float* b_lo, const float* b_hi, int32_t b_ref,
                         int32_t next_record, uint32_t* out12) {
    uint32_t h[6];
    float c[6];
    for (int k = 0; k < 3; ++k) {
        centre_extent(a_lo[k], a_hi[k], &c[k], &h[k]);
        if (b_lo) {
            centre_extent(b_lo[k], b_hi[k], &c[3 + k], &h[3 + k]);
        } else {
            c[3 + k] = 0.0f;
            h[3 + k] = 0x7e00u;
        }
    }
    for (int k = 0; k < 3; ++k) {
        __builtin_memcpy(&out12[k], &c[k], 4);
        __builtin_memcpy(&out12[4 + k], &c[3 + k], 4);
    }
    out12[3] = (uint32_t)a_ref;
    out12[7] = (uint32_t)b_ref;
    out12[8] = h[0] | (h[1] << 16);
    out12[9] = h[2] | (h[3] << 16);
    out12[10] = h[4] | (h[5] << 16);
    out12[11] = (uint32_t)next_record;
}
constexpr uint32_t kMaxStack = 32;

// Climb records ("start at the leaf the ray stands on"): a bounce ray leaves a sphere whose leaf is known, and every
// node on the way from the root to that leaf contains the origin, so a top-down traversal spends most of its steps
// walking down that path (two box tests per level, one of them a foregone conclusion).  A climb starts at the leaf and
// walks UP instead: at each ancestor only the sibling subtree is left to look at, and one record holds TWO levels of
// siblings, so the walk covers two levels per step with the same two box tests an inner node takes.  Records have the
// wide-node layout and live in the same array (`wide`), after the n_inner wide nodes:
//   record of inner node i (i >= 1)        at n_inner + i           (slot n_inner + 0, the root's, is unused: zeros)
//   record of the leaf starting at first   at 2 * n_inner + first   (positions that start no leaf: zeros)
//   box A = the sibling, box B = the uncle, last word = the next record
// sibling = the other child of the node's parent, uncle = the other child of its grandparent (no box, ref -1, when the
// parent is the root), next = record of the grandparent, or kClimbDone when the grandparent is
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

// How the top-down build splits a range of spheres.  Morton: at the highest differing bit of the sorted 30-bit codes (the
// tree the device builder reproduces byte for byte).  Sah: binned surface-area heuristic over the centres (16 bins per axis),
// host only: a slower build (10 k spheres: 4.6 ms against 1.8 ms, 100 k: 53 against 18) for a tree that traverses 1.03-1.10x
// faster on uniformly scattered spheres and 1.13-1.31x faster on clustered ones (profiles/r3_sah_vs_morton.txt); for scenes
// that stay as they are for many launches.  Hits do
// not depend on the tree (lbvh.h top), so both give the scan's (t, index).
enum class LbvhSplit { Morton, Sah };

// spheres_xyzr: n x 4 floats (cx, cy, cz, r)
void build_lbvh(const float* spheres_xyzr, uint32_t n, Lbvh& out, LbvhSplit split = LbvhSplit::Morton);

}  // namespace pt
