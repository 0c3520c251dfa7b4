// pt_lbvh_build.hip — LBVH build on the device, node for node the tree lbvh.cpp builds on the host.
//
// The reference has no acceleration structure (its closest hit is the scan of integrator.rs:106-127); this is the
// device side of the pruning structure described in lbvh.h, for scenes that change between frames (the editor of
// app.rs:59-116 mutates spheres while rendering) and for sphere counts where the host build is the slowest step.
//
// Stages (all on the caller's stream):
//   lb_init / lb_centre_bounds   finite-centre bounds, ordered-integer atomics
//   lb_codes                     30-bit Morton codes with the host's f32 arithmetic (IEEE divide)
//   rocprim::radix_sort_pairs    stable, so equal codes keep ascending original index (the host comparator)
//   lb_radix_tree                Karras 2012: one thread per internal node finds its key range and split;
//                                equal codes continue on the bits of the sorted position
//   lb_boxes / lb_boxes_top      bottom-up: the second child to arrive at a node unions the two boxes (left, right
//                                order, so the zero signs match the host's too) and counts the node's inner nodes;
//                                nodes inside one workgroup's positions first, the straddling ones in one workgroup
//   lb_emit                      nodes of >= 3 spheres become 48-byte wide nodes, numbered in pre-order: a node's
//                                rank is the sum over its ancestors of 1 + (inner nodes of the left sibling subtree);
//                                the same thread writes the climb records of the node's two children (lbvh.h) and the
//                                traversal start of the spheres in its leaf children
//   lb_leaves                    leaf-ordered sphere and index arrays
// Ranges of <= kLeafMax spheres are leaves, exactly as in the host build, so the two builders give byte-identical
// arrays (tests/test_gpu_lbvh_build.py).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "lbvh.h"
#include "pt_lbvh_build.h"

namespace pt {
namespace {

constexpr int kBlock = 256;

// info block (uint32): [0..2] min centre, [3..5] max centre (ordered encoding), [6] max depth, [7] wide node count
constexpr int kInfoWords = kLbvhInfoWords;

__device__ __forceinline__ uint32_t ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// std::min / std::max as the host evaluates them (the first argument wins ties, which fixes the sign of a zero)
__device__ __forceinline__ float min2(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float max2(float a, float b) { return (a < b) ? b : a; }

__device__ __forceinline__ bool finite(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }

constexpr int kSegments = 64;

__global__ void lb_init(uint32_t* info) {
    const int t = threadIdx.x;
    if (t < 3) info[t] = ordered(INFINITY);
    else if (t < 6) info[t] = ordered(-INFINITY);
    else if (t < kInfoWords + kSegments) info[t] = 0u;  // [kInfoWords ...) = queue counters of lb_boxes
}

__global__ void __launch_bounds__(kBlock) lb_centre_bounds(const float4* __restrict__ spheres, uint32_t n, uint32_t* info) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 s = spheres[i];
        const float c[3] = {s.x, s.y, s.z};
        for (int k = 0; k < 3; ++k)
            if (finite(c[k])) {
                lo[k] = fminf(lo[k], c[k]);
                hi[k] = fmaxf(hi[k], c[k]);
            }
    }
    __shared__ uint32_t part[kBlock / 64][6];
    for (int k = 0; k < 3; ++k) {
        uint32_t a = ordered(lo[k]), b = ordered(hi[k]);
        for (int off = 32; off > 0; off >>= 1) {
            a = min(a, (uint32_t)__shfl_xor((int)a, off));
            b = max(b, (uint32_t)__shfl_xor((int)b, off));
        }
        if ((threadIdx.x & 63) == 0) {
            part[threadIdx.x >> 6][k] = a;
            part[threadIdx.x >> 6][3 + k] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        uint32_t v = part[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? min(v, part[w][threadIdx.x]) : max(v, part[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&info[threadIdx.x], v);
        else atomicMax(&info[threadIdx.x], v);
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(kBlock) lb_codes(const float4* __restrict__ spheres, uint32_t n, const uint32_t* __restrict__ info,
                                                   uint32_t* __restrict__ code, uint32_t* __restrict__ index) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 s = spheres[i];
    const float c[3] = {s.x, s.y, s.z};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = unordered(info[k]), hi = unordered(info[3 + k]);
        const float ext = hi - lo;
        float u = (finite(c[k]) && ext > 0.0f) ? (c[k] - lo) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        q[k] = (uint32_t)fminf(1023.0f, u * 1024.0f);
    }
    code[i] = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    index[i] = i;
}

// common-prefix length of the keys (code, position) at sorted positions i and j; -1 outside the array
__device__ __forceinline__ int delta(const uint32_t* __restrict__ code, int n, int i, uint32_t ci, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t cj = code[j];
    return ci != cj ? __clz((int)(ci ^ cj)) : 32 + __clz(i ^ j);
}

// nodes[i] = (first, last, split, parent): the first three words are written by node i's thread, the parent word by the
// parent's thread (-1 for the root, node 0).  leaf_parent[j]: parent of sorted position j.
__global__ void __launch_bounds__(kBlock) lb_radix_tree(const uint32_t* __restrict__ code, int n, int* __restrict__ nodes,
                                                        int* __restrict__ leaf_parent, uint32_t* __restrict__ flags) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n - 1) return;
    const uint32_t ci = code[i];
    const int d = delta(code, n, i, ci, i + 1) > delta(code, n, i, ci, i - 1) ? 1 : -1;
    const int dmin = delta(code, n, i, ci, i - d);
    int lmax = 2;
    while (delta(code, n, i, ci, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(code, n, i, ci, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(code, n, i, ci, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(code, n, i, ci, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int split = i + s * d + min(d, 0);  // last position of the left child
    const int lo = min(i, j), hi = max(i, j);
    nodes[4 * i + 0] = lo;
    nodes[4 * i + 1] = hi;
    nodes[4 * i + 2] = split;
    if (lo == split) leaf_parent[split] = i;
    else nodes[4 * split + 3] = i;
    if (split + 1 == hi) leaf_parent[hi] = i;
    else nodes[4 * (split + 1) + 3] = i;
    if (i == 0) nodes[3] = -1;
    flags[i] = 0u;
}

struct Box {
    float lo[3], hi[3];
};

// box of the sphere at sorted position j: lbvh.cpp Builder::bounds for one sphere
__device__ __forceinline__ Box sphere_box(const float4* __restrict__ spheres, const uint32_t* __restrict__ order, int j) {
    const float4 s = spheres[order[j]];
    const float c[3] = {s.x, s.y, s.z};
    const float r = fabsf(s.w);
    Box b;
    for (int k = 0; k < 3; ++k) {
        float lo = nextafterf(c[k] - r, -INFINITY), hi = nextafterf(c[k] + r, INFINITY);
        if (!(lo == lo) || !(hi == hi)) {
            lo = -INFINITY;
            hi = INFINITY;
        }
        b.lo[k] = lo;
        b.hi[k] = hi;
    }
    return b;
}

__device__ __forceinline__ Box load_box(const float4* boxes, int node, int* inner) {
    const float4 a = boxes[2 * node], c = boxes[2 * node + 1];
    Box b;
    b.lo[0] = a.x; b.lo[1] = a.y; b.lo[2] = a.z;
    b.hi[0] = c.x; b.hi[1] = c.y; b.hi[2] = c.z;
    *inner = __float_as_int(a.w);
    return b;
}

// boxes[2 * node] = (lo.xyz, inner nodes in the subtree), boxes[2 * node + 1] = (hi.xyz, 0)
//
// One climb step at node p for the child whose box is `mine`: the first child to arrive leaves (returns false); its
// stores are released by the atomic and acquired by the second's, which unions the two boxes and carries on.
// SCOPE is the scope both arrivals at the node share.  An agent-scope release / acquire costs an L2 write-back /
// invalidate on a multi-XCD part (with every step at agent scope this pass took 5 ms per million spheres), so
// lb_boxes climbs at workgroup scope as far as a workgroup's positions reach and leaves the rest to lb_boxes_top.
template <int SCOPE>
__device__ __forceinline__ bool climb(const float4* __restrict__ spheres, const uint32_t* __restrict__ order, const int4 nd, int p,
                                      uint32_t* flags, float4* boxes, Box& mine, int& mine_inner, int& mine_lo) {
    if (__hip_atomic_fetch_add(&flags[p], 1u, __ATOMIC_ACQ_REL, SCOPE) == 0u) return false;
    const bool left = mine_lo == nd.x;
    Box other;
    int other_inner = 0;
    if (left) {
        if (nd.z + 1 == nd.y) other = sphere_box(spheres, order, nd.y);
        else other = load_box(boxes, nd.z + 1, &other_inner);
    } else {
        if (nd.x == nd.z) other = sphere_box(spheres, order, nd.x);
        else other = load_box(boxes, nd.z, &other_inner);
    }
    const Box& l = left ? mine : other;
    const Box& r = left ? other : mine;
    Box u;
    for (int k = 0; k < 3; ++k) {
        u.lo[k] = min2(l.lo[k], r.lo[k]);
        u.hi[k] = max2(l.hi[k], r.hi[k]);
    }
    mine = u;
    mine_inner = (nd.y - nd.x + 1 > (int)kLeafMax) ? 1 + mine_inner + other_inner : 0;
    mine_lo = nd.x;
    boxes[2 * p] = make_float4(u.lo[0], u.lo[1], u.lo[2], __int_as_float(mine_inner));
    boxes[2 * p + 1] = make_float4(u.hi[0], u.hi[1], u.hi[2], 0.0f);
    return true;
}

// Pass 1: one thread per sorted position, workgroups of kTopBlock positions.  A thread climbs while the node's positions
// all belong to its workgroup; at the first node that straddles workgroups it queues (node, child) for pass 2.  Queue
// segments (one per blockIdx % kSegments) spread the append counters over addresses.
constexpr int kTopBlock = 1024;

__global__ void __launch_bounds__(kTopBlock) lb_boxes(const float4* __restrict__ spheres, const uint32_t* __restrict__ order, int n,
                                                      const int4* __restrict__ nodes, const int* __restrict__ leaf_parent, uint32_t* flags,
                                                      float4* boxes, uint32_t* queue_count, int2* __restrict__ queue, uint32_t segment_cap) {
    const int j = blockIdx.x * kTopBlock + threadIdx.x;
    if (j >= n) return;
    Box mine = sphere_box(spheres, order, j);
    int mine_inner = 0;
    int mine_lo = j;
    int child = n + j;
    int p = leaf_parent[j];
    while (p >= 0) {
        const int4 nd = nodes[p];
        if (nd.x / kTopBlock != nd.y / kTopBlock) {
            const uint32_t seg = blockIdx.x % kSegments;
            const uint32_t slot = atomicAdd(&queue_count[seg], 1u);
            queue[(size_t)seg * segment_cap + slot] = make_int2(p, child);
            return;
        }
        if (!climb<__HIP_MEMORY_SCOPE_WORKGROUP>(spheres, order, nd, p, flags, boxes, mine, mine_inner, mine_lo)) return;
        child = p;
        p = nd.w;
    }
}

// Pass 2: the nodes that straddle pass-1 workgroups (a few per workgroup boundary and level): one thread per queued
// arrival, agent scope.  It starts from a clean L2 (kernel boundary) and makes few stores, so the write-backs are cheap.
__global__ void __launch_bounds__(kBlock) lb_boxes_top(const float4* __restrict__ spheres, const uint32_t* __restrict__ order, int n,
                                                       const int4* __restrict__ nodes, uint32_t* flags, float4* boxes,
                                                       const uint32_t* __restrict__ queue_count, const int2* __restrict__ queue,
                                                       uint32_t segment_cap) {
    const uint32_t seg = blockIdx.y;
    const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= queue_count[seg]) return;
    const int2 entry = queue[(size_t)seg * segment_cap + e];
    Box mine;
    int mine_inner = 0, mine_lo;
    if (entry.y >= n) {
        mine_lo = entry.y - n;
        mine = sphere_box(spheres, order, mine_lo);
    } else {
        mine = load_box(boxes, entry.y, &mine_inner);
        mine_lo = nodes[entry.y].x;
    }
    int4 nd = nodes[entry.x];
    for (int p = entry.x; p >= 0;) {
        const int up = nd.w;
        const int4 next = nodes[up >= 0 ? up : 0];  // in flight while the atomic and the sibling's box are fetched
        if (!climb<__HIP_MEMORY_SCOPE_AGENT>(spheres, order, nd, p, flags, boxes, mine, mine_inner, mine_lo)) return;
        p = up;
        nd = next;
    }
}

__device__ __forceinline__ void child_entry(const float4* __restrict__ spheres, const uint32_t* __restrict__ order,
                                            const float4* __restrict__ boxes, int lo, int hi, int node, int rank, Box* b, int* ref) {
    int inner;
    if (lo == hi) *b = sphere_box(spheres, order, lo);
    else *b = load_box(boxes, node, &inner);
    const int count = hi - lo + 1;
    *ref = count <= (int)kLeafMax ? -(lo * 8 + count) - 1 : rank;
}

// One 48-byte record (lbvh.h): boxes A and B as centre + f16 half extent, their refs, the next record.
__device__ __forceinline__ void put_record(float4* __restrict__ rec, const Box& a, int a_ref, const Box* b, int b_ref, int next) {
    uint32_t w[12];
    pack_record(a.lo, a.hi, a_ref, b ? b->lo : nullptr, b ? b->hi : nullptr, b_ref, next, w);
    rec[0] = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
    rec[1] = make_float4(__uint_as_float(w[4]), __uint_as_float(w[5]), __uint_as_float(w[6]), __uint_as_float(w[7]));
    rec[2] = make_float4(__uint_as_float(w[8]), __uint_as_float(w[9]), __uint_as_float(w[10]), __uint_as_float(w[11]));
}

__global__ void __launch_bounds__(kBlock) lb_emit(const float4* __restrict__ spheres, const uint32_t* __restrict__ order, int n,
                                                  const int4* __restrict__ nodes, const float4* __restrict__ boxes,
                                                  float4* __restrict__ wide, int2* __restrict__ sphere_start, uint32_t* info) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n - 1) return;
    const int4 nd = nodes[i];
    if (nd.y - nd.x + 1 <= (int)kLeafMax) return;
    int rank = 0, depth = 0, child_lo = nd.x;
    for (int p = nd.w; p >= 0;) {
        const int4 a = nodes[p];
        rank += 1;
        if (child_lo != a.x && a.z - a.x + 1 > (int)kLeafMax) rank += __float_as_int(boxes[2 * a.z].w);
        child_lo = a.x;
        ++depth;
        p = a.w;
    }
    const int left_inner = (nd.z - nd.x + 1 > (int)kLeafMax) ? __float_as_int(boxes[2 * nd.z].w) : 0;
    Box bl, br;
    int rl, rr;
    child_entry(spheres, order, boxes, nd.x, nd.z, nd.z, rank + 1, &bl, &rl);
    child_entry(spheres, order, boxes, nd.z + 1, nd.y, nd.z + 1, rank + 1 + left_inner, &br, &rr);
    put_record(wide + 3 * (size_t)rank, bl, rl, &br, rr, 0);
    atomicMax(&info[6], (uint32_t)depth + 1u);  // the children are one level below
    const int n_inner = __float_as_int(boxes[0].w);
    if (i == 0) info[7] = (uint32_t)n_inner;

    // Climb records of this node's two children: their sibling is the other child, their uncle is this node's sibling,
    // their grandparent this node's parent.
    Box bu = bl;
    int ru = -1, next = kClimbDone;
    const bool has_uncle = nd.w >= 0;
    if (has_uncle) {
        const int4 a = nodes[nd.w];
        const bool i_is_left = nd.x == a.x;
        const int a_left_inner = (a.z - a.x + 1 > (int)kLeafMax) ? __float_as_int(boxes[2 * a.z].w) : 0;
        const int parent_rank = i_is_left ? rank - 1 : rank - 1 - a_left_inner;
        if (i_is_left) child_entry(spheres, order, boxes, a.z + 1, a.y, a.z + 1, parent_rank + 1 + a_left_inner, &bu, &ru);
        else child_entry(spheres, order, boxes, a.x, a.z, a.z, parent_rank + 1, &bu, &ru);
        if (a.w >= 0) next = n_inner + parent_rank;
    }
    const int rec_l = rl >= 0 ? n_inner + rl : 2 * n_inner + nd.x;
    const int rec_r = rr >= 0 ? n_inner + rr : 2 * n_inner + nd.z + 1;
    put_record(wide + 3 * (size_t)rec_l, br, rr, has_uncle ? &bu : nullptr, has_uncle ? ru : -1, next);
    put_record(wide + 3 * (size_t)rec_r, bl, rl, has_uncle ? &bu : nullptr, has_uncle ? ru : -1, next);
    if (rl < 0)
        for (int j = nd.x; j <= nd.z; ++j) sphere_start[order[j]] = make_int2(rl, rec_l);
    if (rr < 0)
        for (int j = nd.z + 1; j <= nd.y; ++j) sphere_start[order[j]] = make_int2(rr, rec_r);
}

__global__ void __launch_bounds__(kBlock) lb_leaves(const float4* __restrict__ spheres, const uint32_t* __restrict__ order, uint32_t n,
                                                    float4* __restrict__ leaf_spheres, int* __restrict__ leaf_index) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = order[i];
    const float4 s = spheres[o];
    leaf_spheres[i] = make_float4(s.x, s.y, s.z, s.w * s.w);  // radius * radius (integrator.rs:111)
    leaf_index[i] = (int)o;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// pass-1 workgroups per queue segment x entries a workgroup can append (one per position)
static size_t queue_segment_cap(uint32_t n) {
    const size_t blocks = ((size_t)n + kTopBlock - 1) / kTopBlock;
    return (blocks + kSegments - 1) / kSegments * kTopBlock;
}

size_t lbvh_build_sort_bytes(uint32_t n) {
    size_t b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n,
                                    0u, 30u, (hipStream_t) nullptr);
    return b;
}

size_t lbvh_build_scratch_bytes(uint32_t n) {
    const size_t m = n;
    return align256((kInfoWords + kSegments) * 4) + 4 * align256(m * 4)   // info, code/index in and out
           + align256(queue_segment_cap(n) * kSegments * sizeof(int2))
           + align256(m * 16) + align256(m * 4)              // nodes, leaf_parent
           + align256(m * 4) + align256(m * 32)              // flags, boxes
           + align256(lbvh_build_sort_bytes(n));
}

hipError_t lbvh_build_device(const float4* d_spheres, uint32_t n, void* scratch, size_t scratch_bytes, float4* d_leaf_spheres,
                             int* d_leaf_index, LbvhDeviceTree* tree, bool refit, hipStream_t stream) {
    tree->failed_stage = "arguments";
    if (n < 3 || scratch_bytes < lbvh_build_scratch_bytes(n)) return hipErrorInvalidValue;
    const size_t m = n;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* q = p;
        p += align256(bytes);
        return q;
    };
    uint32_t* info = (uint32_t*)take((kInfoWords + kSegments) * 4);
    const uint32_t segment_cap = (uint32_t)queue_segment_cap(n);
    int2* queue = (int2*)take((size_t)segment_cap * kSegments * sizeof(int2));
    uint32_t* code_in = (uint32_t*)take(m * 4);
    uint32_t* index_in = (uint32_t*)take(m * 4);
    uint32_t* code = (uint32_t*)take(m * 4);
    uint32_t* order = (uint32_t*)take(m * 4);
    int4* nodes = (int4*)take(m * 16);
    int* leaf_parent = (int*)take(m * 4);
    uint32_t* flags = (uint32_t*)take(m * 4);
    float4* boxes = (float4*)take(m * 32);
    size_t sort_bytes = lbvh_build_sort_bytes(n);
    void* sort_tmp = take(sort_bytes);

    const uint32_t blocks = (n + kBlock - 1) / kBlock;
    hipError_t e;
#define LB_STAGE(name, ...)                       \
    __VA_ARGS__;                                  \
    if ((e = hipGetLastError()) != hipSuccess) {  \
        tree->failed_stage = name;                \
        return e;                                 \
    }
    LB_STAGE("lb_init", lb_init<<<1, 128, 0, stream>>>(info))
    if (refit) {
        // order, nodes and leaf_parent of the previous build of these n spheres are still in the scratch buffer: only the
        // boxes (and with them the wide nodes and the leaf arrays) follow the moved spheres
        if ((e = hipMemsetAsync(flags, 0, m * 4, stream)) != hipSuccess) {
            tree->failed_stage = "memset";
            return e;
        }
    } else {
        LB_STAGE("lb_centre_bounds", lb_centre_bounds<<<min(blocks, 256u), kBlock, 0, stream>>>(d_spheres, n, info))
        LB_STAGE("lb_codes", lb_codes<<<blocks, kBlock, 0, stream>>>(d_spheres, n, info, code_in, index_in))
        e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, code_in, code, index_in, order, (size_t)n, 0u, 30u, stream);
        if (e != hipSuccess) {
            tree->failed_stage = "radix_sort_pairs";
            return e;
        }
        LB_STAGE("lb_radix_tree", lb_radix_tree<<<blocks, kBlock, 0, stream>>>(code, (int)n, (int*)nodes, leaf_parent, flags))
    }
    LB_STAGE("lb_boxes", lb_boxes<<<(n + kTopBlock - 1) / kTopBlock, kTopBlock, 0, stream>>>(d_spheres, order, (int)n, nodes, leaf_parent, flags,
                                                                                            boxes, info + kInfoWords, queue, segment_cap))
    LB_STAGE("lb_boxes_top", lb_boxes_top<<<dim3((segment_cap + kBlock - 1) / kBlock, kSegments), kBlock, 0, stream>>>(
                                 d_spheres, order, (int)n, nodes, flags, boxes, info + kInfoWords, queue, segment_cap))
    LB_STAGE("lb_leaves", lb_leaves<<<blocks, kBlock, 0, stream>>>(d_spheres, order, n, d_leaf_spheres, d_leaf_index))
#undef LB_STAGE
    tree->info = info;
    tree->order = order;
    tree->nodes = nodes;
    tree->boxes = boxes;
    return hipSuccess;
}

hipError_t lbvh_emit_device(const float4* d_spheres, uint32_t n, const LbvhDeviceTree& tree, float4* d_wide, int2* d_sphere_start,
                            hipStream_t stream) {
    const uint32_t blocks = (n + kBlock - 1) / kBlock;
    lb_emit<<<blocks, kBlock, 0, stream>>>(d_spheres, (const uint32_t*)tree.order, (int)n, (const int4*)tree.nodes,
                                           (const float4*)tree.boxes, d_wide, d_sphere_start, (uint32_t*)tree.info);
    return hipGetLastError();
}

}  // namespace pt
