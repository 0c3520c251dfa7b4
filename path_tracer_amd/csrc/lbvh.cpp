// lbvh.cpp — host LBVH build: 30-bit Morton codes of the sphere centres, sort, top-down split at the highest
// differing code bit (the tree a Karras radix build produces), leaves of <= kLeafMax spheres, pre-order emission with
// skip links for stackless traversal.
#include "lbvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace pt {
namespace {

uint32_t expand10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

enum class Split { Morton, Median, Sah };

struct Builder {
    Split mode = Split::Morton;  // Median (depth <= log2(n) + 1): fallback when the Morton tree is too deep; Sah: lbvh.h
    const float* s;
    std::vector<uint32_t> code;
    std::vector<uint32_t> order;
    Lbvh* out;

    void bounds(uint32_t first, uint32_t last, float lo[3], float hi[3]) const {
        for (int k = 0; k < 3; ++k) {
            lo[k] = INFINITY;
            hi[k] = -INFINITY;
        }
        for (uint32_t i = first; i < last; ++i) {
            const float* p = s + 4 * (size_t)order[i];
            const float r = std::fabs(p[3]);
            for (int k = 0; k < 3; ++k) {
                // one ulp outwards so that c +- r rounding never shrinks the box; non-finite spheres make the box
                // infinite (always traversed), which keeps the scan's behaviour for them
                float a = std::nextafter(p[k] - r, -INFINITY), b = std::nextafter(p[k] + r, INFINITY);
                if (!(a == a) || !(b == b)) {
                    a = -INFINITY;
                    b = INFINITY;
                }
                lo[k] = std::min(lo[k], a);
                hi[k] = std::max(hi[k], b);
            }
        }
    }

    // Binned surface-area heuristic over the sphere centres of order[first, last): 16 bins on each axis, cost of a plane =
    // area(left boxes) * count(left) + area(right boxes) * count(right); order[first, last) is partitioned in place (stable,
    // so the result is a function of the input alone) and the split position returned.  Spheres that are not finite, a
    // range whose centres coincide, or a plane that leaves one side empty fall back to the median of the current order.
    uint32_t sah_partition(uint32_t first, uint32_t last) {
        constexpr int kBins = 16;
        const uint32_t count = last - first, median = first + count / 2;
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = first; i < last; ++i) {
            const float* p = s + 4 * (size_t)order[i];
            for (int k = 0; k < 4; ++k)
                if (!std::isfinite(p[k])) return median;
            for (int k = 0; k < 3; ++k) {
                clo[k] = std::min(clo[k], p[k]);
                chi[k] = std::max(chi[k], p[k]);
            }
        }
        struct Bin {
            double lo[3], hi[3];
            uint32_t n;
        };
        auto area = [](const double* lo, const double* hi) {
            const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        double best_cost = INFINITY;
        int best_axis = -1, best_bin = 0;
        for (int axis = 0; axis < 3; ++axis) {
            const float ext = chi[axis] - clo[axis];
            if (!(ext > 0.0f)) continue;
            Bin bins[kBins];
            for (Bin& b : bins) {
                for (int k = 0; k < 3; ++k) b.lo[k] = INFINITY, b.hi[k] = -INFINITY;
                b.n = 0;
            }
            const double scale = (double)kBins / (double)ext;
            for (uint32_t i = first; i < last; ++i) {
                const float* p = s + 4 * (size_t)order[i];
                const int j = std::min(kBins - 1, (int)(((double)p[axis] - (double)clo[axis]) * scale));
                const double r = std::fabs((double)p[3]);
                for (int k = 0; k < 3; ++k) {
                    bins[j].lo[k] = std::min(bins[j].lo[k], (double)p[k] - r);
                    bins[j].hi[k] = std::max(bins[j].hi[k], (double)p[k] + r);
                }
                bins[j].n++;
            }
            double right_area[kBins];
            uint32_t right_n[kBins];
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t acc = 0;
            for (int j = kBins - 1; j > 0; --j) {
                for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], bins[j].lo[k]), hi[k] = std::max(hi[k], bins[j].hi[k]);
                acc += bins[j].n;
                right_area[j] = acc ? area(lo, hi) : 0.0;
                right_n[j] = acc;
            }
            for (int k = 0; k < 3; ++k) lo[k] = INFINITY, hi[k] = -INFINITY;
            acc = 0;
            for (int j = 0; j + 1 < kBins; ++j) {  // plane between bin j and bin j + 1
                for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], bins[j].lo[k]), hi[k] = std::max(hi[k], bins[j].hi[k]);
                acc += bins[j].n;
                if (acc == 0 || right_n[j + 1] == 0) continue;
                const double cost = area(lo, hi) * acc + right_area[j + 1] * right_n[j + 1];
                if (cost < best_cost) best_cost = cost, best_axis = axis, best_bin = j;
            }
        }
        if (best_axis < 0) return median;
        const double scale = (double)kBins / (double)(chi[best_axis] - clo[best_axis]);
        auto mid = std::stable_partition(order.begin() + first, order.begin() + last, [&](uint32_t id) {
            const float* p = s + 4 * (size_t)id;
            return std::min(kBins - 1, (int)(((double)p[best_axis] - (double)clo[best_axis]) * scale)) <= best_bin;
        });
        const uint32_t split = (uint32_t)(mid - order.begin());
        return (split == first || split == last) ? median : split;
    }

    // returns the node index
    uint32_t build(uint32_t first, uint32_t last, uint32_t depth) {
        const uint32_t me = (uint32_t)out->nodes.size();
        out->nodes.emplace_back();
        out->max_depth = std::max(out->max_depth, depth);
        BvhNode nd;
        const uint32_t count = last - first;
        if (count <= kLeafMax) {
            bounds(first, last, nd.lo, nd.hi);
            nd.leaf = (int32_t)(first * 8 + count);
        } else {
            nd.leaf = -1;
            // Binary radix tree over the keys (code, sorted position) — the tree of Karras 2012, which the device
            // builder (pt_lbvh_build.hip) produces node for node: split where the highest bit in which the first
            // and last key differ turns 1.  Equal codes continue on the bits of the sorted position.
            uint32_t split = first + count / 2;
            const uint32_t diff = code[order[first]] ^ code[order[last - 1]];
            if (mode == Split::Sah) {
                split = sah_partition(first, last);
            } else if (mode == Split::Median) {
                // median
            } else if (diff != 0) {
                const int b = 31 - __builtin_clz(diff);
                uint32_t lo_i = first, hi_i = last - 1;  // first element whose bit b is set (codes are sorted)
                while (lo_i < hi_i) {
                    const uint32_t mid = (lo_i + hi_i) / 2;
                    if ((code[order[mid]] >> b) & 1u) hi_i = mid; else lo_i = mid + 1;
                }
                split = lo_i;
            } else {
                const int b = 31 - __builtin_clz(first ^ (last - 1));
                split = ((last - 1) >> b) << b;
            }
            const uint32_t l = build(first, split, depth + 1);
            const uint32_t r = build(split, last, depth + 1);
            for (int k = 0; k < 3; ++k) {  // union of the children's boxes (min/max are exact)
                nd.lo[k] = std::min(out->nodes[l].lo[k], out->nodes[r].lo[k]);
                nd.hi[k] = std::max(out->nodes[l].hi[k], out->nodes[r].hi[k]);
            }
        }
        nd.skip = (int32_t)out->nodes.size();
        out->nodes[me] = nd;
        return me;
    }
};

}  // namespace

static void build_preorder(const float* spheres_xyzr, uint32_t n, Lbvh& out, Split mode) {
    out.nodes.clear();
    out.leaf_spheres.clear();
    out.leaf_index.clear();
    out.max_depth = 0;
    Builder b;
    b.mode = mode;
    b.s = spheres_xyzr;
    b.out = &out;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const float c = spheres_xyzr[4 * (size_t)i + k];
            if (std::isfinite(c)) {
                lo[k] = std::min(lo[k], c);
                hi[k] = std::max(hi[k], c);
            }
        }
    b.code.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t q[3];
        for (int k = 0; k < 3; ++k) {
            const float c = spheres_xyzr[4 * (size_t)i + k];
            const float ext = hi[k] - lo[k];
            float u = (std::isfinite(c) && ext > 0.0f) ? (c - lo[k]) / ext : 0.0f;
            u = std::min(std::max(u, 0.0f), 1.0f);
            q[k] = (uint32_t)std::min(1023.0f, u * 1024.0f);
        }
        b.code[i] = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    }
    b.order.resize(n);
    std::iota(b.order.begin(), b.order.end(), 0u);
    std::sort(b.order.begin(), b.order.end(), [&](uint32_t x, uint32_t y) {
        return b.code[x] != b.code[y] ? b.code[x] < b.code[y] : x < y;
    });
    out.nodes.reserve((size_t)n + 8);
    b.build(0, n, 0);
    out.leaf_spheres.resize((size_t)n * 4);
    out.leaf_index.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float* p = spheres_xyzr + 4 * (size_t)b.order[i];
        float* o = &out.leaf_spheres[4 * (size_t)i];
        o[0] = p[0];
        o[1] = p[1];
        o[2] = p[2];
        o[3] = p[3] * p[3];  // radius * radius (integrator.rs:111)
        out.leaf_index[i] = (int32_t)b.order[i];
    }
}

void build_lbvh(const float* spheres_xyzr, uint32_t n, Lbvh& out, LbvhSplit split) {
    out.wide.clear();
    out.n_inner = 0;
    out.sphere_start.clear();
    out.stack_depth = 0;
    out.nodes.clear();
    out.leaf_spheres.clear();
    out.leaf_index.clear();
    out.max_depth = 0;
    if (n == 0) return;
    build_preorder(spheres_xyzr, n, out, split == LbvhSplit::Sah ? Split::Sah : Split::Morton);
    if (out.max_depth + 1 > kMaxStack) build_preorder(spheres_xyzr, n, out, Split::Median);  // depth <= log2(n) + 1 <= 25

    // wide layout: inner nodes renumbered in pre-order; children of pre-order node i are i + 1 and nodes[i + 1].skip
    const size_t m = out.nodes.size();
    std::vector<int32_t> ref(m);
    int32_t n_inner = 0;
    for (size_t i = 0; i < m; ++i) ref[i] = out.nodes[i].leaf < 0 ? n_inner++ : -(out.nodes[i].leaf + 1);
    auto put = [](WideNode& w, const BvhNode& a, int32_t a_ref, const BvhNode* b, int32_t b_ref, int32_t next_record) {
        uint32_t words[12];
        pack_record(a.lo, a.hi, a_ref, b ? b->lo : nullptr, b ? b->hi : nullptr, b_ref, next_record, words);
        memcpy(w.v, words, sizeof(words));
    };
    if (out.nodes[0].leaf >= 0) {  // the whole tree is one leaf: root with that leaf on the left, an empty leaf on the right
        WideNode w;
        put(w, out.nodes[0], ref[0], &out.nodes[0], -1 /* first 0, count 0 */, 0);
        out.wide.push_back(w);
    } else {
        out.wide.resize((size_t)n_inner);
        for (size_t i = 0; i < m; ++i) {
            if (out.nodes[i].leaf >= 0) continue;
            const size_t l = i + 1, r = (size_t)out.nodes[l].skip;
            put(out.wide[(size_t)ref[i]], out.nodes[l], ref[l], &out.nodes[r], ref[r], 0);
        }
    }
    out.stack_depth = out.max_depth + 1;

    // climb records (lbvh.h) behind the wide nodes
    out.n_inner = (uint32_t)out.wide.size();
    const size_t ni = out.n_inner;
    WideNode zero;
    memset(&zero, 0, sizeof(zero));
    out.wide.resize(2 * ni + n, zero);
    out.sphere_start.assign(2 * (size_t)n, 0);
    if (out.nodes[0].leaf >= 0) {  // a single leaf: nothing to climb to
        for (uint32_t i = 0; i < n; ++i) {
            out.sphere_start[2 * (size_t)out.leaf_index[i]] = ref[0];
            out.sphere_start[2 * (size_t)out.leaf_index[i] + 1] = kClimbDone;
        }
        return;
    }
    std::vector<int32_t> parent(m, -1);
    for (size_t i = 0; i < m; ++i)
        if (out.nodes[i].leaf < 0) {
            parent[i + 1] = (int32_t)i;
            parent[(size_t)out.nodes[i + 1].skip] = (int32_t)i;
        }
    auto sibling = [&](size_t x) {
        const size_t p = (size_t)parent[x], l = p + 1;
        return x == l ? (size_t)out.nodes[l].skip : l;
    };
    auto record_of = [&](size_t x) -> int32_t {
        return out.nodes[x].leaf < 0 ? (int32_t)(ni + (size_t)ref[x]) : (int32_t)(2 * ni + (size_t)(out.nodes[x].leaf >> 3));
    };
    for (size_t x = 1; x < m; ++x) {
        const size_t p = (size_t)parent[x], s = sibling(x);
        const int32_t g = parent[p];
        WideNode& w = out.wide[(size_t)record_of(x)];
        if (g >= 0) {
            const size_t u = sibling(p);
            put(w, out.nodes[s], ref[s], &out.nodes[u], ref[u], parent[(size_t)g] >= 0 ? record_of((size_t)g) : kClimbDone);
        } else {
            put(w, out.nodes[s], ref[s], nullptr, -1, kClimbDone);
        }
        if (out.nodes[x].leaf >= 0) {
            const uint32_t first = (uint32_t)out.nodes[x].leaf >> 3, count = (uint32_t)out.nodes[x].leaf & 7u;
            for (uint32_t k = 0; k < count; ++k) {
                out.sphere_start[2 * (size_t)out.leaf_index[first + k]] = ref[x];
                out.sphere_start[2 * (size_t)out.leaf_index[first + k] + 1] = record_of(x);
            }
        }
    }
}

}  // namespace pt
