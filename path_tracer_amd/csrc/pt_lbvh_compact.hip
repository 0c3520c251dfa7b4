// pt_lbvh_compact.hip — compact (32-byte, 16-bit grid) copy of the LBVH's wide nodes and the breadth-first treelet of its
// top levels (layout and the conservativeness argument: pt_device.h "Compact inner node").  Runs after either builder
// (lbvh.cpp on the host, pt_lbvh_build.hip on the device) and after a refit, on the wide nodes they produced, so both
// builders and their byte-identical arrays stay as they are.
#include "pt_lbvh_build.h"

#include "pt_device.h"

namespace pt {

namespace {

// One thread per wide node.  grid: coordinate = base + q * step per axis (double on the way in; the traversal decodes
// in f32 and its rounding is covered by the spare step taken here).  *bad is raised when a coordinate is not finite.
__global__ __launch_bounds__(256) void lbvh_compact_kernel(const float4* __restrict__ wide, uint32_t n_nodes, double bx, double by,
                                                           double bz, double isx, double isy, double isz, uint4* __restrict__ out,
                                                           uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const float4 n0 = wide[4 * (size_t)i], n1 = wide[4 * (size_t)i + 1], n2 = wide[4 * (size_t)i + 2], n3 = wide[4 * (size_t)i + 3];
    bool finite = true;
    auto qlo = [&](float v, double b, double is) -> uint32_t {
        if (!(__builtin_fabsf(v) <= kF32Max)) { finite = false; return 0u; }
        double q = __builtin_floor(((double)v - b) * is) - 1.0;
        q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
        return (uint32_t)q;
    };
    auto qhi = [&](float v, double b, double is) -> uint32_t {
        if (!(__builtin_fabsf(v) <= kF32Max)) { finite = false; return 65535u; }
        double q = __builtin_ceil(((double)v - b) * is) + 1.0;
        q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
        return (uint32_t)q;
    };
    uint4 a, b;
    a.x = qlo(n0.x, bx, isx) | (qlo(n0.y, by, isy) << 16);
    a.y = qlo(n0.z, bz, isz) | (qhi(n1.x, bx, isx) << 16);
    a.z = qhi(n1.y, by, isy) | (qhi(n1.z, bz, isz) << 16);
    a.w = qlo(n2.x, bx, isx) | (qlo(n2.y, by, isy) << 16);
    b.x = qlo(n2.z, bz, isz) | (qhi(n3.x, bx, isx) << 16);
    b.y = qhi(n3.y, by, isy) | (qhi(n3.z, bz, isz) << 16);
    b.z = f32_bits(n0.w);
    b.w = f32_bits(n1.w);
    out[2 * (size_t)i] = a;
    out[2 * (size_t)i + 1] = b;
    if (!finite) *bad = 1u;  // plain store of the same value by whoever sees one; read by the host after the stream drained
}

// Breadth-first copy of the first kTreeletLevels levels: one workgroup, one level per pass.  Slot 0 is the root; a child
// that is an inner node of a copied level gets the next free slot of the following level (in the order of its parents'
// slots, left before right) and its parent's ref becomes kTreeletTag | slot; refs to deeper inner nodes and to leaves
// stay as they are.  out[2 * slot], out[2 * slot + 1] = the node; *out_count = slots used.
__global__ __launch_bounds__(256) void lbvh_treelet_kernel(const uint4* __restrict__ compact, uint32_t n_nodes, uint4* __restrict__ out,
                                                           uint32_t* __restrict__ out_count) {
    __shared__ int s_node[kTreeletMaxNodes + 1];  // slot -> node index
    __shared__ uint32_t s_scan[256];
    const uint32_t t = threadIdx.x;
    if (t == 0) s_node[0] = 0;
    uint32_t level_begin = 0, level_end = n_nodes ? 1u : 0u;
    __syncthreads();
    for (uint32_t lvl = 0; lvl < kTreeletLevels && level_end > level_begin; ++lvl) {
        const uint32_t n_l = level_end - level_begin;  // <= 128
        const bool have = t < n_l;
        uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
        if (have) {
            const int node = s_node[level_begin + t];
            a = compact[2 * (size_t)node];
            b = compact[2 * (size_t)node + 1];
        }
        const bool last = lvl + 1 == kTreeletLevels;
        const bool cl = have && !last && (int)b.z >= 0, cr = have && !last && (int)b.w >= 0;
        s_scan[t] = (uint32_t)cl + (uint32_t)cr;
        __syncthreads();
        for (uint32_t d = 1; d < 256; d <<= 1) {  // inclusive scan
            const uint32_t v = t >= d ? s_scan[t - d] : 0u;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const uint32_t total = s_scan[255];
        uint32_t slot = level_end + s_scan[t] - ((uint32_t)cl + (uint32_t)cr);
        if (cl) {
            s_node[slot] = (int)b.z;
            b.z = (uint32_t)(kTreeletTag | (int)slot);
            ++slot;
        }
        if (cr) {
            s_node[slot] = (int)b.w;
            b.w = (uint32_t)(kTreeletTag | (int)slot);
        }
        if (have) {
            out[2 * (size_t)(level_begin + t)] = a;
            out[2 * (size_t)(level_begin + t) + 1] = b;
        }
        __syncthreads();
        level_begin = level_end;
        level_end += total;
    }
    if (t == 0) *out_count = level_begin;  // every slot below level_begin was written
}

}  // namespace

bool lbvh_grid_from_root(const float root_node[16], LbvhGrid* g) {
    // root box = union of the root's two child boxes (rows 0/1 = left lo/hi, rows 2/3 = right lo/hi)
    for (int k = 0; k < 3; ++k) {
        const double lo = (double)(root_node[k] < root_node[8 + k] ? root_node[k] : root_node[8 + k]);
        const double hi = (double)(root_node[4 + k] > root_node[12 + k] ? root_node[4 + k] : root_node[12 + k]);
        if (!(lo >= -3.0e38 && hi <= 3.0e38 && hi >= lo)) return false;
        // 65533 steps span the box, so that lo - 1 step and hi + 1 step (the builder's spare step) still fit in 16 bits
        double step = (hi - lo) / 65533.0;
        if (!(step > 0.0)) step = 1.0;  // flat on this axis
        const float stepf = (float)(step * (1.0 + 1.0e-6));  // never below the exact quotient after rounding to f32
        const float basef = (float)(lo - (double)stepf);     // q = 1 decodes to about lo
        if (!(stepf > 0.0f && stepf < 3.0e38f && __builtin_fabsf(basef) < 3.0e38f)) return false;
        // the traversal's error budget assumes a grid that is not tiny next to its distance from the origin
        if (__builtin_fabs(lo) > 1.0e6 * (hi - lo + step) || __builtin_fabs(hi) > 1.0e6 * (hi - lo + step)) return false;
        g->base[k] = basef;
        g->step[k] = stepf;
    }
    return true;
}

hipError_t lbvh_compact_device(const float4* d_wide, uint32_t n_nodes, const LbvhGrid& g, uint4* d_compact, uint4* d_treelet,
                               uint32_t* d_flags, hipStream_t stream) {
    if (n_nodes == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_flags, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lbvh_compact_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, d_wide, n_nodes, (double)g.base[0],
                       (double)g.base[1], (double)g.base[2], 1.0 / (double)g.step[0], 1.0 / (double)g.step[1], 1.0 / (double)g.step[2],
                       d_compact, d_flags);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lbvh_treelet_kernel, dim3(1), dim3(256), 0, stream, d_compact, n_nodes, d_treelet, d_flags + 1);
    return hipGetLastError();
}

}  // namespace pt
