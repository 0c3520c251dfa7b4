// insploray_comm.hip — the path's one exchange step, inside the C ABI: partial radiance buffers of the ranks of one
// node are combined on a root rank with RCCL over xGMI.
//
// Reference analogue: the merge loop of RayTracer::render (core-engine/src/renderer.rs:152-159), where the main thread
// receives every worker's TileAccumulator and adds it into the frame (Accumulator::merge_tile, full_buffer.rs:107-125).
// Here a "worker" is a GPU (one process per GPU, one context per process):
//   * IR_PARTITION_TILES   — ranks own disjoint 64x64 tiles (renderer.rs:105-119), so merging is a GATHER: each rank
//     packs its owned tiles (sums + counts, 20 B per pixel) into one contiguous payload, sends it to the root with one
//     ncclSend, and the root unpacks every payload into the combined film.  1/world of the frame per rank crosses xGMI
//     (16.6 MB at 4K over 8 ranks) instead of a reduce of the zero-padded frame (166 MB);
//   * IR_PARTITION_SAMPLES — every rank holds a partial sum of every pixel: one grouped ncclReduce(sum) of the float4
//     sums and the u32 counts (one launch on the wire, two datatypes).
// The exchange reads a SNAPSHOT of the film taken on the render stream (a device-to-device copy or the pack kernel, tens
// of microseconds) and runs on its own stream, so the next render overlaps it; the combined film lives in a buffer of
// its own on the root, so every rank's partial film stays intact and progressive rendering can go on after a flush.
//
// RCCL is loaded at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, a process that
// already loaded an RCCL (torch) shares that copy, and a host without RCCL still renders on one GPU.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "ir_internal.h"
#include "pt_kernels.h"

namespace irx {

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

template <class F>
bool bind(void* h, const char* name, F& fn, std::string& err) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    if (!fn) err = std::string("librccl has no symbol ") + name;
    return fn != nullptr;
}

const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl& r = g_rccl;
        // A copy the process already holds (torch ships one under the same soname) is shared: RTLD_NOLOAD first.  A fresh
        // copy is loaded RTLD_LOCAL, so that its symbols never interpose on another RCCL that is loaded later.
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // IR_DEBUG_RCCL_LIB=<path>: tests only (like IR_DEBUG_LIB) — a stand-in transport with the same ten entry points
        // (tests/cpp/fake_rccl.cpp), so that the N-rank branches below run on a box whose one GPU RCCL would refuse to share
        if (const char* debug_lib = getenv("IR_DEBUG_RCCL_LIB")) {
            r.handle = dlopen(debug_lib, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) {
                const char* e = dlerror();
                r.error = std::string("cannot load IR_DEBUG_RCCL_LIB ") + debug_lib + ": " + (e ? e : "?");
                return;
            }
        }
        for (const char* n : names) {
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.handle) break;
        }
        for (const char* n : names) {
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "not found");
            return;
        }
        const bool ok = bind(r.handle, "ncclGetUniqueId", r.GetUniqueId, r.error) && bind(r.handle, "ncclCommInitRank", r.CommInitRank, r.error) &&
                        bind(r.handle, "ncclCommDestroy", r.CommDestroy, r.error) &&
                        bind(r.handle, "ncclCommGetAsyncError", r.CommGetAsyncError, r.error) &&
                        bind(r.handle, "ncclGetErrorString", r.GetErrorString, r.error) && bind(r.handle, "ncclReduce", r.Reduce, r.error) &&
                        bind(r.handle, "ncclSend", r.Send, r.error) && bind(r.handle, "ncclRecv", r.Recv, r.error) &&
                        bind(r.handle, "ncclGroupStart", r.GroupStart, r.error) && bind(r.handle, "ncclGroupEnd", r.GroupEnd, r.error);
        if (!ok) r.handle = nullptr;
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

#define IR_NCCL(ctx, call)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return fail(ctx, IR_EDEVICE, "%s: %s", #call, rccl()->GetErrorString(r_));   \
    } while (0)

static_assert(sizeof(ncclUniqueId) == IR_COMM_ID_BYTES, "IR_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

constexpr uint32_t kTilePixels = pt::kTile * pt::kTile;

// ---- tile payloads: slot k of rank r's payload is tile r + k * world, 64x64 pixels row-major (pixels outside a clipped
// edge tile are zero); all sums first, then all counts
__global__ __launch_bounds__(256) void pack_tiles_kernel(const float4* __restrict__ sum, const uint32_t* __restrict__ counts, uint32_t w,
                                                         uint32_t h, uint32_t tiles_x, uint32_t rank, uint32_t world, uint32_t owned,
                                                         float4* __restrict__ out_sum, uint32_t* __restrict__ out_counts) {
    const size_t total = (size_t)owned * kTilePixels;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(i / kTilePixels), p = (uint32_t)(i % kTilePixels);
        const uint32_t tile = rank + k * world;
        const uint32_t x = (tile % tiles_x) * pt::kTile + (p % pt::kTile), y = (tile / tiles_x) * pt::kTile + (p / pt::kTile);
        const bool in = x < w && y < h;
        out_sum[i] = in ? sum[(size_t)y * w + x] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        out_counts[i] = in ? counts[(size_t)y * w + x] : 0u;
    }
}
__global__ __launch_bounds__(256) void unpack_tiles_kernel(const float4* __restrict__ in_sum, const uint32_t* __restrict__ in_counts,
                                                           uint32_t w, uint32_t h, uint32_t tiles_x, uint32_t rank, uint32_t world,
                                                           uint32_t owned, float4* __restrict__ sum, uint32_t* __restrict__ counts) {
    const size_t total = (size_t)owned * kTilePixels;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(i / kTilePixels), p = (uint32_t)(i % kTilePixels);
        const uint32_t tile = rank + k * world;
        const uint32_t x = (tile % tiles_x) * pt::kTile + (p % pt::kTile), y = (tile / tiles_x) * pt::kTile + (p / pt::kTile);
        if (x < w && y < h) {
            sum[(size_t)y * w + x] = in_sum[i];
            counts[(size_t)y * w + x] = in_counts[i];
        }
    }
}

uint32_t tile_count(uint32_t w, uint32_t h) { return ((w + pt::kTile - 1) / pt::kTile) * ((h + pt::kTile - 1) / pt::kTile); }
size_t payload_bytes(uint32_t w, uint32_t h, uint32_t rank, uint32_t world) {
    return (size_t)pt::owned_tiles(tile_count(w, h), rank, world) * kTilePixels * (sizeof(float4) + sizeof(uint32_t));
}
unsigned grid_for(size_t items) {
    const size_t blocks = (items + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

hipError_t launch_pack(const float4* sum, const uint32_t* counts, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, void* payload,
                       hipStream_t stream) {
    const uint32_t owned = pt::owned_tiles(tile_count(w, h), rank, world);
    if (owned == 0) return hipSuccess;
    float4* ps = (float4*)payload;
    uint32_t* pc = (uint32_t*)(ps + (size_t)owned * kTilePixels);
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(grid_for((size_t)owned * kTilePixels)), dim3(256), 0, stream, sum, counts, w, h,
                       (w + pt::kTile - 1) / pt::kTile, rank, world, owned, ps, pc);
    return hipGetLastError();
}
hipError_t launch_unpack(const void* payload, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, float4* sum, uint32_t* counts,
                         hipStream_t stream) {
    const uint32_t owned = pt::owned_tiles(tile_count(w, h), rank, world);
    if (owned == 0) return hipSuccess;
    const float4* ps = (const float4*)payload;
    const uint32_t* pc = (const uint32_t*)(ps + (size_t)owned * kTilePixels);
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3(grid_for((size_t)owned * kTilePixels)), dim3(256), 0, stream, ps, pc, w, h,
                       (w + pt::kTile - 1) / pt::kTile, rank, world, owned, sum, counts);
    return hipGetLastError();
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1;
    hipStream_t stream = nullptr;                                // the exchange runs here, beside the render stream
    hipEvent_t ev_snap = nullptr, ev_start = nullptr, ev_done = nullptr;
    DeviceBuf staging;    // snapshot of this rank's partial film (sample mode) or its packed owned tiles (tile mode)
    DeviceBuf gathered;   // root, tile mode: the other ranks' payloads, rank-major
    DeviceBuf comb_sum, comb_counts;  // root: the combined film
    bool pending = false;  // an exchange was enqueued and ev_done recorded
};

namespace {

// grow-only buffer the exchange stream may still be using: wait for that stream before freeing
int ensure_comm(ir_context* ctx, Comm* c, DeviceBuf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return IR_OK;
    if (b.p) IR_HIP(ctx, hipStreamSynchronize(c->stream));
    return ensure(ctx, b, bytes);
}

int exchange(ir_context* ctx, uint32_t root) {
    Comm* c = ctx->comm;
    const Rccl* r = rccl();
    if (!c || !r) return fail(ctx, IR_ESTATE, "ir_combine: no communicator (call ir_comm_init first)");
    if (root >= c->world) return fail(ctx, IR_EINVAL, "ir_combine: root %u not in [0,%u)", root, c->world);
    const uint32_t w = ctx->width, h = ctx->height;
    const size_t n = (size_t)w * h;
    if (n == 0) return fail(ctx, IR_ESTATE, "ir_combine: nothing rendered yet");
    int mode = ctx->part_mode;
    if (mode == IR_PARTITION_NONE) {
        if (c->world > 1) return fail(ctx, IR_ESTATE, "ir_combine: %u ranks but no partition set (ir_set_partition)", c->world);
        mode = IR_PARTITION_SAMPLES;  // a single rank: the combined film is a copy of its film
    } else if (ctx->rank != c->rank || ctx->world != c->world) {
        return fail(ctx, IR_ESTATE, "ir_combine: partition is rank %u of %u but the communicator is rank %u of %u", ctx->rank, ctx->world,
                    c->rank, c->world);
    }
    IR_TRY(use_device(ctx));
    const bool is_root = c->rank == root;
    // ---- snapshot on the render stream (after the renders enqueued so far, before the next one)
    const size_t my_bytes = mode == IR_PARTITION_TILES ? payload_bytes(w, h, c->rank, c->world) : n * 20;
    IR_TRY(ensure_comm(ctx, c, c->staging, my_bytes));
    if (is_root) {
        IR_TRY(ensure_comm(ctx, c, c->comb_sum, n * sizeof(float4)));
        IR_TRY(ensure_comm(ctx, c, c->comb_counts, n * sizeof(uint32_t)));
        if (mode == IR_PARTITION_TILES) {
            size_t all = 0;
            for (uint32_t k = 0; k < c->world; ++k)
                if (k != root) all += payload_bytes(w, h, k, c->world);
            IR_TRY(ensure_comm(ctx, c, c->gathered, all));
        }
    }
    // the previous exchange may still be reading the staging buffer
    if (c->pending) IR_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
    if (mode == IR_PARTITION_TILES) {
        IR_HIP(ctx, launch_pack(film_sum(ctx), film_counts(ctx), w, h, c->rank, c->world, c->staging.p, ctx->stream));
    } else {
        IR_HIP(ctx, hipMemcpyAsync(c->staging.p, film_sum(ctx), n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
        IR_HIP(ctx, hipMemcpyAsync((char*)c->staging.p + n * sizeof(float4), film_counts(ctx), n * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                   ctx->stream));
    }
    IR_HIP(ctx, hipEventRecord(c->ev_snap, ctx->stream));
    IR_HIP(ctx, hipStreamWaitEvent(c->stream, c->ev_snap, 0));
    IR_HIP(ctx, hipEventRecord(c->ev_start, c->stream));
    // ---- the exchange, on its own stream
    if (mode == IR_PARTITION_TILES) {
        if (c->world > 1) {
            IR_NCCL(ctx, r->GroupStart());
            ncclResult_t res = ncclSuccess;
            if (is_root) {
                size_t off = 0;
                for (uint32_t k = 0; k < c->world && res == ncclSuccess; ++k) {
                    if (k == root) continue;
                    const size_t b = payload_bytes(w, h, k, c->world);
                    if (b) res = r->Recv((char*)c->gathered.p + off, b / 4, ncclFloat32, (int)k, c->comm, c->stream);
                    off += b;
                }
            } else if (my_bytes) {
                res = r->Send(c->staging.p, my_bytes / 4, ncclFloat32, (int)root, c->comm, c->stream);
            }
            const ncclResult_t end = r->GroupEnd();
            if (res != ncclSuccess) return fail(ctx, IR_EDEVICE, "ir_combine: ncclSend/ncclRecv: %s", r->GetErrorString(res));
            if (end != ncclSuccess) return fail(ctx, IR_EDEVICE, "ir_combine: ncclGroupEnd: %s", r->GetErrorString(end));
        }
        if (is_root) {
            size_t off = 0;
            for (uint32_t k = 0; k < c->world; ++k) {
                const void* src = (k == root) ? c->staging.p : (const void*)((char*)c->gathered.p + off);
                IR_HIP(ctx, launch_unpack(src, w, h, k, c->world, (float4*)c->comb_sum.p, (uint32_t*)c->comb_counts.p, c->stream));
                if (k != root) off += payload_bytes(w, h, k, c->world);
            }
        }
    } else {
        float* s_sum = (float*)c->staging.p;
        uint32_t* s_cnt = (uint32_t*)((char*)c->staging.p + n * sizeof(float4));
        // recvbuff is only read on the root; the others pass their own staging buffer (never written)
        void* r_sum = is_root ? c->comb_sum.p : (void*)s_sum;
        void* r_cnt = is_root ? c->comb_counts.p : (void*)s_cnt;
        IR_NCCL(ctx, r->GroupStart());
        const ncclResult_t r1 = r->Reduce(s_sum, r_sum, n * 4, ncclFloat32, ncclSum, (int)root, c->comm, c->stream);
        const ncclResult_t r2 = r1 == ncclSuccess ? r->Reduce(s_cnt, r_cnt, n, ncclUint32, ncclSum, (int)root, c->comm, c->stream) : r1;
        const ncclResult_t end = r->GroupEnd();
        if (r2 != ncclSuccess) return fail(ctx, IR_EDEVICE, "ir_combine: ncclReduce: %s", r->GetErrorString(r2));
        if (end != ncclSuccess) return fail(ctx, IR_EDEVICE, "ir_combine: ncclGroupEnd: %s", r->GetErrorString(end));
    }
    IR_HIP(ctx, hipEventRecord(c->ev_done, c->stream));
    c->pending = true;
    ctx->output_combined = is_root;
    return IR_OK;
}

}  // namespace

void comm_release(ir_context* ctx) {
    Comm* c = ctx->comm;
    if (!c) return;
    ctx->comm = nullptr;
    ctx->output_combined = false;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    const Rccl* r = rccl();
    if (c->comm && r) (void)r->CommDestroy(c->comm);
    release(c->staging); release(c->gathered); release(c->comb_sum); release(c->comb_counts);
    if (c->ev_snap) (void)hipEventDestroy(c->ev_snap);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int output_film(ir_context* ctx, const float4** sum, const uint32_t** counts) {
    Comm* c = ctx->comm;
    if (ctx->output_combined && c && c->comb_sum.p) {
        if (c->pending) IR_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
        *sum = (const float4*)c->comb_sum.p;
        *counts = (const uint32_t*)c->comb_counts.p;
    } else {
        *sum = film_sum(ctx);
        *counts = film_counts(ctx);
    }
    return IR_OK;
}

}  // namespace irx

using namespace irx;

extern "C" {

int ir_comm_unique_id(uint8_t out_id[IR_COMM_ID_BYTES]) {
    if (!out_id) return fail(nullptr, IR_EINVAL, "ir_comm_unique_id: NULL");
    const Rccl* r = rccl();
    if (!r) return fail(nullptr, IR_EDEVICE, "ir_comm_unique_id: %s", g_rccl.error.c_str());
    ncclUniqueId id;
    const ncclResult_t res = r->GetUniqueId(&id);
    if (res != ncclSuccess) return fail(nullptr, IR_EDEVICE, "ncclGetUniqueId: %s", r->GetErrorString(res));
    memcpy(out_id, &id, sizeof(id));
    return IR_OK;
}

int ir_comm_init(ir_context* ctx, const uint8_t id_bytes[IR_COMM_ID_BYTES], uint32_t rank, uint32_t world) {
    if (!ctx) return IR_EINVAL;
    if (!id_bytes) return fail(ctx, IR_EINVAL, "ir_comm_init: NULL id");
    if (world == 0 || rank >= world) return fail(ctx, IR_EINVAL, "ir_comm_init: rank %u not in [0,%u)", rank, world);
    if (ctx->comm) return fail(ctx, IR_ESTATE, "ir_comm_init: the context already has a communicator (ir_comm_destroy first)");
    const Rccl* r = rccl();
    if (!r) return fail(ctx, IR_EDEVICE, "ir_comm_init: %s", g_rccl.error.c_str());
    IR_TRY(use_device(ctx));
    Comm* c = new (std::nothrow) Comm();
    if (!c) return fail(ctx, IR_ENOMEM, "ir_comm_init: out of host memory");
    ctx->comm = c;  // from here on comm_release cleans up
    c->rank = rank;
    c->world = world;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess || (e = hipEventCreate(&c->ev_snap)) != hipSuccess ||
        (e = hipEventCreate(&c->ev_start)) != hipSuccess || (e = hipEventCreate(&c->ev_done)) != hipSuccess) {
        comm_release(ctx);
        return fail(ctx, IR_EDEVICE, "ir_comm_init: stream/event setup: %s", hipGetErrorString(e));
    }
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    const ncclResult_t res = r->CommInitRank(&c->comm, (int)world, id, (int)rank);  // collective: returns when every rank joined
    if (res != ncclSuccess) {
        c->comm = nullptr;
        comm_release(ctx);
        return fail(ctx, IR_EDEVICE, "ncclCommInitRank(rank %u of %u on device %d): %s", rank, world, ctx->device, r->GetErrorString(res));
    }
    return IR_OK;
}

int ir_comm_destroy(ir_context* ctx) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(use_device(ctx));
    comm_release(ctx);
    return IR_OK;
}

int ir_combine_async(ir_context* ctx, uint32_t root) {
    if (!ctx) return IR_EINVAL;
    return exchange(ctx, root);
}

int ir_combine_wait(ir_context* ctx) {
    if (!ctx) return IR_EINVAL;
    Comm* c = ctx->comm;
    if (!c) return fail(ctx, IR_ESTATE, "ir_combine_wait: no communicator");
    if (!c->pending) return IR_OK;
    IR_TRY(use_device(ctx));
    IR_HIP(ctx, hipEventSynchronize(c->ev_done));
    ncclResult_t async = ncclSuccess;
    const Rccl* r = rccl();
    if (r && c->comm && r->CommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess)
        return fail(ctx, IR_EDEVICE, "ir_combine: asynchronous RCCL error: %s", r->GetErrorString(async));
    return IR_OK;
}

int ir_combine(ir_context* ctx, uint32_t root) {
    if (!ctx) return IR_EINVAL;
    IR_TRY(exchange(ctx, root));
    return ir_combine_wait(ctx);
}

double ir_last_combine_ms(const ir_context* ctx) {
    if (!ctx || !ctx->comm || !ctx->comm->pending) return 0.0;
    if (hipSetDevice(ctx->device) != hipSuccess) return 0.0;
    if (hipEventSynchronize(ctx->comm->ev_done) != hipSuccess) return 0.0;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->comm->ev_start, ctx->comm->ev_done) != hipSuccess) return 0.0;
    return (double)ms;
}

size_t ir_tile_payload_bytes(uint32_t width, uint32_t height, uint32_t rank, uint32_t world) {
    if (world == 0 || rank >= world) return 0;
    return payload_bytes(width, height, rank, world);
}

int ir_pack_owned_tiles(ir_context* ctx, void* d_payload, size_t bytes) {
    if (!ctx) return IR_EINVAL;
    const bool by_tiles = ctx->part_mode == IR_PARTITION_TILES;
    const uint32_t rank = by_tiles ? ctx->rank : 0, world = by_tiles ? ctx->world : 1;
    const size_t need = payload_bytes(ctx->width, ctx->height, rank, world);
    if (bytes != need) return fail(ctx, IR_EINVAL, "ir_pack_owned_tiles: payload is %zu bytes, this rank's tiles need %zu", bytes, need);
    if (need == 0) return IR_OK;
    if (!d_payload || ((uintptr_t)d_payload & 15u)) return fail(ctx, IR_EINVAL, "ir_pack_owned_tiles: NULL or misaligned payload");
    IR_TRY(use_device(ctx));
    IR_HIP(ctx, launch_pack(film_sum(ctx), film_counts(ctx), ctx->width, ctx->height, rank, world, d_payload, ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

int ir_unpack_tiles(ir_context* ctx, const void* d_payload, size_t bytes, uint32_t rank, uint32_t world) {
    if (!ctx) return IR_EINVAL;
    if (world == 0 || rank >= world) return fail(ctx, IR_EINVAL, "ir_unpack_tiles: rank %u not in [0,%u)", rank, world);
    const size_t need = payload_bytes(ctx->width, ctx->height, rank, world);
    if (bytes != need) return fail(ctx, IR_EINVAL, "ir_unpack_tiles: payload is %zu bytes, rank %u of %u owns %zu", bytes, rank, world, need);
    if (need == 0) return IR_OK;
    if (!d_payload || ((uintptr_t)d_payload & 15u)) return fail(ctx, IR_EINVAL, "ir_unpack_tiles: NULL or misaligned payload");
    IR_TRY(use_device(ctx));
    ctx->output_combined = false;
    IR_HIP(ctx, launch_unpack(d_payload, ctx->width, ctx->height, rank, world, film_sum(ctx), film_counts(ctx), ctx->stream));
    IR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IR_OK;
}

}  // extern "C"
