// fake_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for the ten RCCL entry points csrc/insploray_comm.hip binds, so that
// the N-rank branches of the exchange step (the root's grouped ncclRecv per peer, the payload offsets per rank, the
// unpack loop, the two-datatype grouped ncclReduce) can EXECUTE on a box with one GPU.  RCCL itself refuses two ranks on
// one device; here the ranks are ordinary processes that share the device and move bytes through a POSIX shared-memory
// segment with host staging.  Selected with IR_DEBUG_RCCL_LIB=<path to libfake_rccl.so> (tests only, like IR_DEBUG_LIB);
// it proves nothing about xGMI or about RCCL's own behaviour — only that the library's calls, buffers, counts and
// offsets are the ones a conforming transport needs to produce the right film (tests/test_gpu_comm_ranks.py).
//
// Semantics kept from the real thing: calls between ncclGroupStart / ncclGroupEnd are queued and issued at GroupEnd;
// every operation is ordered after the work already enqueued on the stream it names (the fake synchronises that
// stream) and is complete when the call that issues it returns (stronger than RCCL's stream ordering, so the caller's
// later stream work sees the data either way).  Reductions add in rank order.
//
// No HIP headers: hipMemcpy / hipStreamSynchronize are looked up in the process (whichever runtime copy it loaded).
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

using hipMemcpy_t = int (*)(void*, const void*, size_t, int);
using hipStreamSynchronize_t = int (*)(void*);
hipMemcpy_t p_hipMemcpy = nullptr;
hipStreamSynchronize_t p_hipStreamSynchronize = nullptr;
constexpr int kHostToDevice = 1, kDeviceToHost = 2;  // hipMemcpyKind

bool bind_hip() {
    if (p_hipMemcpy && p_hipStreamSynchronize) return true;
    void* h = nullptr;
    for (const char* n : {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"}) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    p_hipMemcpy = (hipMemcpy_t)dlsym(h ? h : RTLD_DEFAULT, "hipMemcpy");
    p_hipStreamSynchronize = (hipStreamSynchronize_t)dlsym(h ? h : RTLD_DEFAULT, "hipStreamSynchronize");
    return p_hipMemcpy && p_hipStreamSynchronize;
}

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 48u << 20;  // per rank: what one operation may carry

struct Shared {
    std::atomic<uint32_t> joined;
    // mailbox[src]: one message in flight per source rank
    std::atomic<uint64_t> sent[kMaxRanks];      // messages src has published
    std::atomic<uint64_t> consumed[kMaxRanks];  // messages of src that their receiver has taken
    std::atomic<uint32_t> dest[kMaxRanks];
    std::atomic<uint64_t> bytes[kMaxRanks];
    // reductions: contribution slots double as mailboxes' payload area (an operation is either one or the other)
    std::atomic<uint64_t> red_ready[kMaxRanks];  // reductions rank r has contributed to
    std::atomic<uint64_t> red_done;              // reductions the root has finished
};

struct Comm {
    int rank = 0, world = 1;
    Shared* sh = nullptr;
    char* slots = nullptr;  // world x kSlotBytes behind the header
    size_t map_bytes = 0;
    uint64_t my_sent = 0, red_count = 0;
    uint64_t taken[kMaxRanks] = {0};
    char name[64] = {0};
};

enum OpKind { kSend, kRecv, kReduce };
struct Op {
    OpKind kind;
    const void* send;
    void* recv;
    size_t count;
    int dtype, peer_or_root;
    Comm* comm;
    void* stream;
};
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_queue;

enum { kSuccess = 0, kUnhandled = 1, kSystem = 2, kInternal = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

size_t dtype_bytes(int t) {  // ncclDataType_t
    switch (t) {
        case 0: case 1: return 1;        // int8, uint8
        case 2: case 3: case 7: return 4;  // int32, uint32, float32
        case 4: case 5: case 8: return 8;  // int64, uint64, float64
        case 6: return 2;                // float16
        default: return 0;
    }
}

bool wait_until(const std::atomic<uint64_t>& a, uint64_t at_least) {
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t spin = 0;; ++spin) {
        if (a.load(std::memory_order_acquire) >= at_least) return true;
        if ((spin & 1023u) == 1023u) {
            sched_yield();
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (t1.tv_sec - t0.tv_sec > 60) return false;  // a peer died: fail instead of hanging the test
        }
    }
}

int run(const Op& op) {
    Comm* c = op.comm;
    const size_t bytes = op.count * dtype_bytes(op.dtype);
    if (!bind_hip()) return kSystem;
    if (bytes > kSlotBytes || dtype_bytes(op.dtype) == 0) return kInvalidArgument;
    if (p_hipStreamSynchronize(op.stream) != 0) return kUnhandled;
    Shared* sh = c->sh;
    if (op.kind == kSend) {
        // my slot is free once the receiver took my previous message
        if (!wait_until(sh->consumed[c->rank], c->my_sent)) return kSystem;
        if (p_hipMemcpy(c->slots + (size_t)c->rank * kSlotBytes, op.send, bytes, kDeviceToHost) != 0) return kUnhandled;
        sh->dest[c->rank].store((uint32_t)op.peer_or_root);
        sh->bytes[c->rank].store(bytes);
        c->my_sent++;
        sh->sent[c->rank].store(c->my_sent, std::memory_order_release);
        return kSuccess;
    }
    if (op.kind == kRecv) {
        const int src = op.peer_or_root;
        if (!wait_until(sh->sent[src], c->taken[src] + 1)) return kSystem;
        if ((int)sh->dest[src].load() != c->rank || sh->bytes[src].load() != bytes) return kInvalidUsage;  // mismatched send / recv
        if (p_hipMemcpy(op.recv, c->slots + (size_t)src * kSlotBytes, bytes, kHostToDevice) != 0) return kUnhandled;
        c->taken[src]++;
        sh->consumed[src].store(c->taken[src], std::memory_order_release);
        return kSuccess;
    }
    // reduce(sum) to a root: every rank publishes its contribution, the root adds them in rank order
    const int root = op.peer_or_root;
    c->red_count++;
    if (!wait_until(sh->red_done, c->red_count - 1)) return kSystem;  // the slots of the previous reduction are free
    if (p_hipMemcpy(c->slots + (size_t)c->rank * kSlotBytes, op.send, bytes, kDeviceToHost) != 0) return kUnhandled;
    sh->red_ready[c->rank].store(c->red_count, std::memory_order_release);
    if (c->rank == root) {
        for (int r = 0; r < c->world; ++r)
            if (!wait_until(sh->red_ready[r], c->red_count)) return kSystem;
        std::vector<char> acc(bytes);
        memcpy(acc.data(), c->slots, bytes);
        for (int r = 1; r < c->world; ++r) {
            const char* src = c->slots + (size_t)r * kSlotBytes;
            if (op.dtype == 7) {
                float* a = (float*)acc.data();
                const float* b = (const float*)src;
                for (size_t i = 0; i < op.count; ++i) a[i] = a[i] + b[i];
            } else if (op.dtype == 3 || op.dtype == 2) {
                uint32_t* a = (uint32_t*)acc.data();
                const uint32_t* b = (const uint32_t*)src;
                for (size_t i = 0; i < op.count; ++i) a[i] += b[i];
            } else {
                return kInvalidArgument;
            }
        }
        if (p_hipMemcpy(op.recv, acc.data(), bytes, kHostToDevice) != 0) return kUnhandled;
        sh->red_done.store(c->red_count, std::memory_order_release);
    }
    return kSuccess;
}

int submit(const Op& op) {
    if (g_group_depth > 0) {
        g_queue.push_back(op);
        return kSuccess;
    }
    return run(op);
}

}  // namespace

extern "C" {

struct FakeUniqueId {
    char internal[128];
};

int ncclGetUniqueId(FakeUniqueId* id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof(*id));
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%lld_%ld", (int)getpid(), (long long)t.tv_sec, t.tv_nsec);
    return kSuccess;
}

int ncclCommInitRank(void** comm, int nranks, FakeUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return kInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = nranks;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = 4096 + (size_t)nranks * kSlotBytes;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);  // a fresh segment is zero-filled: every counter starts at 0
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
        if (fd >= 0) close(fd);
        delete c;
        return kSystem;
    }
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return kSystem;
    }
    static_assert(sizeof(Shared) <= 4096, "header page");
    c->sh = (Shared*)p;
    c->slots = (char*)p + 4096;
    // collective: returns when every rank has joined (as ncclCommInitRank does)
    c->sh->joined.fetch_add(1);
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while ((int)c->sh->joined.load() < nranks) {
        sched_yield();
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (t1.tv_sec - t0.tv_sec > 60) {
            munmap(p, c->map_bytes);
            if (rank == 0) shm_unlink(c->name);
            delete c;
            return kSystem;
        }
    }
    *comm = c;
    return kSuccess;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return kInvalidArgument;
    munmap((void*)c->sh, c->map_bytes);
    if (c->rank == 0) shm_unlink(c->name);  // the others keep their mappings until they unmap
    delete c;
    return kSuccess;
}

int ncclCommGetAsyncError(void* comm, int* async_error) {
    if (!comm || !async_error) return kInvalidArgument;
    *async_error = kSuccess;
    return kSuccess;
}

const char* ncclGetErrorString(int r) {
    switch (r) {
        case kSuccess: return "no error";
        case kUnhandled: return "fake rccl: a HIP call failed";
        case kSystem: return "fake rccl: system error or a peer that never arrived (60 s)";
        case kInvalidArgument: return "fake rccl: invalid argument";
        case kInvalidUsage: return "fake rccl: send and recv do not match";
        default: return "fake rccl: internal error";
    }
}

int ncclGroupStart(void) {
    g_group_depth++;
    return kSuccess;
}

int ncclGroupEnd(void) {
    if (g_group_depth <= 0) return kInvalidUsage;
    if (--g_group_depth > 0) return kSuccess;
    std::vector<Op> q;
    q.swap(g_queue);
    // sends first (they never wait for a receiver of THIS group's messages), then receives, then reductions in call order
    for (const Op& op : q)
        if (op.kind == kSend) {
            const int r = run(op);
            if (r != kSuccess) return r;
        }
    for (const Op& op : q)
        if (op.kind != kSend) {
            const int r = run(op);
            if (r != kSuccess) return r;
        }
    return kSuccess;
}

int ncclReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, int root, void* comm, void* stream) {
    if (!comm || op != 0 /* ncclSum */) return kInvalidArgument;
    return submit(Op{kReduce, sendbuff, recvbuff, count, datatype, root, (Comm*)comm, stream});
}

int ncclSend(const void* sendbuff, size_t count, int datatype, int peer, void* comm, void* stream) {
    if (!comm) return kInvalidArgument;
    return submit(Op{kSend, sendbuff, nullptr, count, datatype, peer, (Comm*)comm, stream});
}

int ncclRecv(void* recvbuff, size_t count, int datatype, int peer, void* comm, void* stream) {
    if (!comm) return kInvalidArgument;
    return submit(Op{kRecv, nullptr, recvbuff, count, datatype, peer, (Comm*)comm, stream});
}

}  // extern "C"
