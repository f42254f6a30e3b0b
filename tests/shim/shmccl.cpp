// shmccl - TEST INFRASTRUCTURE, not part of the product.
//
// A same-node collective library with the entry points libdeepmod_hip binds from librccl (deepmod_amd/csrc/deepmod_hip.hip load_rccl():
// ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclReduce, ncclReduceScatter, ncclGroupStart / ncclGroupEnd, ncclCommDestroy,
// ncclGetErrorString, ncclGetVersion), defined against <rccl/rccl.h> itself (round 6) and implemented over one POSIX shared-memory segment and host staging.  RCCL refuses a communicator whose ranks share
// a device ("Duplicate GPU detected"), so on a one-GPU box the product's N > 1 code - the slice arithmetic of dm_summary_reduce_scatter,
// the order of collective calls over the ranks, the rank-sliced BED - could never run through the C ABI.  With DEEPMOD_RCCL_LIBRARY
// pointing here it does: two, three or eight real processes on device 0, every collective call the product makes answered with RCCL's
// documented semantics.  What this does NOT show is RCCL itself over xGMI: that still needs a multi-GPU node.
//
// Every collective is host-synchronous: stream sync, device -> shared slot of this rank, barrier, reduce what this rank must receive,
// host -> device, barrier.  Chunks of SLOT_BYTES.  A rank that waits longer than TIMEOUT_S at a barrier returns ncclSystemError (the
// product turns that into its own error path) instead of hanging the box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // the SAME declarations the product binds against: a signature that drifts is a compile error here

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
constexpr size_t SLOT_BYTES = size_t(2) << 20;
constexpr double TIMEOUT_S = 60.0;
// result codes / data types / operations are rccl.h's enumerators under the names this file was written with
constexpr ncclResult_t OK = ncclSuccess, UNHANDLED_HIP = ncclUnhandledCudaError, SYSTEM_ERROR = ncclSystemError, INTERNAL_ERROR = ncclInternalError,
                       INVALID_ARGUMENT = ncclInvalidArgument, INVALID_USAGE = ncclInvalidUsage;
constexpr int T_INT32 = ncclInt32, T_FLOAT64 = ncclFloat64;
constexpr int OP_SUM = ncclSum, OP_MAX = ncclMax;

struct Header {
    std::atomic<uint32_t> ready;
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    uint32_t nranks;
};
constexpr size_t HEADER_BYTES = 4096;

struct Comm {
    int rank = 0, nranks = 1;
    unsigned char* base = nullptr;
    size_t bytes = 0;
    Header* hdr() const { return reinterpret_cast<Header*>(base); }
    unsigned char* slot(int r) const { return base + HEADER_BYTES + size_t(r) * SLOT_BYTES; }
};

double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}

ncclResult_t barrier(Comm* c) {
    Header* h = c->hdr();
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == uint32_t(c->nranks)) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_release);
        return OK;
    }
    const double t0 = now();
    while (h->generation.load(std::memory_order_acquire) == gen) {
        usleep(20);
        if (now() - t0 > TIMEOUT_S) return SYSTEM_ERROR;
    }
    return OK;
}

size_t elem_size(int dtype) { return dtype == T_INT32 ? 4 : dtype == T_FLOAT64 ? 8 : 0; }

template <typename T>
void reduce_into(T* out, const Comm* c, size_t first, size_t n, int op) {
    for (size_t i = 0; i < n; ++i) {
        T v = reinterpret_cast<const T*>(c->slot(0))[first + i];
        for (int r = 1; r < c->nranks; ++r) {
            const T u = reinterpret_cast<const T*>(c->slot(r))[first + i];
            v = op == OP_SUM ? T(v + u) : (u > v ? u : v);
        }
        out[i] = v;
    }
}

// The common form: `total` elements of every rank's send buffer are reduced; this rank receives the elements [want0, want0 + wantn) of the
// result into recv (element 0 of recv = element want0 of the result).  wantn == 0: this rank takes part and receives nothing.
ncclResult_t collective(Comm* c, const void* send, void* recv, size_t total, size_t want0, size_t wantn, int dtype, int op, hipStream_t stream) {
    const size_t es = elem_size(dtype);
    if (!c || !es || (op != OP_SUM && op != OP_MAX)) return INVALID_ARGUMENT;
    if (hipStreamSynchronize(stream) != hipSuccess) return UNHANDLED_HIP;
    const size_t per = SLOT_BYTES / es;
    unsigned char* tmp = static_cast<unsigned char*>(std::malloc(SLOT_BYTES));
    if (!tmp) return SYSTEM_ERROR;
    ncclResult_t rc = OK;
    for (size_t off = 0; off < total && rc == OK; off += per) {
        const size_t n = total - off < per ? total - off : per;
        if (hipMemcpy(c->slot(c->rank), static_cast<const unsigned char*>(send) + off * es, n * es, hipMemcpyDefault) != hipSuccess) rc = UNHANDLED_HIP;
        const ncclResult_t b0 = barrier(c);
        if (rc == OK) rc = b0;
        const size_t lo = want0 > off ? want0 : off, hi = want0 + wantn < off + n ? want0 + wantn : off + n;
        if (rc == OK && lo < hi) {
            if (dtype == T_INT32) reduce_into(reinterpret_cast<int32_t*>(tmp), c, lo - off, hi - lo, op);
            else reduce_into(reinterpret_cast<double*>(tmp), c, lo - off, hi - lo, op);
            if (hipMemcpy(static_cast<unsigned char*>(recv) + (lo - want0) * es, tmp, (hi - lo) * es, hipMemcpyDefault) != hipSuccess) rc = UNHANDLED_HIP;
        }
        const ncclResult_t b1 = barrier(c);        // nobody overwrites a slot another rank still reads
        if (rc == OK) rc = b1;
    }
    std::free(tmp);
    return rc;
}
}  // namespace

// The functions below DEFINE entry points rccl.h declares: same names, same parameter lists, same return type - the compiler checks each of them
// against the header's prototype.  ncclComm_t (a pointer to RCCL's opaque ncclComm) carries a pointer to this file's Comm.
static Comm* as_comm(ncclComm_t c) { return reinterpret_cast<Comm*>(c); }

extern "C" {

ncclResult_t ncclGetVersion(int* version) {
    if (!version) return INVALID_ARGUMENT;
    *version = NCCL_VERSION_CODE;                  // the header this stand-in was compiled against
    return OK;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return INVALID_ARGUMENT;
    static std::atomic<unsigned> serial{0};
    std::memset(id->internal, 0, sizeof id->internal);
    std::snprintf(id->internal, sizeof id->internal, "/deepmod_shmccl_%d_%u_%lld", int(getpid()), serial.fetch_add(1), (long long)(now() * 1e6));
    return OK;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || id.internal[0] != '/') return INVALID_ARGUMENT;
    const size_t bytes = HEADER_BYTES + size_t(nranks) * SLOT_BYTES;
    int fd = shm_open(id.internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    const bool creator = fd >= 0;
    if (creator) {
        if (ftruncate(fd, off_t(bytes)) != 0) { close(fd); shm_unlink(id.internal); return SYSTEM_ERROR; }
    } else {
        const double t0 = now();
        for (;;) {
            fd = shm_open(id.internal, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && size_t(st.st_size) == bytes) break;
            if (fd >= 0) close(fd);
            if (now() - t0 > TIMEOUT_S) return SYSTEM_ERROR;
            usleep(200);
        }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (creator) shm_unlink(id.internal); return SYSTEM_ERROR; }
    Comm* c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    c->base = static_cast<unsigned char*>(p);
    c->bytes = bytes;
    if (creator) {                                 // a fresh segment is zero-filled: counters start at 0
        c->hdr()->nranks = uint32_t(nranks);
        c->hdr()->ready.store(1, std::memory_order_release);
    } else {
        const double t0 = now();
        while (c->hdr()->ready.load(std::memory_order_acquire) != 1) {
            usleep(200);
            if (now() - t0 > TIMEOUT_S) { munmap(p, bytes); delete c; return SYSTEM_ERROR; }
        }
        if (c->hdr()->nranks != uint32_t(nranks)) { munmap(p, bytes); delete c; return INVALID_USAGE; }
    }
    const ncclResult_t rc = barrier(c);            // everybody has the segment mapped: the name can go
    if (creator) shm_unlink(id.internal);
    if (rc != OK) { munmap(p, bytes); delete c; return rc; }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return OK;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = as_comm(comm);
    if (!c) return INVALID_ARGUMENT;
    munmap(c->base, c->bytes);
    delete c;
    return OK;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    return collective(as_comm(comm), send, recv, count, 0, count, dtype, op, stream);
}

ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t dtype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
    Comm* c = as_comm(comm);
    if (!c || root < 0 || root >= c->nranks) return INVALID_ARGUMENT;
    return collective(c, send, recv, count, 0, c->rank == root ? count : 0, dtype, op, stream);
}

ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm* c = as_comm(comm);
    if (!c) return INVALID_ARGUMENT;
    return collective(c, send, recv, recvcount * size_t(c->nranks), recvcount * size_t(c->rank), recvcount, dtype, op, stream);
}

// every call above has completed when it returns, in the order all ranks issue them: a group is that order, nothing more
ncclResult_t ncclGroupStart(void) { return OK; }
ncclResult_t ncclGroupEnd(void) { return OK; }

const char* ncclGetErrorString(ncclResult_t e) {
    switch (e) {
        case OK: return "no error";
        case UNHANDLED_HIP: return "shmccl: HIP call failed";
        case SYSTEM_ERROR: return "shmccl: system error or a rank did not arrive within the timeout";
        case INVALID_ARGUMENT: return "shmccl: invalid argument";
        case INVALID_USAGE: return "shmccl: invalid usage";
        default: return "shmccl: internal error";
    }
}
}  // extern "C"
