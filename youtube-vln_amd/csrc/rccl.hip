// Direct RCCL binding behind the C ABI: the gradient exchange of the data-parallel path (replaces
// DistributedDataParallel over NCCL, utils/distributed.py:63-104).
//
// librccl is resolved at RUN time with dlopen/dlsym -- libytvln.so has no link dependency on it, single-GPU users never
// load it, and the copy that gets used is the one already mapped into the process (PyTorch's own librccl.so when torch is
// imported: same HIP runtime, so streams and device pointers are interchangeable).  Only the TYPES of
// <rccl/rccl.h> are used at compile time.
//
// A communicator handle is an opaque pointer owned by the caller; every collective is asynchronous on the stream it is
// given.  Nothing here allocates device memory or synchronises a stream.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string.h>

namespace ytvln {
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char path[512] = {0};
};

RcclApi g_api;
std::mutex g_mu;

template <class F>
bool sym(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

// Resolution order: an explicit path; a librccl already mapped into the process (RTLD_NOLOAD: PyTorch's); the loader's default.
int load_api(const char* path) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_api.handle) return 0;
    void* h = nullptr;
    const char* used = nullptr;
    if (path && path[0]) {
        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        used = path;
        if (!h) return fail(-3, "ytvln_rccl_load: dlopen(%s): %s", path, dlerror());
    } else {
        static const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (h) { used = n; break; }
        }
        if (!h) {
            for (const char* n : names) {
                h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (h) { used = n; break; }
            }
        }
        if (!h) return fail(-3, "ytvln_rccl_load: librccl.so not found (%s); pass its path explicitly", dlerror());
    }
    RcclApi a;
    a.handle = h;
    const bool ok = sym(h, "ncclGetVersion", a.GetVersion) && sym(h, "ncclGetUniqueId", a.GetUniqueId) &&
                    sym(h, "ncclCommInitRank", a.CommInitRank) && sym(h, "ncclCommDestroy", a.CommDestroy) &&
                    sym(h, "ncclCommAbort", a.CommAbort) && sym(h, "ncclCommGetAsyncError", a.CommGetAsyncError) &&
                    sym(h, "ncclAllReduce", a.AllReduce) && sym(h, "ncclBroadcast", a.Broadcast) &&
                    sym(h, "ncclGroupStart", a.GroupStart) && sym(h, "ncclGroupEnd", a.GroupEnd) &&
                    sym(h, "ncclGetErrorString", a.GetErrorString);
    if (!ok) {
        dlclose(h);
        return fail(-3, "ytvln_rccl_load: %s lacks an expected nccl* symbol", used);
    }
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(a.AllReduce), &info) && info.dli_fname) strncpy(a.path, info.dli_fname, sizeof(a.path) - 1);
    else strncpy(a.path, used, sizeof(a.path) - 1);
    g_api = a;
    return 0;
}

#define YT_NCCL(call, name)                                                                             \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess) return ::ytvln::fail(-4, "%s: %s", name, g_api.GetErrorString(r_));      \
    } while (0)

bool to_nccl_type(int dtype, ncclDataType_t* t, size_t* size) {
    switch (dtype) {
        case YTVLN_DT_F32: *t = ncclFloat32; *size = 4; return true;
        case YTVLN_DT_F64: *t = ncclFloat64; *size = 8; return true;
        case YTVLN_DT_BF16: *t = ncclBfloat16; *size = 2; return true;
        case YTVLN_DT_I64: *t = ncclInt64; *size = 8; return true;
        case YTVLN_DT_U8: *t = ncclUint8; *size = 1; return true;
    }
    return false;
}

}  // namespace
}  // namespace ytvln

using namespace ytvln;

extern "C" int ytvln_rccl_load(const char* path) { return load_api(path); }

extern "C" const char* ytvln_rccl_library_path(void) { return g_api.handle ? g_api.path : ""; }

extern "C" int ytvln_rccl_version(int* version) {
    YT_REQUIRE(version != nullptr, "ytvln_rccl_version: NULL output");
    if (int rc = load_api(nullptr)) return rc;
    YT_NCCL(g_api.GetVersion(version), "ncclGetVersion");
    return 0;
}

extern "C" int ytvln_rccl_unique_id(void* id_out, int64_t bytes) {
    YT_REQUIRE(id_out != nullptr && bytes == (int64_t)sizeof(ncclUniqueId), "ytvln_rccl_unique_id: need a %d-byte buffer",
               (int)sizeof(ncclUniqueId));
    if (int rc = load_api(nullptr)) return rc;
    ncclUniqueId id;
    YT_NCCL(g_api.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int ytvln_rccl_init(void** comm_out, const void* id, int64_t id_bytes, int rank, int world, int device) {
    YT_REQUIRE(comm_out != nullptr && id != nullptr && id_bytes == (int64_t)sizeof(ncclUniqueId), "ytvln_rccl_init: bad id buffer");
    YT_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ytvln_rccl_init: rank %d outside world %d", rank, world);
    if (int rc = load_api(nullptr)) return rc;
    if (device >= 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail(-2, "ytvln_rccl_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    YT_NCCL(g_api.CommInitRank(&comm, world, uid, rank), "ncclCommInitRank");
    *comm_out = comm;
    return 0;
}

extern "C" int ytvln_rccl_allreduce(void* comm, void* buf, int64_t count, int dtype, int op, void* stream) {
    YT_REQUIRE(comm != nullptr && g_api.handle != nullptr, "ytvln_rccl_allreduce: no communicator");
    YT_REQUIRE(count >= 0 && (buf != nullptr || count == 0), "ytvln_rccl_allreduce: bad buffer");
    ncclDataType_t t;
    size_t sz;
    YT_REQUIRE(to_nccl_type(dtype, &t, &sz), "ytvln_rccl_allreduce: unknown dtype %d", dtype);
    YT_REQUIRE(op == YTVLN_RED_SUM || op == YTVLN_RED_MAX || op == YTVLN_RED_MIN, "ytvln_rccl_allreduce: unknown op %d", op);
    if (count == 0) return 0;
    const ncclRedOp_t rop = op == YTVLN_RED_SUM ? ncclSum : op == YTVLN_RED_MAX ? ncclMax : ncclMin;
    YT_NCCL(g_api.AllReduce(buf, buf, (size_t)count, t, rop, (ncclComm_t)comm, as_stream(stream)), "ncclAllReduce");
    return 0;
}

// Several contiguous slices of one buffer as ONE RCCL group (one launch train, no host round trip between the slices).
extern "C" int ytvln_rccl_allreduce_slices(void* comm, void* base, int dtype, const int64_t* offsets, const int64_t* counts, int nslices,
                                           void* stream) {
    YT_REQUIRE(comm != nullptr && g_api.handle != nullptr, "ytvln_rccl_allreduce_slices: no communicator");
    YT_REQUIRE(nslices >= 0 && (nslices == 0 || (base && offsets && counts)), "ytvln_rccl_allreduce_slices: bad arguments");
    ncclDataType_t dt;
    size_t esz;
    YT_REQUIRE(to_nccl_type(dtype, &dt, &esz), "ytvln_rccl_allreduce_slices: unsupported dtype %d", dtype);
    if (nslices == 0) return 0;
    YT_NCCL(g_api.GroupStart(), "ncclGroupStart");
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < nslices; ++i) {
        if (counts[i] <= 0) continue;
        char* p = static_cast<char*>(base) + (size_t)offsets[i] * esz;
        ncclResult_t r = g_api.AllReduce(p, p, (size_t)counts[i], dt, ncclSum, (ncclComm_t)comm, as_stream(stream));
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    ncclResult_t e = g_api.GroupEnd();
    if (first != ncclSuccess) return fail(-4, "ncclAllReduce (grouped): %s", g_api.GetErrorString(first));
    if (e != ncclSuccess) return fail(-4, "ncclGroupEnd: %s", g_api.GetErrorString(e));
    return 0;
}
extern "C" int ytvln_rccl_allreduce_slices_f32(void* comm, float* base, const int64_t* offsets, const int64_t* counts, int nslices,
                                               void* stream) {
    return ytvln_rccl_allreduce_slices(comm, base, YTVLN_DT_F32, offsets, counts, nslices, stream);
}

extern "C" int ytvln_rccl_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream) {
    YT_REQUIRE(comm != nullptr && g_api.handle != nullptr, "ytvln_rccl_broadcast: no communicator");
    YT_REQUIRE(bytes >= 0 && (buf != nullptr || bytes == 0), "ytvln_rccl_broadcast: bad buffer");
    if (bytes == 0) return 0;
    YT_NCCL(g_api.Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, (ncclComm_t)comm, as_stream(stream)), "ncclBroadcast");
    return 0;
}

extern "C" int ytvln_rccl_async_error(void* comm) {
    YT_REQUIRE(comm != nullptr && g_api.handle != nullptr, "ytvln_rccl_async_error: no communicator");
    ncclResult_t st = ncclSuccess;
    YT_NCCL(g_api.CommGetAsyncError((ncclComm_t)comm, &st), "ncclCommGetAsyncError");
    if (st != ncclSuccess && st != ncclInProgress) return fail(-4, "RCCL asynchronous error: %s", g_api.GetErrorString(st));
    return 0;
}

extern "C" int ytvln_rccl_destroy(void* comm) {
    if (comm == nullptr) return 0;
    YT_REQUIRE(g_api.handle != nullptr, "ytvln_rccl_destroy: RCCL was never loaded");
    YT_NCCL(g_api.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
    return 0;
}
