#include "dist.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace fhs {

namespace {
// the few RCCL entry points used, by their stable C ABI (rccl.h: ncclUniqueId is 128 opaque bytes passed by value,
// ncclResult_t 0 = success, ncclDataType_t ncclUint8 = 1)
struct UniqueId { char internal[128]; };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, UniqueId id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*CommAbort)(void *comm) = nullptr;               // optional
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string load_error;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.load_error = std::string("cannot load librccl.so.1: ") + dlerror();
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString)
            r.load_error = "librccl.so.1 lacks an expected symbol";
    });
    return r;
}
std::string rccl_err(const char *what, int rc) {
    Rccl &r = rccl();
    return std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
}
}  // namespace

bool Dist::available(std::string &err) {
    Rccl &r = rccl();
    if (!r.load_error.empty()) { err = r.load_error; return false; }
    return true;
}

int Dist::unique_id(void *out128, std::string &err) {
    Rccl &r = rccl();
    if (!r.load_error.empty()) { err = r.load_error; return -3; }
    UniqueId id;
    if (int rc = r.GetUniqueId(&id)) { err = rccl_err("ncclGetUniqueId", rc); return -2; }
    std::memcpy(out128, &id, sizeof(id));
    return 0;
}

int Dist::init_rccl(int rk, int wd, const void *id128, std::string &err) {
    if (active()) { err = "distributed transport already initialised"; return -3; }
    Rccl &r = rccl();
    if (!r.load_error.empty()) { err = r.load_error; return -3; }
    UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    void *c = nullptr;
    if (int rc = r.CommInitRank(&c, wd, id, rk)) { err = rccl_err("ncclCommInitRank", rc); return -2; }
    comm_ = c;
    rank = rk;
    world = wd;
    return 0;
}

int Dist::init_host(int rk, int wd, HostAllGatherFn fn, void *user, std::string &err) {
    if (active()) { err = "distributed transport already initialised"; return -3; }
    host_fn_ = fn;
    host_user_ = user;
    rank = rk;
    world = wd;
    return 0;
}

int Dist::all_gather(const void *d_send, void *d_recv, size_t bytes, hipStream_t s, std::string &err) {
    if (bytes == 0) return 0;
    n_gathers++;
    bytes_sent += bytes;
    if (comm_) {
        if (int rc = rccl().AllGather(d_send, d_recv, bytes, /*ncclUint8*/ 1, comm_, s)) {
            err = rccl_err("ncclAllGather", rc);
            return -2;
        }
        return 0;
    }
    if (!host_fn_) { err = "no distributed transport (call fhs_dist_init first)"; return -3; }
    const size_t total = bytes * (size_t)world;
    if (h_cap_ < total) {
        if (h_send_) (void)hipHostFree(h_send_);
        if (h_recv_) (void)hipHostFree(h_recv_);
        h_send_ = h_recv_ = nullptr;
        h_cap_ = 0;
        if (hipHostMalloc(&h_send_, total) != hipSuccess || hipHostMalloc(&h_recv_, total) != hipSuccess) {
            err = "hipHostMalloc (exchange staging) failed";
            return -2;
        }
        h_cap_ = total;
    }
    hipError_t e = hipMemcpyAsync(h_send_, d_send, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { err = std::string("exchange download: ") + hipGetErrorString(e); return -2; }
    if (int rc = host_fn_(host_user_, h_send_, h_recv_, bytes)) {
        err = "host all-gather callback failed (" + std::to_string(rc) + ")";
        return -2;
    }
    e = hipMemcpyAsync(d_recv, h_recv_, total, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // the staging buffer is reused by the next exchange
    if (e != hipSuccess) { err = std::string("exchange upload: ") + hipGetErrorString(e); return -2; }
    return 0;
}

void Dist::shutdown(bool abort) {
    // ncclCommDestroy is collective-ish (it flushes outstanding work and may wait for the peers); after a bring-up that
    // failed on SOME ranks the communicator is half-formed and a destroy can block for good: there the
    // communicator is aborted -- ncclCommAbort frees it without talking to anyone -- or, without that entry point, left
    // to the process exit.
    if (comm_ && !abort) (void)rccl().CommDestroy(comm_);
    else if (comm_ && rccl().CommAbort) (void)rccl().CommAbort(comm_);
    comm_ = nullptr;
    host_fn_ = nullptr;
    if (h_send_) (void)hipHostFree(h_send_);
    if (h_recv_) (void)hipHostFree(h_recv_);
    h_send_ = h_recv_ = nullptr;
    h_cap_ = 0;
    rank = 0;
    world = 1;
    n_gathers = bytes_sent = 0;
}

}  // namespace fhs
