// Multi-GPU exchange inside the library: one process per GPU, one RCCL communicator per context, collectives enqueued
// on the context's own HIP stream (no host synchronisation in the data path).  RCCL is loaded at run time
// (dlopen "librccl.so.1": the copy already in the process if the host program, e.g. PyTorch, brought one), so a
// single-GPU user never needs it.  For ranks that SHARE one GPU (tests on a one-GPU box; RCCL refuses two ranks on
// one device) a host transport can be plugged in instead: the library stages through host memory and calls back.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace fhs {

typedef int (*HostAllGatherFn)(void *user, const void *send, void *recv, size_t bytes_per_rank);

class Dist {
  public:
    int rank = 0, world = 1;
    bool active() const { return comm_ != nullptr || host_fn_ != nullptr; }
    bool stream_ordered() const { return comm_ != nullptr; }

    // librccl.so.1 loadable with every entry point used here (dlopen + dlsym only: no communicator, no GPU call)
    static bool available(std::string &err);
    static int unique_id(void *out128, std::string &err);
    int init_rccl(int rank, int world, const void *id128, std::string &err);   // the caller has made the device current
    int init_host(int rank, int world, HostAllGatherFn fn, void *user, std::string &err);
    // d_send: bytes_per_rank bytes, d_recv: world * bytes_per_rank bytes (rank-major).  RCCL: enqueued on `s`;
    // host transport: synchronises `s`, stages through pinned host memory, calls back, uploads.
    int all_gather(const void *d_send, void *d_recv, size_t bytes_per_rank, hipStream_t s, std::string &err);
    void shutdown(bool abort = false);
    // exchange counters since init: calls of all_gather with a non-empty payload, bytes this rank contributed
    unsigned long long n_gathers = 0, bytes_sent = 0;

  private:
    void *comm_ = nullptr;            // ncclComm_t
    HostAllGatherFn host_fn_ = nullptr;
    void *host_user_ = nullptr;
    void *h_send_ = nullptr, *h_recv_ = nullptr;
    size_t h_cap_ = 0;
};

}  // namespace fhs
