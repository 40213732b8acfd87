// comm.h -- the collectives of the multi-GPU path behind one small interface.
//
// One process per GPU: landmarks and the Cholesky of S are sharded, the exchanges are a handful of
// all-reduces / reduces of device buffers.  Two transports implement them:
//   * RCCL over xGMI (make_rccl_comm)            -- production: ncclAllReduce / ncclReduce / ncclBroadcast on the solver's stream;
//   * host shared memory (make_shm_comm)         -- bring-up and tests: the ranks are processes of ONE node (they may even
//                                                   share one GPU), buffers are staged through a POSIX shared-memory
//                                                   segment and summed in rank order.  Slow by design, bitwise
//                                                   deterministic, and it lets EVERY world > 1 branch of Solver / TilePlan
//                                                   (who reduces what, which flags are max-reduced, error propagation,
//                                                   the sweep time-out agreement) run on a single-GPU box with the real
//                                                   kernels -- where RCCL cannot even be initialised with two ranks.
// All calls are collective, issued in the same order on every rank, and enqueue on / synchronise with `stream`.
#pragma once
#ifdef APEX_COMM_HOST_ONLY   // tests/comm_host_harness.cpp: the shm transport on HOST buffers, built with g++ (no HIP, no GPU)
typedef void* hipStream_t;
#else
#include <hip/hip_runtime.h>
#endif
#include <stddef.h>

#include <memory>
#include <string>

namespace apex {

class Communicator {
   public:
    virtual ~Communicator() = default;
    virtual const char* transport() const = 0;
    virtual bool all_reduce_sum(double* dev, size_t n, hipStream_t s) = 0;
    virtual bool all_reduce_max(int* dev, size_t n, hipStream_t s) = 0;
    virtual bool reduce_sum(double* dev, size_t n, int root, hipStream_t s) = 0;            // result valid on root only
    virtual bool all_gather(const void* dev_send, void* dev_recv, size_t bytes_per_rank, hipStream_t s) = 0;
    virtual bool broadcast(double* dev, size_t n, int root, hipStream_t s) = 0;
    virtual bool group_start() { return true; }    // RCCL: ncclGroupStart / End around a batch of calls
    virtual bool group_end() { return true; }
    const std::string& error() const { return err_; }
    int world() const { return world_; }
    int rank() const { return rank_; }

   protected:
    Communicator(int world, int rank) : world_(world), rank_(rank) {}
    bool fail(const std::string& m) { err_ = m; return false; }
    int world_, rank_;
    std::string err_;
};

// unique_id128: the 128 bytes of apexgpu_get_unique_id, the same on every rank.  nullptr + *err on failure.
std::unique_ptr<Communicator> make_rccl_comm(int world, int rank, const void* unique_id128, std::string* err);
// name: any string shared by the ranks of ONE run and by no other (e.g. "<pid of the launcher>-<counter>")
std::unique_ptr<Communicator> make_shm_comm(int world, int rank, const char* name, std::string* err);

}  // namespace apex
