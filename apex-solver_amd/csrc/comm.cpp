// comm.cpp -- see comm.h
#include "comm.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#ifdef APEX_WITH_RCCL
#include <rccl/rccl.h>
#endif

namespace apex {

// ------------------------------------------------------------------------------------------------------------------
// RCCL
// ------------------------------------------------------------------------------------------------------------------
#ifdef APEX_WITH_RCCL
namespace {
class RcclComm final : public Communicator {
   public:
    RcclComm(int world, int rank, ncclComm_t c) : Communicator(world, rank), c_(c) {}
    ~RcclComm() override { if (c_) ncclCommDestroy(c_); }
    const char* transport() const override { return "rccl"; }
    bool all_reduce_sum(double* d, size_t n, hipStream_t s) override { return ok(ncclAllReduce(d, d, n, ncclDouble, ncclSum, c_, s), "all-reduce (sum)"); }
    bool all_reduce_max(int* d, size_t n, hipStream_t s) override { return ok(ncclAllReduce(d, d, n, ncclInt, ncclMax, c_, s), "all-reduce (max)"); }
    bool reduce_sum(double* d, size_t n, int root, hipStream_t s) override { return ok(ncclReduce(d, d, n, ncclDouble, ncclSum, root, c_, s), "reduce (sum)"); }
    bool all_gather(const void* send, void* recv, size_t bytes, hipStream_t s) override { return ok(ncclAllGather(send, recv, bytes, ncclChar, c_, s), "all-gather"); }
    bool broadcast(double* d, size_t n, int root, hipStream_t s) override { return ok(ncclBroadcast(d, d, n, ncclDouble, root, c_, s), "broadcast"); }
    bool group_start() override { return ok(ncclGroupStart(), "group start"); }
    bool group_end() override { return ok(ncclGroupEnd(), "group end"); }

   private:
    bool ok(ncclResult_t r, const char* what) { return r == ncclSuccess ? true : fail(std::string("RCCL error in ") + what + ": " + ncclGetErrorString(r)); }
    ncclComm_t c_;
};
}  // namespace
#endif

std::unique_ptr<Communicator> make_rccl_comm(int world, int rank, const void* unique_id128, std::string* err) {
#ifdef APEX_WITH_RCCL
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(&id, unique_id128, sizeof id);
    ncclComm_t c;
    const ncclResult_t r = ncclCommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) { if (err) *err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return nullptr; }
    return std::unique_ptr<Communicator>(new RcclComm(world, rank, c));
#else
    (void)world; (void)rank; (void)unique_id128;
    if (err) *err = "library built without RCCL";
    return nullptr;
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// host shared memory
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t kShmChunk = (size_t)4 << 20;   // bytes per rank and round
struct ShmHeader {
    std::atomic<uint32_t> arrive;
    std::atomic<uint32_t> gen;
    std::atomic<uint32_t> attached;
    uint32_t pad[13];
};

class ShmComm final : public Communicator {
   public:
    ShmComm(int world, int rank, std::string name, void* base, size_t bytes)
        : Communicator(world, rank), name_(std::move(name)), base_(static_cast<char*>(base)), bytes_(bytes) {}
    ~ShmComm() override {
        if (base_) munmap(base_, bytes_);
        if (rank_ == 0) shm_unlink(name_.c_str());
    }
    const char* transport() const override { return "host shared memory"; }

    bool all_reduce_sum(double* d, size_t n, hipStream_t s) override {
        return rounds(d, n * sizeof(double), s, [&](char* mine, size_t bytes) {
            double* out = reinterpret_cast<double*>(mine);
            const size_t m = bytes / sizeof(double);
            std::vector<double> acc(m, 0.0);
            for (int r = 0; r < world_; ++r) {   // rank order: every rank computes the same bits
                const double* src = reinterpret_cast<const double*>(slot(r));
                for (size_t i = 0; i < m; ++i) acc[i] += src[i];
            }
            tmp_.assign(reinterpret_cast<char*>(acc.data()), reinterpret_cast<char*>(acc.data()) + bytes);
            (void)out;
        }, -1);
    }
    bool all_reduce_max(int* d, size_t n, hipStream_t s) override {
        return rounds(d, n * sizeof(int), s, [&](char*, size_t bytes) {
            const size_t m = bytes / sizeof(int);
            std::vector<int> acc(m);
            memcpy(acc.data(), slot(0), bytes);
            for (int r = 1; r < world_; ++r) {
                const int* src = reinterpret_cast<const int*>(slot(r));
                for (size_t i = 0; i < m; ++i) acc[i] = acc[i] > src[i] ? acc[i] : src[i];
            }
            tmp_.assign(reinterpret_cast<char*>(acc.data()), reinterpret_cast<char*>(acc.data()) + bytes);
        }, -1);
    }
    bool reduce_sum(double* d, size_t n, int root, hipStream_t s) override {
        return rounds(d, n * sizeof(double), s, [&](char*, size_t bytes) {
            const size_t m = bytes / sizeof(double);
            std::vector<double> acc(m, 0.0);
            for (int r = 0; r < world_; ++r) {
                const double* src = reinterpret_cast<const double*>(slot(r));
                for (size_t i = 0; i < m; ++i) acc[i] += src[i];
            }
            tmp_.assign(reinterpret_cast<char*>(acc.data()), reinterpret_cast<char*>(acc.data()) + bytes);
        }, root);
    }
    bool broadcast(double* d, size_t n, int root, hipStream_t s) override {
        return rounds(d, n * sizeof(double), s, [&](char*, size_t bytes) { tmp_.assign(slot(root), slot(root) + bytes); }, -1);
    }
    bool all_gather(const void* send, void* recv, size_t bytes, hipStream_t s) override {
        if (bytes > kShmChunk) return fail("shm communicator: all_gather piece too large");
        if (hipStreamSynchronize(s) != hipSuccess) return fail("shm communicator: stream error");
        if (hipMemcpy(slot(rank_), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("shm communicator: copy to host failed");
        if (!barrier()) return false;
        std::vector<char> all((size_t)world_ * bytes);
        for (int r = 0; r < world_; ++r) memcpy(all.data() + (size_t)r * bytes, slot(r), bytes);
        if (!barrier()) return false;
        if (hipMemcpy(recv, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return fail("shm communicator: copy to device failed");
        return true;
    }

   private:
    char* slot(int r) const { return base_ + sizeof(ShmHeader) + (size_t)r * kShmChunk; }
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base_); }
    // Sense-reversing barrier over the attached processes; a rank that does not arrive within 120 s fails the collective
    // on the others (instead of hanging them).
    bool barrier() {
        ShmHeader* h = hdr();
        const uint32_t g = h->gen.load(std::memory_order_acquire);
        if (h->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world_) {
            h->arrive.store(0, std::memory_order_relaxed);
            h->gen.fetch_add(1, std::memory_order_acq_rel);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; h->gen.load(std::memory_order_acquire) == g; ++spin) {
            if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
            if ((spin & 0xFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return fail("shm communicator: a rank did not reach the collective within 120 s");
        }
        return true;
    }
    // dev buffer of `bytes`, in rounds of kShmChunk: own piece -> slot, barrier, combine(all slots) -> tmp_, barrier, tmp_ -> dev
    // (on `only` when >= 0).
    template <typename F>
    bool rounds(void* dev, size_t bytes, hipStream_t s, F combine, int only) {
        if (hipStreamSynchronize(s) != hipSuccess) return fail("shm communicator: stream error before the collective");
        char* p = static_cast<char*>(dev);
        for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += kShmChunk) {
            const size_t len = bytes - off < kShmChunk ? bytes - off : kShmChunk;
            if (len && hipMemcpy(slot(rank_), p + off, len, hipMemcpyDeviceToHost) != hipSuccess) return fail("shm communicator: copy to host failed");
            if (!barrier()) return false;
            if (len) combine(slot(rank_), len);
            if (!barrier()) return false;
            if (len && (only < 0 || only == rank_) && hipMemcpy(p + off, tmp_.data(), len, hipMemcpyHostToDevice) != hipSuccess)
                return fail("shm communicator: copy to device failed");
            if (bytes == 0) break;
        }
        return true;
    }
    std::string name_;
    char* base_;
    size_t bytes_;
    std::vector<char> tmp_;
};
}  // namespace

std::unique_ptr<Communicator> make_shm_comm(int world, int rank, const char* name, std::string* err) {
    auto bad = [&](const std::string& m) -> std::unique_ptr<Communicator> { if (err) *err = m; return nullptr; };
    if (world < 1 || rank < 0 || rank >= world || !name || !*name) return bad("shm communicator: bad arguments");
    std::string n = std::string("/apexgpu-") + name;
    for (char& c : n) if (c == '/' && &c != &n[0]) c = '_';
    const size_t bytes = sizeof(ShmHeader) + (size_t)world * kShmChunk;
    const int fd = shm_open(n.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) return bad("shm communicator: shm_open failed");
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return bad("shm communicator: ftruncate failed"); }
    void* base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return bad("shm communicator: mmap failed");
    auto c = std::unique_ptr<ShmComm>(new ShmComm(world, rank, n, base, bytes));
    // rendezvous: nobody proceeds (and rank 0 does not unlink) before everyone is attached
    ShmHeader* h = reinterpret_cast<ShmHeader*>(base);
    h->attached.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    while (h->attached.load(std::memory_order_acquire) < (uint32_t)world) {
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return bad("shm communicator: the other ranks did not attach within 120 s");
    }
    return c;
}

}  // namespace apex
