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
constexpr uint32_t kShmReady = 0x41504558u;     // written by rank 0 once the header of a FRESH segment is zeroed
struct ShmHeader {
    std::atomic<uint32_t> arrive;
    std::atomic<uint32_t> gen;
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> ready;
    std::atomic<uint32_t> go;
    uint32_t pad[11];
};
// the transport's only contact with the device: staging copies and the stream's completion
#ifdef APEX_COMM_HOST_ONLY
bool dev_sync(hipStream_t) { return true; }
bool dev_to_host(void* h, const void* d, size_t n) { memcpy(h, d, n); return true; }
bool host_to_dev(void* d, const void* h, size_t n) { memcpy(d, h, n); return true; }
#else
bool dev_sync(hipStream_t s) { return hipStreamSynchronize(s) == hipSuccess; }
bool dev_to_host(void* h, const void* d, size_t n) { return hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess; }
bool host_to_dev(void* d, const void* h, size_t n) { return hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; }
#endif

class ShmComm final : public Communicator {
   public:
    ShmComm(int world, int rank, std::string name, void* base, size_t bytes)
        : Communicator(world, rank), name_(std::move(name)), base_(static_cast<char*>(base)), bytes_(bytes) {}
    ~ShmComm() override {
        if (base_) munmap(base_, bytes_);   // (the name was unlinked by rank 0 when the rendezvous completed: make_shm_comm)
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
        if (!dev_sync(s)) return fail("shm communicator: stream error");
        if (!dev_to_host(slot(rank_), send, bytes)) return fail("shm communicator: copy to host failed");
        if (!barrier()) return false;
        std::vector<char> all((size_t)world_ * bytes);
        for (int r = 0; r < world_; ++r) memcpy(all.data() + (size_t)r * bytes, slot(r), bytes);
        if (!barrier()) return false;
        if (!host_to_dev(recv, all.data(), all.size())) return fail("shm communicator: copy to device failed");
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
        if (!dev_sync(s)) return fail("shm communicator: stream error before the collective");
        char* p = static_cast<char*>(dev);
        for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += kShmChunk) {
            const size_t len = bytes - off < kShmChunk ? bytes - off : kShmChunk;
            if (len && !dev_to_host(slot(rank_), p + off, len)) return fail("shm communicator: copy to host failed");
            if (!barrier()) return false;
            if (len) combine(slot(rank_), len);
            if (!barrier()) return false;
            if (len && (only < 0 || only == rank_) && !host_to_dev(p + off, tmp_.data(), len))
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

// Rank 0 owns the segment's life: it removes whatever a crashed run may have left under the name, creates the segment
// exclusively, zeroes the header and only then marks it ready; the others open (never create) and attach to a READY
// segment, and while they wait they keep checking that the object behind the name is still the one they mapped -- a rank
// that was quicker than rank 0 and found a leftover moves over to the fresh segment instead of waiting on a dead one.
// Once every rank is attached rank 0 unlinks the name (the memory lives until the last unmap) and then gives the go, so a
// completed rendezvous never leaves a name behind, a failed one is unlinked too, and no leftover can say "go".
std::unique_ptr<Communicator> make_shm_comm(int world, int rank, const char* name, std::string* err) {
    auto bad = [&](const std::string& m) -> std::unique_ptr<Communicator> { if (err) *err = m; return nullptr; };
    if (world < 1 || rank < 0 || rank >= world || !name || !*name) return bad("shm communicator: bad arguments");
    std::string n = std::string("/apexgpu-") + name;
    for (char& c : n) if (c == '/' && &c != &n[0]) c = '_';
    const size_t bytes = sizeof(ShmHeader) + (size_t)world * kShmChunk;
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120); };
    void* base = MAP_FAILED;
    ino_t ino = 0;
    if (rank == 0) {
        (void)shm_unlink(n.c_str());
        const int fd = shm_open(n.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return bad("shm communicator: shm_open failed");
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(n.c_str()); return bad("shm communicator: ftruncate failed"); }
        base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (base == MAP_FAILED) { shm_unlink(n.c_str()); return bad("shm communicator: mmap failed"); }
        ShmHeader* h = reinterpret_cast<ShmHeader*>(base);
        h->arrive.store(0); h->gen.store(0); h->attached.store(1); h->go.store(0);
        h->ready.store(kShmReady, std::memory_order_release);
    }
    auto attach = [&]() -> bool {   // ranks > 0: map the object currently behind the name once rank 0 has made it ready
        for (;;) {
            const int fd = shm_open(n.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == bytes) {
                void* b = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                close(fd);
                if (b != MAP_FAILED) {
                    if (reinterpret_cast<ShmHeader*>(b)->ready.load(std::memory_order_acquire) == kShmReady) {
                        base = b; ino = st.st_ino;
                        reinterpret_cast<ShmHeader*>(b)->attached.fetch_add(1, std::memory_order_acq_rel);
                        return true;
                    }
                    munmap(b, bytes);
                }
            } else if (fd >= 0) {
                close(fd);
            }
            if (timed_out()) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(500));
        }
    };
    if (rank != 0 && !attach()) return bad("shm communicator: rank 0 did not create the segment within 120 s");
    // rendezvous: rank 0 waits until everyone is attached to ITS segment, unlinks the name and only then gives the go; the
    // others wait for the go (never for the attach count: a leftover may show any) and keep checking the name meanwhile
    for (unsigned spin = 0;; ++spin) {
        ShmHeader* h = reinterpret_cast<ShmHeader*>(base);
        if (rank == 0 ? h->attached.load(std::memory_order_acquire) >= (uint32_t)world : h->go.load(std::memory_order_acquire) == 1u) break;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (rank != 0 && (spin & 0x3F) == 0x3F) {   // still the object behind the name?  (a leftover that rank 0 has since replaced)
            struct stat st;
            const int fd = shm_open(n.c_str(), O_RDWR, 0600);
            const bool replaced = fd >= 0 && fstat(fd, &st) == 0 && st.st_ino != ino;
            if (fd >= 0) close(fd);
            if (replaced) {
                munmap(base, bytes); base = MAP_FAILED;
                if (!attach()) return bad("shm communicator: rank 0 did not create the segment within 120 s");
            }
        }
        if (timed_out()) {
            munmap(base, bytes);
            if (rank == 0) shm_unlink(n.c_str());
            return bad("shm communicator: the other ranks did not attach within 120 s");
        }
    }
    auto c = std::unique_ptr<ShmComm>(new ShmComm(world, rank, n, base, bytes));
    if (rank == 0) {   // (in this order: a crash in between leaves no name that says "go")
        shm_unlink(n.c_str());
        reinterpret_cast<ShmHeader*>(base)->go.store(1u, std::memory_order_release);
    }
    return c;
}

}  // namespace apex
