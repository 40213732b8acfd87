// Host build of the multi-rank communicator's shared-memory transport (apex-solver_amd/csrc/comm.cpp with
// APEX_COMM_HOST_ONLY: "device" buffers are host buffers, no HIP, no GPU) + a self-test that every rank of a run calls.
// Built and driven by tests/test_comm_host.py with 2-4 processes under `pytest -m "not gpu"`: the rendezvous (fresh segment,
// leftovers of crashed runs), the sense-reversing barrier, multi-round transfers (> 4 MiB per rank), rank-ordered sums.
#define APEX_COMM_HOST_ONLY 1
#include "../apex-solver_amd/csrc/comm.cpp"

#include <stdio.h>

static double f(int r, size_t i) { return 1.0 / (double)(3 + r) + (double)((i * 2654435761u + (size_t)r * 40503u) % 1000003u) * 1e-7; }

extern "C" int comm_host_selftest(int world, int rank, const char* name, long n_big, char* msg, int msg_len) {
    std::string err;
    auto fail = [&](const std::string& m) { snprintf(msg, (size_t)msg_len, "rank %d: %s", rank, m.c_str()); return 1; };
    std::unique_ptr<apex::Communicator> c = apex::make_shm_comm(world, rank, name, &err);
    if (!c) return fail("make_shm_comm: " + err);
    const size_t n = (size_t)n_big;
    // all-reduce (sum), several rounds, against the sum in rank order (bitwise)
    std::vector<double> v(n), want(n, 0.0);
    for (size_t i = 0; i < n; ++i) v[i] = f(rank, i);
    for (int r = 0; r < world; ++r)
        for (size_t i = 0; i < n; ++i) want[i] += f(r, i);
    if (!c->all_reduce_sum(v.data(), n, nullptr)) return fail("all_reduce_sum: " + c->error());
    for (size_t i = 0; i < n; ++i) if (v[i] != want[i]) return fail("all_reduce_sum differs at " + std::to_string(i));
    // all-reduce (max)
    std::vector<int> m(1000);
    for (size_t i = 0; i < m.size(); ++i) m[i] = (int)((i * 7919u + (size_t)rank * 104729u) % 10007u) - 5000;
    if (!c->all_reduce_max(m.data(), m.size(), nullptr)) return fail("all_reduce_max: " + c->error());
    for (size_t i = 0; i < m.size(); ++i) {
        int w = -1 << 30;
        for (int r = 0; r < world; ++r) { const int q = (int)((i * 7919u + (size_t)r * 104729u) % 10007u) - 5000; w = q > w ? q : w; }
        if (m[i] != w) return fail("all_reduce_max differs at " + std::to_string(i));
    }
    // reduce to a root: the root holds the sum, the others keep what they had
    const int root = world > 1 ? 1 : 0;
    std::vector<double> q(5000);
    for (size_t i = 0; i < q.size(); ++i) q[i] = f(rank, i + 17);
    if (!c->reduce_sum(q.data(), q.size(), root, nullptr)) return fail("reduce_sum: " + c->error());
    for (size_t i = 0; i < q.size(); ++i) {
        double w = 0.0;
        for (int r = 0; r < world; ++r) w += f(r, i + 17);
        if (q[i] != (rank == root ? w : f(rank, i + 17))) return fail("reduce_sum differs at " + std::to_string(i));
    }
    // broadcast from the last rank
    std::vector<double> b(3000);
    for (size_t i = 0; i < b.size(); ++i) b[i] = f(rank, i + 5);
    if (!c->broadcast(b.data(), b.size(), world - 1, nullptr)) return fail("broadcast: " + c->error());
    for (size_t i = 0; i < b.size(); ++i) if (b[i] != f(world - 1, i + 5)) return fail("broadcast differs at " + std::to_string(i));
    // all-gather of 1 KiB per rank, an empty collective, and a group around two calls
    std::vector<char> mine(1024), all((size_t)world * 1024);
    for (size_t i = 0; i < mine.size(); ++i) mine[i] = (char)(rank * 31 + (int)i);
    if (!c->all_gather(mine.data(), all.data(), mine.size(), nullptr)) return fail("all_gather: " + c->error());
    for (int r = 0; r < world; ++r)
        for (size_t i = 0; i < 1024; ++i) if (all[(size_t)r * 1024 + i] != (char)(r * 31 + (int)i)) return fail("all_gather differs");
    if (!c->all_reduce_sum(v.data(), 0, nullptr)) return fail("empty all_reduce_sum: " + c->error());
    double one = (double)(rank + 1); int flag = rank;
    if (!c->group_start() || !c->all_reduce_sum(&one, 1, nullptr) || !c->all_reduce_max(&flag, 1, nullptr) || !c->group_end())
        return fail("grouped collectives: " + c->error());
    if (one != 0.5 * world * (world + 1) || flag != world - 1) return fail("grouped collectives differ");
    snprintf(msg, (size_t)msg_len, "rank %d of %d ok (%s)", rank, world, c->transport());
    return 0;
}
