// The dataflow factorisation of a DENSE block of T x T tiles (the top of final-13682's elimination tree is 18 x 18) against
// the level launches on the same matrix, with the per-unit stamps of k_factor_flow turned into a critical-path table.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I apex-solver_amd/csrc -I include tools/flow_bench.cpp -o tools/flow_bench \
//         -L apex-solver_amd -lapexgpu -Wl,-rpath,'$ORIGIN/../apex-solver_amd'
// usage: tools/flow_bench [T=18] [reps=5]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "tile_plan.h"
using namespace apex;

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 18, reps = argc > 2 ? atoi(argv[2]) : 5;
    const size_t te = (size_t)kNB * kNB;
    std::vector<uint8_t> present((size_t)T * T, 0);
    for (int i = 0; i < T; ++i) for (int j = 0; j <= i; ++j) present[(size_t)i * T + j] = 1;
    hipStream_t st; (void)hipStreamCreate(&st);
    // one SPD matrix: strong diagonal, small pseudo-random rest (lower tiles; diagonal tiles symmetric)
    std::vector<double> host((size_t)T * (T + 1) / 2 * te);
    double best[2] = {1e9, 1e9};
    std::vector<double> ref;
    for (int mode = 0; mode < 2; ++mode) {   // 0: level launches, 1: dataflow
        TilePlan tp;
        tp.set_factor_flow(mode ? 64 : 0, 1000);
        const std::string err = tp.build(T, present, st);
        if (!err.empty()) { printf("build: %s\n", err.c_str()); return 1; }
        for (int I = 0; I < T; ++I)
            for (int J = 0; J <= I; ++J) {
                double* h = host.data() + (size_t)tp.slot(I, J) * te;
                for (int r = 0; r < kNB; ++r)
                    for (int c = 0; c < kNB; ++c) {
                        const int gi = I * kNB + r, gj = J * kNB + c, a = std::max(gi, gj), b = std::min(gi, gj);
                        unsigned x = (unsigned)a * 2654435761u ^ (unsigned)b * 40503u; x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
                        h[(size_t)r * kNB + c] = (gi == gj ? 4.0 * T * kNB * 0.02 + 4.0 : 0.0) + 0.02 * ((double)(x & 0xffff) / 65536.0 - 0.5);
                    }
            }
        if (mode && tp.enable_flow_trace() != hipSuccess) { printf("trace alloc failed\n"); return 1; }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < reps; ++rep) {
            (void)hipMemcpyAsync(tp.tiles(), host.data(), host.size() * 8, hipMemcpyHostToDevice, st);
            (void)hipMemsetAsync(tp.flag_dev(), 0, 16, st);
            (void)hipStreamSynchronize(st);
            int failed = 0;
            (void)hipEventRecord(e0, st);
            const hipError_t fe = tp.factor(&failed);
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (fe != hipSuccess || failed) { printf("factor: %s failed_at %d\n", hipGetErrorString(fe), failed); return 1; }
            if (tp.factor_flow_gave_up()) { printf("dataflow launch TIMED OUT\n"); return 1; }
            if (rep) best[mode] = std::min(best[mode], (double)ms);
        }
        std::vector<double> got(host.size());
        (void)hipMemcpy(got.data(), tp.tiles(), got.size() * 8, hipMemcpyDeviceToHost);
        if (mode == 0) ref = got;
        else {
            size_t diff = 0; double md = 0.0;
            for (int I = 0; I < T; ++I)
                for (int J = 0; J <= I; ++J)
                    for (int r = 0; r < kNB; ++r)
                        for (int c = 0; c < (I == J ? r + 1 : kNB); ++c) {
                            const size_t q = (size_t)tp.slot(I, J) * te + (size_t)r * kNB + c;
                            if (got[q] != ref[q]) { ++diff; md = fmax(md, fabs(got[q] - ref[q])); }
                        }
            printf("factor: %zu entries differ from the level launches' (max %.2e)\n", diff, md);
        }
        printf("T = %d: %s  %.1f us  (groups in the launch %d, units %d; list-schedule model %.0f us)\n", T, mode ? "dataflow      " : "level launches", best[mode] * 1e3,
               tp.factor_flow_groups(), tp.factor_flow_units(), tp.factor_flow_sim_us());
        if (mode) {
            std::vector<FactorUnit> u; std::vector<unsigned long long> s;
            if (tp.read_flow_trace(&u, &s) != hipSuccess) { printf("no trace\n"); return 1; }
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t i = 0; i < u.size(); ++i) { t0 = std::min(t0, s[3 * i]); t1 = std::max(t1, s[3 * i + 2]); }
            printf("launch span by the stamps: %.1f us\n", (t1 - t0) * 0.01);
            auto us = [&](unsigned long long t) { return (t - t0) * 0.01; };
            // per kind: how long units wait after dispatch, how long they work
            double w[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0}; int n[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < u.size(); ++i) { const int k = u[i].kind; w[k] += (s[3 * i + 1] - s[3 * i]) * 0.01; d[k] += (s[3 * i + 2] - s[3 * i + 1]) * 0.01; ++n[k]; }
            for (int k = 0; k < 4; ++k)
                if (n[k]) printf("  kind %d (%s): %5d units, mean wait after dispatch %.1f us, mean work %.1f us\n", k, k == 0 ? "potrf" : k == 1 ? "panel" : k == 2 ? "update 48x48" : "update, whole tile", n[k], w[k] / n[k], d[k] / n[k]);
            // the chain: potrf K -> the nine panel units of the tile below the diagonal -> the nine units of the last update of
            // the next diagonal tile -> potrf K+1.  Per group of nine: last dispatch | first ready .. last ready | last done
            printf("  potrf: dispatched ready done | panel (K+1,K): disp<= ready[first..last] done<= | update (K+1,K+1)<-K: disp<= ready[first..last] done<=\n");
            std::vector<size_t> potrfs;
            for (size_t i = 0; i < u.size(); ++i) if ((u[i].kind & 15) == 0) potrfs.push_back(i);
            for (size_t p = 0; p < potrfs.size(); ++p) {
                const size_t i = potrfs[p];
                double g[2][4] = {{-1, 1e18, -1, -1}, {-1, 1e18, -1, -1}};   // [panel | update][last dispatch, first ready, last ready, last done]
                if (p + 1 < potrfs.size()) {
                    const size_t nx = potrfs[p + 1];
                    const double* below = nullptr;
                    for (size_t q = i + 1; q < u.size() && !below; ++q) if ((u[q].kind & 15) == 1) below = u[q].C;   // first panel tile of column K
                    for (size_t q = 0; q < u.size(); ++q) {
                        int which = -1;
                        if ((u[q].kind & 15) == 1 && u[q].C == below) which = 0;
                        if ((u[q].kind & 15) == 2 && u[q].C == u[nx].C && u[q].A == below) which = 1;
                        if (which < 0) continue;
                        g[which][0] = std::max(g[which][0], us(s[3 * q])); g[which][1] = std::min(g[which][1], us(s[3 * q + 1]));
                        g[which][2] = std::max(g[which][2], us(s[3 * q + 1])); g[which][3] = std::max(g[which][3], us(s[3 * q + 2]));
                    }
                }
                printf("  %3d: %8.1f %8.1f %8.1f | %8.1f [%8.1f .. %8.1f] %8.1f | %8.1f [%8.1f .. %8.1f] %8.1f\n", u[i].strip, us(s[3 * i]), us(s[3 * i + 1]),
                       us(s[3 * i + 2]), g[0][0], g[0][1], g[0][2], g[0][3], g[1][0], g[1][1], g[1][2], g[1][3]);
            }
        }
    }
    return 0;
}
