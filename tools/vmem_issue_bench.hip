// vmem_issue_bench.hip -- what does a vector-memory instruction cost the CU's address / L1 path, by access shape?
//   hipcc --offload-arch=gfx950 -O3 tools/vmem_issue_bench.hip -o tools/vmem_issue_bench && tools/vmem_issue_bench
// Every wave issues LOADS 16-byte (or 8 / 4-byte) loads per trip from a small table (L2 / L1 resident, so that what is
// timed is the path, not the memory), in one of these shapes:
//   0 contiguous        lane L reads 16 B at base + 16 L                      (1 KB per instruction, 16 lines)
//   1 quad-contiguous   quads read 64 contiguous bytes at random places       (16 lines)
//   2 pair-contiguous   lane pairs read 32 contiguous bytes at random places  (32 lines)
//   3 scattered         every lane its own random 16 B                        (64 lines)
//   4 scattered 8 B, 5 scattered 4 B
//   6 octet-contiguous  eight lanes read 128 contiguous, 128-byte aligned bytes at random places (16 x 64 B = 8 L2 lines)
//   7 16-lane-contiguous  sixteen lanes read 256 contiguous bytes at random places               (16 x 64 B = 4 x 256 B)
//   (round 5: does an L2 miss cost per 64-byte request or per 128-byte L2 line?  -- what a landmark BUNDLE would buy)
// Reported: clocks per wave-instruction at the CU level (wall clocks x CUs-worth of waves / instructions issued on a CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ tab, const uint32_t* __restrict__ idx, int table_bytes, int trips, double* out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc = 0.0;
    uint32_t r = idx[(wave * 64 + lane) & 0xFFFF];
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            r = r * 1664525u + 1013904223u;
            uint32_t off;
            if (SHAPE == 0) off = ((r >> 8) * 0 + ((t * 8 + k) * 1024 + wave * 8192) % (table_bytes - 1024)) & ~1023u, off += 16 * lane;
            else if (SHAPE == 1) { const uint32_t q = __shfl(r, lane & ~3, 64); off = ((q >> 4) % (table_bytes / 64)) * 64 + 16 * (lane & 3); }
            else if (SHAPE == 2) { const uint32_t q = __shfl(r, lane & ~1, 64); off = ((q >> 4) % (table_bytes / 32)) * 32 + 16 * (lane & 1); }
            else if (SHAPE == 6) { const uint32_t q = __shfl(r, lane & ~7, 64); off = ((q >> 4) % (table_bytes / 128)) * 128 + 16 * (lane & 7); }
            else if (SHAPE == 7) { const uint32_t q = __shfl(r, lane & ~15, 64); off = ((q >> 4) % (table_bytes / 256)) * 256 + 16 * (lane & 15); }
            else off = ((r >> 4) % (table_bytes / 16)) * 16;
            if (SHAPE <= 3 || SHAPE >= 6) { const double2 v = *reinterpret_cast<const double2*>(tab + off); acc += v.x + v.y; }
            else if (SHAPE == 4) { acc += *reinterpret_cast<const double*>(tab + off); }
            else { acc += *reinterpret_cast<const float*>(tab + off); }
        }
    }
    if (acc == 1.2345e301) out[0] = acc;
}

int main(int argc, char** argv) {
    const int table_bytes = argc > 1 ? atoi(argv[1]) : (1 << 20);
    const int trips = 2000;
    char* tab; uint32_t* idx; double* out;
    hipMalloc(&tab, table_bytes); hipMemset(tab, 0, table_bytes);
    std::vector<uint32_t> h(65536); for (auto& x : h) x = (uint32_t)rand();
    hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[8] = {"contiguous 16 B", "quad-contiguous 64 B", "pair-contiguous 32 B", "scattered 16 B", "scattered 8 B", "scattered 4 B", "octet-contiguous 128 B", "16-lane-contiguous 256 B"};
    for (int wgs_per_cu : {1, 2, 4}) {
        const int grid = 256 * wgs_per_cu;
        printf("table %d KB, %d workgroups (of 4 waves) per CU:\n", table_bytes / 1024, wgs_per_cu);
        for (int s = 0; s < 8; ++s) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (s) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                    case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, tab, idx, table_bytes, trips, out); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            const double inst_per_cu = (double)wgs_per_cu * 4 * trips * 8;
            printf("  %-22s %8.3f ms  = %6.1f ns per wave-instruction per CU (~%5.1f clk at 2.1 GHz)\n", names[s], best, best * 1e6 / inst_per_cu, best * 1e6 / inst_per_cu * 2.1);
        }
    }
    return 0;
}
