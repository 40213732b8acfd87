// hwid_census.hip -- where do workgroups land?  One record per workgroup: XCC_ID and the HW_ID fields (SE, SH, CU).
//   hipcc --offload-arch=gfx950 -O3 tools/hwid_census.hip -o tools/hwid_census && tools/hwid_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(192) void k(unsigned* out, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
int main() {
    const int n = 4096;
    unsigned* d; hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(k, dim3(n), dim3(192), 0, 0, d, 2000);   // 20 us each: the grid fills the chip
    std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cus; std::map<unsigned, int> cu_ids, se_ids, sh_ids, xccs;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; cu_ids[cu]++; se_ids[se]++; sh_ids[sh]++; xccs[xcc]++;
    }
    printf("distinct (xcc, se, sh, cu): %zu\n", cus.size());
    printf("cu_id histogram:"); for (auto& p : cu_ids) printf(" %u:%d", p.first, p.second); printf("\n");
    printf("se_id histogram:"); for (auto& p : se_ids) printf(" %u:%d", p.first, p.second); printf("\n");
    printf("sh_id histogram:"); for (auto& p : sh_ids) printf(" %u:%d", p.first, p.second); printf("\n");
    printf("xcc histogram:"); for (auto& p : xccs) printf(" %u:%d", p.first, p.second); printf("\n");
    printf("first 16 workgroups (xcc se sh cu):"); for (int i = 0; i < 16; ++i) printf(" [%u %u %u %u]", h[2*i+1] & 0xF, (h[2*i] >> 13) & 7, (h[2*i] >> 12) & 1, (h[2*i] >> 8) & 0xF); printf("\n");
    // per xcc: which (se, cu) exist
    for (unsigned x = 0; x < 2; ++x) { printf("xcc %u:", x); for (auto& p : cus) if ((p.first >> 12) == x) printf(" se%u.cu%u(%d)", (p.first >> 8) & 7, p.first & 0xF, p.second); printf("\n"); }
    return 0;
}
