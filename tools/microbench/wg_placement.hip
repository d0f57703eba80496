// Where does the dispatcher put the workgroups of a launch that is about one workgroup per CU?  Each workgroup records
// the XCC id and the HW_ID register (SE / CU / SIMD fields) of its first wave and then spins for ~10 us so that all of
// them are resident together.  Prints, for several grid sizes and LDS footprints, how many CUs hold 0 / 1 / 2 / 3+
// workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/wg_placement.hip -o /tmp/wg_placement && /tmp/wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256) probe(unsigned *out, long long spin_ticks) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID, 4 bits
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        lds[0] = (float)hw;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) {}
}

int main() {
    const int grids[] = {196, 324, 400, 648, 784};
    const int lds_kb[] = {17, 43, 90};
    unsigned *d;
    hipMalloc(&d, 2 * 4096 * sizeof(unsigned));
    for (int kb : lds_kb)
        for (int n : grids) {
            hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
            hipLaunchKernelGGL(probe, dim3(n), dim3(256), kb * 1024, 0, d, 1000LL);     // 10 us at 100 MHz
            hipDeviceSynchronize();
            std::vector<unsigned> h(2 * n);
            hipMemcpy(h.data(), d, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
            std::map<unsigned, int> per_cu;
            for (int i = 0; i < n; ++i) {
                const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
                const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
                per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
            }
            int hist[8] = {0};
            for (auto &kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
            printf("LDS %2d KB  grid %4d : distinct CUs %3zu | CUs holding 1: %3d  2: %3d  3: %3d  4: %3d  5+: %3d\n", kb, n,
                   per_cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7]);
        }
    return 0;
}
