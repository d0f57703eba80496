// Phase timing of the prioritized-replay path kernel (not part of the library): the kernel source compiled with
// RLX_PER_PROFILE, which makes thread 0 stamp the 100 MHz wall clock at the phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Icoach_amd/csrc \
//         tools/per_update_profile.hip coach_amd/csrc/runtime.hip -o /tmp/per_prof && /tmp/per_prof [n]
#define RLX_PER_PROFILE 1
#include "../coach_amd/csrc/sumtree.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32, cap = 1 << 20;
    const bool store = argc > 2;          // any second argument: time rlx_per_store of n consecutive leaves instead
    double *t[3], *maxp, *err;
    int *status, *idx;
    for (auto &p : t) CK(hipMalloc(&p, sizeof(double) * (2 * cap - 1)));
    CK(hipMalloc(&maxp, 8)); CK(hipMalloc(&status, 4)); CK(hipMalloc(&idx, 4 * n)); CK(hipMalloc(&err, 8 * n));
    CK(hipMemset(status, 0, 4));
    if (rlx_per_init(t[0], t[1], t[2], cap, maxp, nullptr)) { printf("init: %s\n", rlx_last_error()); return 1; }
    for (int s = 0; s < cap; s += 1 << 16)
        if (rlx_per_store(t[0], t[1], t[2], cap, s, 1 << 16, 0.6, maxp, status, nullptr)) { printf("store failed\n"); return 1; }
    static const char *names[] = {"clear tables + inputs", "issue prefetch", "pow", "leaf level (3 barriers)",
                                  "prefetch arrival", "20 levels", "root store"};
    std::vector<int> hi(n);
    std::vector<double> he(n);
    srand(1);
    for (int rep = 0; rep < 6; ++rep) {
        for (int i = 0; i < n; ++i) { hi[i] = (int)(((unsigned)rand() * 2654435761u) % cap); he[i] = rand() / (double)RAND_MAX * 3.0; }
        CK(hipMemcpy(idx, hi.data(), 4 * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(err, he.data(), 8 * n, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, nullptr));
        if (store) {
            if (rlx_per_store(t[0], t[1], t[2], cap, (int)(((unsigned)rand() * 2654435761u) % (cap - n)), n, 0.6, maxp, status, nullptr)) { printf("store failed\n"); return 1; }
        } else if (rlx_per_update(t[0], t[1], t[2], cap, idx, err, n, 0.6, 1e-6, maxp, status, nullptr)) { printf("update failed\n"); return 1; }
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long p[8];
        CK(hipMemcpyFromSymbol(p, HIP_SYMBOL(rlx_per::g_per_prof), sizeof(p)));
        printf("%s rep %d  n %d  events %.1f us |", store ? "store " : "update", rep, n, ms * 1e3);
        for (int k = 0; k < 6; ++k) printf("  %s %.2f", names[k], (p[k + 1] - p[k]) * 0.01);
        printf("  | in-kernel total %.2f us\n", (p[6] - p[0]) * 0.01);
    }
    return 0;
}
