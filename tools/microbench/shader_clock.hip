// What does the shader clock do under this kind of load?  A kernel spins on a dependent MFMA chain; wave 0 of every workgroup
// reads the shader-clock counter (s_memtime / clock64) and the constant 100 MHz counter (s_memrealtime / wall_clock64) before
// and after: cycles per 10 ns tick = the clock the wave actually ran at, and cycles per MFMA = the issue interval of a
// dependent v_mfma_f32_32x32x2_f32 chain on ONE wave per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/shader_clock.hip -o /tmp/shader_clock && /tmp/shader_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) spin(long long *out, float *sink, int n_mfma) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n_mfma; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    long long *d; float *sink;
    hipMalloc(&d, 2 * 4096 * sizeof(long long)); hipMalloc(&sink, 4);
    const int grids[] = {1, 256, 1024};
    for (int rep = 0; rep < 2; ++rep)
        for (int g : grids)
            for (int n : {256, 4096}) {
                for (int k = 0; k < (rep ? 20 : 1); ++k) hipLaunchKernelGGL(spin, dim3(g), dim3(256), 0, 0, d, sink, n);   // rep 1: in a train of launches
                hipDeviceSynchronize();
                std::vector<long long> h(2 * g);
                hipMemcpy(h.data(), d, 2 * g * sizeof(long long), hipMemcpyDeviceToHost);
                double c = 0, w = 0;
                for (int i = 0; i < g; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
                printf("%s grid %4d  %5d dependent MFMAs per wave: %7.1f clock64 ticks and %6.2f us per workgroup -> %.3f ticks per ns, %.1f ticks / %.1f ns per MFMA\n",
                       rep ? "train " : "single", g, n, c / g, w / g * 0.01, (c / g) / (w / g * 10.0), c / g / n, w / g * 10.0 / n);
            }
    return 0;
}
