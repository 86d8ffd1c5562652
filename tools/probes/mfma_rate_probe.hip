// Probe (GPU box): issue cost of the MFMA shapes considered for the RBF block of the edge MLPs' first Linear.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/mfma_rate_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void rate_kernel(unsigned long long *cyc, float *sink, int iters) {
    f16x8 a8, b8; f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.001f * (threadIdx.x + i)); b8[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    const float af = 0.001f * threadIdx.x, bf = 0.5f - 0.002f * threadIdx.x;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c3, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    unsigned long long *cyc; float *sink; hipMalloc(&cyc, 1024 * 8); hipMalloc(&sink, 1024 * 64 * 4);
    unsigned long long hc[1024];
    const int iters = 2000;
    const char *names[3] = {"f16 16x16x32", "f16 16x16x16", "f32 16x16x4"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(64), 0, 0, cyc, sink, iters);
            else if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(64), 0, 0, cyc, sink, iters);
            else hipLaunchKernelGGL(rate_kernel<2>, dim3(1024), dim3(64), 0, 0, cyc, sink, iters);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(hc, cyc, 1024 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; ++i) s += hc[i];
        printf("%s: %.2f s_memtime ticks per MFMA (one wave per SIMD, 4 independent accumulators)\n", names[kind], s / 1024 / iters / 4);
    }
    return 0;
}
