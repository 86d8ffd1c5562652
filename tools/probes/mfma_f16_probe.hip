// Probe (GPU box): does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs, and what does it cost per issue?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_f16_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void denorm_kernel(float a_val, float b_val, float *out) {
    const int lane = threadIdx.x;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((lane >> 4) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }   // k = 0 only
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[lane * 4 + 0] = c[0];
}

template <int KIND>
__global__ void rate_kernel(unsigned long long *cyc, float *sink, int iters) {
    f16x8 a, b; bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i));
                                  ab[i] = (__bf16)(0.001f * (threadIdx.x + i)); bb[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    float *out; hipMalloc(&out, 256 * 4);
    float h[256];
    struct { float a, b; const char *what; } cases[] = {
        {1.0f, 9.5367431640625e-07f, "B = 2^-20 (fp16 subnormal), A = 1"},
        {9.5367431640625e-07f, 1.0f, "A = 2^-20 (fp16 subnormal), B = 1"},
        {5.9604644775390625e-08f, 1.0f, "A = 2^-24 (smallest fp16 subnormal), B = 1"},
        {0.00006103515625f, 1.0f, "A = 2^-14 (smallest fp16 normal), B = 1"},
        {9.5367431640625e-07f, 9.5367431640625e-07f, "A = B = 2^-20 (product 2^-40)"}};
    for (auto &cs : cases) {
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, cs.a, cs.b, out);
        hipMemcpy(h, out, 256 * 4, hipMemcpyDeviceToHost);
        printf("%-50s -> C[0][0] = %.10e (expected %.10e)\n", cs.what, h[0], (double)cs.a * cs.b);
    }
    unsigned long long *cyc; float *sink; hipMalloc(&cyc, 1024 * 8); hipMalloc(&sink, 1024 * 64 * 4);
    unsigned long long hc[1024];
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(64), 0, 0, cyc, sink, iters);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(64), 0, 0, cyc, sink, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(hc, cyc, 1024 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; ++i) s += hc[i];
        printf("%s 16x16x32: %.2f cycles per MFMA (one wave per SIMD, 4 independent accumulators)\n", kind ? "bf16" : "f16", s / 1024 / iters / 4);
    }
    return 0;
}
