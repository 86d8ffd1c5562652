// Developer probe (not part of the library): what one SIMD of gfx950 issues per cycle.  Waves of one workgroup per CU run either a
// stream of MFMAs or a stream of vector instructions; the run time of the mixtures against the pure streams says whether (and how
// far) the matrix pipe and the vector ALU of a SIMD overlap.   hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// role 0: nothing; 1: MFMA stream, CH independent accumulator chains; 2: VALU stream, independent; 3: VALU stream, one dependent chain
// 4: MFMA + VALU interleaved in one wave (1 MFMA : 3 fma)
template <int CH>
__device__ void mfma_stream(int iters, float *out) {
    f32x4 acc[CH];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int CH>
__device__ void valu_stream(int iters, float *out) {
    float v[CH];
    const float m = 1.0f + 1e-7f * threadIdx.x, k = 1e-9f * threadIdx.x;
    for (int c = 0; c < CH; ++c) v[c] = threadIdx.x + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / CH * 4; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(m), "v"(k));
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += v[c];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__device__ void mixed_stream(int iters, float *out) {
    f32x4 acc[4];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i); }
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[12];
    const float m = 1.0f + 1e-7f * threadIdx.x, k = 1e-9f * threadIdx.x;
    for (int c = 0; c < 12; ++c) v[c] = threadIdx.x + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3 * c]) : "v"(m), "v"(k));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3 * c + 1]) : "v"(m), "v"(k));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3 * c + 2]) : "v"(m), "v"(k));
            }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0];
    for (int c = 0; c < 12; ++c) s += v[c];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// roles[w & 7 ... ]: role of wave w (waves w, w + 4, w + 8 share SIMD w & 3)
__global__ void __launch_bounds__(1024) probe(const int *roles, int iters, int prio_mask, float *out) {
    extern __shared__ float lds[];
    if (iters < 0) out[0] = lds[threadIdx.x];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = roles[w];
    if ((prio_mask >> w) & 1) __builtin_amdgcn_s_setprio(3);
    switch (role) {
    case 1: mfma_stream<4>(iters, out); break;            // 32 MFMAs per iteration
    case 5: mfma_stream<1>(iters * 4, out); break;        // one dependent chain: 8 per iteration x 4
    case 6: mfma_stream<2>(iters * 2, out); break;
    case 2: valu_stream<8>(iters, out); break;            // 128 fma per iteration, 8 independent chains
    case 3: valu_stream<1>(iters, out); break;            // 128 fma, one dependent chain
    case 7: valu_stream<2>(iters, out); break;
    case 4: mixed_stream(iters, out); break;              // 32 MFMAs + 96 fma per iteration
    default: break;
    }
}

int main(int argc, char **argv) {
    const int iters = 20000;
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);      // one workgroup per CU
    struct Cfg { const char *name; int nw; int roles[16]; int prio; };
    const Cfg cfgs[] = {
        {"mfma x1/SIMD (4 chains)", 4, {1, 1, 1, 1}, 0},
        {"mfma x1/SIMD (1 dependent chain)", 4, {5, 5, 5, 5}, 0},
        {"mfma x1/SIMD (2 chains)", 4, {6, 6, 6, 6}, 0},
        {"mfma x2/SIMD (4 chains each)", 8, {1, 1, 1, 1, 1, 1, 1, 1}, 0},
        {"valu x1/SIMD (8 chains)", 4, {2, 2, 2, 2}, 0},
        {"valu x1/SIMD (1 dependent chain)", 4, {3, 3, 3, 3}, 0},
        {"valu x1/SIMD (2 chains)", 4, {7, 7, 7, 7}, 0},
        {"valu x2/SIMD (8 chains each)", 8, {2, 2, 2, 2, 2, 2, 2, 2}, 0},
        {"valu x2/SIMD (1 chain each)", 8, {3, 3, 3, 3, 3, 3, 3, 3}, 0},
        {"valu x3/SIMD (1 chain each)", 12, {3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3}, 0},
        {"mfma + valu(8ch) on each SIMD", 8, {1, 1, 1, 1, 2, 2, 2, 2}, 0},
        {"mfma + valu(8ch), valu prio 3", 8, {1, 1, 1, 1, 2, 2, 2, 2}, 0xf0},
        {"mfma + valu(8ch), mfma prio 3", 8, {1, 1, 1, 1, 2, 2, 2, 2}, 0x0f},
        {"mfma + valu(1ch) on each SIMD", 8, {1, 1, 1, 1, 3, 3, 3, 3}, 0},
        {"mfma(1ch) + valu(1ch)", 8, {5, 5, 5, 5, 3, 3, 3, 3}, 0},
        {"2 mfma + 1 valu(1ch) per SIMD", 12, {1, 1, 1, 1, 1, 1, 1, 1, 3, 3, 3, 3}, 0},
        {"2 mfma + 1 valu(1ch), valu prio", 12, {1, 1, 1, 1, 1, 1, 1, 1, 3, 3, 3, 3}, 0xf00},
        {"2 mfma + 1 valu(8ch), valu prio", 12, {1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2}, 0xf00},
        {"mixed in one wave (1:3)", 4, {4, 4, 4, 4}, 0},
        {"mixed in one wave x2/SIMD", 8, {4, 4, 4, 4, 4, 4, 4, 4}, 0},
    };
    int *d_roles; float *d_out;
    hipMalloc(&d_roles, 16 * sizeof(int)); hipMalloc(&d_out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // clock warm-up
    { int r[16] = {1, 1, 1, 1, 2, 2, 2, 2}; hipMemcpy(d_roles, r, sizeof(r), hipMemcpyHostToDevice);
      for (int k = 0; k < 20; ++k) probe<<<256, 512, 98304>>>(d_roles, iters, 0, d_out); hipDeviceSynchronize(); }
    printf("%-36s %9s  %s\n", "configuration (256 workgroups)", "ms", "cycles@2.4GHz per iteration (32 MFMA = 512 pipe cycles; 128 fma = 512 issue cycles)");
    for (const Cfg &c : cfgs) {
        hipMemcpy(d_roles, c.roles, sizeof(c.roles), hipMemcpyHostToDevice);
        probe<<<256, c.nw * 64, 98304>>>(d_roles, iters, c.prio, d_out);
        hipEventRecord(e0);
        probe<<<256, c.nw * 64, 98304>>>(d_roles, iters, c.prio, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-36s %9.3f  %8.1f\n", c.name, ms, ms * 1e-3 * 2.4e9 / iters);
    }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
