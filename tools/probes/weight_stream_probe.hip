// Probe (GPU box): how fast can EVERY workgroup of a 256-workgroup launch stream the same 512 KB of weights into its CU?
// This is the number that decides whether the per-node products (512 KB of weights per layer) can move from their own launch
// (node_linear16_kernel: weights stationary, every workgroup reads 128 KB) into the per-workgroup edge kernels, where each of
// the ~252 workgroups would have to stream all 512 KB for its 22 atoms.
//   variant 0: LDS-DMA (global_load_lds_dwordx4), a ring of RING KB refilled in bulk rounds (vmcnt(0) + barrier per round:
//              an upper bound for a loader / consumer ring, which adds its handshake on top)
//   variant 1: register-staged global_load_dwordx4, 8 loads in flight per lane (what the fused `lin_fuse` variants did)
// Both with 12 or 3 issuing waves per workgroup (3 = the helper waves a fused node stage has free).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wsp tools/probes/weight_stream_probe.hip && /tmp/wsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kBytes = 512 * 1024, kRing = 64 * 1024;

template <int VARIANT>
__global__ void __launch_bounds__(768) stream_kernel(const float *w, float *sink, unsigned long long *ticks, int reps, int issue_waves) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0;
    for (int r = 0; r < reps; ++r) {
        if (r == 1) { __syncthreads(); t0 = __builtin_amdgcn_s_memrealtime(); }        // repetition 0 warms L2 / the Infinity Cache
        if (VARIANT == 0) {
            for (int base = 0; base < kBytes; base += kRing) {                          // one ring refill per round
                if (wave < issue_waves)
                    for (int piece = wave; piece < kRing / 1024; piece += issue_waves)   // 1 KB per wave-instruction
                        __builtin_amdgcn_global_load_lds((gbl_void *)(w + (base + piece * 1024) / 4 + lane * 4), (lds_void *)(lds + piece * 256), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const float4 v = reinterpret_cast<const float4 *>(lds)[threadIdx.x];     // touch the data
                acc.x += v.x; acc.y += v.y;
                __syncthreads();
            }
        } else {
            if (wave < issue_waves) {
                const float4 *src = reinterpret_cast<const float4 *>(w);
                const int n4 = kBytes / 16, stride = issue_waves * 64;
                for (int i = wave * 64 + lane; i < n4; i += 8 * stride) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = (i + u * stride < n4) ? src[i + u * stride] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].w; }
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc.x + acc.y == 12345.678f) sink[0] = acc.x;
}

int main() {
    float *w, *sink; unsigned long long *ticks;
    hipMalloc(&w, kBytes); hipMalloc(&sink, 64); hipMalloc(&ticks, 256 * 8);
    std::vector<float> h(kBytes / 4, 0.5f);
    hipMemcpy(w, h.data(), kBytes, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)stream_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    const int reps = 21;
    for (int variant = 0; variant < 2; ++variant)
        for (int iw : {12, 3}) {
            for (int pass = 0; pass < 2; ++pass) {
                if (variant == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(256), dim3(768), kRing, 0, w, sink, ticks, reps, iw);
                else hipLaunchKernelGGL(stream_kernel<1>, dim3(256), dim3(768), 0, 0, w, sink, ticks, reps, iw);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> t(256);
            hipMemcpy(t.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
            double worst = 0, sum = 0;
            for (auto x : t) { worst = x > worst ? x : worst; sum += x; }
            const double us_med = sum / 256 / 100.0 / (reps - 1), us_worst = worst / 100.0 / (reps - 1);      // 100 MHz counter
            printf("%s, %2d issuing waves: 512 KB per workgroup, 256 workgroups at once: mean %.2f us (%.0f GB/s per CU), slowest workgroup %.2f us\n",
                   variant == 0 ? "LDS-DMA ring (64 KB rounds)" : "register-staged loads      ", iw, us_med, kBytes / us_med / 1e3, us_worst);
        }
    return 0;
}
