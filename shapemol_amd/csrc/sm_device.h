// Shared device helpers for the gfx950 kernels: fp32 MFMA tile algebra in the "D layout".
//
// D layout (used for every 16-column activation tile held in registers):
//   a wave holds 16 columns (edges or atoms); lane l = (n = l & 15, g = l >> 4);
//   register 4*t + r of lane (n, g) holds feature 16*t + 4*g + r of column n.
// This is exactly the C/D fragment of v_mfma_f32_16x16x4_f32 when the weight matrix is the A
// operand (rows = output features) and the activations are the B operand (columns): the
// accumulator of one product is therefore the B operand of the next one with no data movement
// (k-step 4*t + r takes register 4*t + r from all four lane groups), and a row-major [col][H]
// array in memory maps onto it with one 16-byte access per tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SM_DEV __device__ __forceinline__

SM_DEV f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

SM_DEV float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
SM_DEV void stg4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// sum over the four lane groups (same n, g = 0..3)
SM_DEV float sum_groups(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// In-register LayerNorm (eps 1e-5, biased variance, two-pass) + ReLU of one D-layout column
// vector; gamma/beta are read from `gb` ([H] gamma followed by [H] beta) at the lane's features.
template <int NT>
SM_DEV void ln_relu_dlayout(float (&v)[NT * 4], const float *gamma, const float *beta, int g) {
    constexpr int H = NT * 16;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) s += v[i];
    const float mean = sum_groups(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) { const float d = v[i] - mean; q += d * d; }
    const float var = sum_groups(q) * (1.0f / H);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 ga = ldg4(gamma + 16 * t + 4 * g);
        const float4 be = ldg4(beta + 16 * t + 4 * g);
        v[4 * t + 0] = fmaxf((v[4 * t + 0] - mean) * rstd * ga.x + be.x, 0.f);
        v[4 * t + 1] = fmaxf((v[4 * t + 1] - mean) * rstd * ga.y + be.y, 0.f);
        v[4 * t + 2] = fmaxf((v[4 * t + 2] - mean) * rstd * ga.z + be.z, 0.f);
        v[4 * t + 3] = fmaxf((v[4 * t + 3] - mean) * rstd * ga.w + be.w, 0.f);
    }
}

// acc[t2] += W2[t2-th 16-row block] * act  with the packed weight image
//   w[((t2 * NT + t) * 64 + lane) * 4 + r] = W2[16*t2 + (lane & 15)][16*t + 4*(lane >> 4) + r]
// (LDS or global; one 16-byte read per lane feeds four k-steps).
template <int NT, int NT2>
SM_DEV void gemm_packed(const float *w, const float (&act)[NT * 4], f32x4 (&acc)[NT2], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int t2 = 0; t2 < NT2; ++t2) {
            const float4 a = ldg4(w + ((t2 * NT + t) * 64 + lane) * 4);
            acc[t2] = mfma16(a.x, act[4 * t + 0], acc[t2]);
            acc[t2] = mfma16(a.y, act[4 * t + 1], acc[t2]);
            acc[t2] = mfma16(a.z, act[4 * t + 2], acc[t2]);
            acc[t2] = mfma16(a.w, act[4 * t + 3], acc[t2]);
        }
    }
}

// Gaussian smearing centres of the reference (models/common.py:19), coeff = -0.5/(mu1-mu0)^2 = -0.5
__constant__ float c_rbf_centres[20] = {0.f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.25f, 2.5f, 2.75f, 3.f,
                                        3.5f, 4.f, 4.5f, 5.f, 5.5f, 6.f, 7.f, 8.f, 9.f, 10.f};

// B operand of the RBF product: lane group g supplies centre 4*s + g at k-step s (5 steps, G = 20)
SM_DEV void rbf_dlayout(float d, int g, float (&rb)[5]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const float u = d - c_rbf_centres[4 * s + g];
        rb[s] = expf(-0.5f * (u * u));
    }
}

// Philox4x32-10 (Salmon et al. 2011), counter-based; used for device-side noise
struct Philox {
    uint32_t k0, k1;
    SM_DEV static void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    SM_DEV void operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) const {
        uint32_t c[4] = {c0, c1, c2, c3};
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, a, b);
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};
SM_DEV float u01_open(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)
SM_DEV float u01_half(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }            // [0,1)
