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

// Diagnostic phase stamps (only in the -DSM_STAMPS build, tools/ only): 100 MHz real-time counter,
// taken after draining the wave's outstanding memory operations so that a stamp closes its phase.
#ifdef SM_STAMPS
#define SM_STAMP(buf, slot)                                                                        \
    do {                                                                                           \
        if ((buf) != nullptr && (threadIdx.x & 63) == 0) {                                         \
            unsigned long long t_;                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            (buf)[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (slot)] = t_; \
        }                                                                                          \
    } while (0)
// same without draining memory operations: when the wave's instruction stream reaches this point
#define SM_TICK(buf, slot)                                                                         \
    do {                                                                                           \
        if ((buf) != nullptr && (threadIdx.x & 63) == 0) {                                         \
            unsigned long long t_;                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            (buf)[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (slot)] = t_; \
        }                                                                                          \
    } while (0)
#else
#define SM_STAMP(buf, slot) do { } while (0)
#define SM_TICK(buf, slot) do { } while (0)
#endif

SM_DEV f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

SM_DEV float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
SM_DEV void stg4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// Cooperative global -> LDS copy of `n4` float4 by all `nthr` threads of the workgroup, eight
// independent 16-byte loads in flight per thread (a plain copy loop is serialised on the L2 latency).
SM_DEV void copy_to_lds(float *lds, const float *g, int n4, int tid, int nthr) {
    const float4 *src = reinterpret_cast<const float4 *>(g);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    constexpr int U = 8;
    int i = tid;
    for (; i + (U - 1) * nthr < n4; i += U * nthr) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * nthr];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * nthr] = v[u];
    }
    for (; i < n4; i += nthr) dst[i] = src[i];
}

// Asynchronous global -> LDS copy of `n4` float4 (a multiple of 64: whole 1 KB pieces) by LDS-DMA: every wave issues
// its share of the pieces back to back (global_load_lds_dwordx4: wave-uniform LDS base + 16 bytes per lane, per-lane
// source address) and carries on; nothing is staged in registers.  The data has landed after the issuing wave's
// s_waitcnt vmcnt(0) -- which __syncthreads() includes -- and the barrier.  (copy_to_lds above falls back to one
// dependent load -> store round per 12 KB for images smaller than its unrolled round: 7 us for 160 KB.)
SM_DEV void dma_to_lds(float *lds, const float *g, int n4, int wave, int nwave, int lane) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    for (int c = wu; c < n4 / 64; c += nwave)
        __builtin_amdgcn_global_load_lds((gbl_void *)(g + ((size_t)c * 64 + lane) * 4), (lds_void *)(lds + (size_t)c * 256), 16, 0, 0);
}

// The same for any length: whole 1 KB pieces by LDS-DMA, the tail (< 64 float4) by one plain load -> store per thread.
SM_DEV void image_to_lds(float *lds, const float *g, int n4, int tid, int wave, int nwave, int lane) {
    dma_to_lds(lds, g, n4, wave, nwave, lane);
    const int done = (n4 / 64) * 64;
    if (tid < n4 - done)
        reinterpret_cast<float4 *>(lds)[done + tid] = reinterpret_cast<const float4 *>(g)[done + tid];
}

// ---- cross-lane helpers without LDS traffic ------------------------------------------------------
// DPP row operations (within a 16-lane row) and the gfx950 permlane swaps (between rows); a
// __shfl_xor would lower to ds_bpermute_b32 (LDS pipe, ~100+ cycles of latency each).
template <int CTRL>
SM_DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7 - i inside each 8-lane half row
constexpr int DPP_ROW_MIRROR = 0x140;   // lane i <-> 15 - i inside each 16-lane row

// all-reduce over the SEGW (4, 8 or 16) consecutive lanes of a row segment
template <int SEGW>
SM_DEV float seg_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    if constexpr (SEGW >= 8) v += dpp_mov<DPP_HALF_MIRROR>(v);
    if constexpr (SEGW == 16) v += dpp_mov<DPP_ROW_MIRROR>(v);
    return v;
}
template <int SEGW>
SM_DEV float seg_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
    if constexpr (SEGW == 16) v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return v;
}
// v(lane) + v(lane ^ 16): v_permlane16_swap exchanges the odd rows of its first operand with the even
// rows of the second; with both operands = v the two results are {r0,r0,r2,r2} and {r1,r1,r3,r3}.
// (inline asm: the builtin folds swap(v, v) to (v, v); the s_nop covers the VALU-write hazard.)
SM_DEV float sum_xor16(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
SM_DEV float sum_xor32(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// sum over the four lane groups (same n, g = 0..3)
SM_DEV float sum_groups(float v) { return sum_xor32(sum_xor16(v)); }

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
    constexpr int CH = NT2 < 4 ? NT2 : 4;       // output blocks in flight: CH independent accumulators
#pragma unroll                                  // between two dependent MFMAs (dependent latency 40 cycles)
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int c0 = 0; c0 < NT2; c0 += CH) {
            float4 a[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = ldg4(w + (((c0 + i) * NT + t) * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[c0 + i] = mfma16(a[i].x, act[4 * t + 0], acc[c0 + i]);
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[c0 + i] = mfma16(a[i].y, act[4 * t + 1], acc[c0 + i]);
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[c0 + i] = mfma16(a[i].z, act[4 * t + 2], acc[c0 + i]);
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[c0 + i] = mfma16(a[i].w, act[4 * t + 3], acc[c0 + i]);
        }
    }
}

// e^x through v_exp_f32 (2^y, 1 ulp): relative error <= (|x| + 2) * 2^-24, enough for softmax terms and
// Gaussian smearing (large |x| only where the value is negligible); ~3 instructions instead of ~12.
SM_DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// Gaussian smearing centres of the reference (models/common.py:19): 0, 1..3 step 0.25, 3.5..6 step 0.5, 7..10
// step 1; coeff = -0.5/(mu1-mu0)^2 = -0.5.  Lane group g supplies centre 4*s + g at k-step s (5 steps, G = 20).
// The five centres of a lane are computed arithmetically ONCE per kernel and kept in registers: indexing a
// __constant__ table by lane costs a global load + full wait per centre inside the job loop.
SM_DEV float rbf_centre(int i) {
    return i == 0 ? 0.f : (i < 10 ? 1.f + 0.25f * (float)(i - 1) : (i < 16 ? 3.5f + 0.5f * (float)(i - 10) : 7.f + (float)(i - 16)));
}
SM_DEV void rbf_centres(int g, float (&cen)[5]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) cen[s] = rbf_centre(4 * s + g);
}
// B operand of the RBF product
SM_DEV void rbf_dlayout(float d, const float (&cen)[5], float (&rb)[5]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const float u = d - cen[s];
        rb[s] = fast_exp(-0.5f * (u * u));
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

// ---- fp32-accurate products on the bf16 matrix cores ------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the VECTOR rate and never co-executes with VALU work (rocprofv3:
// SQ_VALU_MFMA_COEXEC_CYCLES = 0 for every fp32-MFMA kernel here), so an fp32-MFMA kernel pays
// matrix cycles + vector cycles.  v_mfma_f32_16x16x32_bf16 is 16x faster per FLOP and does overlap.
// A float is split EXACTLY into three bf16 pieces by truncation (8 + 8 + 8 significand bits):
//     x = hi + mid + lo,   hi = x & 0xFFFF0000,  mid = (x - hi) & 0xFFFF0000,  lo = trunc_bf16(x - hi - mid)
// and x*w is the sum of the six piece products of total order <= 2 (hh, hm, mh, hl, lh, mm); each dropped
// one (ml, lm: below 2^-24 |x w|; ll: below 2^-32) is smaller than one ulp of the product.  Piece products are exact in fp32 and the
// MFMA accumulates in fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

SM_DEV f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// {low half = upper 16 bits of x0, high half = upper 16 bits of x1}
SM_DEV unsigned pack_hi16(unsigned x0, unsigned x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// eight floats (k = 0..7 of a lane's k-slot) -> three bf16x8 B fragments
SM_DEV void split3_bf16(const float (&v)[8], u32x4 &hi, u32x4 &mid, u32x4 &lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = v[2 * q], x1 = v[2 * q + 1];
        const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
        const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u);
        const float r1 = x1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
        const unsigned m0 = __builtin_bit_cast(unsigned, r0), m1 = __builtin_bit_cast(unsigned, r1);
        const float s0 = r0 - __builtin_bit_cast(float, m0 & 0xFFFF0000u);
        const float s1 = r1 - __builtin_bit_cast(float, m1 & 0xFFFF0000u);
        hi[q] = pack_hi16(u0, u1);
        mid[q] = pack_hi16(m0, m1);
        lo[q] = pack_hi16(__builtin_bit_cast(unsigned, s0), __builtin_bit_cast(unsigned, s1));
    }
}

// two floats -> one u32 (two bf16) per piece
SM_DEV void split3_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xFFFF0000u);
    const float r1 = x1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
    const unsigned m0 = __builtin_bit_cast(unsigned, r0), m1 = __builtin_bit_cast(unsigned, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m0 & 0xFFFF0000u);
    const float s1 = r1 - __builtin_bit_cast(float, m1 & 0xFFFF0000u);
    hi = pack_hi16(u0, u1);
    mid = pack_hi16(m0, m1);
    lo = pack_hi16(__builtin_bit_cast(unsigned, s0), __builtin_bit_cast(unsigned, s1));
}

// acc[t2] += W2[16-row block t2] * act with the split weight image (LDS)
//   w[(((piece * NT2 + t2) * NB + b) * 64 + lane) * 4 + q]  (u32, two bf16 each), NB = NT / 2 k-steps of 32;
//   element j of lane (m, g) at k-step b is piece(W2[16*t2 + m][16*(2b + (j >> 2)) + 4g + (j & 3)]):
//   the k order inside a step is chosen so that the B operand is the lane's own D-layout registers
//   8b .. 8b+7 (tiles 2b and 2b+1), i.e. again no data movement between Linear -> LN -> ReLU -> Linear.
template <int NT, int NT2>
SM_DEV void gemm_bf16x6(const unsigned *w, const float (&act)[NT * 4], f32x4 (&acc)[NT2], int lane) {
    constexpr int NB = NT / 2;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act[8 * b + j];
        u32x4 bh, bm, bl;
        split3_bf16(v, bh, bm, bl);
#pragma unroll
        for (int t2 = 0; t2 < NT2; ++t2) {
            const u32x4 ah = *reinterpret_cast<const u32x4 *>(w + (((0 * NT2 + t2) * NB + b) * 64 + lane) * 4);
            const u32x4 am = *reinterpret_cast<const u32x4 *>(w + (((1 * NT2 + t2) * NB + b) * 64 + lane) * 4);
            const u32x4 al = *reinterpret_cast<const u32x4 *>(w + (((2 * NT2 + t2) * NB + b) * 64 + lane) * 4);
            f32x4 c = acc[t2];
            c = mfma_bf16(al, bh, c);      // smallest terms first
            c = mfma_bf16(am, bm, c);
            c = mfma_bf16(ah, bl, c);
            c = mfma_bf16(am, bh, c);
            c = mfma_bf16(ah, bm, c);
            c = mfma_bf16(ah, bh, c);
            acc[t2] = c;
        }
    }
}

// The same product one OUTPUT tile at a time (activations split once, up front), so that a tile's epilogue
// (logits + softmax, weighted neighbour sums) is independent vector work that can issue under the next
// tile's MFMAs instead of after all of them.
template <int NT>
SM_DEV void split_act(const float (&act)[NT * 4], u32x4 (&bh)[NT / 2], u32x4 (&bm)[NT / 2], u32x4 (&bl)[NT / 2]) {
#pragma unroll
    for (int b = 0; b < NT / 2; ++b) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act[8 * b + j];
        split3_bf16(v, bh[b], bm[b], bl[b]);
    }
}
template <int NT, int NT2>
SM_DEV f32x4 tile_bf16x6(const unsigned *w, int t2, const u32x4 (&bh)[NT / 2], const u32x4 (&bm)[NT / 2],
                         const u32x4 (&bl)[NT / 2], f32x4 c, int lane) {
    constexpr int NB = NT / 2;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const u32x4 ah = *reinterpret_cast<const u32x4 *>(w + (((0 * NT2 + t2) * NB + b) * 64 + lane) * 4);
        const u32x4 am = *reinterpret_cast<const u32x4 *>(w + (((1 * NT2 + t2) * NB + b) * 64 + lane) * 4);
        const u32x4 al = *reinterpret_cast<const u32x4 *>(w + (((2 * NT2 + t2) * NB + b) * 64 + lane) * 4);
        c = mfma_bf16(al, bh[b], c);      // smallest terms first
        c = mfma_bf16(am, bm[b], c);
        c = mfma_bf16(ah, bl[b], c);
        c = mfma_bf16(am, bh[b], c);
        c = mfma_bf16(ah, bm[b], c);
        c = mfma_bf16(ah, bh[b], c);
    }
    return c;
}

// Attention weights of one edge tile from its key rows (D layout) and the centre atoms' query rows:
// logit of head 2t + (g >> 1) = (q . k over the head's 8 dims) / sqrt(8): 4 dims in this lane group and 4 in
// the partner group g ^ 1.  One v_permlane16_swap pairs head blocks t and t + NT/2, so that afterwards the
// even lane groups own the logits of blocks 0 .. NT/2-1 and the odd groups those of NT/2 .. NT-1 (no lane
// repeats another's softmax); the softmax runs over the SEGW neighbour slots of the atom (a DPP row segment).
// alpha[t] belongs to head block t + (NT/2) * (g & 1).
template <int NT, int SEGW>
SM_DEV float attention_weight_pair(float4 qa, float4 qb, f32x4 ka, f32x4 kb, bool ok, float &mx, float &s) {
    float pa = qa.x * ka[0] + qa.y * ka[1] + qa.z * ka[2] + qa.w * ka[3];
    float pb = qb.x * kb[0] + qb.y * kb[1] + qb.z * kb[2] + qb.w * kb[3];
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(pa), "+v"(pb));
    float p = pa + pb;
    p = ok ? p * 0.35355339059327373f : -INFINITY;
    mx = seg_max<SEGW>(p);
    const float e = ok ? fast_exp(p - mx) : 0.f;
    s = seg_sum<SEGW>(e);
    return s > 0.f ? e * __builtin_amdgcn_rcpf(s) : 0.f;
}
template <int NT, int SEGW>
SM_DEV float attention_weight_pair(float4 qa, float4 qb, f32x4 ka, f32x4 kb, bool ok) {
    float pa = qa.x * ka[0] + qa.y * ka[1] + qa.z * ka[2] + qa.w * ka[3];
    float pb = qb.x * kb[0] + qb.y * kb[1] + qb.z * kb[2] + qb.w * kb[3];
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(pa), "+v"(pb));
    float p = pa + pb;
    p = ok ? p * 0.35355339059327373f : -INFINITY;
    const float mx = seg_max<SEGW>(p);
    const float e = ok ? fast_exp(p - mx) : 0.f;
    const float s = seg_sum<SEGW>(e);
    return s > 0.f ? e * __builtin_amdgcn_rcpf(s) : 0.f;
}
template <int NT, int SEGW>
SM_DEV void attention_weights(const float *qrow, const f32x4 (&k)[NT], bool ok, int g, float (&alpha)[NT / 2 > 0 ? NT / 2 : 1]) {
    static_assert(NT % 2 == 0, "head blocks are paired");
#pragma unroll
    for (int t = 0; t < NT / 2; ++t)
        alpha[t] = attention_weight_pair<NT, SEGW>(ldg4(qrow + 16 * t + 4 * g), ldg4(qrow + 16 * (t + NT / 2) + 4 * g), k[t], k[t + NT / 2], ok);
}
