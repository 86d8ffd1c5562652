// Edge kernels: the per-edge MLPs + neighbour softmax + aggregation of one attention layer.
//
// Reference semantics (paths relative to the reference repository):
//   x2h: BaseX2HAttLayer.forward  models/uni_transformer.py:48-81   (k, v, softmax, sum; the
//        node_output MLP + residual, :82-90, runs in the node kernel)
//   h2x: BaseH2XAttLayer.forward  models/uni_transformer.py:121-151 (k, v (x) rel_x, softmax, sum)
//   RBF: GaussianSmearing.forward models/common.py:26-28, distances from uni_transformer.py:300-303
//
// Formulation.  The first Linear of each edge MLP acts on [rbf(20) | h_i | h_j | inv_shape_i]; it is
// split as  W_r rbf + (W_i h_i + W_s s_i + b) + W_j h_j : the two bracketed terms are per-NODE
// products computed once per atom by the node kernel ("pre" buffer: [N][4][H] = A_k, B_k, A_v, B_v)
// and gathered here as the initial accumulator; only the 20-wide RBF product and the H x H second
// Linear are evaluated per edge.  All products run on v_mfma_f32_16x16x4_f32 (exact fp32) with the
// weights as the A operand from LDS and 16 edge columns per wave in the D layout (sm_device.h).
//
// Work split: one job = the KP neighbour slots of max(1, 16/KP) centre atoms = max(1, KP/16) tiles of
// 16 edge columns.  Jobs are dealt wave-major across workgroups (job = block + grid * wave) so that
// the SIMDs of all CUs receive an equal share when there are only a few thousand jobs.
#pragma once
#include "sm_device.h"

struct EdgeArgs {
    const float *blob;      // packed weights of this layer/kernel (see EdgeBlob)
    const float *pre;       // [N][4][H]  A_k | B_k | A_v | B_v   (node pre-products)
    const float *q;         // [N][H]
    const float *x;         // [N][3]   current coordinates (this layer's input)
    const int *nbr;         // [N][KP]  neighbour atom index or -1
    const float *ew;        // [N][KP]  edge weight sigma(...)
    float *out;             // x2h: [N][H] attention output; h2x: [N][16][3] rows 4g+r (permuted heads)
    int n_atoms;
    int ld_pre;             // row stride of `pre` in floats (4H, or 8H when two products share a buffer)
    unsigned long long *stamps;   // diagnostic build only
};

// LDS image of one edge kernel's weights, in floats.
template <int H, bool H2X>
struct EdgeBlob {
    static constexpr int NT = H / 16;
    static constexpr int NT2V = H2X ? 1 : NT;
    static constexpr int WR = NT * 5 * 64;          // [NT][5][64]   W1[:, 0:20] as A fragments
    static constexpr int W2K = NT * NT * 256;       // [NT][NT][64][4]
    static constexpr int W2V = NT2V * NT * 256;
    // offsets
    static constexpr int K_WR = 0;
    static constexpr int K_W2 = K_WR + WR;
    static constexpr int K_G = K_W2 + W2K;          // gamma[H], beta[H], b2[H]
    static constexpr int K_B = K_G + H;
    static constexpr int K_B2 = K_B + H;
    static constexpr int V_WR = K_B2 + H;
    static constexpr int V_W2 = V_WR + WR;
    static constexpr int V_G = V_W2 + W2V;
    static constexpr int V_B = V_G + H;
    static constexpr int V_B2 = V_B + H;
    static constexpr int TOTAL = V_B2 + NT2V * 16;  // floats (multiple of 4)
};

// hidden = ReLU(LN(A_i + B_j + W_r rbf))  for one tile; returns it in D layout.
template <int NT>
SM_DEV void edge_hidden(const float *a_row, const float *b_row, const float (&rb)[5],
                        const float *wr, const float *gamma, const float *beta,
                        int lane, int g, float (&hid)[NT * 4]) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 a = ldg4(a_row + 16 * t + 4 * g);
        const float4 b = ldg4(b_row + 16 * t + 4 * g);
        acc[t] = f32x4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(wr[(t * 5 + s) * 64 + lane], rb[s], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        hid[4 * t + 0] = acc[t][0]; hid[4 * t + 1] = acc[t][1];
        hid[4 * t + 2] = acc[t][2]; hid[4 * t + 3] = acc[t][3];
    }
    ln_relu_dlayout<NT>(hid, gamma, beta, g);
}

template <int H, int KP, bool H2X>
__global__ void __launch_bounds__(768)
edge_attention_kernel(EdgeArgs a) {
    using BL = EdgeBlob<H, H2X>;
    constexpr int NT = BL::NT;
    constexpr int NT2V = BL::NT2V;
    constexpr int APJ = (KP >= 16) ? 1 : 16 / KP;   // atoms per job
    constexpr int TPJ = (KP >= 16) ? KP / 16 : 1;   // tiles per job
    constexpr int SEGW = (KP >= 16) ? 16 : KP;      // lanes (columns) of one atom inside a tile
    extern __shared__ __attribute__((aligned(16))) float lds[];

    SM_STAMP(a.stamps, 0);
    copy_to_lds(lds, a.blob, BL::TOTAL / 4, threadIdx.x, blockDim.x);
    __syncthreads();
    SM_STAMP(a.stamps, 1);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    float cen[5];
    rbf_centres(g, cen);
    const int njobs = (a.n_atoms + APJ - 1) / APJ;
    const float inv_sqrt_dh = 0.35355339059327373f;   // 1/sqrt(8)

    for (int job = blockIdx.x + gridDim.x * wave; job < njobs; job += gridDim.x * nwave) {
        const int atom_raw = job * APJ + (KP >= 16 ? 0 : n / SEGW);
        const bool atom_ok = atom_raw < a.n_atoms;
        const int atom = atom_ok ? atom_raw : a.n_atoms - 1;
        const float xi0 = a.x[atom * 3 + 0], xi1 = a.x[atom * 3 + 1], xi2 = a.x[atom * 3 + 2];
        const float *pre_i = a.pre + (size_t)atom * a.ld_pre;

        // ---- pass 1: keys and logits of every tile of the job --------------------------------
        float logit[TPJ][NT];
        int nb[TPJ];
        float rel[TPJ][3];
        float mx[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) mx[t] = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < TPJ; ++tt) {
            // the weight image in LDS is loop-invariant; without this the compiler hoists all of its
            // reads out of the job loop (hundreds of live registers, spilled to scratch)
            asm volatile("" ::: "memory");
            const int slot = (KP >= 16) ? tt * 16 + n : n % SEGW;
            const int jraw = a.nbr[atom * KP + slot];
            const bool ok = atom_ok && jraw >= 0;
            const int j = ok ? jraw : atom;
            nb[tt] = ok ? j : -1;
            rel[tt][0] = xi0 - a.x[j * 3 + 0];
            rel[tt][1] = xi1 - a.x[j * 3 + 1];
            rel[tt][2] = xi2 - a.x[j * 3 + 2];
            const float d = sqrtf(rel[tt][0] * rel[tt][0] + rel[tt][1] * rel[tt][1] + rel[tt][2] * rel[tt][2]);
            float rb[5];
            rbf_dlayout(d, cen, rb);
            float hid[NT * 4];
            edge_hidden<NT>(pre_i, a.pre + (size_t)j * a.ld_pre + H, rb, lds + BL::K_WR, lds + BL::K_G,
                            lds + BL::K_B, lane, g, hid);
            SM_STAMP(a.stamps, 2);
            f32x4 kacc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 b2 = ldg4(lds + BL::K_B2 + 16 * t + 4 * g);
                kacc[t] = f32x4{b2.x, b2.y, b2.z, b2.w};
            }
            gemm_packed<NT, NT>(lds + BL::K_W2, hid, kacc, lane);
            SM_STAMP(a.stamps, 3);
            // logit of head 2t + (g >> 1): 4 dims here + 4 dims in the partner lane group (g ^ 1)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 qq = ldg4(a.q + (size_t)atom * H + 16 * t + 4 * g);
                float p = qq.x * kacc[t][0] + qq.y * kacc[t][1] + qq.z * kacc[t][2] + qq.w * kacc[t][3];
                p = sum_xor16(p);
                p = ok ? p * inv_sqrt_dh : -INFINITY;
                logit[tt][t] = p;
                mx[t] = fmaxf(mx[t], p);
            }
        }
        // ---- softmax over the KP slots of each atom (lanes of the segment, then tiles) ---------
        float den[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            mx[t] = seg_max<SEGW>(mx[t]);
            float s = 0.f;
#pragma unroll
            for (int tt = 0; tt < TPJ; ++tt) {
                const float e = (nb[tt] >= 0) ? expf(logit[tt][t] - mx[t]) : 0.f;
                logit[tt][t] = e;
                s += e;
            }
            s = seg_sum<SEGW>(s);
            den[t] = s > 0.f ? 1.0f / s : 0.f;
        }

        SM_STAMP(a.stamps, 4);
        // ---- pass 2: values, weighted by alpha * e_w, summed over the slots ---------------------
        constexpr int NOUT = H2X ? 12 : NT * 4;
        float osum[NOUT];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) osum[i] = 0.f;
#pragma unroll
        for (int tt = 0; tt < TPJ; ++tt) {
            asm volatile("" ::: "memory");
            const int slot = (KP >= 16) ? tt * 16 + n : n % SEGW;
            const bool ok = nb[tt] >= 0;
            const int j = ok ? nb[tt] : atom;
            const float d = sqrtf(rel[tt][0] * rel[tt][0] + rel[tt][1] * rel[tt][1] + rel[tt][2] * rel[tt][2]);
            float rb[5];
            rbf_dlayout(d, cen, rb);
            float hid[NT * 4];
            edge_hidden<NT>(pre_i + 2 * H, a.pre + (size_t)j * a.ld_pre + 3 * H, rb, lds + BL::V_WR,
                            lds + BL::V_G, lds + BL::V_B, lane, g, hid);
            f32x4 vacc[NT2V];
#pragma unroll
            for (int t = 0; t < NT2V; ++t) {
                const float4 b2 = ldg4(lds + BL::V_B2 + 16 * t + 4 * g);
                vacc[t] = f32x4{b2.x, b2.y, b2.z, b2.w};
            }
            SM_STAMP(a.stamps, 5);
            gemm_packed<NT, NT2V>(lds + BL::V_W2, hid, vacc, lane);
            SM_STAMP(a.stamps, 6);
            const float w = ok ? a.ew[atom * KP + slot] : 0.f;
            if constexpr (!H2X) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float aw = logit[tt][t] * den[t] * w;
                    osum[4 * t + 0] += aw * vacc[t][0]; osum[4 * t + 1] += aw * vacc[t][1];
                    osum[4 * t + 2] += aw * vacc[t][2]; osum[4 * t + 3] += aw * vacc[t][3];
                }
            } else {
                // value row 4g + r belongs to head 2*((NT/2)*(g&1) + r) + (g>>1), whose alpha this lane
                // holds in logit[.][(NT/2)*(g&1) + r]; rows with r >= NT/2 are zero padding.
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float al = 0.f;
                    if (r < NT / 2) {
                        const float lo = logit[tt][r] * den[r];
                        const float hi = logit[tt][(NT / 2 + r) % NT] * den[(NT / 2 + r) % NT];
                        al = (g & 1) ? hi : lo;
                    }
                    const float av = al * w * vacc[0][r];
                    osum[3 * r + 0] += av * rel[tt][0];
                    osum[3 * r + 1] += av * rel[tt][1];
                    osum[3 * r + 2] += av * rel[tt][2];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NOUT; ++i) osum[i] = seg_sum<SEGW>(osum[i]);
        if (atom_ok && (n % SEGW) == 0) {
            if constexpr (!H2X) {
                float *o = a.out + (size_t)atom * H;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    stg4(o + 16 * t + 4 * g, float4{osum[4 * t], osum[4 * t + 1], osum[4 * t + 2], osum[4 * t + 3]});
            } else {
                float *o = a.out + (size_t)atom * 48 + 12 * g;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    stg4(o + 4 * i, float4{osum[4 * i], osum[4 * i + 1], osum[4 * i + 2], osum[4 * i + 3]});
            }
        }
        SM_STAMP(a.stamps, 7);
    }
}
