// Node kernels: every per-atom Linear / MLP of the score network as one generic MFMA kernel.
//
// Reference semantics (paths relative to the reference repository):
//   MLP = Linear -> LayerNorm -> ReLU -> Linear          models/common.py:47-67
//   hq_func / xq_func (queries)                          models/uni_transformer.py:74,144
//   node_output MLP + residual                           models/uni_transformer.py:82-88
//   v_inference = Linear -> ShiftedSoftplus -> Linear    models/molopt_score_model.py:262-266,305
//   the per-node halves W_i h_i, W_j h_j of the edge MLPs' first Linear (see sm_edge.h)
//
// One workgroup owns 16 atoms (one D-layout tile, sm_device.h); its waves split the OUTPUT
// features (16-row weight blocks) between them and stream their weight rows straight from
// global memory/L2 as the MFMA A operand (row-major [out][in], one 16-byte load = 4 k-steps),
// while the 16 x K activation tile is the B operand held in registers by every wave.
// A two-layer job exchanges the hidden tile through LDS so each wave sees all H hidden features.
#pragma once
#include "sm_device.h"

enum NodeMode { NODE_LINEAR = 0, NODE_LN_RELU = 1, NODE_SSP = 2 };

struct NodeJob {
    const float *in0;     // [N][H]
    const float *in1;     // [N][H] second half of the input (K = 2H) or nullptr (K = H)
    const float *w1;      // [n_out1][ldw1] row-major, n_out1 % 16 == 0
    const float *add_mol; // per-molecule additive term [B][n_out1] (bias folded in) or nullptr
    const float *b1;      // [n_out1] or nullptr
    const float *ln_g;    // NODE_LN_RELU
    const float *ln_b;
    const float *w2;      // [n_out2 padded to 16][H] row-major
    const float *b2;      // [n_out2 padded to 16]
    const float *resid;   // [N][H] added to the output, or nullptr
    float *out;           // [N][ld_out]
    int ldw1, n_out1, mode, n_out2, ld_out, n_store;
};

struct NodeArgs {
    NodeJob job[2];       // blockIdx.y selects
    const int *mol_of;    // [N]
    int n_atoms;
};

template <int H>
__global__ void __launch_bounds__(512)
node_mlp_kernel(NodeArgs args) {
    constexpr int NT = H / 16;
    constexpr int XS = H + 16;                        // LDS row stride of the exchange tile
    __shared__ __attribute__((aligned(16))) float xch[16 * XS];
    const NodeJob &J = args.job[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int atom_raw = blockIdx.x * 16 + n;
    const bool atom_ok = atom_raw < args.n_atoms;
    const int atom = atom_ok ? atom_raw : args.n_atoms - 1;
    const bool two = J.in1 != nullptr;

    // B operand: the 16 x K input tile in D layout (k-step 4t + r <- register 4t + r)
    float bin[2 * NT * 4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 v = ldg4(J.in0 + (size_t)atom * H + 16 * t + 4 * g);
        bin[4 * t] = v.x; bin[4 * t + 1] = v.y; bin[4 * t + 2] = v.z; bin[4 * t + 3] = v.w;
    }
    if (two) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 v = ldg4(J.in1 + (size_t)atom * H + 16 * t + 4 * g);
            bin[4 * (NT + t)] = v.x; bin[4 * (NT + t) + 1] = v.y; bin[4 * (NT + t) + 2] = v.z; bin[4 * (NT + t) + 3] = v.w;
        }
    }
    const int mol = J.add_mol ? args.mol_of[atom] : 0;
    const int nt1 = J.n_out1 / 16;

    for (int t1 = wave; t1 < nt1; t1 += nwave) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (J.add_mol) {
            const float4 s = ldg4(J.add_mol + (size_t)mol * J.n_out1 + 16 * t1 + 4 * g);
            acc = f32x4{s.x, s.y, s.z, s.w};
        }
        if (J.b1) {
            const float4 s = ldg4(J.b1 + 16 * t1 + 4 * g);
            acc += f32x4{s.x, s.y, s.z, s.w};
        }
        const float *wrow = J.w1 + (size_t)(16 * t1 + n) * J.ldw1 + 4 * g;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 w = ldg4(wrow + 16 * t);
            acc = mfma16(w.x, bin[4 * t + 0], acc);
            acc = mfma16(w.y, bin[4 * t + 1], acc);
            acc = mfma16(w.z, bin[4 * t + 2], acc);
            acc = mfma16(w.w, bin[4 * t + 3], acc);
        }
        if (two) {
#pragma unroll
            for (int t = NT; t < 2 * NT; ++t) {
                const float4 w = ldg4(wrow + 16 * t);
                acc = mfma16(w.x, bin[4 * t + 0], acc);
                acc = mfma16(w.y, bin[4 * t + 1], acc);
                acc = mfma16(w.z, bin[4 * t + 2], acc);
                acc = mfma16(w.w, bin[4 * t + 3], acc);
            }
        }
        if (J.mode == NODE_LINEAR) {
            if (atom_ok) stg4(J.out + (size_t)atom * J.ld_out + 16 * t1 + 4 * g, float4{acc[0], acc[1], acc[2], acc[3]});
        } else {
            stg4(xch + n * XS + 16 * t1 + 4 * g, float4{acc[0], acc[1], acc[2], acc[3]});
        }
    }
    if (J.mode == NODE_LINEAR) return;
    __syncthreads();

    float hid[NT * 4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 v = ldg4(xch + n * XS + 16 * t + 4 * g);
        hid[4 * t] = v.x; hid[4 * t + 1] = v.y; hid[4 * t + 2] = v.z; hid[4 * t + 3] = v.w;
    }
    if (J.mode == NODE_LN_RELU) {
        ln_relu_dlayout<NT>(hid, J.ln_g, J.ln_b, g);
    } else {   // ShiftedSoftplus: softplus(x) - ln 2   (models/common.py:39-45; torch threshold 20)
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) {
            const float v = hid[i];
            hid[i] = (v > 20.f ? v : log1pf(expf(v))) - 0.6931471805599453f;
        }
    }
    const int nt2 = (J.n_out2 + 15) / 16;
    for (int t2 = wave; t2 < nt2; t2 += nwave) {
        const float4 bb = ldg4(J.b2 + 16 * t2 + 4 * g);
        f32x4 acc = {bb.x, bb.y, bb.z, bb.w};
        const float *wrow = J.w2 + (size_t)(16 * t2 + n) * H + 4 * g;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 w = ldg4(wrow + 16 * t);
            acc = mfma16(w.x, hid[4 * t + 0], acc);
            acc = mfma16(w.y, hid[4 * t + 1], acc);
            acc = mfma16(w.z, hid[4 * t + 2], acc);
            acc = mfma16(w.w, hid[4 * t + 3], acc);
        }
        if (!atom_ok) continue;
        const int f0 = 16 * t2 + 4 * g;
        if (J.resid) {
            const float4 r = ldg4(J.resid + (size_t)atom * H + f0);
            acc += f32x4{r.x, r.y, r.z, r.w};
        }
        if (f0 + 4 <= J.n_store && (J.ld_out & 3) == 0) {
            stg4(J.out + (size_t)atom * J.ld_out + f0, float4{acc[0], acc[1], acc[2], acc[3]});
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (f0 + r < J.n_store) J.out[(size_t)atom * J.ld_out + f0 + r] = acc[r];
        }
    }
}
