// Node kernels: every per-atom Linear / MLP of the score network.  The default path is the bf16x6 family in
// the second half of this file (node_linear6_kernel, node_chain6_kernel, node_prologue6_kernel: exactly split
// operands on the bf16 matrix cores, activations kept in LDS as ready-made fragments); the fp32-MFMA kernels
// described next are the first version, kept behind shapemol_set_option("lin_bf16" / "chain_bf16", 0).
//
// Reference semantics (paths relative to the reference repository):
//   MLP = Linear -> LayerNorm -> ReLU -> Linear          models/common.py:47-67
//   hq_func / xq_func (queries)                          models/uni_transformer.py:74,144
//   node_output MLP + residual                           models/uni_transformer.py:82-88
//   v_inference = Linear -> ShiftedSoftplus -> Linear    models/molopt_score_model.py:262-266,305
//   the per-node halves W_i h_i, W_j h_j of the edge MLPs' first Linear (see sm_edge.h)
//
// Weights are pre-packed on the host into MFMA A-fragment images
//     img[((t2 * NTK + t) * 64 + lane) * 4 + r] = W[16*t2 + (lane & 15)][16*t + 4*(lane >> 4) + r]
// (NTK = K / 16).  Both kernels are WEIGHT-STATIONARY IN REGISTERS: a wave owns one 16-row block of
// output features, reads its A fragments once (8 KB, perfectly coalesced) and keeps them in VGPRs
// while 16-atom column tiles stream through as the B operand in the D layout (sm_device.h).
// With only ~5.5k atoms per batch the problem is latency- and balance-bound, not FLOP-bound:
// no LDS weight copy, no barrier in the linear kernel, grids sized to the CU count.
//
//   node_linear_kernel : out[N][OT*16] = in[N][H] W^T (+ per-molecule term).  16 waves = 16 output
//                        blocks per workgroup; the workgroup walks the column tiles of its atom group.
//   node_mlp2_kernel   : Linear(K->H) -> LN+ReLU | SSP -> Linear(H->R2) [+ residual] for 16/NT column
//                        tiles per workgroup; the NT waves of a team split the hidden features and
//                        exchange the hidden tile through LDS once (LayerNorm needs all of them).
#pragma once
#include "sm_device.h"

enum NodeMode { NODE_LN_RELU = 1, NODE_SSP = 2 };
constexpr int kNodeThreads = 1024;

struct NodeLinArgs {
    const float *in;        // [N][H]
    const float *wimg;      // packed image of [OT*16][H]
    const float *add_mol;   // per-molecule additive term [B][OT*16] (bias folded in) or nullptr
    const int *mol_of;      // [N]
    float *out;             // [N][OT*16]
    int n_atoms, n_out_tiles, tiles_per_group;
    int ld_add, ld_out;           // row strides of add_mol and out (floats)
    unsigned long long *stamps;   // diagnostic build only
    int nwave;                    // waves per workgroup (node_linear16_kernel reads it here instead of blockDim)
};

constexpr int kLinChunk = 8;      // column tiles staged in LDS at a time (shared by the workgroup's waves)

template <int H>
__global__ void __launch_bounds__(kNodeThreads)
node_linear_kernel(NodeLinArgs a) {
    constexpr int NT = H / 16;
    constexpr int XS = H + 16;
    __shared__ __attribute__((aligned(16))) float tiles[kLinChunk * 16 * XS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nwave = blockDim.x >> 6;
    const int ogroups = (a.n_out_tiles + nwave - 1) / nwave;
    const int ot_raw = (blockIdx.x % ogroups) * nwave + wave;
    const bool ot_ok = ot_raw < a.n_out_tiles;
    const int ot = ot_ok ? ot_raw : a.n_out_tiles - 1;
    const int ag = blockIdx.x / ogroups;
    const int n_ct = (a.n_atoms + 15) / 16;
    const int ct0 = ag * a.tiles_per_group, ct1 = min(ct0 + a.tiles_per_group, n_ct);
    SM_STAMP(a.stamps, 0);

    float4 w[NT];                                                   // this wave's weight block, resident
#pragma unroll
    for (int t = 0; t < NT; ++t) w[t] = ldg4(a.wimg + ((size_t)(ot * NT + t) * 64 + lane) * 4);

    for (int cb = ct0; cb < ct1; cb += kLinChunk) {
        const int nc = min(kLinChunk, ct1 - cb);
        __syncthreads();                                            // previous chunk fully consumed
        // the workgroup's activation tiles -> LDS once (every wave needs all of them as its B operand)
        for (int idx = threadIdx.x; idx < nc * 16 * (H / 4); idx += blockDim.x) {
            const int ar = idx / (H / 4), c4 = idx % (H / 4);
            const int at = min(cb * 16 + ar, a.n_atoms - 1);
            stg4(tiles + ar * XS + 4 * c4, ldg4(a.in + (size_t)at * H + 4 * c4));
        }
        __syncthreads();
        // per-molecule term of the first tile; the next tile's is fetched while this one multiplies
        float4 add_cur = {0.f, 0.f, 0.f, 0.f}, add_nxt = {0.f, 0.f, 0.f, 0.f};
        if (a.add_mol) add_cur = ldg4(a.add_mol + (size_t)a.mol_of[min(cb * 16 + n, a.n_atoms - 1)] * a.ld_add + 16 * ot + 4 * g);
        for (int c = 0; c < nc; ++c) {
            if (a.add_mol && c + 1 < nc)
                add_nxt = ldg4(a.add_mol + (size_t)a.mol_of[min((cb + c + 1) * 16 + n, a.n_atoms - 1)] * a.ld_add + 16 * ot + 4 * g);
            const float *xrow = tiles + (c * 16 + n) * XS;
            f32x4 acc = {add_cur.x, add_cur.y, add_cur.z, add_cur.w};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 x = ldg4(xrow + 16 * t + 4 * g);
                acc = mfma16(w[t].x, x.x, acc);
                acc = mfma16(w[t].y, x.y, acc);
                acc = mfma16(w[t].z, x.z, acc);
                acc = mfma16(w[t].w, x.w, acc);
            }
            const int atom = (cb + c) * 16 + n;
            if (ot_ok && atom < a.n_atoms) stg4(a.out + (size_t)atom * a.ld_out + 16 * ot + 4 * g, float4{acc[0], acc[1], acc[2], acc[3]});
            add_cur = add_nxt;
        }
    }
    SM_STAMP(a.stamps, 2);
}

// The same product on the bf16 matrix cores with exactly split operands (gemm_bf16x6 arithmetic, sm_device.h):
// the wave's weight block is resident as three bf16 piece fragments (12 KB per wave, host-split image
//   wimg[(((ot * 3 + piece) * NB + b) * 64 + lane) * 4 + q], element order as in gemm_bf16x6),
// the workgroup splits each activation tile once while staging it in LDS (one pair of float4 per thread)
// and every wave reads ready-made B fragments: no vector work in the multiply loop at all.
constexpr int kLin6Chunk = 8;     // column tiles staged at a time: 3 * H * 32 bytes each (12 KB at H = 128)

template <int H>
__global__ void __launch_bounds__(kNodeThreads)
node_linear6_kernel(NodeLinArgs a) {
    constexpr int NB = H / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lin6_lds[];
    u32x4 *frag = reinterpret_cast<u32x4 *>(lin6_lds);              // [tile][piece][NB][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nwave = a.nwave;      // (reading blockDim costs two dependent loads from the implicit kernel arguments at the head of the launch)
    const int ogroups = (a.n_out_tiles + nwave - 1) / nwave;
    const int ot_raw = (blockIdx.x % ogroups) * nwave + wave;
    const bool ot_ok = ot_raw < a.n_out_tiles;
    const int ot = ot_ok ? ot_raw : a.n_out_tiles - 1;
    const int ag = blockIdx.x / ogroups;
    const int n_ct = (a.n_atoms + 15) / 16;
    const int ct0 = ag * a.tiles_per_group, ct1 = min(ct0 + a.tiles_per_group, n_ct);

    u32x4 w[3][NB];
    {
        const u32x4 *wi = reinterpret_cast<const u32x4 *>(a.wimg) + (size_t)ot * 3 * NB * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int b = 0; b < NB; ++b) w[p][b] = wi[(p * NB + b) * 64];
    }

    for (int cb = ct0; cb < ct1; cb += kLin6Chunk) {
        const int nc = min(kLin6Chunk, ct1 - cb);
        __syncthreads();                                            // previous chunk fully consumed
        for (int idx = threadIdx.x; idx < nc * NB * 64; idx += nwave * 64) {
            const int sl = idx & 63, sb = (idx >> 6) % NB, sc = idx / (64 * NB);
            const int at = min((cb + sc) * 16 + (sl & 15), a.n_atoms - 1);
            const float *src = a.in + (size_t)at * H + 32 * sb + 4 * (sl >> 4);
            const float4 v0 = ldg4(src), v1 = ldg4(src + 16);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            u32x4 hi, mid, lo;
            split3_bf16(v, hi, mid, lo);
            u32x4 *dst = frag + ((size_t)(sc * 3) * NB + sb) * 64 + sl;
            dst[0] = hi; dst[NB * 64] = mid; dst[2 * NB * 64] = lo;
        }
        __syncthreads();
        for (int c = 0; c < nc; c += 2) {                           // two tiles at a time: independent MFMA chains
            const bool two = c + 1 < nc;
            const int atom0 = (cb + c) * 16 + n, atom1 = atom0 + 16;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (a.add_mol) {
                const float4 t0 = ldg4(a.add_mol + (size_t)a.mol_of[min(atom0, a.n_atoms - 1)] * a.ld_add + 16 * ot + 4 * g);
                const float4 t1 = ldg4(a.add_mol + (size_t)a.mol_of[min(atom1, a.n_atoms - 1)] * a.ld_add + 16 * ot + 4 * g);
                acc0 = f32x4{t0.x, t0.y, t0.z, t0.w}; acc1 = f32x4{t1.x, t1.y, t1.z, t1.w};
            }
            const u32x4 *f0 = frag + (size_t)(c * 3) * NB * 64 + lane;
            const u32x4 *f1 = frag + (size_t)((two ? c + 1 : c) * 3) * NB * 64 + lane;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const u32x4 h0 = f0[b * 64], m0 = f0[(NB + b) * 64], l0 = f0[(2 * NB + b) * 64];
                const u32x4 h1 = f1[b * 64], m1 = f1[(NB + b) * 64], l1 = f1[(2 * NB + b) * 64];
                acc0 = mfma_bf16(w[2][b], h0, acc0); acc1 = mfma_bf16(w[2][b], h1, acc1);      // smallest terms first
                acc0 = mfma_bf16(w[1][b], m0, acc0); acc1 = mfma_bf16(w[1][b], m1, acc1);
                acc0 = mfma_bf16(w[0][b], l0, acc0); acc1 = mfma_bf16(w[0][b], l1, acc1);
                acc0 = mfma_bf16(w[1][b], h0, acc0); acc1 = mfma_bf16(w[1][b], h1, acc1);
                acc0 = mfma_bf16(w[0][b], m0, acc0); acc1 = mfma_bf16(w[0][b], m1, acc1);
                acc0 = mfma_bf16(w[0][b], h0, acc0); acc1 = mfma_bf16(w[0][b], h1, acc1);
            }
            if (ot_ok && atom0 < a.n_atoms) stg4(a.out + (size_t)atom0 * a.ld_out + 16 * ot + 4 * g, float4{acc0[0], acc0[1], acc0[2], acc0[3]});
            if (ot_ok && two && atom1 < a.n_atoms) stg4(a.out + (size_t)atom1 * a.ld_out + 16 * ot + 4 * g, float4{acc1[0], acc1[1], acc1[2], acc1[3]});
        }
    }
}

struct NodeMlpArgs {
    const float *in0;      // [N][H]
    const float *in1;      // [N][H] second half of the input when K = 2H, else unused
    const float *w1img;    // packed image of [H][K]
    const float *b1;       // [H]
    const float *ln_g, *ln_b;
    const float *w2img;    // packed image of [16*nt2][H]
    const float *b2;       // [16*nt2]
    const float *resid;    // [N][H] added to the output, or nullptr
    float *out;            // [N][ld_out]
    int ld_out, n_store, mode, nt2, n_atoms;
};

template <int H>
struct NodeMlpLds {
    static constexpr int NT = H / 16;
    static constexpr int TEAMS = (16 / NT) < 4 ? (16 / NT) : 4;   // column tiles per workgroup (LDS budget)
    static constexpr int XS = H + 16;                    // row stride of the exchange tile
    static constexpr int TOTAL = TEAMS * 16 * XS;        // floats
};

template <int H, int KT>
__global__ void __launch_bounds__(kNodeThreads)
node_mlp2_kernel(NodeMlpArgs a) {
    constexpr int NT = H / 16;
    using L = NodeMlpLds<H>;
    __shared__ __attribute__((aligned(16))) float xch[L::TOTAL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    // waves beyond TEAMS * NT (only when NT < 4) mirror an existing team: identical values, stores disabled
    const int team = (wave / NT) % L::TEAMS, ot = wave % NT;
    const bool mirror = (wave / NT) >= L::TEAMS;
    const int ct = blockIdx.x * L::TEAMS + team;
    const bool tile_ok = ct * 16 < a.n_atoms;
    const int atom_raw = ct * 16 + n;
    const bool atom_ok = atom_raw < a.n_atoms && !mirror;
    const int atom = atom_raw < a.n_atoms ? atom_raw : a.n_atoms - 1;
    float *xrow = xch + (team * 16 + n) * L::XS;

    // ---- first Linear: this wave's 16 hidden features, K in phases of H ------------------------
    {
        const float4 b = ldg4(a.b1 + 16 * ot + 4 * g);
        f32x4 acc = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int kp = 0; kp < KT; ++kp) {
            const float *src = kp == 0 ? a.in0 : a.in1;
            float4 w[NT], x[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                w[t] = ldg4(a.w1img + ((size_t)(ot * KT * NT + kp * NT + t) * 64 + lane) * 4);
                x[t] = ldg4(src + (size_t)atom * H + 16 * t + 4 * g);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc = mfma16(w[t].x, x[t].x, acc);
                acc = mfma16(w[t].y, x[t].y, acc);
                acc = mfma16(w[t].z, x[t].z, acc);
                acc = mfma16(w[t].w, x[t].w, acc);
            }
        }
        stg4(xrow + 16 * ot + 4 * g, float4{acc[0], acc[1], acc[2], acc[3]});
    }
    // second-layer weights: issue the loads before the barrier so their latency overlaps it
    const bool has2 = ot < a.nt2;
    float4 w2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        w2[t] = has2 ? ldg4(a.w2img + ((size_t)(ot * NT + t) * 64 + lane) * 4) : float4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (!tile_ok || !has2) return;

    float hid[NT * 4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 v = ldg4(xrow + 16 * t + 4 * g);
        hid[4 * t] = v.x; hid[4 * t + 1] = v.y; hid[4 * t + 2] = v.z; hid[4 * t + 3] = v.w;
    }
    if (a.mode == NODE_LN_RELU) {
        ln_relu_dlayout<NT>(hid, a.ln_g, a.ln_b, g);
    } else {   // ShiftedSoftplus: softplus(x) - ln 2   (models/common.py:39-45; torch threshold 20)
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) {
            const float v = hid[i];
            hid[i] = (v > 20.f ? v : log1pf(expf(v))) - 0.6931471805599453f;
        }
    }
    const float4 bb = ldg4(a.b2 + 16 * ot + 4 * g);
    f32x4 acc2 = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc2 = mfma16(w2[t].x, hid[4 * t + 0], acc2);
        acc2 = mfma16(w2[t].y, hid[4 * t + 1], acc2);
        acc2 = mfma16(w2[t].z, hid[4 * t + 2], acc2);
        acc2 = mfma16(w2[t].w, hid[4 * t + 3], acc2);
    }
    if (!atom_ok) return;
    const int f0 = 16 * ot + 4 * g;
    if (a.resid) {
        const float4 rr = ldg4(a.resid + (size_t)atom * H + f0);
        acc2 += f32x4{rr.x, rr.y, rr.z, rr.w};
    }
    if (f0 + 4 <= a.n_store && (a.ld_out & 3) == 0) {
        stg4(a.out + (size_t)atom * a.ld_out + f0, float4{acc2[0], acc2[1], acc2[2], acc2[3]});
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (f0 + k < a.n_store) a.out[(size_t)atom * a.ld_out + f0 + k] = acc2[k];
    }
}

// -------------------------------------------------------------------------------------------------
// node_chain_kernel: everything that follows the x2h attention of a layer on the node side, in one launch:
//     h' = h + MLP_out([att | h])                                  (uni_transformer.py:82-88)
//     up to two follow-up MLPs on h':  the query MLP of this layer's h2x attention, and either the
//     query MLP of the NEXT layer's x2h attention (h does not change in h2x) or, on the last layer,
//     the atom-type head v_inference.
// A team of NT waves owns one 16-atom column tile and splits the feature blocks; the hidden tiles and
// h' cross LDS (three buffers, three barriers in total).
// -------------------------------------------------------------------------------------------------
struct NodeFollow {
    const float *w1img, *b1, *ln_g, *ln_b, *w2img, *b2;
    const float *w1img6, *w2img6;                         // split bf16 images (node_chain6_kernel)
    float *out;
    int ld_out, n_store, mode, nt2;
};
struct NodeChainArgs {
    const float *att, *h;                                 // [N][H] each
    const float *w1img, *b1, *ln_g, *ln_b, *w2img, *b2;   // node_output MLP (K = 2H)
    const float *w1img6, *w2img6;                         // split bf16 images (node_chain6_kernel)
    float *h_out;                                         // [N][H]
    NodeFollow f[2];
    int n_follow, n_atoms;
    unsigned long long *stamps;   // diagnostic build only
    // node_chain16_kernel only: the per-node products of the NEXT attentions (out[N][n_lin_tiles * 16] = h' W^T + per-molecule
    // term) as a last use of the h' fragments, instead of a separate node_linear launch (n_lin_tiles = 0: none)
    const float *lin_img16, *add_mol;
    const int *mol_of;
    float *pre_out;
    int n_lin_tiles, ld_add, ld_out;
};

// One workgroup = NT waves (one per 16-row block of output features) x CHAIN_COLS column tiles: every weight
// block is read from L2 once per workgroup and applied to all its column tiles (the per-CU L2 read rate,
// ~70 GB/s, is what bounds this kernel: 448 KB of weights per workgroup at H = 128).
constexpr int CHAIN_COLS = 2;

template <int H>
__global__ void __launch_bounds__(H * 4)
node_chain_kernel(NodeChainArgs a) {
    constexpr int NT = H / 16, CC = CHAIN_COLS;
    constexpr int XS = H + 16;                            // row stride of the exchange tiles
    constexpr int XS2 = 2 * H + 16;                       // row stride of the [att | h] input tile
    __shared__ __attribute__((aligned(16))) float bufA[CC * 16 * XS], bufB[CC * 16 * XS], bufC[CC * 16 * XS];
    __shared__ __attribute__((aligned(16))) float tile[CC * 16 * XS2];
    const int lane = threadIdx.x & 63, ot = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int ct0 = blockIdx.x * CC;
    const int f0 = 16 * ot + 4 * g;

    auto load_w = [&](const float *wimg, int ntk, int kt0, float4 (&w)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = ldg4(wimg + ((size_t)(ot * ntk + kt0 + t) * 64 + lane) * 4);
    };
    // acc[c] += W * X_c for both column tiles, B operand streamed from LDS rows (stride xs)
    auto gemm_lds = [&](const float4 (&w)[NT], const float *x0, int xs, f32x4 (&acc)[CC]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float4 x = ldg4(x0 + (c * 16 + n) * xs + 16 * t + 4 * g);
                acc[c] = mfma16(w[t].x, x.x, acc[c]);
                acc[c] = mfma16(w[t].y, x.y, acc[c]);
                acc[c] = mfma16(w[t].z, x.z, acc[c]);
                acc[c] = mfma16(w[t].w, x.w, acc[c]);
            }
        }
    };
    auto read_row = [&](const float *buf, int c, float (&v)[NT * 4]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 q = ldg4(buf + (c * 16 + n) * XS + 16 * t + 4 * g);
            v[4 * t] = q.x; v[4 * t + 1] = q.y; v[4 * t + 2] = q.z; v[4 * t + 3] = q.w;
        }
    };
    auto activate = [&](float (&v)[NT * 4], int mode, const float *gam, const float *bet) {
        if (mode == NODE_LN_RELU) {
            ln_relu_dlayout<NT>(v, gam, bet, g);
        } else {
#pragma unroll
            for (int i = 0; i < NT * 4; ++i) v[i] = (v[i] > 20.f ? v[i] : log1pf(expf(v[i]))) - 0.6931471805599453f;
        }
    };
    auto gemm_hid = [&](const float4 (&w)[NT], const float (&v)[NT * 4], f32x4 acc) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc = mfma16(w[t].x, v[4 * t + 0], acc);
            acc = mfma16(w[t].y, v[4 * t + 1], acc);
            acc = mfma16(w[t].z, v[4 * t + 2], acc);
            acc = mfma16(w[t].w, v[4 * t + 3], acc);
        }
        return acc;
    };
    auto atom_of = [&](int c) { return min((ct0 + c) * 16 + n, a.n_atoms - 1); };
    auto atom_ok = [&](int c) { return (ct0 + c) * 16 + n < a.n_atoms; };

    SM_TICK(a.stamps, 0);
    // ---- stage 1: node_output MLP; every stage's weight block is requested one stage ahead ----------
    float4 wa[NT], wb[NT], wc[NT];
    load_w(a.w1img, 2 * NT, 0, wa);
    load_w(a.w1img, 2 * NT, NT, wb);
    {   // [att | h] tiles of the workgroup -> LDS
        constexpr int R4 = 2 * H / 4;
        for (int idx = threadIdx.x; idx < CC * 16 * R4; idx += NT * 64) {
            const int ar = idx / R4, c4 = idx % R4;
            const int at = min(ct0 * 16 + ar, a.n_atoms - 1);
            const float4 v = c4 < H / 4 ? ldg4(a.att + (size_t)at * H + 4 * c4) : ldg4(a.h + (size_t)at * H + 4 * (c4 - H / 4));
            stg4(tile + ar * XS2 + 4 * c4, v);
        }
    }
    const float4 b1 = ldg4(a.b1 + f0);
    load_w(a.w2img, NT, 0, wc);                                  // second Linear of the output MLP
    __syncthreads();
    {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b1.x, b1.y, b1.z, b1.w};
        gemm_lds(wa, tile, XS2, acc);
        gemm_lds(wb, tile + H, XS2, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c) stg4(bufA + (c * 16 + n) * XS + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    }
    if (a.n_follow > 0) load_w(a.f[0].w1img, NT, 0, wa);         // first Linears of the follow-up MLPs
    if (a.n_follow > 1) load_w(a.f[1].w1img, NT, 0, wb);
    SM_TICK(a.stamps, 1);
    __syncthreads();
    SM_TICK(a.stamps, 2);
    {
        const float4 b = ldg4(a.b2 + f0);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float hid[NT * 4];
            read_row(bufA, c, hid);
            activate(hid, NODE_LN_RELU, a.ln_g, a.ln_b);
            const f32x4 acc = gemm_hid(wc, hid, f32x4{b.x, b.y, b.z, b.w});
            const float4 hres = ldg4(tile + (c * 16 + n) * XS2 + H + f0);      // residual: h is in the input tile
            const float4 hn = {acc[0] + hres.x, acc[1] + hres.y, acc[2] + hres.z, acc[3] + hres.w};
            stg4(bufB + (c * 16 + n) * XS + f0, hn);
            if (atom_ok(c)) stg4(a.h_out + (size_t)atom_of(c) * H + f0, hn);
        }
    }
    SM_TICK(a.stamps, 3);
    __syncthreads();
    SM_TICK(a.stamps, 4);
    if (a.n_follow == 0) return;

    // ---- stage 2: follow-up MLPs on the new h ------------------------------------------------------
    const bool on0 = ot < a.f[0].nt2, on1 = a.n_follow > 1 && ot < a.f[1].nt2;
    if (on0) load_w(a.f[0].w2img, NT, 0, wc);
    {
        const float4 b = ldg4(a.f[0].b1 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b.x, b.y, b.z, b.w};
        gemm_lds(wa, bufB, XS, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c) stg4(bufA + (c * 16 + n) * XS + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    }
    if (on1) load_w(a.f[1].w2img, NT, 0, wa);
    if (a.n_follow > 1) {
        const float4 b = ldg4(a.f[1].b1 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b.x, b.y, b.z, b.w};
        gemm_lds(wb, bufB, XS, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c) stg4(bufC + (c * 16 + n) * XS + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    }
    SM_TICK(a.stamps, 5);
    __syncthreads();
    SM_TICK(a.stamps, 6);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!(k == 0 ? on0 : on1)) continue;
        const NodeFollow &F = a.f[k];
        const float4 b = ldg4(F.b2 + f0);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float hid[NT * 4];
            read_row(k == 0 ? bufA : bufC, c, hid);
            activate(hid, F.mode, F.ln_g, F.ln_b);
            const f32x4 acc = gemm_hid(k == 0 ? wc : wa, hid, f32x4{b.x, b.y, b.z, b.w});
            if (!atom_ok(c)) continue;
            const int atom = atom_of(c);
            if (f0 + 4 <= F.n_store && (F.ld_out & 3) == 0) {
                stg4(F.out + (size_t)atom * F.ld_out + f0, float4{acc[0], acc[1], acc[2], acc[3]});
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f0 + r < F.n_store) F.out[(size_t)atom * F.ld_out + f0 + r] = acc[r];
            }
        }
    }
    SM_TICK(a.stamps, 7);
}

// -------------------------------------------------------------------------------------------------
// node_chain6_kernel: the same chain on the bf16 matrix cores with exactly split operands (the
// gemm_bf16x6 arithmetic of sm_device.h).  Differences to node_chain_kernel:
//   * every activation that feeds a Linear lives in LDS as ready-made B fragments (three bf16 pieces),
//     written once by whoever produces it: the staging pass ([att | h]), the normalise pass (hidden
//     tiles) or the producing wave itself (h': each lane owns half a fragment);
//   * LayerNorm + ReLU (or shifted softplus) is no longer repeated by every wave of the team: the
//     pre-activations cross LDS in fp32 and the team's lanes normalise disjoint slices (H / 8 lanes per
//     column, two-pass statistics over a DPP row segment) -- one more barrier per MLP, ~8x less vector work;
//   * weight blocks are resident per wave as 3 x K/32 fragments of 4 VGPRs, requested a stage ahead.
// -------------------------------------------------------------------------------------------------
// Slot of lane `l` inside the 64-slot block of k-step `b` of a fragment buffer.  Consumers read a block with all 64
// lanes (any bijection is conflict-free for them); the normalise pass WRITES it with the 16 lanes of a column, which hold
// the same n and different (k-step, lane group): un-swizzled those stores hit one bank group (8-way conflicts, 45 % of
// the kernel's LDS cycles per SQ_LDS_BANK_CONFLICT).  XOR-ing n with (group, k-step parity) spreads them over the banks.
SM_DEV int frag_slot(int b, int l) { return (l & 48) | ((l ^ (((l >> 4) << 1) | (b & 1))) & 15); }

template <int H>
struct Chain6Lds {
    static constexpr int NB = H / 32, CC = CHAIN_COLS;
    static constexpr int FRAG = 3 * NB * CC * 64;          // u32x4 per fragment buffer of K = H
    static constexpr int XS = H + 4;                       // fp32 pre-activation row stride: rows 16 B apart modulo 128 B,
                                                           // so the eight rows of a store group use different banks
    static constexpr int PRE = CC * 16 * XS;               // floats per pre-activation buffer
    static constexpr size_t BYTES = (size_t)3 * FRAG * 16 + (size_t)2 * PRE * 4;
};

template <int H>
__global__ void __launch_bounds__(H * 4)
node_chain6_kernel(NodeChainArgs a) {
    using L = Chain6Lds<H>;
    constexpr int NT = H / 16, NB = H / 32, CC = CHAIN_COLS, XS = L::XS;
    constexpr int LPC = NB * 4;                            // lanes per column in the normalise pass
    static_assert(CC == 2, "the normalise pass covers exactly 32 columns");
    extern __shared__ __attribute__((aligned(16))) unsigned char chain6_lds[];
    u32x4 *fin = reinterpret_cast<u32x4 *>(chain6_lds);    // [att | h] fragments (K = 2H); later the two hidden tiles
    u32x4 *fhid0 = fin, *fhid1 = fin + L::FRAG;
    u32x4 *fh = fin + 2 * L::FRAG;                         // fragments of the new h
    float *pre0 = reinterpret_cast<float *>(fin + 3 * L::FRAG), *pre1 = pre0 + L::PRE;
    const int lane = threadIdx.x & 63, ot = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int ct0 = blockIdx.x * CC;
    const int f0 = 16 * ot + 4 * g;

    auto load_w = [&](const float *img, auto &w) {         // w[3][KB]: this wave's block of a split image
        constexpr int KB = sizeof(w[0]) / sizeof(u32x4);
        const u32x4 *wi = reinterpret_cast<const u32x4 *>(img) + (size_t)ot * 3 * KB * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int b = 0; b < KB; ++b) w[p][b] = wi[(p * KB + b) * 64];
    };
    // acc[c] += W * X_c over KB k-steps; fragments at f[((piece * KB + b) * CC + c) * 64 + lane]
    auto gemm6 = [&](const auto &w, const u32x4 *f, f32x4 (&acc)[CC]) {
        constexpr int KB = sizeof(w[0]) / sizeof(u32x4);
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            u32x4 xh[CC], xm[CC], xl[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                xh[c] = f[((0 * KB + b) * CC + c) * 64 + frag_slot(b, lane)];
                xm[c] = f[((1 * KB + b) * CC + c) * 64 + frag_slot(b, lane)];
                xl[c] = f[((2 * KB + b) * CC + c) * 64 + frag_slot(b, lane)];
            }
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[2][b], xh[c], acc[c]);      // smallest terms first
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[1][b], xm[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[0][b], xl[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[1][b], xh[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[0][b], xm[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[0][b], xh[c], acc[c]);
        }
    };
    auto store_pre = [&](float *pre, const f32x4 (&acc)[CC]) {
#pragma unroll
        for (int c = 0; c < CC; ++c) stg4(pre + (c * 16 + n) * XS + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    };
    // activation of a pre-activation buffer -> fragments: LPC lanes per column, 8 features per lane
    auto normalise = [&](const float *pre, int mode, const float *gam, const float *bet, u32x4 *fo) {
        const int col = ot * (64 / LPC) + lane / LPC, ln = lane % LPC;
        const int b = ln >> 2, gg = ln & 3;
        const int fa = 32 * b + 4 * gg;
        const float4 p0 = ldg4(pre + col * XS + fa), p1 = ldg4(pre + col * XS + fa + 16);
        float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        if (mode == NODE_LN_RELU) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
            const float mean = seg_sum<LPC>(s) * (1.0f / H);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
            const float var = seg_sum<LPC>(q) * (1.0f / H);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            const float4 g0 = ldg4(gam + fa), g1 = ldg4(gam + fa + 16), b0 = ldg4(bet + fa), b1 = ldg4(bet + fa + 16);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf((v[i] - mean) * rstd * ga[i] + be[i], 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (v[i] > 20.f ? v[i] : log1pf(expf(v[i]))) - 0.6931471805599453f;
        }
        u32x4 hi, mid, lo;
        split3_bf16(v, hi, mid, lo);
        u32x4 *dst = fo + (b * CC + (col >> 4)) * 64 + frag_slot(b, gg * 16 + (col & 15));
        dst[0] = hi; dst[NB * CC * 64] = mid; dst[2 * NB * CC * 64] = lo;
    };
    auto atom_of = [&](int c) { return min((ct0 + c) * 16 + n, a.n_atoms - 1); };
    auto atom_ok = [&](int c) { return (ct0 + c) * 16 + n < a.n_atoms; };

    SM_TICK(a.stamps, 0);
    // ---- stage 0: weights of the output MLP; [att | h] tiles -> fragments ----------------------------
    // The activation rows are requested BEFORE the weights: vector memory operations complete in order, and behind the 48 weight
    // loads of a wave the staging pass waits for them too (18.96 -> 18.61 us per launch at B = 256).
    u32x4 w1[3][2 * NB], w2[3][NB];
    constexpr int SITER = CC * 2 * NB * 64 / (NT * 64);
    float4 sv0[SITER], sv1[SITER];
#pragma unroll
    for (int it = 0; it < SITER; ++it) {
        const int idx = threadIdx.x + it * NT * 64;
        const int sl = idx & 63, sb = (idx >> 6) % (2 * NB), sc = idx / (64 * 2 * NB);
        const int at = min((ct0 + sc) * 16 + (sl & 15), a.n_atoms - 1);
        const float *src = (sb < NB ? a.att + (size_t)at * H + 32 * sb : a.h + (size_t)at * H + 32 * (sb - NB)) + 4 * (sl >> 4);
        sv0[it] = ldg4(src); sv1[it] = ldg4(src + 16);
    }
    float4 hres[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) hres[c] = ldg4(a.h + (size_t)atom_of(c) * H + f0);    // residual
    const float4 b1 = ldg4(a.b1 + f0), b2 = ldg4(a.b2 + f0);
    load_w(a.w1img6, w1);
#pragma unroll
    for (int it = 0; it < SITER; ++it) {
        const int idx = threadIdx.x + it * NT * 64;
        const int sl = idx & 63, sb = (idx >> 6) % (2 * NB), sc = idx / (64 * 2 * NB);
        const float v[8] = {sv0[it].x, sv0[it].y, sv0[it].z, sv0[it].w, sv1[it].x, sv1[it].y, sv1[it].z, sv1[it].w};
        u32x4 hi, mid, lo;
        split3_bf16(v, hi, mid, lo);
        u32x4 *dst = fin + (sb * CC + sc) * 64 + frag_slot(sb, sl);
        dst[0] = hi; dst[2 * NB * CC * 64] = mid; dst[2 * 2 * NB * CC * 64] = lo;
    }
    load_w(a.w2img6, w2);
    __syncthreads();
    SM_TICK(a.stamps, 1);

    // ---- stage 1: h' = h + W2 relu(LN(W1 [att | h] + b1)) + b2 ---------------------------------------
    {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b1.x, b1.y, b1.z, b1.w};
        gemm6(w1, fin, acc);
        store_pre(pre0, acc);
    }
    u32x4 wf0[3][NB], wf1[3][NB];                                    // first Linears of the follow-up MLPs
    if (a.n_follow > 0) load_w(a.f[0].w1img6, wf0);
    if (a.n_follow > 1) load_w(a.f[1].w1img6, wf1);
    __syncthreads();
    SM_TICK(a.stamps, 2);
    normalise(pre0, NODE_LN_RELU, a.ln_g, a.ln_b, fhid0);
    __syncthreads();
    SM_TICK(a.stamps, 3);
    {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b2.x, b2.y, b2.z, b2.w};
        gemm6(w2, fhid0, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float4 hn = {acc[c][0] + hres[c].x, acc[c][1] + hres[c].y, acc[c][2] + hres[c].z, acc[c][3] + hres[c].w};
            if (atom_ok(c)) stg4(a.h_out + (size_t)atom_of(c) * H + f0, hn);
            // this lane's four features are half of fragment (k-step ot / 2, lane) of column tile c
            unsigned ph[2], pm[2], pl[2];
            split3_pair(hn.x, hn.y, ph[0], pm[0], pl[0]);
            split3_pair(hn.z, hn.w, ph[1], pm[1], pl[1]);
            uint2 *dst = reinterpret_cast<uint2 *>(fh + ((ot >> 1) * CC + c) * 64 + frag_slot(ot >> 1, lane)) + (ot & 1);
            dst[0] = uint2{ph[0], ph[1]};
            dst[NB * CC * 64 * 2] = uint2{pm[0], pm[1]};
            dst[2 * NB * CC * 64 * 2] = uint2{pl[0], pl[1]};
        }
    }
    if (a.n_follow == 0) return;
    const bool on0 = ot < a.f[0].nt2, on1 = a.n_follow > 1 && ot < a.f[1].nt2;
    u32x4 wg0[3][NB], wg1[3][NB];                                    // second Linears of the follow-up MLPs
    if (on0) load_w(a.f[0].w2img6, wg0);
    __syncthreads();
    SM_TICK(a.stamps, 4);

    // ---- stage 2: follow-up MLPs on the new h ---------------------------------------------------------
    {
        const float4 ba = ldg4(a.f[0].b1 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{ba.x, ba.y, ba.z, ba.w};
        gemm6(wf0, fh, acc);
        store_pre(pre0, acc);
    }
    if (a.n_follow > 1) {
        const float4 bb = ldg4(a.f[1].b1 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{bb.x, bb.y, bb.z, bb.w};
        gemm6(wf1, fh, acc);
        store_pre(pre1, acc);
    }
    if (on1) load_w(a.f[1].w2img6, wg1);
    __syncthreads();
    SM_TICK(a.stamps, 5);
    normalise(pre0, a.f[0].mode, a.f[0].ln_g, a.f[0].ln_b, fhid0);
    if (a.n_follow > 1) normalise(pre1, a.f[1].mode, a.f[1].ln_g, a.f[1].ln_b, fhid1);
    __syncthreads();
    SM_TICK(a.stamps, 6);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!(k == 0 ? on0 : on1)) continue;
        const NodeFollow &F = a.f[k];
        const float4 b = ldg4(F.b2 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b.x, b.y, b.z, b.w};
        if (k == 0) gemm6(wg0, fhid0, acc); else gemm6(wg1, fhid1, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            if (!atom_ok(c)) continue;
            const int atom = atom_of(c);
            if (f0 + 4 <= F.n_store && (F.ld_out & 3) == 0) {
                stg4(F.out + (size_t)atom * F.ld_out + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f0 + r < F.n_store) F.out[(size_t)atom * F.ld_out + f0 + r] = acc[c][r];
            }
        }
    }
    SM_STAMP(a.stamps, 7);
}

// -------------------------------------------------------------------------------------------------
// node_prologue6_kernel: everything on the node side that precedes the first attention of an evaluation, in
// one launch (it replaces atom_embed_kernel + node_mlp2_kernel + node_linear6_kernel at layer 0):
//     h0 = ligand_atom_emb([one_hot(v) | time_emb])          (molopt_score_model.py:292-301), stored and split
//     q  = query MLP of the first x2h attention on h0         (uni_transformer.py:68)
//     pre0 = per-node halves of that attention's edge MLPs:   h0 W^T + per-molecule term  ([N][4H])
// plus the per-evaluation bookkeeping (latch the step counter, clear the batch-norm accumulators).
// Same team layout and bf16x6 arithmetic as node_chain6_kernel.
// -------------------------------------------------------------------------------------------------
struct NodePrologueArgs {
    const float *emb_wT, *emb_b;  // [C + D][H] (transposed: a lane's four features are one 16-byte load), [H]
    const int64_t *v;             // [N]
    const int *mol_of;
    const float *ttab;            // [T][D]
    const float *etab;            // [T][C][H] embedding of every (timestep, atom type) pair (emb_table_kernel), or nullptr
    const int *t_mol;             // [B] timestep per molecule (score API)
    const int *step_ptr;          // sampling: device step counter (t = t_first - step), else nullptr
    int *step_cur;
    double *bn_acc;               // zeroed here
    float *h_out;                 // [N][H]
    NodeFollow q;                 // query MLP -> q.out
    const float *lin_img6;        // split image of [n_lin_tiles * 16][H]
    const float *add_mol;         // [B][ld_add] per-molecule term of the linear outputs
    float *pre_out;               // [N][ld_out]
    int n_lin_tiles, ld_add, ld_out;
    int n_atoms, C, D, t_first, bn_acc_len;
    unsigned long long *stamps;   // diagnostic build only
};

template <int H>
__global__ void __launch_bounds__(H * 4)
node_prologue6_kernel(NodePrologueArgs a) {
    using L = Chain6Lds<H>;
    constexpr int NT = H / 16, NB = H / 32, CC = CHAIN_COLS, XS = L::XS;
    constexpr int LPC = NB * 4;
    static_assert(CC == 2, "the normalise pass covers exactly 32 columns");
    extern __shared__ __attribute__((aligned(16))) unsigned char chain6_lds[];
    u32x4 *fh = reinterpret_cast<u32x4 *>(chain6_lds);      // fragments of h0
    u32x4 *fhid = fh + L::FRAG;                             // hidden tile of the query MLP
    float *pre0 = reinterpret_cast<float *>(fh + 2 * L::FRAG);
    const int lane = threadIdx.x & 63, ot = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int ct0 = blockIdx.x * CC;
    const int f0 = 16 * ot + 4 * g;

    auto load_w = [&](const float *img, int tile, u32x4 (&w)[3][NB]) {
        const u32x4 *wi = reinterpret_cast<const u32x4 *>(img) + (size_t)tile * 3 * NB * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int b = 0; b < NB; ++b) w[p][b] = wi[(p * NB + b) * 64];
    };
    auto gemm6 = [&](const u32x4 (&w)[3][NB], const u32x4 *f, f32x4 (&acc)[CC]) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            u32x4 xh[CC], xm[CC], xl[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                xh[c] = f[((0 * NB + b) * CC + c) * 64 + frag_slot(b, lane)];
                xm[c] = f[((1 * NB + b) * CC + c) * 64 + frag_slot(b, lane)];
                xl[c] = f[((2 * NB + b) * CC + c) * 64 + frag_slot(b, lane)];
            }
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[2][b], xh[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[1][b], xm[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[0][b], xl[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[1][b], xh[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[0][b], xm[c], acc[c]);
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_bf16(w[0][b], xh[c], acc[c]);
        }
    };
    auto atom_of = [&](int c) { return min((ct0 + c) * 16 + n, a.n_atoms - 1); };
    auto atom_ok = [&](int c) { return (ct0 + c) * 16 + n < a.n_atoms; };

    // ---- bookkeeping of the evaluation ------------------------------------------------------------------
    const int step = a.step_ptr ? *a.step_ptr : 0;
    {
        const int gid = blockIdx.x * (H * 4) + threadIdx.x;           // (the launch uses H * 4 threads: no blockDim read)
        if (gid == 0 && a.step_ptr) *a.step_cur = step;
        for (int i = gid; i < a.bn_acc_len; i += gridDim.x * (H * 4)) a.bn_acc[i] = 0.0;
    }
    // ---- stage 0: embedding of the workgroup's atoms -> global h0 and LDS fragments ----------------------
    u32x4 wq1[3][NB], wl[3][NB];
    load_w(a.q.w1img6, ot, wq1);
    for (int idx = threadIdx.x; idx < CC * NB * 64; idx += NT * 64) {
        const int sl = idx & 63, sb = (idx >> 6) % NB, sc = idx / (64 * NB);
        const int at_raw = (ct0 + sc) * 16 + (sl & 15);
        const int at = min(at_raw, a.n_atoms - 1);
        const int t = a.step_ptr ? a.t_first - step : a.t_mol[a.mol_of[at]];
        const int vi = min(max((int)a.v[at], 0), a.C - 1);       // out-of-range types are flagged by v_check_kernel
        const int fa = 32 * sb + 4 * (sl >> 4);
        float vv[8];
        {   // the precomputed row of (t, v) (emb_table_kernel, built once per context with the sum y = (b + W[:, v]) + sum_k W[:, C + k] te[k],
            // k ascending, that the reference's Linear evaluates per atom: molopt_score_model.py:292-301)
            const float *row = a.etab + ((size_t)t * a.C + vi) * H + fa;
            const float4 r0 = ldg4(row), r1 = ldg4(row + 16);
            vv[0] = r0.x; vv[1] = r0.y; vv[2] = r0.z; vv[3] = r0.w; vv[4] = r1.x; vv[5] = r1.y; vv[6] = r1.z; vv[7] = r1.w;
        }
        if (at_raw < a.n_atoms) {
            stg4(a.h_out + (size_t)at * H + fa, float4{vv[0], vv[1], vv[2], vv[3]});
            stg4(a.h_out + (size_t)at * H + fa + 16, float4{vv[4], vv[5], vv[6], vv[7]});
        }
        u32x4 hi, mid, lo;
        split3_bf16(vv, hi, mid, lo);
        u32x4 *dst = fh + (sb * CC + sc) * 64 + frag_slot(sb, sl);
        dst[0] = hi; dst[NB * CC * 64] = mid; dst[2 * NB * CC * 64] = lo;
    }
    if (a.n_lin_tiles > 0) load_w(a.lin_img6, ot, wl);
    const float4 b1 = ldg4(a.q.b1 + f0);
    __syncthreads();

    // ---- stage 1: first Linear of the query MLP; the per-node linear outputs of the edge MLPs ------------
    {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b1.x, b1.y, b1.z, b1.w};
        gemm6(wq1, fh, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c) stg4(pre0 + (c * 16 + n) * XS + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    }
    const bool onq = ot < a.q.nt2;
    if (onq) load_w(a.q.w2img6, ot, wq1);                    // second Linear of the query MLP (reuses the registers)
    // linear outputs: tiles ot, ot + NT, ... (4 per wave for the [N][4H] products); weights alternate between two
    // register sets so that the next block is in flight during the current product
    auto lin_tile = [&](int tile, const u32x4 (&w)[3][NB]) {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float4 t = a.add_mol ? ldg4(a.add_mol + (size_t)a.mol_of[atom_of(c)] * a.ld_add + 16 * tile + 4 * g)
                                       : float4{0.f, 0.f, 0.f, 0.f};
            acc[c] = f32x4{t.x, t.y, t.z, t.w};
        }
        gemm6(w, fh, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (atom_ok(c)) stg4(a.pre_out + (size_t)atom_of(c) * a.ld_out + 16 * tile + 4 * g, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    };
    for (int tile = ot; tile < a.n_lin_tiles; tile += 2 * NT) {
        u32x4 wn[3][NB];
        const bool more1 = tile + NT < a.n_lin_tiles, more2 = tile + 2 * NT < a.n_lin_tiles;
        if (more1) load_w(a.lin_img6, tile + NT, wn);
        lin_tile(tile, wl);
        if (more2) load_w(a.lin_img6, tile + 2 * NT, wl);
        if (more1) lin_tile(tile + NT, wn);
    }
    __syncthreads();
    {   // LayerNorm + ReLU of the hidden tile -> fragments (LPC lanes per column, as in node_chain6_kernel)
        const int col = ot * (64 / LPC) + lane / LPC, ln = lane % LPC;
        const int b = ln >> 2, gg = ln & 3;
        const int fa = 32 * b + 4 * gg;
        const float4 p0 = ldg4(pre0 + col * XS + fa), p1 = ldg4(pre0 + col * XS + fa + 16);
        float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
        const float mean = seg_sum<LPC>(s) * (1.0f / H);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
        const float var = seg_sum<LPC>(q) * (1.0f / H);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const float4 g0 = ldg4(a.q.ln_g + fa), g1 = ldg4(a.q.ln_g + fa + 16), e0 = ldg4(a.q.ln_b + fa), e1 = ldg4(a.q.ln_b + fa + 16);
        const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float be[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf((v[i] - mean) * rstd * ga[i] + be[i], 0.f);
        u32x4 hi, mid, lo;
        split3_bf16(v, hi, mid, lo);
        u32x4 *dst = fhid + (b * CC + (col >> 4)) * 64 + frag_slot(b, gg * 16 + (col & 15));
        dst[0] = hi; dst[NB * CC * 64] = mid; dst[2 * NB * CC * 64] = lo;
    }
    __syncthreads();
    if (onq) {
        const float4 b = ldg4(a.q.b2 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b.x, b.y, b.z, b.w};
        gemm6(wq1, fhid, acc);
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (atom_ok(c)) stg4(a.q.out + (size_t)atom_of(c) * a.q.ld_out + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    }
}
