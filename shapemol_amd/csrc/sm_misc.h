// Small kernels around the attention layers: batch indexing, step-invariant shape terms,
// time/atom embedding, kNN graph + edge weights, the vector-neuron coordinate update with
// train-mode batch statistics, and the DDPM posterior step.
#pragma once
#include "sm_device.h"

// ---------------------------------------------------------------------------------------------
// batch (N,) i64 sorted -> mol_of (N,) i32 and mol_off (B+1,) i32
// (reference: batch_ligand = repeat_interleave(arange(B), counts), scripts/sample_diffusion.py:72)
// ---------------------------------------------------------------------------------------------
// status flags of a context (device int[8], sticky until the next _score/_sample; read by shapemol_status)
enum StatusFlag { ST_VN_BARRIER = 0, ST_BATCH = 1, ST_ATOM_TYPE = 2, ST_TIME = 3, ST_RANGE = 4, ST_SPAN = 5 };

__global__ void mol_index_kernel(const int64_t *batch, int n, int n_mols, int *mol_of, int *mol_off, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t braw = batch[i], praw = i == 0 ? -1 : batch[i - 1];
    // a batch vector that is unsorted or names a molecule >= n_mols would corrupt the kNN ranges: flag it and clamp
    if (braw < 0 || braw >= n_mols || braw < praw) status[ST_BATCH] = 1;
    const int b = (int)min(max(braw, (int64_t)0), (int64_t)n_mols - 1);
    const int prev = (int)min(max(praw, (int64_t)-1), (int64_t)n_mols - 1);
    mol_of[i] = b;
    for (int m = prev + 1; m <= b; ++m) mol_off[m] = i;       // also covers empty molecules
    if (i == n - 1)
        for (int m = b + 1; m <= n_mols; ++m) mol_off[m] = n;
}

// first / one-past-last atom of every atom's molecule (step-invariant; read by the folded coordinate update of the x2h
// kernel, which would otherwise chase mol_of -> mol_off before it can request anything)
__global__ void mol_span_kernel(const int *mol_of, const int *mol_off, int n, int2 *span) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) span[i] = int2{mol_off[mol_of[i]], mol_off[mol_of[i] + 1]};
}

__global__ void t_convert_kernel(const int64_t *t, int n_mols, int n_timesteps, int *t_mol, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mols) return;
    const int64_t tr = t[i];
    if (tr < 0 || tr >= n_timesteps) status[ST_TIME] = 1;
    t_mol[i] = (int)min(max(tr, (int64_t)0), (int64_t)n_timesteps - 1);
}

// atom types must index the embedding table: flag and leave (the embedding stage clamps its own read)
__global__ void v_check_kernel(const int64_t *v, int n, int n_classes, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (v[i] < 0 || v[i] >= n_classes)) status[ST_ATOM_TYPE] = 1;
}

// ---------------------------------------------------------------------------------------------
// InvariantShapeEmbLayer.forward (models/uni_transformer.py:181-189): one block per molecule.
// ---------------------------------------------------------------------------------------------
struct ShapeInvArgs {
    const float *shape;              // [B][S][3]
    const float *w1, *b1, *g, *be;   // Linear S->S, LayerNorm(S)
    const float *w2, *b2;            // Linear S->SL
    float *inv;                      // [B][SL]
    int S, SL;
};
__global__ void shape_invariant_kernel(ShapeInvArgs a) {
    __shared__ float raw[64], hid[64], red[4];
    const int b = blockIdx.x, c = threadIdx.x, S = a.S;
    const float *sh = a.shape + (size_t)b * S * 3;
    if (c < 3) {
        float m = 0.f;
        for (int i = 0; i < S; ++i) m += sh[i * 3 + c];
        red[c] = m / S;
    }
    __syncthreads();
    if (c == 0) red[3] = red[0] * red[0] + red[1] * red[1] + red[2] * red[2] + 1e-6f;
    __syncthreads();
    if (c < S)
        raw[c] = sh[c * 3] * (red[0] / red[3]) + sh[c * 3 + 1] * (red[1] / red[3]) + sh[c * 3 + 2] * (red[2] / red[3]);
    __syncthreads();
    float y = 0.f;
    if (c < S) {
        y = a.b1[c];
        for (int i = 0; i < S; ++i) y += a.w1[c * S + i] * raw[i];
    }
    // LayerNorm over S values (S <= 64: one wave)
    float s = c < S ? y : 0.f;
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
    const float mean = s / S;
    float q = c < S ? (y - mean) * (y - mean) : 0.f;
    for (int m = 1; m < 64; m <<= 1) q += __shfl_xor(q, m, 64);
    const float rstd = 1.0f / sqrtf(q / S + 1e-5f);
    if (c < S) hid[c] = fmaxf((y - mean) * rstd * a.g[c] + a.be[c], 0.f);
    __syncthreads();
    if (c < a.SL) {
        float o = a.b2[c];
        for (int i = 0; i < S; ++i) o += a.w2[c * S + i] * hid[i];
        a.inv[(size_t)b * a.SL + c] = o;
    }
}

// Step-invariant per-molecule term of the edge MLPs' first Linear:
//   add[b][blk*H + f] = W1[f][G+2H : G+2H+SL] . inv_b + b1[f]   for the two "A" blocks (k: blk 0, v: blk 2)
//   and 0 for the "B" blocks (1, 3).            (the inv_shape[dst] columns of kv_input, uni_transformer.py:61-63)
struct ShapeTermArgs {
    const float *inv;      // [B][SL]
    const float *wk, *bk;  // W1 of the k MLP [H][ldw] (pointer already at column G+2H), bias [H]
    const float *wv, *bv;
    float *add;            // [B][ld] (4H columns written)
    int ldw, H, SL, ld;
};
__global__ void shape_term_kernel(ShapeTermArgs a) {
    const int b = blockIdx.x;
    for (int f = threadIdx.x; f < 4 * a.H; f += blockDim.x) {
        const int blk = f / a.H, ff = f % a.H;
        float v = 0.f;
        if ((blk & 1) == 0) {
            const float *w = (blk == 0 ? a.wk : a.wv) + (size_t)ff * a.ldw;
            v = (blk == 0 ? a.bk : a.bv)[ff];
            for (int i = 0; i < a.SL; ++i) v += w[i] * a.inv[(size_t)b * a.SL + i];
        }
        a.add[(size_t)b * a.ld + f] = v;
    }
}

// all shape terms of a chain in one launch: grid (B, n_terms), the argument blocks in device memory
__global__ void shape_term_multi_kernel(const ShapeTermArgs *terms) {
    const ShapeTermArgs a = terms[blockIdx.y];
    const int b = blockIdx.x;
    for (int f = threadIdx.x; f < 4 * a.H; f += blockDim.x) {
        const int blk = f / a.H, ff = f % a.H;
        float v = 0.f;
        if ((blk & 1) == 0) {
            const float *w = (blk == 0 ? a.wk : a.wv) + (size_t)ff * a.ldw;
            v = (blk == 0 ? a.bk : a.bv)[ff];
            for (int i = 0; i < a.SL; ++i) v += w[i] * a.inv[(size_t)b * a.SL + i];
        }
        a.add[(size_t)b * a.ld + f] = v;
    }
}

// The same with the molecule loop inside: a thread keeps its weight row (SL floats) in registers and applies it to the
// kShapeTermMols molecules of its workgroup, whose invariant embeddings sit in LDS (the kernel above re-reads the row, 32
// dependent strided loads, for every molecule: 134 us for 23 terms x 256 molecules against ~10 us here; once per chain).
// Same sums in the same order (bias first, then the SL products ascending).
constexpr int kShapeTermMols = 32;
template <int SL>
__global__ void __launch_bounds__(256) shape_term_multi_tiled_kernel(const ShapeTermArgs *terms, int n_mols) {
    const ShapeTermArgs a = terms[blockIdx.y];
    __shared__ float invs[kShapeTermMols][SL];
    const int b0 = blockIdx.x * kShapeTermMols, nb = min(kShapeTermMols, n_mols - b0);
    for (int idx = threadIdx.x; idx < nb * SL; idx += blockDim.x) invs[idx / SL][idx % SL] = a.inv[(size_t)b0 * SL + idx];
    __syncthreads();
    for (int f = threadIdx.x; f < 4 * a.H; f += blockDim.x) {
        const int blk = f / a.H, ff = f % a.H;
        if (blk & 1) {
            for (int b = 0; b < nb; ++b) a.add[(size_t)(b0 + b) * a.ld + f] = 0.f;
            continue;
        }
        const float *w = (blk == 0 ? a.wk : a.wv) + (size_t)ff * a.ldw;
        const float bias = (blk == 0 ? a.bk : a.bv)[ff];
        float wr[SL];
#pragma unroll
        for (int i = 0; i < SL; ++i) wr[i] = w[i];
        for (int b = 0; b < nb; ++b) {
            float v = bias;
#pragma unroll
            for (int i = 0; i < SL; ++i) v += wr[i] * invs[b][i];
            a.add[(size_t)(b0 + b) * a.ld + f] = v;
        }
    }
}

// Shape part of the VN-linear inputs (step-invariant): ps[b][which][c][dim] =
//   sum_s W_which[c][1 + heads + s] * shape[b][s][dim]      (tmp_output concat, uni_transformer.py:154)
struct VnShapeArgs {
    const float *shape;    // [B][S][3]
    const float *wf, *wd;  // [heads][1 + heads + S]
    float *ps;             // [B][2][heads][3]
    int S, heads;
};
__global__ void vn_shape_kernel(VnShapeArgs a) {
    const int b = blockIdx.x, per = a.heads * 3, cin = 1 + a.heads + a.S;
    for (int idx = threadIdx.x; idx < 2 * per; idx += blockDim.x) {
        const int which = idx / per, c = (idx % per) / 3, dim = idx % 3;
        const float *w = (which ? a.wd : a.wf) + (size_t)c * cin + 1 + a.heads;
        float v = 0.f;
        for (int s = 0; s < a.S; ++s) v += w[s] * a.shape[((size_t)b * a.S + s) * 3 + dim];
        a.ps[(size_t)b * 2 * per + idx] = v;
    }
}

// all layers in one launch: grid (B, L); `shape` is the call's argument, the rest comes from the device-side blocks
__global__ void vn_shape_multi_kernel(const float *shape, const VnShapeArgs *layers) {
    VnShapeArgs a = layers[blockIdx.y];
    const int b = blockIdx.x, per = a.heads * 3, cin = 1 + a.heads + a.S;
    for (int idx = threadIdx.x; idx < 2 * per; idx += blockDim.x) {
        const int which = idx / per, c = (idx % per) / 3, dim = idx % 3;
        const float *w = (which ? a.wd : a.wf) + (size_t)c * cin + 1 + a.heads;
        float v = 0.f;
        for (int s = 0; s < a.S; ++s) v += w[s] * shape[((size_t)b * a.S + s) * 3 + dim];
        a.ps[(size_t)b * 2 * per + idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// time embedding (molopt_score_model.py:154-166,247-252).  It depends on the timestep only, so the
// table over all T timesteps is built once per context; a score evaluation just indexes it.
// ---------------------------------------------------------------------------------------------
struct TimeTableArgs {
    const float *w1, *b1, *w2, *b2;  // Linear D->2D, Linear 2D->D
    float *table;                    // [T][D]
    int T, D;
};
__global__ void time_table_kernel(TimeTableArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.T) return;
    const int D = a.D, half = D / 2;
    float e[16], hdn[32];
    const float c = (float)(9.210340371976184 / (half - 1));    // ln(10000) / (half - 1)
    for (int i = 0; i < half; ++i) {
        const float fr = expf((float)i * -c);
        const float arg = (float)t * fr;
        e[i] = sinf(arg);
        e[half + i] = cosf(arg);
    }
    for (int o = 0; o < 2 * D; ++o) {
        float y = a.b1[o];
        for (int i = 0; i < D; ++i) y += a.w1[o * D + i] * e[i];
        hdn[o] = y / (1.0f + expf(-y));                            // SiLU
    }
    for (int o = 0; o < D; ++o) {
        float y = a.b2[o];
        for (int i = 0; i < 2 * D; ++i) y += a.w2[o * 2 * D + i] * hdn[i];
        a.table[(size_t)t * D + o] = y;
    }
}

// The atom embedding of every (timestep, atom type) pair, built once: etab[t][v][:] = (b + W[:, v]) + sum_k W[:, C + k] te_t[k],
// k ascending -- the arithmetic of the prologue's embedding stage, which then reads one 512-byte row per atom instead of
// running the sum behind three dependent loads (step counter -> time row; atom type -> weight row).  T x C x H floats (7.7 MB).
__global__ void emb_table_kernel(const float *emb_wT, const float *emb_b, const float *ttab, float *etab, int T, int C, int D, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;         // (t, v, f / 4)
    const int q4 = H / 4;
    if (idx >= T * C * q4) return;
    const int f = 4 * (idx % q4), v = (idx / q4) % C, t = idx / (q4 * C);
    const float *te = ttab + (size_t)t * D;
    const float4 bb = ldg4(emb_b + f), wv = ldg4(emb_wT + (size_t)v * H + f);
    float y[4] = {bb.x + wv.x, bb.y + wv.y, bb.z + wv.z, bb.w + wv.w};
    for (int k = 0; k < D; ++k) {
        const float4 wk = ldg4(emb_wT + (size_t)(C + k) * H + f);
        const float tk = te[k];
        y[0] += wk.x * tk; y[1] += wk.y * tk; y[2] += wk.z * tk; y[3] += wk.w * tk;
    }
    stg4(etab + ((size_t)t * C + v) * H + f, float4{y[0], y[1], y[2], y[3]});
}

// ligand_atom_emb(cat[one_hot(v), time_emb[batch]])  (molopt_score_model.py:292-301) + the
// per-evaluation bookkeeping: latch the step counter, clear the batch-norm accumulators.
struct AtomEmbArgs {
    const float *w, *b;     // [H][C + D], [H]
    const int64_t *v;       // [N]
    const int *mol_of;
    const float *ttab;      // [T][D] time-embedding table
    const int *t_mol;       // [B] timestep per molecule (score API)
    const int *step_ptr;    // sampling: device step counter, t = t_first - step for every molecule; else nullptr
    int *step_cur;          // stable copy of the counter for the rest of this step's kernels
    double *bn_acc;         // zeroed here
    float *h;               // [N][H]
    int n_atoms, H, C, D, t_first, bn_acc_len;
};
__global__ void atom_embed_kernel(AtomEmbArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    if (idx == 0 && a.step_ptr) *a.step_cur = step;
    for (int i = idx; i < a.bn_acc_len; i += gridDim.x * blockDim.x) a.bn_acc[i] = 0.0;
    if (idx >= a.n_atoms * a.H) return;
    const int i = idx / a.H, f = idx % a.H, ld = a.C + a.D;
    const int t = a.step_ptr ? a.t_first - step : a.t_mol[a.mol_of[i]];
    const float *w = a.w + (size_t)f * ld;
    float y = a.b[f] + w[min(max((int)a.v[i], 0), a.C - 1)];
    const float *te = a.ttab + (size_t)t * a.D;
    for (int k = 0; k < a.D; ++k) y += w[a.C + k] * te[k];
    a.h[idx] = y;
}

// ---------------------------------------------------------------------------------------------
// kNN graph (uni_transformer.py:466-468; torch_geometric knn_graph semantics): one wave per
// centre atom.  The rank of candidate c is the number of candidates of the same molecule with a
// smaller (squared distance, index) key; rank < k selects slot `rank`.  The squared distance is
// (dx*dx + dy*dy) + dz*dz with every operation rounded (no FMA contraction), the same
// arithmetic as the oracle, so neighbour sets agree bit for bit.
// ---------------------------------------------------------------------------------------------
SM_DEV float dist2_rounded(float dx, float dy, float dz) {
#pragma clang fp contract(off)
    const float a = dx * dx, b = dy * dy, c = dz * dz;
    return (a + b) + c;
}

__global__ void knn_kernel(const float *x, const int *mol_of, const int *mol_off, int n_atoms, int k,
                           int kp, int *nbr) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_atoms) return;
    const int m = mol_of[i], s = mol_off[m], cnt = mol_off[m + 1] - s;
    const float xi = x[i * 3], yi = x[i * 3 + 1], zi = x[i * 3 + 2];
    const int deg = min(k, cnt - 1);
    for (int c = lane; c < cnt; c += 64) {
        const int j = s + c;
        if (j == i) continue;
        float dx = x[j * 3] - xi, dy = x[j * 3 + 1] - yi, dz = x[j * 3 + 2] - zi;
        const float dc = dist2_rounded(dx, dy, dz);
        int rank = 0;
        for (int o = 0; o < cnt; ++o) {
            const int jo = s + o;
            dx = x[jo * 3] - xi; dy = x[jo * 3 + 1] - yi; dz = x[jo * 3 + 2] - zi;
            const float d_o = dist2_rounded(dx, dy, dz);
            rank += (jo != i) && (o != c) && (d_o < dc || (d_o == dc && o < c));
        }
        if (rank < k) nbr[(size_t)i * kp + rank] = j;
    }
    for (int sl = lane; sl < kp; sl += 64)
        if (sl >= deg) nbr[(size_t)i * kp + sl] = -1;
}

// e_w = sigmoid(MLP_{G->H->1}(rbf(|x_i - x_j|)))   (uni_transformer.py:475-481), one 16-slot tile per wave
struct EdgeWeightArgs {
    const float *x;
    const int *nbr;         // [N][KP]
    const float *w1, *b1;   // [H][G], [H]
    const float *g, *be;    // LayerNorm
    const float *w2, *b2;   // [1][H], [1]
    float *ew;              // [N][KP]
    int n_slots, kp;        // n_slots = N * KP
};
template <int H>
__global__ void __launch_bounds__(256)
edge_weight_kernel(EdgeWeightArgs a) {
    constexpr int NT = H / 16;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    float cen[5];
    rbf_centres(g, cen);
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int e_raw = tile * 16 + n;
    if (tile * 16 >= a.n_slots) return;
    const bool in_range = e_raw < a.n_slots;
    const int e = in_range ? e_raw : a.n_slots - 1;
    const int i = e / a.kp;
    const int jraw = a.nbr[e];
    const bool ok = in_range && jraw >= 0;
    const int j = ok ? jraw : i;
    const float r0 = a.x[i * 3] - a.x[j * 3], r1 = a.x[i * 3 + 1] - a.x[j * 3 + 1], r2 = a.x[i * 3 + 2] - a.x[j * 3 + 2];
    float rb[5];
    rbf_dlayout(sqrtf(r0 * r0 + r1 * r1 + r2 * r2), cen, rb);
    float hid[NT * 4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 b = ldg4(a.b1 + 16 * t + 4 * g);
        f32x4 acc = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int s = 0; s < 5; ++s) acc = mfma16(a.w1[(16 * t + n) * 20 + 4 * s + g], rb[s], acc);
        hid[4 * t] = acc[0]; hid[4 * t + 1] = acc[1]; hid[4 * t + 2] = acc[2]; hid[4 * t + 3] = acc[3];
    }
    ln_relu_dlayout<NT>(hid, a.g, a.be, g);
    float p = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 w = ldg4(a.w2 + 16 * t + 4 * g);
        p += w.x * hid[4 * t] + w.y * hid[4 * t + 1] + w.z * hid[4 * t + 2] + w.w * hid[4 * t + 3];
    }
    p = sum_groups(p) + a.b2[0];
    if (g == 0 && in_range) a.ew[e] = ok ? 1.0f / (1.0f + expf(-p)) : 0.f;
}

// ---------------------------------------------------------------------------------------------
// Vector-neuron coordinate update of BaseH2XAttLayer (uni_transformer.py:153-156) =
// VNLinearLeakyReLU (shape_vn_layers.py:95-110) with VNBatchNorm in TRAIN mode (:50-61):
// pass 1 computes p = W_f z, d = W_d z per atom and accumulates the batch sums of the vector
// norms; pass 2 normalises with the batch mean / biased variance and updates x.
// z = [x_i | o_i (heads vectors) | shape_b (S vectors)]; the shape part is pre-reduced (vn_shape_kernel).
// o3 is stored with permuted head rows (row 4g + r, see sm_edge.h); `wf_o`/`wd_o` are permuted to match.
// ---------------------------------------------------------------------------------------------
struct VnArgs {
    const float *x;         // [N][3]
    const float *o3;        // [N][16][3]
    const float *ps;        // [B][2][heads][3]
    const float *wf_x, *wd_x;   // [heads]      column 0 of the VN weights
    const float *wf_o, *wd_o;   // [heads][16]  columns of the attention rows (permuted, zero for padding rows)
    const float *bn_g, *bn_b;   // [heads]
    const int *mol_of;
    float *pd;              // [N][heads][6]  p (3) and d (3)
    double *acc;            // [kBnReplicas][2][heads] sums of norms / squared norms (this layer), spread over
                            // replicas because same-address atomics serialise at the memory side
    float *x_out;           // [N][3]
    int n_atoms, heads;
};

// thread = (atom, channel c); 256 threads = 256/heads atoms per block.  The attention rows of the
// block's atoms and the VN weights are staged in LDS with coalesced 16-byte loads (each thread used to
// issue ~150 scalar global loads); the double sums go to global memory with one atomic per block and channel.
constexpr int kVnThreads = 256;
constexpr int kBnReplicas = 16;
__global__ void __launch_bounds__(kVnThreads) vn_stats_kernel(VnArgs a) {
    __shared__ __attribute__((aligned(16))) float s_o[64 * 48];      // up to 64 atoms x 16 rows x 3
    __shared__ __attribute__((aligned(16))) float s_w[2 * 32 * 16];  // wf_o | wd_o, [heads][16]
    __shared__ double red[2][kVnThreads];
    const int heads = a.heads, per_blk = kVnThreads / heads;
    const int la = threadIdx.x / heads, c = threadIdx.x % heads;
    const int atom0 = blockIdx.x * per_blk;
    const int n_here = min(per_blk, a.n_atoms - atom0);
    for (int i = threadIdx.x; i < n_here * 12; i += kVnThreads)
        reinterpret_cast<float4 *>(s_o)[i] = reinterpret_cast<const float4 *>(a.o3 + (size_t)atom0 * 48)[i];
    for (int i = threadIdx.x; i < heads * 16; i += kVnThreads) { s_w[i] = a.wf_o[i]; s_w[32 * 16 + i] = a.wd_o[i]; }
    __syncthreads();
    const int i = atom0 + la;
    double nv = 0.0, nv2 = 0.0;
    if (la < n_here) {
        const float *o = s_o + la * 48;
        const float *wf = s_w + c * 16, *wd = s_w + 32 * 16 + c * 16;
        const float *psf = a.ps + ((size_t)a.mol_of[i] * 2 * heads + c) * 3;
        const float *psd = psf + heads * 3;
        const float wfx = a.wf_x[c], wdx = a.wd_x[c];
        float p[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xk = a.x[i * 3 + k];
            float pp = wfx * xk, dd = wdx * xk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pp += wf[r] * o[r * 3 + k];
                dd += wd[r] * o[r * 3 + k];
            }
            p[k] = pp + psf[k];
            d[k] = dd + psd[k];
        }
        float *out = a.pd + ((size_t)i * heads + c) * 6;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
        nv = (double)nrm;
        nv2 = (double)nrm * (double)nrm;
    }
    red[0][threadIdx.x] = nv;
    red[1][threadIdx.x] = nv2;
    __syncthreads();
    if (threadIdx.x < heads) {
        double s = 0.0, s2 = 0.0;
        for (int k = 0; k < per_blk; ++k) { s += red[0][k * heads + threadIdx.x]; s2 += red[1][k * heads + threadIdx.x]; }
        double *acc = a.acc + (size_t)(blockIdx.x % kBnReplicas) * 2 * heads;
        atomicAdd(acc + threadIdx.x, s);
        atomicAdd(acc + heads + threadIdx.x, s2);
    }
}

// Evaluation-mode batch-norm (module.eval(): the running statistics instead of the batch's, shape_vn_layers.py:50-61 with
// BatchNorm1d in eval mode): the consumers of the batch sums (VnFold, DdpmFold, vn_apply_kernel) take mean and variance
// from sums over the batch, so they are handed sums that reproduce the running statistics for this batch size:
// s1 = mean N, s2 = (var + mean^2) N in replica 0, zeros elsewhere (double: mean and var come back to the last float bit).
__global__ void bn_eval_fill_kernel(const float *run_mean, const float *run_var, int n_layers, int heads, int n_atoms, double *acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = kBnReplicas * 2 * heads;
    if (i >= n_layers * per) return;
    const int l = i / per, r = (i % per) / (2 * heads), which = (i % (2 * heads)) / heads, c = i % heads;
    double v = 0.0;
    if (r == 0) {
        const double m = (double)run_mean[l * heads + c], var = (double)run_var[l * heads + c];
        v = which == 0 ? m * (double)n_atoms : (var + m * m) * (double)n_atoms;
    }
    acc[i] = v;
}

__global__ void vn_apply_kernel(VnArgs a) {
    __shared__ float red[256][3];
    const int heads = a.heads, per_blk = 256 / heads;
    const int la = threadIdx.x / heads, c = threadIdx.x % heads;
    const int i = blockIdx.x * per_blk + la;
    const bool ok = la < per_blk && i < a.n_atoms;
    float o[3] = {0.f, 0.f, 0.f};
    if (ok) {
        const double cnt = (double)a.n_atoms;
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < kBnReplicas; ++r) { s1 += a.acc[(size_t)r * 2 * heads + c]; s2 += a.acc[(size_t)r * 2 * heads + heads + c]; }
        const double mean = s1 / cnt;
        double var = s2 / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float meanf = (float)mean;
        const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
        const float *pd = a.pd + ((size_t)i * heads + c) * 6;
        float p[3] = {pd[0], pd[1], pd[2]};
        const float d[3] = {pd[3], pd[4], pd[5]};
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
        const float nbn = (nrm - meanf) * rstd * a.bn_g[c] + a.bn_b[c];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = p[k] / nrm * nbn;
        const float dot = p[0] * d[0] + p[1] * d[1] + p[2] * d[2];
        const float dsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float coef = dot / (dsq + 1e-6f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float neg = p[k] - coef * d[k];
            o[k] = 0.2f * p[k] + 0.8f * (dot >= 0.f ? p[k] : neg);
        }
    }
    red[threadIdx.x][0] = o[0]; red[threadIdx.x][1] = o[1]; red[threadIdx.x][2] = o[2];
    __syncthreads();
    if (ok && c < 3) {
        float res = 0.f, att = 0.f;
        for (int h = 0; h < heads; ++h) res += red[la * heads + h][c];
        const float *ob = a.o3 + (size_t)i * 48;
        for (int r = 0; r < 16; ++r) att += ob[r * 3 + c];          // padding rows are zero
        a.x_out[i * 3 + c] = a.x[i * 3 + c] + (att / heads + res / heads);
    }
}

// kNN graph and edge weights in one launch (knn_kernel + edge_weight_kernel: the weights of an atom's slots need its own
// neighbour list only).  One wave per atom: the molecule's span comes from the precomputed table (no mol_of -> mol_off
// chase), every candidate's squared distance is computed once and published in an LDS row, the rank loop compares against
// broadcast reads of that row (the two kernels re-read the coordinates of every atom of the molecule from global memory
// for every comparison), the neighbour list reaches the weight stage through LDS, and the first Linear's weights are
// staged in LDS once per workgroup (edge_weight_kernel: forty dependent-address global loads per lane).  Same arithmetic,
// bit for bit: the distances, the tie rule (lower index first) and the fp32 MFMA sequence are those of the two kernels.
// Molecules of up to kGraphCap atoms (the host falls back to the two kernels otherwise).
constexpr int kGraphCap = 128;
constexpr int kGraphWaves = 8;
struct GraphArgs {
    const float *x;
    const int2 *mol_span;   // [N] first / one-past-last atom of the atom's molecule
    int n_atoms, k, kp;
    int *nbr;               // [N][KP]
    const float *w1, *b1, *g, *be, *w2, *b2;   // edge-weight MLP (as EdgeWeightArgs)
    float *ew;              // [N][KP]
    int *span_flag;         // status flag raised if a molecule exceeds kGraphCap atoms (the max_mol_atoms hint was wrong)
    unsigned long long *stamps;   // diagnostic build only
};
template <int H, int KP>
__global__ void __launch_bounds__(kGraphWaves * 64)
graph_kernel(GraphArgs a) {
    constexpr int NT = H / 16;
    constexpr int APW = KP >= 16 ? 1 : 16 / KP;          // atoms per wave = the atoms of one 16-slot tile
    constexpr int LPA = 64 / APW;                         // candidate lanes per atom
    constexpr int NQ = kGraphCap / LPA;                   // candidate chunks per lane
    constexpr int TPW = KP > 16 ? KP / 16 : 1;            // tiles per wave
    __shared__ float w1s[H * 20];
    __shared__ unsigned long long dkey[kGraphWaves][APW][kGraphCap];   // (distance bits, index) of the molecule's atoms
    __shared__ float xs[kGraphWaves][APW][kGraphCap][3];  // the molecule's coordinates (the weight stage's neighbours live there)
    __shared__ int row[kGraphWaves][APW * KP];
    __shared__ __attribute__((aligned(16))) float prm[4][H];        // b1 | gamma | beta | w2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int sub = lane / LPA, l = lane % LPA;
    const int i_raw = (blockIdx.x * kGraphWaves + wave) * APW + sub;
    const bool valid = i_raw < a.n_atoms;
    const int i = valid ? i_raw : a.n_atoms - 1;
    SM_TICK(a.stamps, 0);
    const int2 span = a.mol_span[i];
    for (int idx = threadIdx.x; idx < H * 20; idx += kGraphWaves * 64) w1s[idx] = a.w1[idx];
    // the weight stage's small operands go through LDS as well (requested now: no L2 round trip behind the graph stage,
    // and no registers held across it)
    for (int idx = threadIdx.x; idx < 4 * H; idx += kGraphWaves * 64) {
        const int which = idx / H, f = idx % H;
        prm[which][f] = (which == 0 ? a.b1 : which == 1 ? a.g : which == 2 ? a.be : a.w2)[f];
    }
    const float b2 = a.b2[0];
    const float *x = a.x;
    const int s = span.x, cnt = min(span.y - span.x, kGraphCap), self = i - s;
    if (span.y - span.x > kGraphCap && l == 0) *a.span_flag = 1;
    const float xi = x[i * 3], yi = x[i * 3 + 1], zi = x[i * 3 + 2];
    float dc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = l + LPA * q;
        dc[q] = 0.f;
        if (c < cnt) {
            const int j = s + c;
            const float xj = x[j * 3], yj = x[j * 3 + 1], zj = x[j * 3 + 2];
            const float dx = xj - xi, dy = yj - yi, dz = zj - zi;
            dc[q] = dist2_rounded(dx, dy, dz);
            // order by distance, ties by index: squared distances are non-negative floats, whose bit patterns order like
            // unsigned integers; the atom itself gets the largest key, so it precedes no candidate
            dkey[wave][sub][c] = c == self ? ~0ull : ((unsigned long long)__builtin_bit_cast(unsigned, dc[q]) << 32) | (unsigned)c;
            xs[wave][sub][c][0] = xj; xs[wave][sub][c][1] = yj; xs[wave][sub][c][2] = zj;
        }
    }
    for (int sl = l; sl < KP; sl += LPA) row[wave][sub * KP + sl] = -1;
    SM_STAMP(a.stamps, 1);
    __syncthreads();                       // w1s staged; this wave's rows of distances written
    SM_TICK(a.stamps, 2);
    int rank[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) rank[q] = 0;
    // candidate chunks the wave's molecules really have (a wave-uniform bound: the compare chain of an empty chunk is pure
    // waste, and with six waves per SIMD this loop is bound by vector issue: its phase 6.0 -> 2.8 us for molecules below
    // 32 atoms; taking the distances from the candidates' registers by v_readlane instead of the LDS row changes nothing,
    // packing (distance, index) into one 64-bit key -- one compare instead of five -- does)
    int nq_w = __builtin_amdgcn_readlane((cnt + LPA - 1) / LPA, 0);
    if constexpr (APW > 1) nq_w = max(nq_w, __builtin_amdgcn_readlane((cnt + LPA - 1) / LPA, LPA));
    unsigned long long kc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) kc[q] = ((unsigned long long)__builtin_bit_cast(unsigned, dc[q]) << 32) | (unsigned)(l + LPA * q);
    for (int o = 0; o < cnt; ++o) {
        const unsigned long long k_o = dkey[wave][sub][o];      // one 64-bit compare per candidate: (d_o, o) < (d_c, c)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q >= nq_w) break;
            rank[q] += k_o < kc[q];
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = l + LPA * q;
        if (c < cnt && c != self && rank[q] < a.k) row[wave][sub * KP + rank[q]] = s + c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    SM_TICK(a.stamps, 3);
    if (valid)
        for (int sl = l; sl < KP; sl += LPA) a.nbr[(size_t)i * KP + sl] = row[wave][sub * KP + sl];
    // ---- edge weights of the wave's slots: edge_weight_kernel's tile (column n = slot n of the wave's atoms) ----
    float cen[5];
    rbf_centres(g, cen);
    const int atom0 = (blockIdx.x * kGraphWaves + wave) * APW;
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int slot = tt * 16 + n;                     // slot of the wave: atom slot / KP, neighbour slot % KP
        const int asub = slot / KP;                       // which of the wave's atoms
        const int ia_raw = atom0 + asub;
        const bool in_range = ia_raw < a.n_atoms;
        const int ia = in_range ? ia_raw : a.n_atoms - 1;
        const int jraw = in_range ? row[wave][slot] : -1;
        const bool ok = in_range && jraw >= 0;
        // both ends from the LDS copy of the molecule (ia's own molecule starts at its span; lanes of the other atom of the
        // wave read that atom's table): index of the centre and of the neighbour inside the molecule
        const int s_a = __shfl(s, asub * LPA), self_a = __shfl(self, asub * LPA);
        const int cj = ok ? jraw - s_a : self_a;
        const float *pa = xs[wave][asub][self_a], *pj = xs[wave][asub][cj];
        const float r0 = pa[0] - pj[0], r1 = pa[1] - pj[1], r2 = pa[2] - pj[2];
        float rb[5];
        rbf_dlayout(sqrtf(r0 * r0 + r1 * r1 + r2 * r2), cen, rb);
        float hid[NT * 4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 b = ldg4(prm[0] + 16 * t + 4 * g);
            f32x4 acc = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int s5 = 0; s5 < 5; ++s5) acc = mfma16(w1s[(16 * t + n) * 20 + 4 * s5 + g], rb[s5], acc);
            hid[4 * t] = acc[0]; hid[4 * t + 1] = acc[1]; hid[4 * t + 2] = acc[2]; hid[4 * t + 3] = acc[3];
        }
        SM_TICK(a.stamps, 4);
        ln_relu_dlayout<NT>(hid, prm[1], prm[2], g);
        SM_TICK(a.stamps, 5);
        float p = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 w = ldg4(prm[3] + 16 * t + 4 * g);
            p += w.x * hid[4 * t] + w.y * hid[4 * t + 1] + w.z * hid[4 * t + 2] + w.w * hid[4 * t + 3];
        }
        p = sum_groups(p) + b2;
        if (g == 0 && in_range) a.ew[(size_t)ia * KP + slot % KP] = ok ? 1.0f / (1.0f + expf(-p)) : 0.f;
    }
    SM_STAMP(a.stamps, 6);
}

// ---------------------------------------------------------------------------------------------
// DDPM posterior step (molopt_score_model.py:653-681): q_pos_posterior (:400-404) + noise,
// log_softmax, index_to_log_onehot (:64-68), q_v_posterior (:377-385) with the uniform mixing
// of q_v_pred / q_v_pred_one_timestep (:323-364), Gumbel-argmax sampling (:98-104).
// ---------------------------------------------------------------------------------------------
// Per-chain parameters that change from call to call (noise source, trajectory buffers).  They live in DEVICE memory,
// written by set_chain_params_kernel before the first step, so that the captured step graph does not depend on them:
// a new seed or new trajectory buffers replay the same executable.
struct ChainParams {
    unsigned long long seed;
    const float *eps, *u;    // host-fed noise of ALL steps ([S][N][3], [S][N][C]) or nullptr
    float *tr_pos; int64_t *tr_v; float *tr_v0; float *tr_vt; float *tr_pos_cond; float *tr_v_cond;  // trajectories or nullptr
    const double *guide_draws;   // point-cloud guidance: host-fed uniforms [S][5][N] (parity mode) or nullptr (device Philox)
    int step_base;           // index of the chain's first reverse step (0 unless a chain is resumed mid-way): noise and
                             // trajectory rows are indexed by step - step_base
};
__global__ void set_chain_params_kernel(ChainParams *dst, ChainParams v, int *step_counter) { *dst = v; *step_counter = v.step_base; }

// The coordinate update of the LAST layer (vn_apply_kernel's work: batch-norm of ||p||, VN-leaky-ReLU, mean over the channels,
// x += ...) inside the DDPM kernel: 16 lanes per atom there are the 16 channels here, and the predicted position is consumed
// nowhere else (chains without guidance).  Same arithmetic as the update folded into the x2h kernels (sm_edge16.h, VnFold).
struct DdpmFold {
    const float *pd;          // [N][heads][6]
    const double *acc;        // [kBnReplicas][2][heads] batch sums of the last layer
    const float *bn_g, *bn_b; // [heads]
    const float *xsum;        // [N][3] sum over the attention rows (h2x epilogue)
    const float *x_old;       // [N][3]
    float *pred_out;          // [N][3] the predicted position (kept for readers of the score's output)
    int heads, enable;
};
struct DdpmArgs {
    const float *pred_pos;   // [N][3]
    const float *pred_v;     // [N][C]
    const float *x_t;        // [N][3]
    const int64_t *v_t;      // [N]
    const int *mol_of;
    int t_first;             // t = t_first - step for every molecule
    const float *c0, *ct, *logvar, *log_a, *log_1ma, *log_abar, *log_1mabar;   // [T] tables
    const ChainParams *cp;   // noise source and trajectory buffers of this chain (device memory)
    const int *step_cur;     // this step's index (written by time_embed_kernel) or nullptr (= 0)
    int *step_ptr;           // device step counter, advanced here for the NEXT step (nobody reads it in this kernel)
    float *x_next;           // [N][3]
    int64_t *v_next;         // [N]
    int n_atoms, C;
    DdpmFold vf;             // ddpm_step16_kernel only
};
template <int MAXC>
__global__ void ddpm_step_kernel(DdpmArgs aa) {
    struct : DdpmArgs, ChainParams {} a;
    static_cast<DdpmArgs &>(a) = aa;
    static_cast<ChainParams &>(a) = *aa.cp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int step = a.step_cur ? *a.step_cur : 0;
    if (i < a.n_atoms) {
        const int C = a.C;
        const int t = a.t_first - step;
        const size_t so = (size_t)(step - a.step_base) * a.n_atoms + i;
        float e3[3], uu[MAXC];
        if (a.eps) {
#pragma unroll
            for (int k = 0; k < 3; ++k) e3[k] = a.eps[so * 3 + k];
            for (int c = 0; c < C; ++c) uu[c] = a.u[so * C + c];
        } else {
            Philox ph{(uint32_t)a.seed, (uint32_t)(a.seed >> 32)};
            uint32_t r[4];
            ph((uint32_t)i, (uint32_t)step, 0u, 0x5eedu, r);
            const float r0 = sqrtf(-2.0f * logf(u01_open(r[0]))), r1 = sqrtf(-2.0f * logf(u01_open(r[2])));
            e3[0] = r0 * cosf(6.283185307179586f * u01_half(r[1]));
            e3[1] = r0 * sinf(6.283185307179586f * u01_half(r[1]));
            e3[2] = r1 * cosf(6.283185307179586f * u01_half(r[3]));
            for (int c0 = 0; c0 < C; c0 += 4) {
                ph((uint32_t)i, (uint32_t)step, (uint32_t)(1 + c0 / 4), 0x5eedu, r);
                for (int k = 0; k < 4 && c0 + k < C; ++k) uu[c0 + k] = u01_half(r[k]);
            }
        }
        // ---- positions
        const float c0 = a.c0[t], ct = a.ct[t];
        const float sig = t != 0 ? expf(0.5f * a.logvar[t]) : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float pp = a.pred_pos[i * 3 + k];
            const float xn = (c0 * pp + ct * a.x_t[i * 3 + k]) + sig * e3[k];
            a.x_next[i * 3 + k] = xn;
            if (a.tr_pos) a.tr_pos[so * 3 + k] = xn;
            if (a.tr_pos_cond) a.tr_pos_cond[so * 3 + k] = pp;
        }
        // ---- atom types
        float lg[MAXC];
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) { lg[c] = a.pred_v[(size_t)i * C + c]; mx = fmaxf(mx, lg[c]); }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(lg[c] - mx);
        const float lse = mx + logf(se);
        const int tm1 = t > 0 ? t - 1 : 0;
        const float logC = logf((float)C);
        const float la_prev = a.log_abar[tm1], l1_prev = a.log_1mabar[tm1] - logC;
        const float la_t = a.log_a[t], l1_t = a.log_1ma[t] - logC;
        const int vt = (int)a.v_t[i];
        float un[MAXC];
        float umx = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float lv0 = lg[c] - lse;
            if (a.tr_v_cond) a.tr_v_cond[so * C + c] = lg[c];
            if (a.tr_v0) a.tr_v0[so * C + c] = lv0;
            const float A1 = lv0 + la_prev;
            const float m1 = fmaxf(A1, l1_prev);
            const float q1 = m1 + logf(expf(A1 - m1) + expf(l1_prev - m1));
            const float lvt = (c == vt ? 0.f : -69.07755278982137f) + la_t;   // log(clamp(onehot, 1e-30))
            const float m2 = fmaxf(lvt, l1_t);
            const float q2 = m2 + logf(expf(lvt - m2) + expf(l1_t - m2));
            un[c] = q1 + q2;
            umx = fmaxf(umx, un[c]);
        }
        float us = 0.f;
        for (int c = 0; c < C; ++c) us += expf(un[c] - umx);
        const float ulse = umx + logf(us);
        int best = 0;
        float bestv = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float lp = un[c] - ulse;
            if (a.tr_vt) a.tr_vt[so * C + c] = lp;
            const float gum = -logf(-logf(uu[c] + 1e-30f) + 1e-30f);
            const float sc = gum + lp;
            if (sc > bestv) { bestv = sc; best = c; }
        }
        a.v_next[i] = best;
        if (a.tr_v) a.tr_v[so] = best;
    }
    if (a.step_ptr && blockIdx.x == 0 && threadIdx.x == 0) *a.step_ptr = step + 1;
}

// Same step with 16 lanes per atom (one class per lane, C <= 16): the per-class transcendental chains
// run in parallel and the reductions are DPP row operations.
__global__ void __launch_bounds__(256) ddpm_step16_kernel(DdpmArgs aa) {
    struct : DdpmArgs, ChainParams {} a;
    static_cast<DdpmArgs &>(a) = aa;
    static_cast<ChainParams &>(a) = *aa.cp;
    const int gid = blockIdx.x * 256 + threadIdx.x;               // (launched with 256 threads: no blockDim read)
    const int i_raw = gid >> 4, c = gid & 15;
    const int step = a.step_cur ? *a.step_cur : 0;
    const bool atom_ok = i_raw < a.n_atoms;
    const int i = atom_ok ? i_raw : a.n_atoms - 1;
    const int C = a.C;
    const bool cls = c < C;
    const int t = a.t_first - step;
    const size_t so = (size_t)(step - a.step_base) * a.n_atoms + i;
    float eps = 0.f, uu = 0.5f;
    if (a.eps) {
        if (c < 3) eps = a.eps[so * 3 + c];
        if (cls) uu = a.u[so * C + c];
    } else {
        Philox ph{(uint32_t)a.seed, (uint32_t)(a.seed >> 32)};
        uint32_t r[4];
        if (c < 3) {
            ph((uint32_t)i, (uint32_t)step, 0u, 0x5eedu, r);
            const float rad = sqrtf(-2.0f * logf(u01_open(c < 2 ? r[0] : r[2])));
            const float ang = 6.283185307179586f * u01_half(c < 2 ? r[1] : r[3]);
            eps = rad * (c == 1 ? sinf(ang) : cosf(ang));
        }
        ph((uint32_t)i, (uint32_t)step, (uint32_t)(1 + (c >> 2)), 0x5eedu, r);
        uu = u01_half(r[c & 3]);
    }
    // ---- predicted position: given, or the last layer's coordinate update done here (lane = channel)
    float pp_fold = 0.f;
    if (a.vf.enable) {
        __shared__ double sred[32];
        const int HD = a.vf.heads;
        if (threadIdx.x < 32) {
            const int cc = threadIdx.x & 15, which = threadIdx.x >> 4;
            double t = 0.0;
            if (cc < HD) for (int r = 0; r < kBnReplicas; ++r) t += a.vf.acc[(size_t)r * 2 * HD + which * HD + cc];
            sred[threadIdx.x] = t;
        }
        __syncthreads();
        float o[3] = {0.f, 0.f, 0.f};
        if (c < HD) {
            const double cnt = (double)a.n_atoms;
            const double mean = sred[c] / cnt;
            double var = sred[16 + c] / cnt - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float meanf = (float)mean;
            const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
            const float *pdp = a.vf.pd + ((size_t)i * HD + c) * 6;
            float p[3] = {pdp[0], pdp[1], pdp[2]};
            const float d[3] = {pdp[3], pdp[4], pdp[5]};
            const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
            const float nbn = (nrm - meanf) * rstd * a.vf.bn_g[c] + a.vf.bn_b[c];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = p[k] / nrm * nbn;
            const float dot = p[0] * d[0] + p[1] * d[1] + p[2] * d[2];
            const float dsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const float coef = dot / (dsq + 1e-6f);
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = 0.2f * p[k] + 0.8f * (dot >= 0.f ? p[k] : p[k] - coef * d[k]);
        }
        float res[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) res[k] = seg_sum<16>(o[k]);              // over the channels of the atom
        if (c < 3) {
            const float r = c == 0 ? res[0] : (c == 1 ? res[1] : res[2]);
            pp_fold = a.vf.x_old[i * 3 + c] + (a.vf.xsum[i * 3 + c] / HD + r / HD);
            if (atom_ok) a.vf.pred_out[i * 3 + c] = pp_fold;
        }
    }
    // ---- positions (lanes 0..2)
    if (c < 3) {
        const float pp = a.vf.enable ? pp_fold : a.pred_pos[i * 3 + c];
        const float sig = t != 0 ? expf(0.5f * a.logvar[t]) : 0.f;
        const float xn = (a.c0[t] * pp + a.ct[t] * a.x_t[i * 3 + c]) + sig * eps;
        if (atom_ok) {
            a.x_next[i * 3 + c] = xn;
            if (a.tr_pos) a.tr_pos[so * 3 + c] = xn;
            if (a.tr_pos_cond) a.tr_pos_cond[so * 3 + c] = pp;
        }
    }
    // ---- atom types (one class per lane)
    const float lg = cls ? a.pred_v[(size_t)i * C + c] : -INFINITY;
    const float mx = seg_max<16>(lg);
    const float se = seg_sum<16>(cls ? expf(lg - mx) : 0.f);
    const float lv0 = lg - (mx + logf(se));
    const int tm1 = t > 0 ? t - 1 : 0;
    const float logC = logf((float)C);
    const float la_prev = a.log_abar[tm1], l1_prev = a.log_1mabar[tm1] - logC;
    const float la_t = a.log_a[t], l1_t = a.log_1ma[t] - logC;
    const int vt = (int)a.v_t[i];
    const float A1 = lv0 + la_prev;
    const float m1 = fmaxf(A1, l1_prev);
    const float q1 = m1 + logf(expf(A1 - m1) + expf(l1_prev - m1));
    const float lvt = (c == vt ? 0.f : -69.07755278982137f) + la_t;   // log(clamp(onehot, 1e-30))
    const float m2 = fmaxf(lvt, l1_t);
    const float q2 = m2 + logf(expf(lvt - m2) + expf(l1_t - m2));
    const float un = cls ? q1 + q2 : -INFINITY;
    const float umx = seg_max<16>(un);
    const float us = seg_sum<16>(cls ? expf(un - umx) : 0.f);
    const float lp = un - (umx + logf(us));
    const float sc = cls ? (-logf(-logf(uu + 1e-30f) + 1e-30f)) + lp : -INFINITY;
    const float best = seg_max<16>(sc);
    // first index attaining the maximum (torch argmax tie rule): min over lanes of (sc == best ? c : 99)
    const float cand = (sc == best) ? (float)c : 99.f;
    const float win = -seg_max<16>(-cand);
    if (atom_ok && cls) {
        if (a.tr_v_cond) a.tr_v_cond[so * C + c] = lg;
        if (a.tr_v0) a.tr_v0[so * C + c] = lv0;
        if (a.tr_vt) a.tr_vt[so * C + c] = lp;
    }
    if (atom_ok && c == 0) {
        a.v_next[i] = (int64_t)win;
        if (a.tr_v) a.tr_v[so] = (int64_t)win;
    }
    if (a.step_ptr && gid == 0) *a.step_ptr = step + 1;
}

// log_sample_categorical (molopt_score_model.py:98-104)
__global__ void gumbel_argmax_kernel(const float *logits, const float *u, int n, int C, uint64_t seed, int64_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Philox ph{(uint32_t)seed, (uint32_t)(seed >> 32)};
    int best = 0;
    float bestv = -INFINITY;
    uint32_t r[4];
    for (int c = 0; c < C; ++c) {
        float uc;
        if (u) uc = u[(size_t)i * C + c];
        else { if ((c & 3) == 0) ph((uint32_t)i, 0xFFFFFFFFu, (uint32_t)(c >> 2), 0xca7u, r); uc = u01_half(r[c & 3]); }
        const float sc = -logf(-logf(uc + 1e-30f) + 1e-30f) + logits[(size_t)i * C + c];
        if (sc > bestv) { bestv = sc; best = c; }
    }
    out[i] = best;
}

__global__ void copy_state_kernel(const float *x_src, const int64_t *v_src, float *x_dst, int64_t *v_dst, int n_atoms) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_atoms * 3 && x_dst) x_dst[i] = x_src[i];
    if (i < n_atoms && v_dst) v_dst[i] = v_src[i];
}

// diagnostic: shader-clock / real-time stamp pairs (clock = d(memtime) / d(memrealtime) * 100 MHz)
__global__ void clock_stamp_kernel(unsigned long long *out, const int *step_cur, int cap) {
    const int i = step_cur ? *step_cur : 0;
    if (threadIdx.x == 0 && i < cap) {
        out[2 * i] = __builtin_amdgcn_s_memtime();
        out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------
// Point-cloud shape guidance (models/molopt_score_model.py:699-740, applied to the predicted x0 of the steps with
// t > grad_step, :583-586): an atom whose three nearest cloud points are on average farther than `radius` is pulled
// towards their mean by a random fraction in [ratio, 0.8), re-checked, and pulled again up to five times.
// The reference does this on the host (sklearn KD-tree, numpy float64, one D2H + H2D round trip per step); here it is a
// kernel between the score evaluation and the posterior step: 16 lanes per atom scan the cloud (brute force, float64 as
// the KD-tree), every atom runs its own five-iteration loop.  Uniform draws: the reference consumes np.random.random()
// in the order of the currently-far atoms of each iteration; parity mode is fed the recorded draw of every
// (step, iteration, atom), throughput mode uses Philox keyed by the same triple.
// ---------------------------------------------------------------------------------------------
struct PcGuideArgs {
    float *pred_pos;          // [N][3] in/out
    const double *cloud;      // [P][3]
    const ChainParams *cp;
    const int *step_cur;
    int n_atoms, n_points, t_first, grad_step;
    double radius;
    double ratio;             // lower end of the pull fraction: u * (0.8 - ratio) + ratio (the reference's default: 0.2)
};

// Three nearest cloud points of p.  Every candidate is one 64-bit key: the bits of its squared distance (non-negative
// doubles order like unsigned integers) with the low 12 bits replaced by ... nothing: keys stay exact; the point index
// travels beside the key.  Each of the 16 lanes of an atom scans every 16th point (the cloud sits in LDS) and keeps its
// three smallest keys in a branch-free sorted triple; the lanes' triples are merged pairwise over four xor-shuffle steps
// (smallest three of six, ties broken by the index so that both partners end with the same triple).
struct Top3 { double d[3]; int i[3]; };
SM_DEV bool key_less(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }
SM_DEV void top3_insert(Top3 &t, double d, int i) {
    // compare against the three slots from the back; selects instead of branches
    const bool l2 = key_less(d, i, t.d[2], t.i[2]), l1 = key_less(d, i, t.d[1], t.i[1]), l0 = key_less(d, i, t.d[0], t.i[0]);
    const double n2 = l1 ? t.d[1] : (l2 ? d : t.d[2]); const int j2 = l1 ? t.i[1] : (l2 ? i : t.i[2]);
    const double n1 = l0 ? t.d[0] : (l1 ? d : t.d[1]); const int j1 = l0 ? t.i[0] : (l1 ? i : t.i[1]);
    const double n0 = l0 ? d : t.d[0];                  const int j0 = l0 ? i : t.i[0];
    t.d[0] = n0; t.i[0] = j0; t.d[1] = n1; t.i[1] = j1; t.d[2] = n2; t.i[2] = j2;
}
SM_DEV double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
// cloud: LDS copy [P][3] doubles; identical result in all 16 lanes of the atom
SM_DEV Top3 pc_query(const double *cloud, int n_points, const double (&p)[3], int l16) {
    Top3 t{{1e300, 1e300, 1e300}, {0x7ffffff0, 0x7ffffff1, 0x7ffffff2}};
    for (int c = l16; c < n_points; c += 16) {
        const double dx = p[0] - cloud[c * 3], dy = p[1] - cloud[c * 3 + 1], dz = p[2] - cloud[c * 3 + 2];
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        if (__any(d2 <= t.d[2])) top3_insert(t, d2, c);       // most points beat nobody's third best: skip the insert wave-wide
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        Top3 o;
#pragma unroll
        for (int k = 0; k < 3; ++k) { o.d[k] = shfl_xor_f64(t.d[k], m); o.i[k] = __shfl_xor(t.i[k], m, 64); }
#pragma unroll
        for (int k = 0; k < 3; ++k) top3_insert(t, o.d[k], o.i[k]);      // partner entries beyond the third smallest fall off the end
    }
    return t;
}

__global__ void __launch_bounds__(256) pc_guidance_kernel(PcGuideArgs a) {
    extern __shared__ double pc_cloud[];                            // [P][3]
    const int step = a.step_cur ? *a.step_cur : 0;
    if (a.t_first - step <= a.grad_step) return;                   // `if i > grad_step` (molopt_score_model.py:585)
    for (int i = threadIdx.x; i < a.n_points * 3; i += blockDim.x) pc_cloud[i] = a.cloud[i];
    __syncthreads();
    const double *cloud = pc_cloud;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int atom_raw = gid >> 4, l16 = gid & 15;
    const int atom = atom_raw < a.n_atoms ? atom_raw : a.n_atoms - 1;
    double p[3] = {(double)a.pred_pos[atom * 3], (double)a.pred_pos[atom * 3 + 1], (double)a.pred_pos[atom * 3 + 2]};
    Top3 t = pc_query(cloud, a.n_points, p, l16);
    bool far = (sqrt(t.d[0]) + sqrt(t.d[1]) + sqrt(t.d[2])) / 3.0 > a.radius;
    bool changed = false;
    const ChainParams cp = *a.cp;
    for (int j = 0; j < 5; ++j) {
        if (!__any(far)) break;
        if (far) {
            double u;
            if (cp.guide_draws) {
                u = cp.guide_draws[((size_t)(step - cp.step_base) * 5 + j) * a.n_atoms + atom];
            } else {
                Philox ph{(uint32_t)cp.seed, (uint32_t)(cp.seed >> 32)};
                uint32_t r[4];
                ph((uint32_t)atom, (uint32_t)step, (uint32_t)(100 + j), 0x9c1du, r);
                u = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
            }
            const double scalar = u * (0.8 - a.ratio) + a.ratio;   // np.random.random() * (0.8 - ratio) + ratio
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double nearest = (cloud[t.i[0] * 3 + k] + cloud[t.i[1] * 3 + k] + cloud[t.i[2] * 3 + k]) / 3.0;
                p[k] = p[k] - scalar * (p[k] - nearest);
            }
            changed = true;
        }
        // the query is wave-uniform control flow (shuffles): lanes of atoms that are done run it on their final point
        t = pc_query(cloud, a.n_points, p, l16);
        if (far && (sqrt(t.d[0]) + sqrt(t.d[1]) + sqrt(t.d[2])) / 3.0 < a.radius) far = false;
    }
    if (changed && atom_raw < a.n_atoms && l16 < 3) a.pred_pos[atom * 3 + l16] = (float)(l16 == 0 ? p[0] : (l16 == 1 ? p[1] : p[2]));
}
