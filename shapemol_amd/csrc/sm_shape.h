// Frozen shape encoder on the device: VN_DGCNN_Encoder.forward (reference: models/shape_pointcloud_modelAE.py:231-255)
// with get_graph_feature_cross / dense knn (models/shape_vn_layers.py:257-292) and VNLinearLeakyReLU with train-mode
// VNBatchNorm (models/shape_vn_layers.py:41-61,95-124; the auto-encoder is never put in eval mode, utils/shape.py:226-238).
//
// Formulation.  A DGCNN block applies one Linear over the channels of the edge feature [x_j - x_i | x_i]:
//     W [x_j - x_i ; x_i] = W1 x_j + (W2 - W1) x_i
// so the per-EDGE product (N k columns) becomes two per-POINT products (N columns) and an add per edge: k = 20 times
// fewer FLOPs (512 points, k = 20, 4 blocks of 256 -> 128 channels).  Per block:
//     se_knn_kernel      d2 = |x_i|^2 + |x_j|^2 - 2 x_i.x_j over the 3C-dim point features (fp32 MFMA Gram tiles), 20 smallest
//     se_point_linear    Y = [W1 ; W2 - W1 ; D1 ; D2 - D1] h        (fp32 MFMA, per point and vector component)
//     se_edge_stats      p_ij = Yf1[j] + Yf2[i]; batch sums of ||p|| per channel (train-mode batch-norm)
//     se_edge_apply      normalise, VN-leaky-ReLU against d_ij = Yd1[j] + Yd2[i], mean over the k neighbours -> next h
// and se_head_* for conv_c (Linear 4C -> latent, BatchNorm1d, shared direction, mean over the points).
// Layouts: h [B][N][C][3] fp32; Y [B][N][4 C'][3]; idx [B][N][k] i32.
#pragma once
#include "sm_device.h"

constexpr int kSeK = 20;            // neighbours (num_k of the shipped checkpoint's config; checked on the host)
constexpr int kSeReplicas = 16;     // replicated double accumulators (same-address atomics serialise)

// ---- squared norms of the point features ------------------------------------------------------------------------
__global__ void se_sqnorm_kernel(const float *h, int n_points_total, int D, int ld, float *xx) {
    const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= n_points_total) return;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) { const float v = h[(size_t)p * ld + k]; s += v * v; }
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) xx[p] = s;
}

// ---- k nearest neighbours in feature space (self included, as the reference's topk of -d2) -----------------------
// One workgroup (4 waves) per (shape, 16-row block): d2 of the 16 rows against all N points into LDS, then every wave
// selects the 20 smallest of 4 rows.  D % 16 == 0: Gram tiles on the fp32 matrix cores, k-order permuted so that a lane
// group reads a contiguous quarter of the feature vector; D == 3: direct differences.
template <int D>
__global__ void __launch_bounds__(256) se_knn_kernel(const float *h, int ld, const float *xx, int N, int *idx) {
    extern __shared__ float d2s[];                      // [16][N]
    const int b = blockIdx.y, r0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const float *hb = h + (size_t)b * N * ld;
    if constexpr (D == 3) {
        for (int e = threadIdx.x; e < 16 * N; e += 256) {
            const int r = e / N, c = e % N;
            const float *pi = hb + (size_t)min(r0 + r, N - 1) * ld, *pj = hb + (size_t)c * ld;
            const float dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
            d2s[e] = dx * dx + dy * dy + dz * dz;
        }
    } else {
        static_assert(D % 16 == 0, "feature dimension");
        constexpr int Q = D / 4;                         // floats per lane group
        const float *arow = hb + (size_t)min(r0 + n, N - 1) * ld + g * Q;
        for (int ct = wave; ct < N / 16; ct += 4) {
            const float *brow = hb + (size_t)(ct * 16 + n) * ld + g * Q;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < Q; s += 4) {
                const float4 a = ldg4(arow + s), bq = ldg4(brow + s);
                acc = mfma16(a.x, bq.x, acc); acc = mfma16(a.y, bq.y, acc);
                acc = mfma16(a.z, bq.z, acc); acc = mfma16(a.w, bq.w, acc);
            }
            const float xj = xx[(size_t)b * N + ct * 16 + n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                // accumulator register r of lane (n, g): row 4g + r, column n
                const int row = 4 * g + r;
                const float xi = xx[(size_t)b * N + min(r0 + row, N - 1)];
                d2s[row * N + ct * 16 + n] = (xi + xj) - 2.f * acc[r];
            }
        }
    }
    __syncthreads();
    for (int rr = wave; rr < 16; rr += 4) {
        if (r0 + rr >= N) continue;
        float *row = d2s + rr * N;
        for (int kk = 0; kk < kSeK; ++kk) {
            float best = INFINITY; int bi = 0x7fffffff;
            for (int c = lane; c < N; c += 64) { const float v = row[c]; if (v < best) { best = v; bi = c; } }
            for (int m = 1; m < 64; m <<= 1) {
                const float ov = __shfl_xor(best, m, 64); const int oi = __shfl_xor(bi, m, 64);
                if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { idx[((size_t)b * N + r0 + rr) * kSeK + kk] = bi; row[bi] = INFINITY; }
            __builtin_amdgcn_s_waitcnt(0);               // the removal must be visible to the next scan of this wave
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- per-point products Y[p][m][xyz] = sum_c W[m][c] h[p][c][xyz]  (M rows, K = C) on the fp32 matrix cores ----------
// wimg: A fragments, wimg[((t * (K/16) + s4) * 64 + lane) * 4 + r] = W[16 t + (lane & 15)][4 (4 s4 + r) + (lane >> 4)]
// One workgroup (4 waves) per 16 columns (column = (point, component)); the waves split the row tiles.
template <int K>
__global__ void __launch_bounds__(256) se_point_linear_kernel(const float *h, int ld, const float *wimg, int n_cols, int M, float *y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int col = min(blockIdx.x * 16 + n, n_cols - 1);
    const int p = col / 3, c3 = col % 3;
    float x[K / 4];                                       // B operand: k = 4 s + g
#pragma unroll
    for (int s = 0; s < K / 4; ++s) x[s] = h[(size_t)p * ld + (4 * s + g) * 3 + c3];
    for (int t = wave; t < M / 16; t += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < K / 16; ++s4) {
            const float4 a = ldg4(wimg + ((size_t)(t * (K / 16) + s4) * 64 + lane) * 4);
            acc = mfma16(a.x, x[4 * s4 + 0], acc); acc = mfma16(a.y, x[4 * s4 + 1], acc);
            acc = mfma16(a.z, x[4 * s4 + 2], acc); acc = mfma16(a.w, x[4 * s4 + 3], acc);
        }
        if (blockIdx.x * 16 + n < n_cols) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[((size_t)p * M + 16 * t + 4 * g + r) * 3 + c3] = acc[r];
        }
    }
}

// ---- edge stage -----------------------------------------------------------------------------------------------------
struct SeEdgeArgs {
    const float *y;          // [P][4 C][3]: Yf1 | Yf2 | Yd1 | Yd2 (layer 0: unused, see x / w0)
    const float *x;          // layer 0: points [P][3]
    const float *w0f, *w0d;  // layer 0: [C][2] weights of map_to_feat / map_to_dir
    const int *idx;          // [P][k] neighbour index inside the shape
    const float *bn_g, *bn_b;
    double *acc;             // [kSeReplicas][2][C]
    float *h_out;            // [P][C][3] (row stride h_ld floats per point: the four blocks' outputs interleave for conv_c)
    int n_total, N, C, h_ld, h_off;
};

// p, d of edge (i, j) for channel c
template <bool L0>
SM_DEV void se_edge_pd(const SeEdgeArgs &a, int i, int j, int c, float (&p)[3], float (&d)[3]) {
    if constexpr (L0) {
        const float wf0 = a.w0f[c * 2], wf1 = a.w0f[c * 2 + 1], wd0 = a.w0d[c * 2], wd1 = a.w0d[c * 2 + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xi = a.x[(size_t)i * 3 + k], dx = a.x[(size_t)j * 3 + k] - xi;
            p[k] = wf0 * dx + wf1 * xi;
            d[k] = wd0 * dx + wd1 * xi;
        }
    } else {
        const float *yj = a.y + ((size_t)j * 4 * a.C + c) * 3, *yi = a.y + ((size_t)i * 4 * a.C + a.C + c) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = yj[k] + yi[k];
            d[k] = yj[2 * a.C * 3 + k] + yi[2 * a.C * 3 + k];
        }
    }
}

// one workgroup of C threads per point; pass 1: batch sums of ||p|| + EPS per channel
template <bool L0>
__global__ void se_edge_stats_kernel(SeEdgeArgs a) {
    const int i = blockIdx.x, c = threadIdx.x;
    const int base = (i / a.N) * a.N;
    double s1 = 0.0, s2 = 0.0;
    for (int kk = 0; kk < kSeK; ++kk) {
        const int j = base + a.idx[(size_t)i * kSeK + kk];
        float p[3], d[3];
        se_edge_pd<L0>(a, i, j, c, p, d);
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
        s1 += (double)nrm; s2 += (double)nrm * (double)nrm;
    }
    double *acc = a.acc + (size_t)(blockIdx.x % kSeReplicas) * 2 * a.C;
    atomicAdd(acc + c, s1);
    atomicAdd(acc + a.C + c, s2);
}

// pass 2: VNBatchNorm (train mode) + VN-leaky-ReLU, mean over the k neighbours
template <bool L0>
__global__ void se_edge_apply_kernel(SeEdgeArgs a) {
    const int i = blockIdx.x, c = threadIdx.x;
    const int base = (i / a.N) * a.N;
    double t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < kSeReplicas; ++r) { t1 += a.acc[(size_t)r * 2 * a.C + c]; t2 += a.acc[(size_t)r * 2 * a.C + a.C + c]; }
    const double cnt = (double)a.n_total * kSeK;
    const double mean = t1 / cnt;
    double var = t2 / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float meanf = (float)mean, rstd = 1.0f / sqrtf((float)var + 1e-5f), bg = a.bn_g[c], bb = a.bn_b[c];
    float o[3] = {0.f, 0.f, 0.f};
    for (int kk = 0; kk < kSeK; ++kk) {
        const int j = base + a.idx[(size_t)i * kSeK + kk];
        float p[3], d[3];
        se_edge_pd<L0>(a, i, j, c, p, d);
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
        const float nbn = (nrm - meanf) * rstd * bg + bb;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = p[k] / nrm * nbn;
        const float dot = p[0] * d[0] + p[1] * d[1] + p[2] * d[2];
        const float dsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float coef = dot / (dsq + 1e-6f);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] += 0.2f * p[k] + 0.8f * (dot >= 0.f ? p[k] : p[k] - coef * d[k]);
    }
    float *ho = a.h_out + (size_t)i * a.h_ld + a.h_off + (size_t)c * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) ho[k] = o[k] / kSeK;
}

// ---- conv_c: Linear(4C -> LAT) + BatchNorm1d (train) + shared-direction VN-leaky-ReLU, mean over the points --------------
struct SeHeadArgs {
    const float *hcat;       // [P][KC][3]
    const float *wf;         // [LAT][KC]
    const float *wd;         // [KC]
    const float *bn_g, *bn_b;
    float *pd;               // [P][LAT + 1][3]: p rows then the shared d
    double *acc;             // [kSeReplicas][2][LAT]
    float *out;              // [B][LAT][3]
    int n_total, N, KC, LAT;
};
// one wave per (point, output row): row LAT is the shared direction
__global__ void se_head_linear_kernel(SeHeadArgs a) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int p = wv / (a.LAT + 1), m = wv % (a.LAT + 1);
    if (p >= a.n_total) return;
    const float *w = m < a.LAT ? a.wf + (size_t)m * a.KC : a.wd;
    float s[3] = {0.f, 0.f, 0.f};
    for (int c = lane; c < a.KC; c += 64) {
        const float wc = w[c];
        const float *hp = a.hcat + ((size_t)p * a.KC + c) * 3;
        s[0] += wc * hp[0]; s[1] += wc * hp[1]; s[2] += wc * hp[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int mm = 1; mm < 64; mm <<= 1) s[k] += __shfl_xor(s[k], mm, 64);
    if (lane == 0) {
        float *o = a.pd + ((size_t)p * (a.LAT + 1) + m) * 3;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        if (m < a.LAT) {
            const float nrm = sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) + 1e-6f;
            double *acc = a.acc + (size_t)(p % kSeReplicas) * 2 * a.LAT;
            atomicAdd(acc + m, (double)nrm);
            atomicAdd(acc + a.LAT + m, (double)nrm * (double)nrm);
        }
    }
}
// one workgroup per (shape, output row): normalise, leaky-ReLU, mean over the N points
__global__ void se_head_apply_kernel(SeHeadArgs a) {
    __shared__ float red[3][256];
    const int b = blockIdx.x / a.LAT, m = blockIdx.x % a.LAT;
    double t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < kSeReplicas; ++r) { t1 += a.acc[(size_t)r * 2 * a.LAT + m]; t2 += a.acc[(size_t)r * 2 * a.LAT + a.LAT + m]; }
    const double cnt = (double)a.n_total, mean = t1 / cnt;
    double var = t2 / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float meanf = (float)mean, rstd = 1.0f / sqrtf((float)var + 1e-5f), bg = a.bn_g[m], bb = a.bn_b[m];
    float o[3] = {0.f, 0.f, 0.f};
    for (int nn = threadIdx.x; nn < a.N; nn += blockDim.x) {
        const float *pp = a.pd + ((size_t)(b * a.N + nn) * (a.LAT + 1) + m) * 3;
        const float *dd = a.pd + ((size_t)(b * a.N + nn) * (a.LAT + 1) + a.LAT) * 3;
        float p[3] = {pp[0], pp[1], pp[2]};
        const float d[3] = {dd[0], dd[1], dd[2]};
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
        const float nbn = (nrm - meanf) * rstd * bg + bb;
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = p[k] / nrm * nbn;
        const float dot = p[0] * d[0] + p[1] * d[1] + p[2] * d[2];
        const float dsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float coef = dot / (dsq + 1e-6f);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] += 0.2f * p[k] + 0.8f * (dot >= 0.f ? p[k] : p[k] - coef * d[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = o[k];
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int t = 0; t < (int)blockDim.x; ++t) s += red[threadIdx.x][t];
        a.out[((size_t)b * a.LAT + m) * 3 + threadIdx.x] = s / a.N;
    }
}
