// Training building block (first milestone of the backward pass): the MLP block of the score
// network -- Linear -> LayerNorm(eps 1e-5, affine) -> ReLU -> Linear, /root/reference/models/common.py:47-67, the shared
// inner block of every edge and node function (58 of them per evaluation, ~95 % of its FLOPs) -- forward WITH the
// quantities its backward needs, and the backward itself: gradients of the input rows and of all six parameter tensors.
//
//   forward:   z = x W1^T + b1;  xhat = (z - mean) * rstd;  a = relu(xhat * gamma + beta);  y = a W2^T + b2
//   backward:  da = dy W2;  dpre = da * [a > 0];  dgamma = sum_rows dpre * xhat;  dbeta = sum_rows dpre;
//              dh = dpre * gamma;  dz = rstd * (dh - mean(dh) - xhat * mean(dh * xhat));
//              dx = dz W1;  dW1 = dz^T x;  db1 = sum_rows dz;  dW2 = dy^T a;  db2 = sum_rows dy
//
// Arithmetic: fp32 throughout, products on v_mfma_f32_16x16x4_f32 (exact fp32 multiply-add), so that gradients can be
// held to 1e-4 of the reference's (tests/golden/grad_b12.npz).  Reductions over the rows (the parameter gradients) are
// deterministic: every workgroup writes a partial, a second kernel adds the partials in a fixed order (no atomics).
// These kernels are a correct first version of the training path, not a tuned one: one generic strided GEMM serves all
// five products (rows x small, small x small with the reduction over the rows split over workgroups).
#pragma once
#include "sm_device.h"

// C[m][n] = sum_k A(m, k) B(k, n) (+ bias[n]);  A(m, k) = A[m * sam + k * sak],  B(k, n) = B[k * sbk + n * sbn];
// C row-major with leading dimension ldc.  grid = (ceil(N / 64), ceil(M / 64), splits): split z covers
// k in [z * kchunk, min(K, (z + 1) * kchunk)) and writes its tile to C + z * M * ldc (partials when splits > 1).
struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int M, N, K;
    long long sam, sak, sbk, sbn;
    int ldc, kchunk;
    int accum;                     // 1: C += (only with one split)
};

constexpr int kGemmKC = 16;

__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs a) {
    __shared__ float As[64][kGemmKC + 1];
    __shared__ float Bs[kGemmKC][64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                     // 2 x 2 waves, 32 x 32 outputs each
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_begin = blockIdx.z * a.kchunk, k_end = min(a.K, k_begin + a.kchunk);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool a_kfast = a.sak == 1, b_nfast = a.sbn == 1;
    // the 64 x 16 tile of A and the 16 x 64 tile of B (zero-padded), consecutive threads along the unit stride; the NEXT chunk's
    // elements are requested into registers before the current chunk's products (they fly under the MFMAs)
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int m = a_kfast ? e / kGemmKC : e % 64, k = a_kfast ? e % kGemmKC : e / 64;
            const int gm = m0 + m, gk = k0 + k;
            ra[i] = (gm < a.M && gk < k_end) ? a.A[(long long)gm * a.sam + (long long)gk * a.sak] : 0.f;
            const int n = b_nfast ? e % 64 : e / kGemmKC, kb = b_nfast ? e / 64 : e % kGemmKC;
            const int gn = n0 + n, gkb = k0 + kb;
            rb[i] = (gn < a.N && gkb < k_end) ? a.B[(long long)gkb * a.sbk + (long long)gn * a.sbn] : 0.f;
        }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += kGemmKC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            As[a_kfast ? e / kGemmKC : e % 64][a_kfast ? e % kGemmKC : e / 64] = ra[i];
            Bs[b_nfast ? e / 64 : e % kGemmKC][b_nfast ? e % 64 : e / kGemmKC] = rb[i];
        }
        __syncthreads();
        if (k0 + kGemmKC < k_end) fetch(k0 + kGemmKC);
#pragma unroll
        for (int kk = 0; kk < kGemmKC / 4; ++kk) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[wm * 32 + i * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[kk * 4 + (lane >> 4)][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }
    float *C = a.C + (size_t)blockIdx.z * a.M * a.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + (lane & 15);
            const float b = (a.bias && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + 4 * (lane >> 4) + r;      // C/D fragment: row 4 g + r, column n
                if (row < a.M && col < a.N) {
                    float *dst = C + (size_t)row * a.ldc + col;
                    *dst = acc[i][j][r] + b + (a.accum ? *dst : 0.f);
                }
            }
        }
}

// out[i] = sum_s part[s * n + i] in a fixed order (which makes the parameter gradients deterministic): a workgroup owns 32
// outputs, its 8 thread groups add every 8th partial (float64: partials of sums that cancel -- a bias in front of a
// LayerNorm), the first group adds the 8 group sums.  (One thread per output walking all partials was a chain of up to 256
// dependent loads: 27 % of a training step.)
// The n outputs are a [n / cols][cols] matrix stored with row stride ldo (a column block of a wider gradient matrix).
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float *part, int n_parts, long long n, float *out, int cols, int ldo) {
    __shared__ double grp[8][32];
    const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + col;
    double s = 0.0;
    if (i < n)
        for (int p = g; p < n_parts; p += 8) s += (double)part[(size_t)p * n + i];
    grp[g][col] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += grp[k][col];
        out[(i / cols) * ldo + i % cols] = (float)t;
    }
}

// LayerNorm + ReLU of the rows of z [rows][H] (in place: z becomes the activation a), keeping xhat and rstd for the backward.
// One wave per row, biased variance, two passes, as torch.nn.LayerNorm.  With pd / ps (the edge form, EdgeMLP below) row e first
// receives the per-node terms of its two atoms: z[e] += pd[dst[e]] + ps[src[e]].
__global__ void __launch_bounds__(256) ln_relu_fwd_kernel(float *z, const float *gamma, const float *beta, float *xhat, float *rstd, long long rows, int H,
                                                          const float *pd, const float *ps, const long long *dst, const long long *src) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float *zr = z + row * H, *xr = xhat + row * H;
    float s = 0.f;
    if (pd) {
        const float *pdr = pd + dst[row] * H, *psr = ps + src[row] * H;
        for (int c = lane; c < H; c += 64) { const float v = zr[c] + (pdr[c] + psr[c]); zr[c] = v; s += v; }
    } else
        for (int c = lane; c < H; c += 64) s += zr[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / H;
    float q = 0.f;
    for (int c = lane; c < H; c += 64) { const float d = zr[c] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rs = 1.0f / sqrtf(q / H + 1e-5f);
    if (lane == 0) rstd[row] = rs;
    for (int c = lane; c < H; c += 64) {
        const float xh = (zr[c] - mean) * rs;
        xr[c] = xh;
        zr[c] = fmaxf(xh * gamma[c] + beta[c], 0.f);
    }
}

// a = relu(xhat * gamma + beta) from the saved xhat (the backward recomputes the activation instead of keeping it)
__global__ void __launch_bounds__(256) relu_affine_kernel(const float *xhat, const float *gamma, const float *beta, float *act, long long rows, int H) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * H) return;
    const int c = (int)(i % H);
    act[i] = fmaxf(xhat[i] * gamma[c] + beta[c], 0.f);
}

// Backward of LayerNorm + ReLU: da [rows][H] -> dz in place; the workgroup's sums of dpre * xhat and dpre over its rows go to
// part[wg][2 H] (dgamma | dbeta partials).  kLnRows rows per workgroup, one wave per row at a time.
constexpr int kLnRows = 64;
__global__ void __launch_bounds__(256) ln_relu_bwd_kernel(float *da, const float *xhat, const float *rstd, const float *gamma, const float *beta,
                                                          long long rows, int H, float *part) {
    extern __shared__ float ln_sums[];                         // [4 waves][2 H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mine = ln_sums + wave * 2 * H;
    for (int c = lane; c < 2 * H; c += 64) mine[c] = 0.f;
    const long long r0 = (long long)blockIdx.x * kLnRows;
    for (int rr = wave; rr < kLnRows; rr += 4) {
        const long long row = r0 + rr;
        if (row >= rows) break;
        float *dr = da + row * H;
        const float *xr = xhat + row * H;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < H; c += 64) {
            const float xh = xr[c];
            const float dpre = (xh * gamma[c] + beta[c] > 0.f) ? dr[c] : 0.f;
            mine[c] += dpre * xh;                              // each lane owns its columns: no race inside the wave
            mine[H + c] += dpre;
            const float dh = dpre * gamma[c];
            dr[c] = dh;                                        // (dz is finished below, once the row means are known)
            s1 += dh; s2 += dh * xh;
        }
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float m1 = s1 / H, m2 = s2 / H, rs = rstd[row];
        for (int c = lane; c < H; c += 64) dr[c] = rs * (dr[c] - m1 - xr[c] * m2);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256)
        part[(size_t)blockIdx.x * 2 * H + c] = (ln_sums[c] + ln_sums[2 * H + c]) + (ln_sums[4 * H + c] + ln_sums[6 * H + c]);
}

// Column sums of x [rows][cols] over blocks of kLnRows rows: part[wg][cols] (bias gradients).
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float *x, long long rows, int cols, float *part) {
    const long long r0 = (long long)blockIdx.x * kLnRows;
    const long long r1 = r0 + kLnRows < rows ? r0 + kLnRows : rows;
    for (int c = threadIdx.x; c < cols; c += 256) {
        double s = 0.0;
        for (long long r = r0; r < r1; ++r) s += (double)x[r * cols + c];
        part[(size_t)blockIdx.x * cols + c] = (float)s;
    }
}

// out[i][:] = sum of the rows x[perm ? perm[t] : t], t in [ptr[i], ptr[i+1]) -- the gradient of a per-node term that was gathered
// to the edges (by centre atom: the edges are contiguous; by neighbour atom: through the permutation that groups them).
// One wave per node, lanes over the columns; fixed order, no atomics.
__global__ void __launch_bounds__(256) seg_rowsum_kernel(const float *x, const long long *ptr, const long long *perm, float *out, long long n_nodes, int H) {
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_nodes) return;
    const long long t0 = ptr[i], t1 = ptr[i + 1];
    for (int c = lane; c < H; c += 64) {
        float s = 0.f;
        for (long long t = t0; t < t1; ++t) s += x[(perm ? perm[t] : t) * H + c];
        out[i * H + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Segment attention of the training path (models/uni_transformer.py:71-81 for x2h, :141-151 for h2x): for centre atom i with
// incoming edges e in [ptr[i], ptr[i+1]) (edges grouped by centre, as the graph builder emits them) and head h
//     logit_e = <q_i[h], k_e[h]> / sqrt(dh);   alpha = softmax over the atom's edges;   out_i[h][:] = sum_e alpha_e vals_e[h][:]
// forward and backward in one pass each, one thread per (atom, head) (an atom has <= 32 edges; the backward recomputes the
// softmax instead of storing alpha):  dvals_e = alpha_e dout_i;  dalpha_e = <dout_i, vals_e>;
//     dlogit_e = alpha_e (dalpha_e - sum_e' alpha_e' dalpha_e');   dq_i = sum_e dlogit_e k_e / sqrt(dh);   dk_e = dlogit_e q_i / sqrt(dh).
// q [N][heads dh], k [E][heads dh], vals [E][heads][W] (W = dh for x2h, 3 for h2x: value times relative position), W <= 8.
struct SegAttnArgs {
    const float *q, *k, *vals;
    const long long *ptr;          // [N + 1]
    float *out;                    // forward: [N][heads][W]
    const float *dout;             // backward
    float *dq, *dk, *dvals;
    int n_atoms, heads, dh, W;
};
constexpr int kSegAttnMaxW = 8, kSegAttnMaxDh = 8;

template <bool BWD>
__global__ void __launch_bounds__(256) seg_attention_kernel(SegAttnArgs a) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.n_atoms * a.heads) return;
    const int i = gid / a.heads, h = gid % a.heads, H = a.heads * a.dh;
    const long long e0 = a.ptr[i], e1 = a.ptr[i + 1];
    const float scale = 1.0f / sqrtf((float)a.dh);
    float qv[kSegAttnMaxDh];
    for (int d = 0; d < a.dh; ++d) qv[d] = a.q[(size_t)i * H + h * a.dh + d];
    // pass 1: maximum of the logits;  pass 2: sum of exponentials (and the forward's weighted sum)
    float mx = -INFINITY;
    for (long long e = e0; e < e1; ++e) {
        float l = 0.f;
        for (int d = 0; d < a.dh; ++d) l += qv[d] * a.k[(size_t)e * H + h * a.dh + d];
        mx = fmaxf(mx, l * scale);
    }
    float den = 0.f, acc[kSegAttnMaxW];
    for (int w = 0; w < a.W; ++w) acc[w] = 0.f;
    for (long long e = e0; e < e1; ++e) {
        float l = 0.f;
        for (int d = 0; d < a.dh; ++d) l += qv[d] * a.k[(size_t)e * H + h * a.dh + d];
        const float ex = expf(l * scale - mx);
        den += ex;
        if (!BWD)
            for (int w = 0; w < a.W; ++w) acc[w] += ex * a.vals[((size_t)e * a.heads + h) * a.W + w];
    }
    if (!BWD) {
        for (int w = 0; w < a.W; ++w) a.out[((size_t)i * a.heads + h) * a.W + w] = e1 > e0 ? acc[w] / den : 0.f;
        return;
    }
    float dov[kSegAttnMaxW];
    for (int w = 0; w < a.W; ++w) dov[w] = a.dout[((size_t)i * a.heads + h) * a.W + w];
    // pass 3: s = sum_e alpha_e dalpha_e;  pass 4: the gradients
    float s = 0.f;
    for (long long e = e0; e < e1; ++e) {
        float l = 0.f, da = 0.f;
        for (int d = 0; d < a.dh; ++d) l += qv[d] * a.k[(size_t)e * H + h * a.dh + d];
        for (int w = 0; w < a.W; ++w) da += dov[w] * a.vals[((size_t)e * a.heads + h) * a.W + w];
        s += expf(l * scale - mx) / den * da;
    }
    float dqv[kSegAttnMaxDh];
    for (int d = 0; d < a.dh; ++d) dqv[d] = 0.f;
    for (long long e = e0; e < e1; ++e) {
        float l = 0.f, da = 0.f;
        for (int d = 0; d < a.dh; ++d) l += qv[d] * a.k[(size_t)e * H + h * a.dh + d];
        for (int w = 0; w < a.W; ++w) da += dov[w] * a.vals[((size_t)e * a.heads + h) * a.W + w];
        const float al = expf(l * scale - mx) / den, dl = al * (da - s) * scale;
        for (int w = 0; w < a.W; ++w) a.dvals[((size_t)e * a.heads + h) * a.W + w] = al * dov[w];
        for (int d = 0; d < a.dh; ++d) {
            dqv[d] += dl * a.k[(size_t)e * H + h * a.dh + d];
            a.dk[(size_t)e * H + h * a.dh + d] = dl * qv[d];
        }
    }
    for (int d = 0; d < a.dh; ++d) a.dq[(size_t)i * H + h * a.dh + d] = dqv[d];
}
