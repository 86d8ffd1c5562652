// Training building blocks (the backward pass).  First: the MLP block of the score
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
// One strided GEMM (two kernels: float4 operand moves where strides and alignment allow, scalar otherwise) serves all five
// products (rows x small, small x small with the reduction over the rows split over workgroups).  Further down: the edge
// form's gather / segment sums, the segment attention and the vector-neuron coordinate update, each forward and backward.
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

// The same product for operands that have a unit stride and 16-byte aligned rows (every product of the training path but the
// ones with a one-column operand): each thread moves one float4 of A and one of B per K chunk instead of four scalars of each
// with 64-bit index arithmetic per element.  AK: A is k-contiguous (sak == 1), else m-contiguous (sam == 1); BN: B is
// n-contiguous (sbn == 1), else k-contiguous (sbk == 1).  LDS rows are padded to 20 / 80 floats: float4 stores stay aligned
// and the MFMA operand reads (16 rows x 4 k, 4 k x 16 columns) touch 64 distinct banks.
__device__ __forceinline__ f32x4 gemm_load4(const float *p, int nvalid) {
    if (nvalid >= 4) return *reinterpret_cast<const f32x4 *>(p);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (nvalid > 0) v[0] = p[0];
    if (nvalid > 1) v[1] = p[1];
    if (nvalid > 2) v[2] = p[2];
    return v;
}

template <bool AK, bool BN>
__global__ void __launch_bounds__(256) gemm_f32_vec_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[64][20];
    __shared__ __attribute__((aligned(16))) float Bs[kGemmKC][80];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_begin = blockIdx.z * a.kchunk, k_end = min(a.K, k_begin + a.kchunk);
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this thread's float4 of the A tile and of the B tile
    const int a_m = AK ? tid >> 2 : (tid & 15) * 4, a_k = AK ? (tid & 3) * 4 : tid >> 4;
    const int b_n = BN ? (tid & 15) * 4 : tid >> 2, b_k = BN ? tid >> 4 : (tid & 3) * 4;
    const float *pa = a.A + (long long)(m0 + a_m) * a.sam + (long long)(k_begin + a_k) * a.sak;
    const float *pb = a.B + (long long)(k_begin + b_k) * a.sbk + (long long)(n0 + b_n) * a.sbn;
    const long long step_a = (long long)kGemmKC * a.sak, step_b = (long long)kGemmKC * a.sbk;
    const int a_mvalid = a.M - (m0 + a_m), b_nvalid = a.N - (n0 + b_n);
    f32x4 ra, rb;
    auto fetch = [&](int k0) {
        ra = AK ? gemm_load4(pa, a_mvalid > 0 ? k_end - (k0 + a_k) : 0) : gemm_load4(pa, k0 + a_k < k_end ? a_mvalid : 0);
        rb = BN ? gemm_load4(pb, k0 + b_k < k_end ? b_nvalid : 0) : gemm_load4(pb, b_nvalid > 0 ? k_end - (k0 + b_k) : 0);
        pa += step_a; pb += step_b;
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += kGemmKC) {
        if (AK) *reinterpret_cast<f32x4 *>(&As[a_m][a_k]) = ra;
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) As[a_m + i][a_k] = ra[i];
        }
        if (BN) *reinterpret_cast<f32x4 *>(&Bs[b_k][b_n]) = rb;
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) Bs[b_k + i][b_n] = rb[i];
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
                const int row = m0 + wm * 32 + i * 16 + 4 * (lane >> 4) + r;
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
// The n outputs are a [n / cols][cols] matrix stored with row stride ldo (a column block of a wider gradient matrix), or -- seg > 0
// -- up to three separate arrays of seg outputs each (dgamma, dbeta, db1 of one MLP: one launch, three caller-owned tensors).
struct ReduceOut { float *p[3]; long long seg; int cols, ldo; };
// GROUPS thread groups of 256 / GROUPS outputs each: 8 x 32 for a few partials, 32 x 8 for many (the per-thread chain of dependent
// loads is n_parts / GROUPS long, and with 173 partials of a small matrix that chain, not the bytes, is the kernel's time)
template <int GROUPS>
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float *part, int n_parts, long long n, ReduceOut o) {
    constexpr int COLS = 256 / GROUPS;
    __shared__ double grp[GROUPS][COLS];
    const int col = threadIdx.x % COLS, g = threadIdx.x / COLS;
    const long long i = (long long)blockIdx.x * COLS + col;
    double s = 0.0;
    if (i < n)
        for (int p = g; p < n_parts; p += GROUPS) s += (double)part[(size_t)p * n + i];
    grp[g][col] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < GROUPS; ++k) t += grp[k][col];
        if (o.seg > 0) {
            const int which = (int)(i / o.seg);
            (which == 0 ? o.p[0] : which == 1 ? o.p[1] : o.p[2])[i - which * o.seg] = (float)t;
        } else o.p[0][(i / o.cols) * o.ldo + i % o.cols] = (float)t;
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

// Backward of LayerNorm + ReLU: da [rows][H] -> dz in place; the workgroup's sums over its rows of dpre * xhat, dpre and dz go to
// part[wg][3 H] (dgamma | dbeta | db1 partials: the first Linear's bias gradient is the column sum of dz).  kLnRows rows per
// workgroup, one wave per row at a time.
constexpr int kLnRows = 64, kLnBwdWaves = 16;       // 16 waves: four rows each (with four waves a row's loads waited for the previous row's: 1.6 TB/s)
__global__ void __launch_bounds__(kLnBwdWaves * 64) ln_relu_bwd_kernel(float *da, const float *xhat, const float *rstd, const float *gamma, const float *beta,
                                                                        long long rows, int H, float *part) {
    extern __shared__ float ln_sums[];                         // [waves][3 H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mine = ln_sums + wave * 3 * H;
    for (int c = lane; c < 3 * H; c += 64) mine[c] = 0.f;
    const long long r0 = (long long)blockIdx.x * kLnRows;
    for (int rr = wave; rr < kLnRows; rr += kLnBwdWaves) {
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
        for (int c = lane; c < H; c += 64) {
            const float dz = rs * (dr[c] - m1 - xr[c] * m2);
            dr[c] = dz;
            mine[2 * H + c] += dz;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * H; c += kLnBwdWaves * 64) {
        float t = 0.f;
        for (int w = 0; w < kLnBwdWaves; ++w) t += ln_sums[w * 3 * H + c];
        part[(size_t)blockIdx.x * 3 * H + c] = t;
    }
}

// Column sums of x [rows][cols] over blocks of kLnRows rows: part[wg][cols] (bias gradients).  The 256 threads are
// 256 / cols row groups x cols columns (float64 sums, combined through LDS in a fixed order).
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float *x, long long rows, int cols, float *part) {
    __shared__ double grp[256];
    const long long r0 = (long long)blockIdx.x * kLnRows;
    const long long r1 = r0 + kLnRows < rows ? r0 + kLnRows : rows;
    for (int c0 = 0; c0 < cols; c0 += 256) {
        const int here = min(256, cols - c0), groups = 256 / here;
        const int c = threadIdx.x % here, g = threadIdx.x / here;
        double s = 0.0;
        if (g < groups)
            for (long long r = r0 + g; r < r1; r += groups) s += (double)x[r * cols + c0 + c];
        grp[threadIdx.x] = s;
        __syncthreads();
        if (g == 0) {
            double t = 0.0;
            for (int k = 0; k < groups; ++k) t += grp[k * here + c];
            part[(size_t)blockIdx.x * cols + c0 + c] = (float)t;
        }
        __syncthreads();
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
// and its backward (the softmax is recomputed, not stored):  dvals_e = alpha_e dout_i;  dalpha_e = <dout_i, vals_e>;
//     dlogit_e = alpha_e (dalpha_e - sum_e' alpha_e' dalpha_e');   dq_i = sum_e dlogit_e k_e / sqrt(dh);   dk_e = dlogit_e q_i / sqrt(dh).
// q [N][heads dh], k [E][heads dh], vals [E][heads][W] (W = dh for x2h, 3 for h2x: value times relative position), dh, W <= 8.
// Eight lanes per (atom, head): lane `sub` owns component sub of the head's q / k slice and of its value row, dot products are
// 8-lane butterfly sums -- consecutive lanes read consecutive floats, and there are 8 x as many threads as (atom, head) pairs
// (one thread per pair walked its edges with four dependent passes at 1.4 waves per SIMD: 126 us per backward at 5.5 k atoms).
// Forward: one online-softmax pass.  Backward: one online pass for max, denominator and s = sum alpha dalpha, one for the gradients.
struct SegAttnArgs {
    const float *q, *k, *vals;
    const long long *ptr;          // [N + 1]
    float *out;                    // forward: [N][heads][W]
    const float *dout;             // backward
    float *dq, *dk, *dvals;
    int n_atoms, heads, dh, W;
};
constexpr int kSegAttnMaxW = 8, kSegAttnMaxDh = 8;

__device__ __forceinline__ float sum8(float v) {
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

template <bool BWD>
__global__ void __launch_bounds__(256) seg_attention_kernel(SegAttnArgs a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long grp = gid >> 3;
    const int sub = (int)(gid & 7);
    // (a whole 8-lane group is in or out together; out-of-range groups keep shuffling with zeros so that no lane of a wave is missing)
    const bool live = grp < (long long)a.n_atoms * a.heads;
    const long long i = live ? grp / a.heads : 0;
    const int h = live ? (int)(grp % a.heads) : 0, H = a.heads * a.dh;
    const long long e0 = live ? a.ptr[i] : 0, e1 = live ? a.ptr[i + 1] : 0;
    const bool kd = sub < a.dh, vw = sub < a.W;
    const float scale = 1.0f / sqrtf((float)a.dh);
    const float qv = (live && kd) ? a.q[(size_t)i * H + h * a.dh + sub] : 0.f;
    const float dov = (BWD && live && vw) ? a.dout[((size_t)i * a.heads + h) * a.W + sub] : 0.f;
    const float *kp = a.k + h * a.dh + sub, *vp = a.vals + (size_t)h * a.W + sub;
    float mx = -INFINITY, den = 0.f, acc = 0.f;             // acc: the forward's weighted sum (component sub) / the backward's sum of ex * dalpha
    for (long long e = e0; e < e1; ++e) {
        const float kx = kd ? kp[(size_t)e * H] : 0.f, vx = vw ? vp[(size_t)e * a.heads * a.W] : 0.f;
        const float l = sum8(qv * kx) * scale;
        const float term = BWD ? sum8(dov * vx) : vx;
        const float mn = fmaxf(mx, l), c = expf(mx - mn), ex = expf(l - mn);     // first edge: exp(-inf) = 0
        den = den * c + ex;
        acc = acc * c + ex * term;
        mx = mn;
    }
    if (!BWD) {
        if (live && vw) a.out[((size_t)i * a.heads + h) * a.W + sub] = e1 > e0 ? acc / den : 0.f;
        return;
    }
    const float s = e1 > e0 ? acc / den : 0.f;
    float dqv = 0.f;
    for (long long e = e0; e < e1; ++e) {
        const float kx = kd ? kp[(size_t)e * H] : 0.f, vx = vw ? vp[(size_t)e * a.heads * a.W] : 0.f;
        const float l = sum8(qv * kx) * scale, da = sum8(dov * vx);
        const float al = expf(l - mx) / den, dl = al * (da - s) * scale;
        if (vw) a.dvals[((size_t)e * a.heads + h) * a.W + sub] = al * dov;
        if (kd) { dqv += dl * kx; a.dk[(size_t)e * H + h * a.dh + sub] = dl * qv; }
    }
    if (live && kd) a.dq[(size_t)i * H + h * a.dh + sub] = dqv;
}

// ---------------------------------------------------------------------------------------------------------------------
// The coordinate update's vector-neuron block on the training path: VNLinearLeakyReLU with VNBatchNorm
// (models/shape_vn_layers.py:41-61, 95-110) on z_n = [x_n | o3_n (rows_o rows) | shape_{mol(n)} (rows_s rows)], rows in R^3,
// followed by the mean over the C output channels (models/uni_transformer.py:157-160):
//     pf = Wf z;  nrm = |pf| + 1e-6;  nbn = BatchNorm_over_atoms(nrm) (train: batch mean / biased variance, eps 1e-5);
//     p = pf nbn / nrm;  d = Wd z;  q = <p, d> / (|d|^2 + 1e-6);  y = p - 0.8 [<p, d> < 0] q d;  out_n = mean_c y
// (0.2 p + 0.8 (p or p - q d) written as one expression).  Forward: vn_lin_kernel -> vn_bn_stats_kernel -> vn_act_kernel;
// backward: vn_bwd_a_kernel (everything per (atom, channel) up to the batch-norm) -> vn_bwd_stats_kernel (the two batch sums of
// the batch-norm's backward, its weight / bias gradients) -> vn_bwd_b_kernel (dpf, then dz = Wf^T dpf + Wd^T dd for the x and
// o3 rows, and the workgroup's partial of dWf | dWd) -> reduce_partials_kernel.  One thread per (atom, channel), 256 / C atoms
// per workgroup; every sum in a fixed order.
struct VnTrainArgs {
    const float *x, *o3, *shape;       // [N][3], [N][rows_o][3], [B][rows_s][3]
    const long long *batch;            // [N]
    const float *wf, *wd, *bn_w, *bn_b;   // [C][Cin] x 2, [C] x 2
    float *run_mean, *run_var;         // [C] or null
    float *pf, *dir, *nrm;             // [N][C][3] x 2, [N][C]
    float *stats;                      // [2][C]: the mean and variance the normalisation used
    float *out;                        // [N][3]
    // backward
    const float *gout;                 // [N][3]
    float *dd, *dpfd, *dnd, *dnbn, *xhat;   // [N][C][3] x 2, [N][C] x 3
    float *sums;                       // [2][C]: sum dnbn, sum dnbn xhat
    float *dbn_w, *dbn_b;              // [C]
    float *dx, *do3;                   // [N][3], [N][rows_o][3]
    float *wpart;                      // [workgroups][2][C][Cin]
    long long n_atoms;
    int rows_o, rows_s, C, Cin, training;
};
constexpr float kVnEps = 1e-6f, kVnLeak = 0.2f, kBnEps = 1e-5f;

__device__ __forceinline__ void vn_zrow(const VnTrainArgs &a, long long n, long long b, int i, float z[3]) {
    const float *p = i == 0 ? a.x + n * 3 : i <= a.rows_o ? a.o3 + (n * a.rows_o + (i - 1)) * 3 : a.shape + (b * a.rows_s + (i - 1 - a.rows_o)) * 3;
    z[0] = p[0]; z[1] = p[1]; z[2] = p[2];
}

__global__ void __launch_bounds__(256) vn_lin_kernel(VnTrainArgs a) {
    extern __shared__ float vn_w[];                       // wf | wd
    for (int i = threadIdx.x; i < 2 * a.C * a.Cin; i += 256) vn_w[i] = i < a.C * a.Cin ? a.wf[i] : a.wd[i - a.C * a.Cin];
    __syncthreads();
    const int per = 256 / a.C, la = threadIdx.x / a.C, c = threadIdx.x % a.C;
    const long long n = (long long)blockIdx.x * per + la;
    if (la >= per || n >= a.n_atoms) return;
    const long long b = a.batch[n];
    const float *wf = vn_w + c * a.Cin, *wd = vn_w + (a.C + c) * a.Cin;
    float p[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < a.Cin; ++i) {
        float z[3];
        vn_zrow(a, n, b, i, z);
#pragma unroll
        for (int k = 0; k < 3; ++k) { p[k] += wf[i] * z[k]; d[k] += wd[i] * z[k]; }
    }
    const size_t o = ((size_t)n * a.C + c) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.pf[o + k] = p[k]; a.dir[o + k] = d[k]; }
    a.nrm[(size_t)n * a.C + c] = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + kVnEps;
}

__device__ __forceinline__ double vn_block_sum(double v, double *red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// grid = C: batch mean and biased variance of channel c (two passes, float64), running estimates updated as nn.BatchNorm1d
// does (momentum 0.1, unbiased variance); evaluation mode: the running estimates are what the normalisation uses
__global__ void __launch_bounds__(256) vn_bn_stats_kernel(VnTrainArgs a) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    if (!a.training) {
        if (threadIdx.x == 0) { a.stats[c] = a.run_mean[c]; a.stats[a.C + c] = a.run_var[c]; }
        return;
    }
    double s = 0.0;
    for (long long n = threadIdx.x; n < a.n_atoms; n += 256) s += (double)a.nrm[n * a.C + c];
    const double mean = vn_block_sum(s, red) / (double)a.n_atoms;
    double q = 0.0;
    for (long long n = threadIdx.x; n < a.n_atoms; n += 256) { const double d = (double)a.nrm[n * a.C + c] - mean; q += d * d; }
    const double var = vn_block_sum(q, red) / (double)a.n_atoms;
    if (threadIdx.x == 0) {
        a.stats[c] = (float)mean;
        a.stats[a.C + c] = (float)var;
        if (a.run_mean) {
            const double unb = var * (double)a.n_atoms / (double)(a.n_atoms > 1 ? a.n_atoms - 1 : 1);
            a.run_mean[c] = 0.9f * a.run_mean[c] + 0.1f * (float)mean;
            a.run_var[c] = 0.9f * a.run_var[c] + 0.1f * (float)unb;
        }
    }
}

// the per-(atom, channel) quantities both passes need
struct VnPoint { float pf[3], d[3], p[3], nrm, xh, nbn, s, dot, dsq, q; bool neg; };
__device__ __forceinline__ VnPoint vn_point(const VnTrainArgs &a, long long n, int c) {
    VnPoint v;
    const size_t o = ((size_t)n * a.C + c) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) { v.pf[k] = a.pf[o + k]; v.d[k] = a.dir[o + k]; }
    v.nrm = sqrtf(v.pf[0] * v.pf[0] + v.pf[1] * v.pf[1] + v.pf[2] * v.pf[2]) + kVnEps;
    const float rs = 1.0f / sqrtf(a.stats[a.C + c] + kBnEps);
    v.xh = (v.nrm - a.stats[c]) * rs;
    v.nbn = v.xh * a.bn_w[c] + a.bn_b[c];
    v.s = v.nbn / v.nrm;
    v.dot = 0.f; v.dsq = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { v.p[k] = v.pf[k] * v.s; v.dot += v.p[k] * v.d[k]; v.dsq += v.d[k] * v.d[k]; }
    v.neg = !(v.dot >= 0.f);
    v.q = v.dot / (v.dsq + kVnEps);
    return v;
}

__global__ void __launch_bounds__(256) vn_act_kernel(VnTrainArgs a) {
    __shared__ float red[256][3];
    const int per = 256 / a.C, la = threadIdx.x / a.C, c = threadIdx.x % a.C;
    const long long n = (long long)blockIdx.x * per + la;
    const bool ok = la < per && n < a.n_atoms;
    float y[3] = {0.f, 0.f, 0.f};
    if (ok) {
        const VnPoint v = vn_point(a, n, c);
#pragma unroll
        for (int k = 0; k < 3; ++k) y[k] = kVnLeak * v.p[k] + (1.f - kVnLeak) * (v.neg ? v.p[k] - v.q * v.d[k] : v.p[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) red[threadIdx.x][k] = y[k];
    __syncthreads();
    if (ok && c == 0) {
        float s[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < a.C; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] += red[la * a.C + j][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) a.out[n * 3 + k] = s[k] / (float)a.C;
    }
}

// dy = gout / C per channel.  y = p - 0.8 m q d (m = [dot < 0]):
//   dp = dy - 0.8 m <dy, d> d / (dsq + eps);   dd = -0.8 m (q dy + <dy, d> (p - 2 q d) / (dsq + eps));
//   p = pf s, s = nbn / nrm:  dpf (direct) = s dp;  ds = <dp, pf>;  dnbn = ds / nrm;  dnrm (direct) = -ds nbn / nrm^2
__global__ void __launch_bounds__(256) vn_bwd_a_kernel(VnTrainArgs a) {
    const int per = 256 / a.C, la = threadIdx.x / a.C, c = threadIdx.x % a.C;
    const long long n = (long long)blockIdx.x * per + la;
    if (la >= per || n >= a.n_atoms) return;
    const VnPoint v = vn_point(a, n, c);
    float dy[3], dyd = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { dy[k] = a.gout[n * 3 + k] / (float)a.C; dyd += dy[k] * v.d[k]; }
    const float m = v.neg ? (1.f - kVnLeak) : 0.f, inv = 1.0f / (v.dsq + kVnEps);
    float ds = 0.f;
    const size_t o = ((size_t)n * a.C + c) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float dp = dy[k] - m * dyd * inv * v.d[k];
        a.dd[o + k] = -m * (v.q * dy[k] + dyd * inv * (v.p[k] - 2.f * v.q * v.d[k]));
        a.dpfd[o + k] = v.s * dp;
        ds += dp * v.pf[k];
    }
    const size_t o1 = (size_t)n * a.C + c;
    a.dnbn[o1] = ds / v.nrm;
    a.dnd[o1] = -ds * v.nbn / (v.nrm * v.nrm);
    a.xhat[o1] = v.xh;
}

// grid = C: S1 = sum_n dnbn, S2 = sum_n dnbn xhat (float64);  dbn_b = S1, dbn_w = S2
__global__ void __launch_bounds__(256) vn_bwd_stats_kernel(VnTrainArgs a) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (long long n = threadIdx.x; n < a.n_atoms; n += 256) {
        const double g = (double)a.dnbn[n * a.C + c];
        s1 += g; s2 += g * (double)a.xhat[n * a.C + c];
    }
    s1 = vn_block_sum(s1, red);
    s2 = vn_block_sum(s2, red);
    if (threadIdx.x == 0) { a.sums[c] = (float)s1; a.sums[a.C + c] = (float)s2; a.dbn_b[c] = (float)s1; a.dbn_w[c] = (float)s2; }
}

// batch-norm backward (train: dnrm = rs g (dnbn - S1 / N - xhat S2 / N); eval: rs g dnbn), nrm = |pf| + eps -> dpf += dnrm pf / |pf|;
// then the rows of dz this block's atoms own and the block's partial of dWf | dWd
__global__ void __launch_bounds__(256) vn_bwd_b_kernel(VnTrainArgs a) {
    extern __shared__ float vn_l[];                       // wf | wd [2][C][Cin], then dpf | dd [2][per][C][3]
    const int per = 256 / a.C, la = threadIdx.x / a.C, c = threadIdx.x % a.C, WN = a.C * a.Cin;
    float *w = vn_l, *g = vn_l + 2 * WN;
    for (int i = threadIdx.x; i < 2 * WN; i += 256) w[i] = i < WN ? a.wf[i] : a.wd[i - WN];
    const long long n0 = (long long)blockIdx.x * per, n = n0 + la;
    const int n_here = (int)min((long long)per, a.n_atoms - n0);
    if (la < per) {
        float dpf[3] = {0.f, 0.f, 0.f}, dd[3] = {0.f, 0.f, 0.f};
        if (n < a.n_atoms) {
            const size_t o = ((size_t)n * a.C + c) * 3, o1 = (size_t)n * a.C + c;
            const float rs = 1.0f / sqrtf(a.stats[a.C + c] + kBnEps);
            const float cnt = (float)a.n_atoms;
            float dn = a.dnbn[o1];
            if (a.training) dn -= a.sums[c] / cnt + a.xhat[o1] * a.sums[a.C + c] / cnt;
            const float dnrm = a.dnd[o1] + rs * a.bn_w[c] * dn;
            const float p0 = a.pf[o], p1 = a.pf[o + 1], p2 = a.pf[o + 2];
            const float len = fmaxf(sqrtf(p0 * p0 + p1 * p1 + p2 * p2), 1e-30f);
            dpf[0] = a.dpfd[o] + dnrm * p0 / len; dpf[1] = a.dpfd[o + 1] + dnrm * p1 / len; dpf[2] = a.dpfd[o + 2] + dnrm * p2 / len;
            dd[0] = a.dd[o]; dd[1] = a.dd[o + 1]; dd[2] = a.dd[o + 2];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { g[(la * a.C + c) * 3 + k] = dpf[k]; g[((per + la) * a.C + c) * 3 + k] = dd[k]; }
    }
    __syncthreads();
    // dz rows 0 .. rows_o of the block's atoms: dz[n][i][k] = sum_c wf[c][i] dpf[n][c][k] + wd[c][i] dd[n][c][k]
    const int rows = 1 + a.rows_o;
    for (int t = threadIdx.x; t < n_here * rows * 3; t += 256) {
        const int at = t / (rows * 3), i = (t / 3) % rows, k = t % 3;
        float s = 0.f;
        for (int cc = 0; cc < a.C; ++cc) s += w[cc * a.Cin + i] * g[(at * a.C + cc) * 3 + k] + w[WN + cc * a.Cin + i] * g[((per + at) * a.C + cc) * 3 + k];
        if (i == 0) a.dx[(n0 + at) * 3 + k] = s;
        else a.do3[((n0 + at) * a.rows_o + (i - 1)) * 3 + k] = s;
    }
    // the block's partial of dWf | dWd: [mat][c][i] = sum over its atoms and the three components of (dpf | dd)[n][c] z[n][i]
    for (int t = threadIdx.x; t < 2 * WN; t += 256) {
        const int mat = t / WN, cc = (t % WN) / a.Cin, i = t % a.Cin;
        float s = 0.f;
        for (int at = 0; at < n_here; ++at) {
            float z[3];
            vn_zrow(a, n0 + at, a.batch[n0 + at], i, z);
            const float *gg = g + ((mat * per + at) * a.C + cc) * 3;
            s += gg[0] * z[0] + gg[1] * z[1] + gg[2] * z[2];
        }
        a.wpart[(size_t)blockIdx.x * 2 * WN + t] = s;
    }
}
