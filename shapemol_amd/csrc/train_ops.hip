// C ABI of the training building block (include/shapemol_hip.h, shapemol_mlp_*): forward and backward of the MLP block
// Linear -> LayerNorm -> ReLU -> Linear (/root/reference/models/common.py:47-67).  Kernels: sm_train.h.
#include "../../include/shapemol_hip.h"
#include "sm_train.h"

#include <algorithm>
#include <cstdint>
#include <string>

extern "C" void shapemol_set_error_(const char *msg);     // shapemol_hip.hip: stores the thread's last error

namespace {
constexpr int kMaxHidden = 256;      // ln_relu_bwd_kernel keeps 16 waves x 3 hidden floats in LDS (48 KB at 256)
int tr_fail(const std::string &m) { shapemol_set_error_(m.c_str()); return 1; }
#define TRCHK(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return tr_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// C[M][N] = A B (+ bias); see GemmArgs.  splits > 1: partial tiles C + z * M * ldc
int gemm(hipStream_t s, const float *A, long long sam, long long sak, const float *B, long long sbk, long long sbn, const float *bias,
         float *C, int ldc, int M, int N, int K, int splits, bool accum = false) {
    if (M < 1 || N < 1 || K < 1) return 0;
    GemmArgs g{A, B, bias, C, M, N, K, sam, sak, sbk, sbn, ldc, 0, accum ? 1 : 0};
    splits = accum ? 1 : std::max(1, splits);
    g.kchunk = ((K + splits - 1) / splits + kGemmKC - 1) / kGemmKC * kGemmKC;
    const int nz = (K + g.kchunk - 1) / g.kchunk;
    const dim3 grid((N + 63) / 64, (M + 63) / 64, nz);
    // the float4 kernel needs a unit stride in each operand and 16-byte aligned rows
    auto al = [](const float *p, long long other) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && other % 4 == 0; };
    const int am = (sak == 1 && al(A, sam)) ? 1 : (sam == 1 && al(A, sak)) ? 0 : -1;
    const int bm = (sbn == 1 && al(B, sbk)) ? 1 : (sbk == 1 && al(B, sbn)) ? 0 : -1;
    if (am < 0 || bm < 0) hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, g);
    else if (am == 1 && bm == 1) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, true>), grid, dim3(256), 0, s, g);
    else if (am == 1) hipLaunchKernelGGL((gemm_f32_vec_kernel<true, false>), grid, dim3(256), 0, s, g);
    else if (bm == 1) hipLaunchKernelGGL((gemm_f32_vec_kernel<false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_vec_kernel<false, false>), grid, dim3(256), 0, s, g);
    TRCHK(hipGetLastError());
    return 0;
}
int reduce_parts(hipStream_t s, const float *part, int n_parts, long long n, float *out, int cols = 0, int ldo = 0) {
    if (cols < 1) { cols = (int)std::min<long long>(n, 1 << 30); ldo = cols; }      // a plain vector
    if (n_parts > 32) hipLaunchKernelGGL(reduce_partials_kernel<32>, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, s, part, n_parts, n, ReduceOut{{out, nullptr, nullptr}, 0, cols, ldo});
    else hipLaunchKernelGGL(reduce_partials_kernel<8>, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, part, n_parts, n, ReduceOut{{out, nullptr, nullptr}, 0, cols, ldo});
    TRCHK(hipGetLastError());
    return 0;
}
// the same into up to three separate arrays of seg outputs each
int reduce_parts3(hipStream_t s, const float *part, int n_parts, long long seg, float *o0, float *o1, float *o2) {
    const long long n = seg * (o2 ? 3 : o1 ? 2 : 1);
    if (n_parts > 32) hipLaunchKernelGGL(reduce_partials_kernel<32>, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, s, part, n_parts, n, ReduceOut{{o0, o1, o2}, seg, 0, 0});
    else hipLaunchKernelGGL(reduce_partials_kernel<8>, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, part, n_parts, n, ReduceOut{{o0, o1, o2}, seg, 0, 0});
    TRCHK(hipGetLastError());
    return 0;
}
// row-reduction splits of the two weight-gradient products: many short splits (the strided GEMM is latency-bound per workgroup:
// 32 splits made a training step 10 ms slower than 256), at least 256 rows each
int row_splits(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(256, rows / 256)); }
// the number of splits gemm() will really launch for K = rows
int real_splits(int64_t rows) {
    const int sp = row_splits(rows);
    const int64_t kchunk = ((rows + sp - 1) / sp + kGemmKC - 1) / kGemmKC * kGemmKC;
    return (int)((rows + kchunk - 1) / kchunk);
}
// LayerNorm + ReLU backward on da [rows][H] (-> dz in place) and the three parameter gradients that are row sums of it:
// dgamma | dbeta | db1, reduced into the caller's three arrays by one launch
int ln_backward(hipStream_t s, float *dz, const float *xhat, const float *rstd, const float *gamma, const float *beta, int64_t rows, int H, int nwg,
                float *pln, float *dgamma, float *dbeta, float *db1) {
    hipLaunchKernelGGL(ln_relu_bwd_kernel, dim3(nwg), dim3(kLnBwdWaves * 64), (size_t)kLnBwdWaves * 3 * H * sizeof(float), s, dz, xhat, rstd, gamma, beta, (long long)rows, H, pln);
    TRCHK(hipGetLastError());
    return reduce_parts3(s, pln, nwg, H, dgamma, dbeta, db1);
}
bool bad_dims(int64_t rows, int k_in, int hidden, int n_out) {
    // rows: gemm() puts ceil(rows / 64) on grid.y, which HIP limits to 65535 (4.19 M rows: 8x the edges of a B = 4096 batch)
    return rows < 1 || rows > 65535ll * 64 || k_in < 1 || k_in > 4096 || hidden < 1 || hidden > kMaxHidden || n_out < 1 || n_out > 4096;
}
}  // namespace

extern "C" {

size_t shapemol_mlp_backward_workspace(int64_t rows, int32_t k_in, int32_t hidden, int32_t n_out) {
    if (bad_dims(rows, k_in, hidden, n_out)) return 0;
    const size_t nwg = (size_t)((rows + kLnRows - 1) / kLnRows), sp = (size_t)real_splits(rows);
    return 2 * (size_t)rows * hidden                              // activation a (recomputed) | da -> dz
           + sp * ((size_t)hidden * k_in + (size_t)n_out * hidden)   // split partials of dW1, dW2
           + nwg * (3 * (size_t)hidden + n_out)                   // partials of dgamma | dbeta | db1, db2
           + 64;
}

int shapemol_mlp_forward(const float *d_x, int64_t rows, int32_t k_in, int32_t hidden, int32_t n_out, const float *d_w1,
                         const float *d_b1, const float *d_gamma, const float *d_beta, const float *d_w2, const float *d_b2,
                         float *d_y, float *d_xhat, float *d_rstd, float *d_act, void *stream) {
    if (!d_x || !d_w1 || !d_b1 || !d_gamma || !d_beta || !d_w2 || !d_b2 || !d_y || !d_xhat || !d_rstd || !d_act)
        return tr_fail("shapemol_mlp_forward: null argument");
    if (bad_dims(rows, k_in, hidden, n_out)) return tr_fail("shapemol_mlp_forward: dimensions out of range (hidden <= 256)");
    hipStream_t s = (hipStream_t)stream;
    // z = x W1^T + b1  (B(k, n) = W1[n][k])
    if (gemm(s, d_x, k_in, 1, d_w1, 1, k_in, d_b1, d_act, hidden, (int)rows, hidden, k_in, 1)) return 1;
    hipLaunchKernelGGL(ln_relu_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, d_act, d_gamma, d_beta, d_xhat, d_rstd, (long long)rows, hidden,
                       (const float *)nullptr, (const float *)nullptr, (const long long *)nullptr, (const long long *)nullptr);
    TRCHK(hipGetLastError());
    // y = a W2^T + b2
    return gemm(s, d_act, hidden, 1, d_w2, 1, hidden, d_b2, d_y, n_out, (int)rows, n_out, hidden, 1);
}

int shapemol_mlp_backward(const float *d_x, const float *d_dy, int64_t rows, int32_t k_in, int32_t hidden, int32_t n_out,
                          const float *d_w1, const float *d_gamma, const float *d_beta, const float *d_w2, const float *d_xhat,
                          const float *d_rstd, const float *d_act, float *d_dx, float *d_dw1, float *d_db1, float *d_dgamma, float *d_dbeta,
                          float *d_dw2, float *d_db2, float *d_work, size_t work_floats, void *stream) {
    if (!d_x || !d_dy || !d_w1 || !d_gamma || !d_beta || !d_w2 || !d_xhat || !d_rstd || !d_dw1 || !d_db1 || !d_dgamma || !d_dbeta ||
        !d_dw2 || !d_db2 || !d_work)
        return tr_fail("shapemol_mlp_backward: null argument (only d_dx and d_act may be NULL)");
    if (bad_dims(rows, k_in, hidden, n_out)) return tr_fail("shapemol_mlp_backward: dimensions out of range (hidden <= 256)");
    if (work_floats < shapemol_mlp_backward_workspace(rows, k_in, hidden, n_out)) return tr_fail("shapemol_mlp_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int R = (int)rows, H = hidden;
    const int nwg = (int)((rows + kLnRows - 1) / kLnRows), sp = row_splits(rows), spr = real_splits(rows);
    float *act = d_work, *dz = act + (size_t)rows * H, *pw1 = dz + (size_t)rows * H, *pw2 = pw1 + (size_t)spr * H * k_in,
          *pln = pw2 + (size_t)spr * n_out * H, *pb2 = pln + (size_t)nwg * 3 * H;
    // da = dy W2  (A = dy [R][O], B(k, n) = W2[k][n])
    if (gemm(s, d_dy, n_out, 1, d_w2, H, 1, nullptr, dz, H, R, H, n_out, 1)) return 1;
    // db2 partials (column sums of dy)
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nwg), dim3(256), 0, s, d_dy, (long long)rows, n_out, pb2);
    TRCHK(hipGetLastError());
    if (reduce_parts(s, pb2, nwg, n_out, d_db2)) return 1;
    // the activation a = relu(xhat * gamma + beta): the forward's, when the caller kept it, else recomputed from xhat;
    // dW2 = dy^T a:  A(m, k) = dy[k][m], B(k, n) = a[k][n], reduction over the rows in splits
    if (!d_act) {
        hipLaunchKernelGGL(relu_affine_kernel, dim3((unsigned)(((long long)rows * H + 255) / 256)), dim3(256), 0, s, d_xhat, d_gamma, d_beta, act, (long long)rows, H);
        TRCHK(hipGetLastError());
        d_act = act;
    }
    if (gemm(s, d_dy, 1, n_out, d_act, H, 1, nullptr, pw2, H, n_out, H, R, sp)) return 1;
    if (reduce_parts(s, pw2, spr, (long long)n_out * H, d_dw2)) return 1;
    // LayerNorm + ReLU backward: da -> dz in place, dgamma | dbeta | db1 partials
    if (ln_backward(s, dz, d_xhat, d_rstd, d_gamma, d_beta, rows, H, nwg, pln, d_dgamma, d_dbeta, d_db1)) return 1;
    // dW1 = dz^T x:  A(m, k) = dz[k][m], B(k, n) = x[k][n]
    if (gemm(s, dz, 1, H, d_x, k_in, 1, nullptr, pw1, k_in, H, k_in, R, sp)) return 1;
    if (reduce_parts(s, pw1, spr, (long long)H * k_in, d_dw1)) return 1;
    // dx = dz W1  (B(k, n) = W1[k][n])
    if (d_dx && gemm(s, dz, H, 1, d_w1, k_in, 1, nullptr, d_dx, k_in, R, k_in, H, 1)) return 1;
    return 0;
}

// ---- the edge form of the MLP block: its input row for edge e = (centre i, neighbour j) is the reference's concatenation
// [r_e | h_i | h_j | s_i] (models/uni_transformer.py:60-66, 131-137), so the first Linear splits into an edge term and three
// per-node terms (W1 = [Wr | Wd | Ws | Wi] by columns):  z_e = r_e Wr^T + (h Wd^T + s Wi^T + b1)[i] + (h Ws^T)[j].  The per-node
// products are computed once per atom instead of once per edge (8-32x fewer rows for 288 of the 308 input columns), and the
// backward sums dz over each atom's edges before its node-level products.
struct EdgeDims { int64_t E, N; int kr, H, Si, hidden, n_out, K1; };
static bool bad_edge_dims(const EdgeDims &d) {
    return d.E < 1 || d.E > 65535ll * 64 || d.N < 1 || d.N > 65535ll * 64 || d.kr < 1 || d.kr > 1024 || d.H < 1 || d.H > 1024 || d.Si < 0 || d.Si > 1024 ||
           d.hidden < 1 || d.hidden > kMaxHidden || d.n_out < 1 || d.n_out > 4096;
}
struct EdgeWork { size_t act, dz, dpd, dps, part, pln, pb, gb, total; };
static EdgeWork edge_work(const EdgeDims &d) {
    const size_t spE = (size_t)real_splits(d.E), spN = (size_t)real_splits(d.N), hid = (size_t)d.hidden;
    const size_t nwgE = (size_t)((d.E + kLnRows - 1) / kLnRows);
    EdgeWork w;
    size_t o = 0;
    w.act = o; o += (size_t)d.E * hid;
    w.dz = o; o += (size_t)d.E * hid;
    w.dpd = o; o += (size_t)d.N * hid;
    w.dps = o; o += (size_t)d.N * hid;
    w.part = o; o += std::max({spE * d.n_out * hid, spE * hid * d.kr, spN * hid * (size_t)std::max(d.H, d.Si)});
    w.pln = o; o += nwgE * 3 * hid;
    w.pb = o; o += nwgE * d.n_out;
    w.gb = o; o += 64;
    w.total = o;
    return w;
}

size_t shapemol_edge_mlp_backward_workspace(int64_t n_edges, int64_t n_nodes, int32_t k_edge, int32_t k_node, int32_t k_shape, int32_t hidden, int32_t n_out) {
    const EdgeDims d{n_edges, n_nodes, k_edge, k_node, k_shape, hidden, n_out, k_edge + 2 * k_node + k_shape};
    return bad_edge_dims(d) ? 0 : edge_work(d).total;
}

int shapemol_edge_mlp_forward(const float *d_r, const float *d_h, const float *d_s, const int64_t *d_dst, const int64_t *d_src, int64_t n_edges,
                              int64_t n_nodes, int32_t k_edge, int32_t k_node, int32_t k_shape, int32_t hidden, int32_t n_out, const float *d_w1,
                              const float *d_b1, const float *d_gamma, const float *d_beta, const float *d_w2, const float *d_b2, float *d_y,
                              float *d_xhat, float *d_rstd, float *d_act, float *d_pd, float *d_ps, void *stream) {
    if (!d_r || !d_h || !d_dst || !d_src || !d_w1 || !d_b1 || !d_gamma || !d_beta || !d_w2 || !d_b2 || !d_y || !d_xhat || !d_rstd || !d_act || !d_pd || !d_ps ||
        (k_shape > 0 && !d_s))
        return tr_fail("shapemol_edge_mlp_forward: null argument");
    const EdgeDims d{n_edges, n_nodes, k_edge, k_node, k_shape, hidden, n_out, k_edge + 2 * k_node + k_shape};
    if (bad_edge_dims(d)) return tr_fail("shapemol_edge_mlp_forward: dimensions out of range");
    hipStream_t s = (hipStream_t)stream;
    const int N = (int)n_nodes, E = (int)n_edges, K1 = d.K1;
    const float *wr = d_w1, *wd = d_w1 + k_edge, *ws = wd + k_node, *wi = ws + k_node;      // column blocks of W1 [hidden][K1]
    // pd = h Wd^T + b1 (+ s Wi^T);  ps = h Ws^T;  z = r Wr^T
    if (gemm(s, d_h, k_node, 1, wd, 1, K1, d_b1, d_pd, hidden, N, hidden, k_node, 1)) return 1;
    if (k_shape > 0 && gemm(s, d_s, k_shape, 1, wi, 1, K1, nullptr, d_pd, hidden, N, hidden, k_shape, 1, true)) return 1;
    if (gemm(s, d_h, k_node, 1, ws, 1, K1, nullptr, d_ps, hidden, N, hidden, k_node, 1)) return 1;
    if (gemm(s, d_r, k_edge, 1, wr, 1, K1, nullptr, d_act, hidden, E, hidden, k_edge, 1)) return 1;
    hipLaunchKernelGGL(ln_relu_fwd_kernel, dim3((unsigned)((n_edges + 3) / 4)), dim3(256), 0, s, d_act, d_gamma, d_beta, d_xhat, d_rstd, (long long)n_edges, hidden,
                       (const float *)d_pd, (const float *)d_ps, reinterpret_cast<const long long *>(d_dst), reinterpret_cast<const long long *>(d_src));
    TRCHK(hipGetLastError());
    return gemm(s, d_act, hidden, 1, d_w2, 1, hidden, d_b2, d_y, n_out, E, n_out, hidden, 1);
}

int shapemol_edge_mlp_backward(const float *d_r, const float *d_h, const float *d_s, const int64_t *d_ptr_dst, const int64_t *d_perm_src,
                               const int64_t *d_ptr_src, const float *d_dy, int64_t n_edges, int64_t n_nodes, int32_t k_edge, int32_t k_node,
                               int32_t k_shape, int32_t hidden, int32_t n_out, const float *d_w1, const float *d_gamma, const float *d_beta,
                               const float *d_w2, const float *d_xhat, const float *d_rstd, const float *d_act, float *d_dr, float *d_dh, float *d_ds, float *d_dw1,
                               float *d_db1, float *d_dgamma, float *d_dbeta, float *d_dw2, float *d_db2, float *d_work, size_t work_floats, void *stream) {
    if (!d_r || !d_h || !d_ptr_dst || !d_perm_src || !d_ptr_src || !d_dy || !d_w1 || !d_gamma || !d_beta || !d_w2 || !d_xhat || !d_rstd || !d_dr || !d_dh ||
        !d_dw1 || !d_db1 || !d_dgamma || !d_dbeta || !d_dw2 || !d_db2 || !d_work || (k_shape > 0 && (!d_s || !d_ds)))
        return tr_fail("shapemol_edge_mlp_backward: null argument");
    const EdgeDims d{n_edges, n_nodes, k_edge, k_node, k_shape, hidden, n_out, k_edge + 2 * k_node + k_shape};
    if (bad_edge_dims(d)) return tr_fail("shapemol_edge_mlp_backward: dimensions out of range");
    const EdgeWork w = edge_work(d);
    if (work_floats < w.total) return tr_fail("shapemol_edge_mlp_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int N = (int)n_nodes, E = (int)n_edges, K1 = d.K1, Hd = hidden;
    const int nwgE = (int)((n_edges + kLnRows - 1) / kLnRows);
    const int spE = row_splits(n_edges), sprE = real_splits(n_edges), spN = row_splits(n_nodes), sprN = real_splits(n_nodes);
    float *act = d_work + w.act, *dz = d_work + w.dz, *dpd = d_work + w.dpd, *dps = d_work + w.dps, *part = d_work + w.part, *pln = d_work + w.pln,
          *pb = d_work + w.pb;
    const float *wr = d_w1, *wd = d_w1 + k_edge, *ws = wd + k_node, *wi = ws + k_node;
    const long long *ptr_dst = reinterpret_cast<const long long *>(d_ptr_dst), *perm_src = reinterpret_cast<const long long *>(d_perm_src),
                    *ptr_src = reinterpret_cast<const long long *>(d_ptr_src);
    // second Linear and LayerNorm + ReLU: as shapemol_mlp_backward
    if (gemm(s, d_dy, n_out, 1, d_w2, Hd, 1, nullptr, dz, Hd, E, Hd, n_out, 1)) return 1;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nwgE), dim3(256), 0, s, d_dy, (long long)n_edges, n_out, pb);
    TRCHK(hipGetLastError());
    if (reduce_parts(s, pb, nwgE, n_out, d_db2)) return 1;
    if (!d_act) {
        hipLaunchKernelGGL(relu_affine_kernel, dim3((unsigned)(((long long)n_edges * Hd + 255) / 256)), dim3(256), 0, s, d_xhat, d_gamma, d_beta, act, (long long)n_edges, Hd);
        TRCHK(hipGetLastError());
        d_act = act;
    }
    if (gemm(s, d_dy, 1, n_out, d_act, Hd, 1, nullptr, part, Hd, n_out, Hd, E, spE)) return 1;
    if (reduce_parts(s, part, sprE, (long long)n_out * Hd, d_dw2)) return 1;
    if (ln_backward(s, dz, d_xhat, d_rstd, d_gamma, d_beta, n_edges, Hd, nwgE, pln, d_dgamma, d_dbeta, d_db1)) return 1;      // db1 = sum_e dz_e
    // the edge term: dWr = dz^T r (column block 0 of dW1), dr = dz Wr
    if (gemm(s, dz, 1, Hd, d_r, k_edge, 1, nullptr, part, k_edge, Hd, k_edge, E, spE)) return 1;
    if (reduce_parts(s, part, sprE, (long long)Hd * k_edge, d_dw1, k_edge, K1)) return 1;
    if (gemm(s, dz, Hd, 1, wr, K1, 1, nullptr, d_dr, k_edge, E, k_edge, Hd, 1)) return 1;
    // the per-node terms: dz summed over each atom's edges as centre (contiguous) and as neighbour (through perm_src)
    hipLaunchKernelGGL(seg_rowsum_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, s, (const float *)dz, ptr_dst, (const long long *)nullptr, dpd, (long long)n_nodes, Hd);
    hipLaunchKernelGGL(seg_rowsum_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, s, (const float *)dz, ptr_src, perm_src, dps, (long long)n_nodes, Hd);
    TRCHK(hipGetLastError());
    if (gemm(s, dpd, 1, Hd, d_h, k_node, 1, nullptr, part, k_node, Hd, k_node, N, spN)) return 1;                 // dWd = dpd^T h
    if (reduce_parts(s, part, sprN, (long long)Hd * k_node, d_dw1 + k_edge, k_node, K1)) return 1;
    if (gemm(s, dps, 1, Hd, d_h, k_node, 1, nullptr, part, k_node, Hd, k_node, N, spN)) return 1;                 // dWs = dps^T h
    if (reduce_parts(s, part, sprN, (long long)Hd * k_node, d_dw1 + k_edge + k_node, k_node, K1)) return 1;
    if (gemm(s, dpd, Hd, 1, wd, K1, 1, nullptr, d_dh, k_node, N, k_node, Hd, 1)) return 1;                        // dh = dpd Wd + dps Ws
    if (gemm(s, dps, Hd, 1, ws, K1, 1, nullptr, d_dh, k_node, N, k_node, Hd, 1, true)) return 1;
    if (k_shape > 0) {
        if (gemm(s, dpd, 1, Hd, d_s, k_shape, 1, nullptr, part, k_shape, Hd, k_shape, N, spN)) return 1;          // dWi = dpd^T s
        if (reduce_parts(s, part, sprN, (long long)Hd * k_shape, d_dw1 + k_edge + 2 * k_node, k_shape, K1)) return 1;
        if (gemm(s, dpd, Hd, 1, wi, K1, 1, nullptr, d_ds, k_shape, N, k_shape, Hd, 1)) return 1;                  // ds = dpd Wi
    }
    return 0;
}

// ---- the coordinate update's vector-neuron block (VNLinearLeakyReLU + VNBatchNorm + mean over channels), sm_train.h
static bool bad_vn_dims(int64_t n, int rows_o, int rows_s, int C) {
    // the VN kernels stage 2 C (1 + rows_o + rows_s) weights (and, backward, 2 * per-thread tiles) in dynamic LDS without raising the
    // 64 KB default limit: keep the weight block within 48 KB
    if (n < 1 || n > 65535ll * 64 || rows_o < 0 || rows_o > 256 || rows_s < 0 || rows_s > 1024 || C < 1 || C > 64) return true;
    return (size_t)2 * C * (1 + rows_o + rows_s) * sizeof(float) > 48 * 1024;
}
size_t shapemol_vn_backward_workspace(int64_t n_atoms, int32_t rows_o, int32_t rows_s, int32_t channels) {
    if (bad_vn_dims(n_atoms, rows_o, rows_s, channels)) return 0;
    const size_t NC = (size_t)n_atoms * channels, per = 256 / channels, nwg = ((size_t)n_atoms + per - 1) / per;
    return 9 * NC + 2 * (size_t)channels + nwg * 2 * channels * (size_t)(1 + rows_o + rows_s) + 64;
}

int shapemol_vn_forward(const float *d_x, const float *d_o3, const float *d_shape, const int64_t *d_batch, int64_t n_atoms, int32_t rows_o,
                        int32_t rows_s, int32_t channels, const float *d_wf, const float *d_wd, const float *d_bn_w, const float *d_bn_b,
                        float *d_run_mean, float *d_run_var, int32_t training, float *d_out, float *d_pf, float *d_dir, float *d_stats,
                        float *d_nrm, void *stream) {
    if (!d_x || !d_batch || !d_wf || !d_wd || !d_bn_w || !d_bn_b || !d_out || !d_pf || !d_dir || !d_stats || !d_nrm || (rows_o > 0 && !d_o3) ||
        (rows_s > 0 && !d_shape) || (!training && (!d_run_mean || !d_run_var)))
        return tr_fail("shapemol_vn_forward: null argument");
    if (bad_vn_dims(n_atoms, rows_o, rows_s, channels)) return tr_fail("shapemol_vn_forward: dimensions out of range (channels <= 64)");
    hipStream_t s = (hipStream_t)stream;
    VnTrainArgs a{};
    a.x = d_x; a.o3 = d_o3; a.shape = d_shape; a.batch = reinterpret_cast<const long long *>(d_batch);
    a.wf = d_wf; a.wd = d_wd; a.bn_w = d_bn_w; a.bn_b = d_bn_b; a.run_mean = d_run_mean; a.run_var = d_run_var;
    a.pf = d_pf; a.dir = d_dir; a.nrm = d_nrm; a.stats = d_stats; a.out = d_out;
    a.n_atoms = n_atoms; a.rows_o = rows_o; a.rows_s = rows_s; a.C = channels; a.Cin = 1 + rows_o + rows_s; a.training = training ? 1 : 0;
    const int per = 256 / channels;
    const unsigned nwg = (unsigned)((n_atoms + per - 1) / per);
    hipLaunchKernelGGL(vn_lin_kernel, dim3(nwg), dim3(256), (size_t)2 * channels * a.Cin * sizeof(float), s, a);
    hipLaunchKernelGGL(vn_bn_stats_kernel, dim3(channels), dim3(256), 0, s, a);
    hipLaunchKernelGGL(vn_act_kernel, dim3(nwg), dim3(256), 0, s, a);
    TRCHK(hipGetLastError());
    return 0;
}

int shapemol_vn_backward(const float *d_x, const float *d_o3, const float *d_shape, const int64_t *d_batch, int64_t n_atoms, int32_t rows_o,
                         int32_t rows_s, int32_t channels, const float *d_wf, const float *d_wd, const float *d_bn_w, const float *d_bn_b,
                         const float *d_pf, const float *d_dir, const float *d_stats, int32_t training, const float *d_gout, float *d_dx,
                         float *d_do3, float *d_dwf, float *d_dwd, float *d_dbn_w, float *d_dbn_b, float *d_work, size_t work_floats, void *stream) {
    if (!d_x || !d_batch || !d_wf || !d_wd || !d_bn_w || !d_bn_b || !d_pf || !d_dir || !d_stats || !d_gout || !d_dx || !d_dwf || !d_dwd || !d_dbn_w || !d_dbn_b ||
        !d_work || (rows_o > 0 && (!d_o3 || !d_do3)) || (rows_s > 0 && !d_shape))
        return tr_fail("shapemol_vn_backward: null argument");
    if (bad_vn_dims(n_atoms, rows_o, rows_s, channels)) return tr_fail("shapemol_vn_backward: dimensions out of range (channels <= 64)");
    if (work_floats < shapemol_vn_backward_workspace(n_atoms, rows_o, rows_s, channels)) return tr_fail("shapemol_vn_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t NC = (size_t)n_atoms * channels;
    const int per = 256 / channels, Cin = 1 + rows_o + rows_s;
    const unsigned nwg = (unsigned)((n_atoms + per - 1) / per);
    VnTrainArgs a{};
    a.x = d_x; a.o3 = d_o3; a.shape = d_shape; a.batch = reinterpret_cast<const long long *>(d_batch);
    a.wf = d_wf; a.wd = d_wd; a.bn_w = d_bn_w; a.bn_b = d_bn_b;
    a.pf = const_cast<float *>(d_pf); a.dir = const_cast<float *>(d_dir); a.stats = const_cast<float *>(d_stats);
    a.gout = d_gout;
    a.dd = d_work; a.dpfd = a.dd + 3 * NC; a.dnd = a.dpfd + 3 * NC; a.dnbn = a.dnd + NC; a.xhat = a.dnbn + NC; a.sums = a.xhat + NC; a.wpart = a.sums + 2 * channels;
    a.dbn_w = d_dbn_w; a.dbn_b = d_dbn_b; a.dx = d_dx; a.do3 = d_do3;
    a.n_atoms = n_atoms; a.rows_o = rows_o; a.rows_s = rows_s; a.C = channels; a.Cin = Cin; a.training = training ? 1 : 0;
    hipLaunchKernelGGL(vn_bwd_a_kernel, dim3(nwg), dim3(256), 0, s, a);
    hipLaunchKernelGGL(vn_bwd_stats_kernel, dim3(channels), dim3(256), 0, s, a);
    hipLaunchKernelGGL(vn_bwd_b_kernel, dim3(nwg), dim3(256), ((size_t)2 * channels * Cin + (size_t)2 * per * channels * 3) * sizeof(float), s, a);
    TRCHK(hipGetLastError());
    return reduce_parts3(s, a.wpart, (int)nwg, (long long)channels * Cin, d_dwf, d_dwd, nullptr);
}

int shapemol_seg_attention_forward(const float *d_q, const float *d_k, const float *d_vals, const int64_t *d_ptr, int64_t n_atoms,
                                   int32_t heads, int32_t dh, int32_t width, float *d_out, void *stream) {
    if (!d_q || !d_k || !d_vals || !d_ptr || !d_out) return tr_fail("shapemol_seg_attention_forward: null argument");
    if (n_atoms < 1 || heads < 1 || dh < 1 || dh > kSegAttnMaxDh || width < 1 || width > kSegAttnMaxW) return tr_fail("shapemol_seg_attention_forward: dimensions out of range (dh, width <= 8)");
    SegAttnArgs a{d_q, d_k, d_vals, reinterpret_cast<const long long *>(d_ptr), d_out, nullptr, nullptr, nullptr, nullptr, (int)n_atoms, heads, dh, width};
    hipLaunchKernelGGL(seg_attention_kernel<false>, dim3((unsigned)((n_atoms * heads * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    TRCHK(hipGetLastError());
    return 0;
}

int shapemol_seg_attention_backward(const float *d_q, const float *d_k, const float *d_vals, const int64_t *d_ptr, const float *d_dout,
                                    int64_t n_atoms, int32_t heads, int32_t dh, int32_t width, float *d_dq, float *d_dk, float *d_dvals,
                                    void *stream) {
    if (!d_q || !d_k || !d_vals || !d_ptr || !d_dout || !d_dq || !d_dk || !d_dvals) return tr_fail("shapemol_seg_attention_backward: null argument");
    if (n_atoms < 1 || heads < 1 || dh < 1 || dh > kSegAttnMaxDh || width < 1 || width > kSegAttnMaxW) return tr_fail("shapemol_seg_attention_backward: dimensions out of range (dh, width <= 8)");
    SegAttnArgs a{d_q, d_k, d_vals, reinterpret_cast<const long long *>(d_ptr), nullptr, d_dout, d_dq, d_dk, d_dvals, (int)n_atoms, heads, dh, width};
    hipLaunchKernelGGL(seg_attention_kernel<true>, dim3((unsigned)((n_atoms * heads * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    TRCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
