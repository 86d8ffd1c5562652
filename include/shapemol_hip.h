/*
 * shapemol_hip.h -- C ABI of the MI355X (gfx950) implementation of ShapeMol's denoising hot path.
 *
 * The reference has no FFI of its own (it is pure Python on third-party torch ops), so the
 * boundary is its Python call surface; each entry point below names the reference call it
 * replaces (paths relative to the reference repository root):
 *
 *   shapemol_create / _destroy     ScorePosNet3D.__init__ + load_state_dict
 *                                  models/molopt_score_model.py:171-283, scripts/sample_diffusion.py:211-215
 *   shapemol_score                 ScorePosNet3D.forward        models/molopt_score_model.py:286-320
 *                                  (-> UniTransformerO2TwoUpdateGeneral.forward, models/uni_transformer.py:483-540)
 *   shapemol_sample                ScorePosNet3D.sample_diffusion  models/molopt_score_model.py:533-697
 *   shapemol_log_sample_categorical  log_sample_categorical     models/molopt_score_model.py:98-104
 *
 * Conventions
 *   - plain pointers and sizes only; every `d_` pointer is DEVICE memory on the context's device,
 *     caller-owned, float32 / int64 exactly as the reference's tensors (row-major, contiguous);
 *     outputs are caller-allocated.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - no hidden host synchronisation in _score/_sample/_log_sample_categorical: work is enqueued on
 *     `stream` and the call returns; the caller synchronises.
 *   - return 0 on success, non-zero on error; shapemol_last_error() returns the message of the last
 *     failing call on this thread.
 *   - one context per device; a context is not thread-safe.
 */
#ifndef SHAPEMOL_HIP_H
#define SHAPEMOL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHAPEMOL_ABI_VERSION 5

typedef struct shapemol_ctx shapemol_ctx;

/* `model` section of the training YAML (the YAML files under config/training, lines 21-74), reduced to what the path uses. */
typedef struct shapemol_config {
    int32_t hidden_dim;        /* 128 */
    int32_t n_heads;           /* 16  (hidden_dim / n_heads must be 8) */
    int32_t num_layers;        /* 8   */
    int32_t knn;               /* 8   (1..32) */
    int32_t num_r_gaussian;    /* 20  (the reference hard-codes 20 centres) */
    int32_t shape_dim;         /* 32  point-cloud latent vectors per molecule, each in R^3 */
    int32_t shape_latent_dim;  /* 32  */
    int32_t time_emb_dim;      /* 8   */
    int32_t num_classes;       /* 15  */
    int32_t num_timesteps;     /* 1000 */
} shapemol_config;

int shapemol_abi_version(void);
const char *shapemol_last_error(void);

/* Number of float32 values shapemol_create expects in `weights` for this config: every float entry
 * of the reference state dict, in registration order, densely concatenated (the int64
 * `num_batches_tracked` counters are skipped).  shapemol_amd.pack_state_dict builds it. */
size_t shapemol_weight_count(const shapemol_config *cfg);

/* weights: HOST pointer to shapemol_weight_count() floats.  device: HIP device ordinal. */
int shapemol_create(const shapemol_config *cfg, const float *weights, size_t n_weights,
                    int device, shapemol_ctx **out);
void shapemol_destroy(shapemol_ctx *ctx);

/* Pre-size the workspace (otherwise it grows on demand, which allocates). */
int shapemol_reserve(shapemol_ctx *ctx, int64_t max_atoms, int64_t max_mols);

/* One score evaluation.
 *   d_pos   (N,3) f32   ligand_pos_perturbed      d_v     (N,) i64  ligand_v_perturbed
 *   d_batch (N,) i64 sorted molecule id per atom  d_shape (B,shape_dim,3) f32  ligand_shape
 *   d_t     (B,) i64    time_step
 *   out_pos (N,3) f32   pred_ligand_pos   out_h (N,H) f32 pred_ligand_h (may be NULL)
 *   out_v   (N,C) f32   pred_ligand_v */
int shapemol_score(shapemol_ctx *ctx, const float *d_pos, const int64_t *d_v, const int64_t *d_batch,
                   int64_t n_atoms, int64_t n_mols, const float *d_shape, const int64_t *d_t,
                   float *out_pos, float *out_h, float *out_v, void *stream);

/* Trajectory buffers of shapemol_sample; any pointer may be NULL (that trajectory is not kept).
 * All are DEVICE buffers with a leading num_steps dimension. */
typedef struct shapemol_traj {
    float   *pos_traj;       /* (S,N,3)  x_{t-1} after each step          (pos_traj)       */
    int64_t *v_traj;         /* (S,N)    v_{t-1} after each step          (v_traj)         */
    float   *v0_traj;        /* (S,N,C)  log_softmax of predicted logits  (v0_traj)        */
    float   *vt_traj;        /* (S,N,C)  log posterior q(v_{t-1}|v_t,v0)  (vt_traj)        */
    float   *pos_cond_traj;  /* (S,N,3)  raw network x0 prediction        (pos_cond_traj)  */
    float   *v_cond_traj;    /* (S,N,C)  raw network logits               (v_cond_traj)    */
} shapemol_traj;

/* Reverse chain t = T-1 ... T-num_steps (center_pos_mode 'none', no guidance).
 *   d_init_pos (N,3) f32, d_init_v (N,) i64, d_batch, d_shape as in shapemol_score.
 *   Noise: if d_eps != NULL and d_u != NULL they are host-chosen draws, eps (S,N,3) ~ N(0,1) and
 *   u (S,N,C) ~ U[0,1), consumed per step in the reference's order (randn_like then rand_like,
 *   models/molopt_score_model.py:662,99).  Otherwise noise is generated on the device
 *   (Philox4x32-10 keyed by `seed`, Box-Muller).
 *   out_pos (N,3) f32, out_v (N,) i64: final state.  use_graph != 0 replays a captured hipGraph of one
 *   reverse step (and of twenty steps back to back).  The capture depends on (n_atoms, n_mols) only: seed,
 *   noise and trajectory pointers are passed through device memory, so chains with new seeds or new result
 *   buffers replay the same executable. */
int shapemol_sample(shapemol_ctx *ctx, const float *d_init_pos, const int64_t *d_init_v,
                    const int64_t *d_batch, int64_t n_atoms, int64_t n_mols, const float *d_shape,
                    int32_t num_steps, const float *d_eps, const float *d_u, uint64_t seed,
                    const shapemol_traj *traj, float *out_pos, int64_t *out_v,
                    int32_t use_graph, void *stream);

/* Point-cloud shape guidance of the following _sample calls (pointcloud_shape_guidance,
 * models/molopt_score_model.py:699-740; applied to the predicted x0 of every step with t > grad_step, :583-586):
 * an atom whose three nearest cloud points are on average farther than `radius` is pulled towards their mean by a random
 * fraction in [0.2, 0.8), up to five times.  h_cloud: HOST (n_points,3) float64 (copied); n_points = 0 switches it off.
 * d_draws: DEVICE (S,5,N) float64 uniforms, the np.random.random() draw of every (step, iteration, atom) (parity mode;
 * entries of atoms that are not pulled are ignored), or NULL: Philox(seed of the chain).  Replaces the reference's
 * per-step D2H + KD-tree + H2D round trip by one kernel inside the step graph. */
int shapemol_set_guidance(shapemol_ctx *ctx, const double *h_cloud, int64_t n_points, double radius,
                          int32_t grad_step, const double *d_draws);

/* pointcloud_shape_guidance on its own (models/molopt_score_model.py:699-740): guide d_pos (N,3) f32 in place against the
 * cloud of shapemol_set_guidance; d_draws DEVICE (5,N) float64 or NULL (Philox(seed)). */
int shapemol_guide_points(shapemol_ctx *ctx, float *d_pos, int64_t n_points, const double *d_draws, uint64_t seed, void *stream);

/* The same as the reference's MODULE-level function pointcloud_shape_guidance(use_pointcloud_data, pred_ligand_pos, k=3,
 * ratio=0.2) (models/molopt_score_model.py:699-740), which has no model object at hand: no context, the cloud comes with
 * the call (h_cloud: HOST (n_cloud,3) float64, 3 .. 2048 points) on the CURRENT device; d_pos (n_atoms,3) f32 DEVICE is
 * guided in place; `ratio` is the lower end of the pull fraction u * (0.8 - ratio) + ratio; d_draws / seed as above.
 * Unlike the other entry points this one synchronises `stream` before it returns (it owns a temporary device block); the
 * reference's function is synchronous as well (D2H copy, host KD-tree, H2D copy). */
int shapemol_pointcloud_guidance(const double *h_cloud, int64_t n_cloud, double radius, double ratio, float *d_pos,
                                 int64_t n_atoms, const double *d_draws, uint64_t seed, void *stream);

/* Input validation happens on the device (no host synchronisation in _score/_sample): an unsorted or
 * out-of-range d_batch, an atom type outside [0, num_classes) or a time step outside [0, num_timesteps)
 * sets a sticky flag (the offending index is clamped, so nothing is read or written out of bounds).
 * shapemol_status synchronises the device and returns non-zero (message in shapemol_last_error) if the
 * last _score/_sample on this context saw such an input, an activation beyond the fp16 range of the two-piece f16 node
 * kernels (option node_f16) or a timed-out grid barrier (vn_fuse = 1);
 * flags_out (may be NULL) receives the eight raw flags {barrier, batch, atom type, time step, fp16 range, 0...}.
 * The reference raises from the corresponding torch indexing ops (models/molopt_score_model.py:292-301). */
int shapemol_status(shapemol_ctx *ctx, int32_t *flags_out);
/* The same, synchronising only `stream` (the one the last _score/_sample of this context was enqueued on) instead of the whole
 * device: chains of other contexts running beside it are not waited for. */
int shapemol_status_stream(shapemol_ctx *ctx, int32_t *flags_out, void *stream);

/* Evaluation-mode batch-norm (the module after .eval(): scripts/train_diffusion.py:172-173 puts it there for validate(),
 * models/shape_vn_layers.py:50-61 then normalises the vector norms with BatchNorm1d's running statistics instead of the
 * batch's).  h_mean, h_var: HOST [num_layers][n_heads] float32, the buffers
 * refine_net.base_block.{l}.h2x_layers.0.shape_linear.batchnorm.bn.running_{mean,var}; count = num_layers * n_heads.
 * Takes effect with shapemol_set_option(ctx, "bn_eval", 1); training mode (the default, and what sampling uses: the
 * reference never leaves it while sampling) keeps using the statistics of the batch.  The running statistics
 * are read, never updated (no training step on this path). */
int shapemol_set_bn_running(shapemol_ctx *ctx, const float *h_mean, const float *h_var, int64_t count);

/* argmax_c( logits[n,c] - log(-log(u[n,c] + 1e-30) + 1e-30) ); d_u NULL -> device Philox(seed). */
int shapemol_log_sample_categorical(shapemol_ctx *ctx, const float *d_logits, const float *d_u,
                                    int64_t n_rows, int32_t n_classes, uint64_t seed,
                                    int64_t *out_index, void *stream);

/* ---- frozen shape encoder ------------------------------------------------------------
 * VN_DGCNN_Encoder.forward (models/shape_pointcloud_modelAE.py:207-255) with get_graph_feature_cross / dense knn
 * (models/shape_vn_layers.py:257-292): point clouds (B,N,3) -> shape latents (B,latent_dim,3), the `shape_emb` the
 * diffusion model is conditioned on (utils/shape.py:240-283).  Batch-norm runs on batch statistics, as the reference's
 * auto-encoder does (it is never put in eval mode).  hidden_dim 128, num_k 20 (the shipped checkpoint's config).
 * weights (HOST, float32, shapemol_se_weight_count of them): conv_pos {map_to_feat (C,2), bn weight (C), bn bias (C),
 * map_to_dir (C,2)}; per block {map_to_feat (C,2C), bn weight, bn bias, map_to_dir (C,2C)}; conv_c {map_to_feat
 * (latent,L*C), bn weight (latent), bn bias (latent), map_to_dir (L*C)}. */
typedef struct shapemol_se_ctx shapemol_se_ctx;
size_t shapemol_se_weight_count(int32_t hidden_dim, int32_t latent_dim, int32_t layer_num);
int shapemol_se_create(int32_t hidden_dim, int32_t latent_dim, int32_t layer_num, int32_t num_k, const float *weights,
                       size_t n_weights, int device, shapemol_se_ctx **out);
void shapemol_se_destroy(shapemol_se_ctx *ctx);
/* d_points (B,N,3) f32 DEVICE, N a multiple of 16; d_out (B,latent_dim,3) f32 DEVICE. */
int shapemol_se_encode(shapemol_se_ctx *ctx, const float *d_points, int64_t n_shapes, int64_t n_points, float *d_out, void *stream);

/* ---- training building blocks (the operators of a layer, forward and backward) ---------------
 * The MLP block of models/common.py:47-67 -- y = W2 relu(LayerNorm(W1 x + b1)) + b2, eps 1e-5, affine LayerNorm -- forward
 * with the quantities its backward needs, and the backward: what torch.autograd does for the 58 MLPs of one score
 * evaluation when scripts/train_diffusion.py:135-147 calls loss.backward().  fp32 arithmetic (fp32 MFMA products), device
 * pointers, row-major: x (rows,k_in), w1 (hidden,k_in), b1/gamma/beta (hidden), w2 (n_out,hidden), b2 (n_out), y (rows,n_out).
 * _forward also writes xhat (rows,hidden) = the normalised pre-activation, rstd (rows) and act (rows,hidden) = the ReLU output
 * (xhat and rstd must be kept for _backward; act may be kept and handed back, or dropped: _backward recomputes it from xhat when
 * d_act is NULL).
 * _backward: dy (rows,n_out) -> dx (rows,k_in; may be NULL), dw1, db1, dgamma, dbeta, dw2, db2 (OVERWRITTEN, not
 * accumulated); d_work: shapemol_mlp_backward_workspace() floats.  Reductions over the rows are deterministic. */
size_t shapemol_mlp_backward_workspace(int64_t rows, int32_t k_in, int32_t hidden, int32_t n_out);
int shapemol_mlp_forward(const float *d_x, int64_t rows, int32_t k_in, int32_t hidden, int32_t n_out, const float *d_w1,
                         const float *d_b1, const float *d_gamma, const float *d_beta, const float *d_w2, const float *d_b2,
                         float *d_y, float *d_xhat, float *d_rstd, float *d_act, void *stream);
int shapemol_mlp_backward(const float *d_x, const float *d_dy, int64_t rows, int32_t k_in, int32_t hidden, int32_t n_out,
                          const float *d_w1, const float *d_gamma, const float *d_beta, const float *d_w2, const float *d_xhat,
                          const float *d_rstd, const float *d_act, float *d_dx, float *d_dw1, float *d_db1, float *d_dgamma, float *d_dbeta,
                          float *d_dw2, float *d_db2, float *d_work, size_t work_floats, void *stream);

/* The same block on edge rows whose input is the reference's concatenation [r_e | h_i | h_j | s_i] for edge e = (centre i = dst[e],
 * neighbour j = src[e]) (models/uni_transformer.py:60-66 / :131-137: hk_func, hv_func, xk_func, xv_func): r (n_edges, k_edge) edge
 * features, h (n_nodes, k_node) atom features, s (n_nodes, k_shape) per-atom shape embedding (k_shape may be 0, d_s NULL),
 * W1 (hidden, k_edge + 2 k_node + k_shape) with the reference's column order.  The concatenated rows are never formed: the
 * first Linear is evaluated as an edge term plus per-atom terms (pd, ps: (n_nodes, hidden) scratch), and the backward sums over
 * each atom's edges before its per-atom products.  Edges must be grouped by centre atom (ptr_dst: n_nodes + 1 CSR offsets);
 * perm_src / ptr_src list the edges grouped by neighbour atom (perm_src[t] = edge index).  Outputs and workspace as in
 * shapemol_mlp_*; dW1 is written whole, dr / dh / ds are the gradients of r / h / s. */
size_t shapemol_edge_mlp_backward_workspace(int64_t n_edges, int64_t n_nodes, int32_t k_edge, int32_t k_node, int32_t k_shape, int32_t hidden, int32_t n_out);
int shapemol_edge_mlp_forward(const float *d_r, const float *d_h, const float *d_s, const int64_t *d_dst, const int64_t *d_src, int64_t n_edges,
                              int64_t n_nodes, int32_t k_edge, int32_t k_node, int32_t k_shape, int32_t hidden, int32_t n_out, const float *d_w1,
                              const float *d_b1, const float *d_gamma, const float *d_beta, const float *d_w2, const float *d_b2, float *d_y,
                              float *d_xhat, float *d_rstd, float *d_act, float *d_pd, float *d_ps, void *stream);
int shapemol_edge_mlp_backward(const float *d_r, const float *d_h, const float *d_s, const int64_t *d_ptr_dst, const int64_t *d_perm_src,
                               const int64_t *d_ptr_src, const float *d_dy, int64_t n_edges, int64_t n_nodes, int32_t k_edge, int32_t k_node,
                               int32_t k_shape, int32_t hidden, int32_t n_out, const float *d_w1, const float *d_gamma, const float *d_beta,
                               const float *d_w2, const float *d_xhat, const float *d_rstd, const float *d_act, float *d_dr, float *d_dh, float *d_ds, float *d_dw1,
                               float *d_db1, float *d_dgamma, float *d_dbeta, float *d_dw2, float *d_db2, float *d_work, size_t work_floats, void *stream);

/* The coordinate update's vector-neuron block on the training path: VNLinearLeakyReLU with VNBatchNorm
 * (models/shape_vn_layers.py:41-61, 95-110) applied to [x_n | o3_n | shape_mol(n)] and averaged over the output channels
 * (models/uni_transformer.py:157-160).  x (n_atoms, 3), o3 (n_atoms, rows_o, 3), shape (n_molecules, rows_s, 3), batch (n_atoms)
 * molecule index of every atom, wf / wd (channels, 1 + rows_o + rows_s), out (n_atoms, 3).  training != 0: batch statistics
 * (running estimates, when given, updated with momentum 0.1 as nn.BatchNorm1d does); 0: the running estimates.  pf, dir
 * (n_atoms, channels, 3) and stats (2, channels) are kept for the backward; nrm (n_atoms, channels) is scratch.  _backward: dx,
 * do3, dwf, dwd (channels, 1 + rows_o + rows_s each), dbn_w, dbn_b from gout (n_atoms, 3); no gradient for shape (it comes
 * from the frozen encoder). */
size_t shapemol_vn_backward_workspace(int64_t n_atoms, int32_t rows_o, int32_t rows_s, int32_t channels);
int shapemol_vn_forward(const float *d_x, const float *d_o3, const float *d_shape, const int64_t *d_batch, int64_t n_atoms, int32_t rows_o,
                        int32_t rows_s, int32_t channels, const float *d_wf, const float *d_wd, const float *d_bn_w, const float *d_bn_b,
                        float *d_run_mean, float *d_run_var, int32_t training, float *d_out, float *d_pf, float *d_dir, float *d_stats,
                        float *d_nrm, void *stream);
int shapemol_vn_backward(const float *d_x, const float *d_o3, const float *d_shape, const int64_t *d_batch, int64_t n_atoms, int32_t rows_o,
                         int32_t rows_s, int32_t channels, const float *d_wf, const float *d_wd, const float *d_bn_w, const float *d_bn_b,
                         const float *d_pf, const float *d_dir, const float *d_stats, int32_t training, const float *d_gout, float *d_dx,
                         float *d_do3, float *d_dwf, float *d_dwd, float *d_dbn_w, float *d_dbn_b, float *d_work, size_t work_floats, void *stream);

/* The attention of one layer on the training path (models/uni_transformer.py:71-81 / :141-151): for every centre atom i, whose
 * incoming edges are e in [ptr[i], ptr[i+1]) (edges grouped by centre), and head h: logits <q_i[h], k_e[h]> / sqrt(dh), softmax
 * over the atom's edges, out_i[h][:] = sum_e alpha_e vals_e[h][:].  q (n_atoms, heads*dh), k (n_edges, heads*dh), vals
 * (n_edges, heads, width) with width = dh (x2h) or 3 (h2x: value times relative position); dh, width <= 8.  _backward writes
 * dq, dk, dvals (every edge belongs to exactly one atom: no accumulation, deterministic). */
int shapemol_seg_attention_forward(const float *d_q, const float *d_k, const float *d_vals, const int64_t *d_ptr, int64_t n_atoms,
                                   int32_t heads, int32_t dh, int32_t width, float *d_out, void *stream);
int shapemol_seg_attention_backward(const float *d_q, const float *d_k, const float *d_vals, const int64_t *d_ptr, const float *d_dout,
                                    int64_t n_atoms, int32_t heads, int32_t dh, int32_t width, float *d_dq, float *d_dk, float *d_dvals,
                                    void *stream);

/* ---- diagnostics (used by the parity tests and the bench; not needed by a caller) ---- */
/* (options marked "_sample only" do not affect _score)
 * options: "first_step" (the following _sample calls resume a chain at reverse step v, i.e. at t = T-1-v, from the
 *                        state given as d_init_pos / d_init_v; noise and trajectory rows stay indexed from 0; default 0.
 *                        Used by the windowed full-length parity test; the reference always starts at T-1),
 *          "stop_layer" (run only the first v layers of the next _score; -1 = all),
 *          "edge_bf16"  (3 = fused key/value edge kernel on two-piece f16 operands, both MLP images resident;
 *                        2 = streaming kernels on exactly split bf16 operands (three pieces = the 24 bits of fp32, six products):
 *                            weights stationary in the registers of consumer waves, producer waves stream edge tiles through
 *                            LDS (sm_edge_stream.h; k <= 16) -- the reference-precision path;
 *                        1 = fused phase kernel on exactly split bf16 operands (weight swap between the phases; k <= 16),
 *                        0 = fp32-MFMA edge kernels),
 *          "node_f16"   (1 = node kernels (prologue, chain, per-node products) on two-piece f16 operands [default]: the
 *                        residual stream then passes through fp16 pieces, |x| >= 6e4 raises a status flag;
 *                        0 = the same kernels on exactly split bf16 operands, six products per term),
 *          "feat_f16"   (1 = "f16 features": every matrix product of the f16 kernels on the leading f16 piece of both operands
 *                        only -- one product per term instead of three, 11 significand bits in the operands; accumulation,
 *                        LayerNorm, softmax, coordinates, batch statistics stay fp32.  A reduced-precision throughput mode
 *                        (forward error ~1e-3), outside the parity gates; 0 = two-piece operands [default]),
 *          "lin_bf16", "chain_bf16" (1 = node kernels on the matrix cores with split operands [default],
 *                        0 = fp32-MFMA node kernels),
 *          "edge_tiles" (f16 edge kernels when the waves have several jobs (batches beyond ~6k atoms, k > 16): 1 = one looping
 *                        launch, eight waves per workgroup over consecutive jobs with the next job's rows prefetched,
 *                        0 = sliced launches of the one-job kernel, -1 = automatic (= 1) [default]; k > 16 runs every atom
 *                        as two 16-slot tiles merged by an online-softmax combine kernel in either form),
 *          "max_mol_atoms" (_sample only: largest molecule, in atoms, of the batches of the following chains; 0 = unknown [default].  With it
 *                        the coordinate update of every layer but the last runs in the prologue of the next layer's x2h
 *                        kernel (each workgroup recomputes the coordinates of the molecules its atoms belong to) instead
 *                        of as a launch of its own; a value smaller than the truth raises a status flag.  Does not touch the
 *                        captured graph unless it changes that decision),
 *          "vn_fold"    (1 = fold as above when max_mol_atoms allows [default], 0 = always launch vn_apply),
 *          "vn_fuse"    (2 = VN-linear + batch-norm statistics in the epilogue of the h2x attention, vn_apply as its
 *                        own launch [default]; 1 = the whole coordinate update behind h2x with an in-kernel grid
 *                        barrier (no faster: measured); 0 = separate vn_stats / vn_apply launches),
 *          "bn_eval"    (1 = evaluation-mode batch-norm: the running statistics of shapemol_set_bn_running; 0 = the
 *                        statistics of the batch [default, and what the reference's sampling runs with]),
 *          "x2h_chain"  (1 = x2h attention and node stage of a layer in one launch when every wave has one job in a
 *                        single launch (up to ~6k atoms) [default], 0 = separate launches),
 *          "graph_fuse" (1 = kNN graph + edge weights in one launch when max_mol_atoms is known and <= 128 [default]),
 *          "ddpm_fold"  (_sample only: 1 = the last layer's coordinate update inside the posterior-step kernel when the
 *                        fold above applies and no guidance is set [default]),
 *          "edge_waves" (waves per workgroup of the edge kernels, 1..12; 0 = automatic: ceil(jobs / CUs) [default]; a
 *                        value other than 0 also selects the separate launches),
 *          "lin_waves"  (1..16 waves per workgroup of node_linear_kernel, tuning),
 *          "stream_whole_rounds" (streaming edge kernels, k <= 16: 1 = tiles per workgroup rounded up to whole rounds of two; 0 =
 *                        ceil(tiles / CUs), the last round of an odd count has one tile [default]; tuning, same results),
 *          "stamps", "kstamp_sel" (clock-stamp diagnostics; only meaningful in the --stamps build).
 * Changing an option invalidates a captured graph (the next _sample re-captures). */
int shapemol_set_option(shapemol_ctx *ctx, const char *name, int64_t value);
/* Diagnostic of the parity tests: pin the kNN graph of the following _sample calls at given (reverse step, atom) pairs to given
 * neighbour lists (the reference's own, recorded where its k-th / (k+1)-th choice is closer than a float32 implementation can
 * reproduce: models/uni_transformer.py:446-473 builds one graph per score evaluation, and a flipped neighbour sends the molecule
 * down another trajectory for good).  h_off: HOST [n_steps + 1] CSR offsets of the pins of reverse step s = 0 .. n_steps - 1
 * (s = num_timesteps - 1 - t); h_atom [n_pins] atom index; h_nbr [n_pins][k] neighbour indices (global atom indices, the
 * reference's order).  n_pins = 0 removes the pins.  While pins are set the graph is built by the separate kNN / edge-weight
 * launches (the pinned rows are overwritten between them). */
int shapemol_set_knn_pins(shapemol_ctx *ctx, const int32_t *h_off, int32_t n_steps, const int32_t *h_atom, const int32_t *h_nbr,
                          int64_t n_pins, int32_t k);
/* Diagnostic (host only, no device needed): the exact three-way bf16 split of the default precision mode -- every matrix operand x is
 * carried as hi + mid + lo with hi = x truncated to bf16, mid = (x - hi) truncated, lo = x - hi - mid (8 + 8 + 8 significand bits,
 * fp32 exponent range), so that x == hi + mid + lo EXACTLY for every fp32 value of magnitude 2^-110 (7.7e-34) and above (below that the
 * third piece falls under bf16's smallest subnormal and the error is bounded by 2^-133) (fp32 nn.Linear operands of the reference:
 * models/common.py:47-67).  pieces: three bf16 bit patterns.  tests/test_host.py checks the identity over random and edge values. */
void shapemol_debug_split_exact(float x, uint16_t *pieces);
/* Copy an internal device buffer of the last _score to HOST memory (synchronises the device).
 * names: "nbr" (N,KP) i32, "ew" (N,KP) f32, "h" (N,H), "x" (N,3), "pre" (N,4H), "q" (N,H),
 *        "att" (N,H), "o3" (N,48), "bnstat" (L,16,2,heads) f64, "dims" (8,) i64,
 *        "captures" (1,) i64 hipGraph captures of this context so far.
 * Returns the number of bytes written, or -1. */
int64_t shapemol_debug_read(shapemol_ctx *ctx, const char *name, void *host_dst, size_t max_bytes);
/* Per-kernel launch-time accounting with HIP events on the launch stream (bench only).
 * shapemol_profile_begin() arms it for the following _score/_sample calls (forces eager launches);
 * shapemol_profile_end() synchronises and writes, for each of up to `cap` kernel classes,
 * name / total milliseconds / launch count.  Returns the number of classes. */
int shapemol_profile_begin(shapemol_ctx *ctx);
int shapemol_profile_end(shapemol_ctx *ctx, char (*names)[32], double *total_ms, int64_t *launches, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SHAPEMOL_HIP_H */
