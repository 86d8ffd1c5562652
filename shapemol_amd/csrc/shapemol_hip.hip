// libshapemol_hip.so -- C ABI (include/shapemol_hip.h) over the gfx950 kernels.
// Host side: weight repacking into MFMA-friendly images, workspace, launch sequences for one
// score evaluation (ScorePosNet3D.forward) and for the reverse chain (sample_diffusion), hipGraph
// capture of one chain step, per-kernel event timing.
#include "../../include/shapemol_hip.h"
#include "sm_device.h"
#include "sm_edge.h"
#include "sm_edge_bf16.h"
#include "sm_edge16.h"
#include "sm_node.h"
#include "sm_node16.h"
#include "sm_edge_stream.h"
#include "sm_misc.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string &m) { g_err = m; return 1; }

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

constexpr int kEdgeThreadsDefault = 768;
#ifndef SM_GRAPH_UNROLL
#define SM_GRAPH_UNROLL 20
#endif
constexpr int kGraphUnroll = SM_GRAPH_UNROLL;   // reverse steps per graph launch

// ---- host view of the packed weight array (order documented in shapemol_amd/packing.py) ----
struct Lin { const float *w = nullptr, *b = nullptr; int out = 0, in = 0; };
struct Mlp { Lin l1; const float *g = nullptr, *be = nullptr; Lin l2; };
struct Cursor {
    const float *p; size_t left;
    bool ok = true;
    const float *take(size_t n) {
        if (n > left) { ok = false; return p; }
        const float *r = p; p += n; left -= n; return r;
    }
    Lin lin(int out, int in, bool bias = true) {
        Lin l; l.out = out; l.in = in; l.w = take((size_t)out * in); l.b = bias ? take(out) : nullptr; return l;
    }
    Mlp mlp(int in, int hid, int out) {
        Mlp m; m.l1 = lin(hid, in); m.g = take(hid); m.be = take(hid); m.l2 = lin(out, hid); return m;
    }
};
struct HostLayer { Mlp hk, hv, hq, no, xk, xv, xq; const float *vn_f, *bn_g, *bn_b, *vn_d; };
struct HostModel {
    const float *tab[7];
    Lin te1, te2, emb;
    Mlp ew;
    std::vector<HostLayer> layer;
    Mlp inv;
    Lin v1, v2;
};

size_t weight_count(const shapemol_config &c) {
    const size_t H = c.hidden_dim, G = c.num_r_gaussian, S = c.shape_dim, SL = c.shape_latent_dim,
                 D = c.time_emb_dim, C = c.num_classes, T = c.num_timesteps, hd = c.n_heads;
    const size_t kv = G + 2 * H + SL, cin = 1 + hd + S;
    auto mlp = [](size_t in, size_t hid, size_t out) { return hid * in + hid + 2 * hid + out * hid + out; };
    size_t n = 7 * T;
    n += 2 * D * D + 2 * D + D * 2 * D + D;
    n += H * (C + D) + H;
    n += mlp(G, H, 1);
    const size_t per_layer = 2 * mlp(kv, H, H) + mlp(H, H, H) + mlp(2 * H, H, H) + mlp(kv, H, H) +
                             mlp(kv, H, hd) + mlp(H, H, H) + 2 * hd * cin + 2 * hd;
    n += (size_t)c.num_layers * per_layer;
    n += mlp(S, S, SL);
    n += H * H + H + C * H + C;
    return n;
}

bool parse_weights(const shapemol_config &c, const float *w, size_t n, HostModel &m) {
    const int H = c.hidden_dim, G = c.num_r_gaussian, S = c.shape_dim, SL = c.shape_latent_dim,
              D = c.time_emb_dim, C = c.num_classes, T = c.num_timesteps, hd = c.n_heads;
    const int kv = G + 2 * H + SL, cin = 1 + hd + S;
    Cursor cu{w, n};
    for (int i = 0; i < 7; ++i) m.tab[i] = cu.take(T);
    m.te1 = cu.lin(2 * D, D);
    m.te2 = cu.lin(D, 2 * D);
    m.emb = cu.lin(H, C + D);
    m.ew = cu.mlp(G, H, 1);
    m.layer.resize(c.num_layers);
    for (auto &L : m.layer) {
        L.hk = cu.mlp(kv, H, H); L.hv = cu.mlp(kv, H, H); L.hq = cu.mlp(H, H, H); L.no = cu.mlp(2 * H, H, H);
        L.xk = cu.mlp(kv, H, H); L.xv = cu.mlp(kv, H, hd); L.xq = cu.mlp(H, H, H);
        L.vn_f = cu.take((size_t)hd * cin); L.bn_g = cu.take(hd); L.bn_b = cu.take(hd); L.vn_d = cu.take((size_t)hd * cin);
    }
    m.inv = cu.mlp(S, S, SL);
    m.v1 = cu.lin(H, H);
    m.v2 = cu.lin(C, H);
    return cu.ok && cu.left == 0;
}

// ---- device image builder --------------------------------------------------------------------
struct Image {
    std::vector<float> d;
    size_t alloc(size_t n) {
        const size_t off = (d.size() + 63) & ~size_t(63);
        d.resize(off + n, 0.f);
        return off;
    }
    size_t put(const float *src, size_t n) { const size_t o = alloc(n); std::memcpy(&d[o], src, n * sizeof(float)); return o; }
};

struct DevMlp { size_t w1, b1, g, be, w2, b2; };            // raw row-major (VALU kernels)
struct DevMlpImg { size_t w1img, b1, g, be, w2img, b2; int nt2; size_t w1img6, w2img6, w1img16, w2img16; };   // MFMA A-fragment images (sm_node.h)
struct DevLayer {
    size_t pre_x2h, pre_h2x;          // images of [4H][H]: first-layer node blocks (k_i, k_j, v_i, v_j)
    size_t lin_img;                   // image of [8H][H]: pre_h2x of this layer followed by pre_x2h of the next
    size_t lin6_img, pre6_x2h;        // the same (and pre_x2h alone) as split bf16 images (node_linear6_kernel)
    size_t lin16_img, pre16_x2h;      // ... and as two-piece f16 images (node_linear16_kernel)
    size_t sk_x2h, sv_x2h, sk_h2x, sv_h2x;   // [H][SL] shape columns of the first layers
    size_t bk_x2h, bv_x2h, bk_h2x, bv_h2x;   // first-layer biases [H]
    DevMlpImg q_x2h, q_h2x, no;
    size_t blob_x2h, blob_h2x;        // fp32 edge kernels (sm_edge.h): both MLPs of a kernel in one LDS image
    size_t img_kx, img_vx, img_kh, img_vh;   // bf16-split phase kernels (sm_edge_bf16.h): one image per MLP
    size_t i16_kx, i16_vx, i16_kh, i16_vh;   // two-piece f16 images (sm_edge16.h)
    size_t st_kx, st_vx, st_kh, st_vh;       // streaming kernels (sm_edge_stream.h): producer parts (LDS images) ...
    size_t sw2_kx, sw2_vx, sw2_kh;           // ... and the second Linears as three bf16 pieces (consumers' registers)
    size_t sb2_vx;                           // bias of the x2h value MLP's second Linear [H]
    size_t vn_f, vn_d;                // original [heads][cin]
    size_t wf_x, wd_x, wf_o, wd_o, bn_g, bn_b;
};
struct DevModel {
    size_t tab[7];
    size_t te1w, te1b, te2w, te2b, embw, embb, embwT;
    DevMlp ew, inv;
    DevMlpImg vhead;             // Linear -> SSP -> Linear (second image padded to 16 rows)
    std::vector<DevLayer> layer;
};

// A-fragment image of W[rows][K] taken from src[r * ld + col0 + c]; rows padded with zeros to rows_pad
size_t pack_image(Image &im, const float *src, int rows, int rows_pad, int K, int ld, int col0) {
    const int ntk = K / 16;
    const size_t o = im.alloc((size_t)rows_pad * K);
    for (int t2 = 0; t2 < rows_pad / 16; ++t2)
        for (int t = 0; t < ntk; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t2 + (lane & 15), col = 16 * t + 4 * (lane >> 4) + r;
                    im.d[o + ((size_t)(t2 * ntk + t) * 64 + lane) * 4 + r] = row < rows ? src[(size_t)row * ld + col0 + col] : 0.f;
                }
    return o;
}
size_t put_padded(Image &im, const float *src, int n, int n_pad) {
    const size_t o = im.alloc(n_pad);
    std::memcpy(&im.d[o], src, n * sizeof(float));
    return o;
}
void split3_host(float w, uint16_t (&p)[3]);
size_t pack_linear6_image(Image &im, size_t src, int rows, int K);
size_t pack_linear16_image(Image &im, size_t src, int rows, int K);
DevMlpImg put_mlp_img(Image &im, const Mlp &m) {
    DevMlpImg d;
    const int r2 = (m.l2.out + 15) / 16 * 16;
    d.w1img = pack_image(im, m.l1.w, m.l1.out, m.l1.out, m.l1.in, m.l1.in, 0);
    d.b1 = im.put(m.l1.b, m.l1.out);
    d.g = m.g ? im.put(m.g, m.l1.out) : 0; d.be = m.be ? im.put(m.be, m.l1.out) : 0;
    d.w2img = pack_image(im, m.l2.w, m.l2.out, r2, m.l2.in, m.l2.in, 0);
    d.b2 = put_padded(im, m.l2.b, m.l2.out, r2);
    d.nt2 = r2 / 16;
    d.w1img6 = pack_linear6_image(im, d.w1img, m.l1.out, m.l1.in);
    d.w2img6 = pack_linear6_image(im, d.w2img, r2, m.l2.in);
    d.w1img16 = pack_linear16_image(im, d.w1img, m.l1.out, m.l1.in);
    d.w2img16 = pack_linear16_image(im, d.w2img, r2, m.l2.in);
    return d;
}

int head_of_row(int m, int nt) {     // value row 4g + r of the h2x edge kernel -> head index, -1 = padding
    const int g = m >> 2, r = m & 3;
    if (r >= nt / 2) return -1;
    return 2 * ((nt / 2) * (g & 1) + r) + (g >> 1);
}

// pack one edge MLP into the EdgeBlob image (see sm_edge.h)
void pack_edge_mlp(const Mlp &m, int H, int kv_in, bool perm_heads, float *wr, float *w2, float *gam, float *bet, float *b2) {
    const int NT = H / 16;
    const int nt2 = perm_heads ? 1 : NT;
    for (int t = 0; t < NT; ++t)
        for (int s = 0; s < 5; ++s)
            for (int lane = 0; lane < 64; ++lane)
                wr[(t * 5 + s) * 64 + lane] = m.l1.w[(size_t)(16 * t + (lane & 15)) * kv_in + 4 * s + (lane >> 4)];
    for (int t2 = 0; t2 < nt2; ++t2)
        for (int t = 0; t < NT; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    int row = 16 * t2 + (lane & 15);
                    if (perm_heads) row = head_of_row(lane & 15, NT);
                    const int col = 16 * t + 4 * (lane >> 4) + r;
                    w2[((t2 * NT + t) * 64 + lane) * 4 + r] = row < 0 ? 0.f : m.l2.w[(size_t)row * H + col];
                }
    std::memcpy(gam, m.g, H * sizeof(float));
    std::memcpy(bet, m.be, H * sizeof(float));
    for (int i = 0; i < nt2 * 16; ++i) {
        int row = i;
        if (perm_heads) row = head_of_row(i, NT);
        b2[i] = row < 0 ? 0.f : m.l2.b[row];
    }
}

// exact 3-way bf16 split of a float by truncation; returns the three 16-bit patterns
void split3_host(float w, uint16_t (&p)[3]) {
    float r = w;
    for (int i = 0; i < 3; ++i) {
        uint32_t u; std::memcpy(&u, &r, 4);
        const uint32_t hi = u & 0xFFFF0000u;
        p[i] = (uint16_t)(hi >> 16);
        float h; std::memcpy(&h, &hi, 4);
        r = r - h;
    }
}

// one edge MLP -> EdgePhaseImage (sm_edge_bf16.h)
template <int H>
size_t pack_phase_image(Image &im, const Mlp &m, int kv_in, bool perm_heads) {
    constexpr int NT = H / 16, NB = NT / 2;
    const int nt2 = perm_heads ? 1 : NT;
    const int G4 = (NT + 3) / 4;
    const bool bf1 = nt2 > 1;          // EdgePhaseImage::BF1
    const int o_wr = 0, o_g = o_wr + (bf1 ? 3 * NT * 256 : 5 * G4 * 256), o_b = o_g + H, o_b2 = o_b + H, o_w2 = o_b2 + nt2 * 16;
    const int total = o_w2 + 3 * nt2 * NB * 256;
    const size_t o = im.alloc(total);
    float *d = &im.d[o];
    for (int t = 0; t < NT; ++t)
        for (int s = 0; s < 5; ++s)
            for (int lane = 0; lane < 64; ++lane)
                if (!bf1) d[o_wr + ((s * G4 + t / 4) * 64 + lane) * 4 + (t & 3)] = m.l1.w[(size_t)(16 * t + (lane & 15)) * kv_in + 4 * s + (lane >> 4)];
    if (bf1) {
        uint32_t *wr = reinterpret_cast<uint32_t *>(d + o_wr);
        for (int t = 0; t < NT; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 4; ++q) {
                    uint16_t pc[2][3];
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * q + e;
                        split3_host(j < 5 ? m.l1.w[(size_t)(16 * t + (lane & 15)) * kv_in + 4 * j + (lane >> 4)] : 0.f, pc[e]);
                    }
                    for (int piece = 0; piece < 3; ++piece)
                        wr[((size_t)(piece * NT + t) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
                }
    }
    std::memcpy(d + o_g, m.g, H * sizeof(float));
    std::memcpy(d + o_b, m.be, H * sizeof(float));
    for (int i = 0; i < nt2 * 16; ++i) {
        const int row = perm_heads ? head_of_row(i, NT) : i;
        d[o_b2 + i] = row < 0 ? 0.f : m.l2.b[row];
    }
    uint32_t *w2 = reinterpret_cast<uint32_t *>(d + o_w2);
    for (int t2 = 0; t2 < nt2; ++t2)
        for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane) {
                const int mrow = lane & 15, g = lane >> 4;
                const int row = perm_heads ? head_of_row(mrow, NT) : 16 * t2 + mrow;
                for (int q = 0; q < 4; ++q) {
                    uint16_t pc[2][3];
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * q + e;
                        const int col = 16 * (2 * b + (j >> 2)) + 4 * g + (j & 3);
                        split3_host(row < 0 ? 0.f : m.l2.w[(size_t)row * H + col], pc[e]);
                    }
                    for (int piece = 0; piece < 3; ++piece)
                        w2[(((size_t)(piece * nt2 + t2) * NB + b) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
                }
            }
    return o;
}

// hi + lo two-piece f16 split (both round-to-nearest): the 16-bit patterns
void split2_host(float w, uint16_t (&p)[2]) {
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)(w - (float)h);
    std::memcpy(&p[0], &h, 2);
    std::memcpy(&p[1], &l, 2);
}

// one edge MLP -> EdgeImage16 (sm_edge16.h); returns the largest |hidden activation| the LayerNorm of this MLP can
// produce (the fp16 range check of the caller)
template <int H>
float pack_image16(Image &im, const Mlp &m, int kv_in, bool perm_heads, size_t &img) {
    constexpr int NT = H / 16, NB = NT / 2;
    const int nt2 = perm_heads ? 1 : NT;
    const int o_w1 = 0, o_w2 = 2 * NT * 192, o_g = o_w2 + 2 * nt2 * NB * 256, o_b = o_g + H, o_b2 = o_b + H;
    const int total = (o_b2 + nt2 * 16 + 255) / 256 * 256;
    img = im.alloc(total);
    uint32_t *d = reinterpret_cast<uint32_t *>(&im.d[img]);
    for (int t = 0; t < NT; ++t)
        for (int lane = 0; lane < 64; ++lane)
            for (int q = 0; q < 3; ++q) {
                uint16_t pc[2][2];
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * q + e;
                    split2_host(j < 5 ? m.l1.w[(size_t)(16 * t + (lane & 15)) * kv_in + 4 * j + (lane >> 4)] : 0.f, pc[e]);
                }
                for (int piece = 0; piece < 2; ++piece)
                    d[o_w1 + ((size_t)(piece * NT + t) * 3 + q) * 64 + lane] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
            }
    for (int t2 = 0; t2 < nt2; ++t2)
        for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane) {
                const int mrow = lane & 15, g = lane >> 4;
                const int row = perm_heads ? head_of_row(mrow, NT) : 16 * t2 + mrow;
                for (int q = 0; q < 4; ++q) {
                    uint16_t pc[2][2];
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * q + e;
                        const int col = 16 * (2 * b + (j >> 2)) + 4 * g + (j & 3);
                        split2_host(row < 0 ? 0.f : m.l2.w[(size_t)row * H + col], pc[e]);
                    }
                    for (int piece = 0; piece < 2; ++piece)
                        d[o_w2 + (((size_t)(piece * nt2 + t2) * NB + b) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
                }
            }
    float *pp = &im.d[img];
    std::memcpy(pp + o_g, m.g, H * sizeof(float));
    std::memcpy(pp + o_b, m.be, H * sizeof(float));
    for (int i = 0; i < nt2 * 16; ++i) {
        const int row = perm_heads ? head_of_row(i, NT) : i;
        pp[o_b2 + i] = row < 0 ? 0.f : m.l2.b[row];
    }
    float gmax = 0.f, bmax = 0.f;
    for (int i = 0; i < H; ++i) { gmax = std::max(gmax, std::fabs(m.g[i])); bmax = std::max(bmax, std::fabs(m.be[i])); }
    return gmax * std::sqrt((float)(H - 1)) + bmax;
}

// fp32 A-fragment image of W[rows][K] (pack_image) -> split bf16 image of node_linear6_kernel:
//   [(((ot * 3 + piece) * NB + b) * 64 + lane) * 4 + q] u32, element order of gemm_bf16x6
size_t pack_linear6_image(Image &im, size_t src, int rows, int K) {
    const int ntk = K / 16, NB = K / 32, nto = rows / 16;
    const size_t o = im.alloc((size_t)nto * 3 * NB * 256);
    uint32_t *w = reinterpret_cast<uint32_t *>(&im.d[o]);
    for (int ot = 0; ot < nto; ++ot)
        for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 4; ++q) {
                    uint16_t pc[2][3];
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * q + e, t = 2 * b + (j >> 2), r = j & 3;
                        split3_host(im.d[src + ((size_t)(ot * ntk + t) * 64 + lane) * 4 + r], pc[e]);
                    }
                    for (int piece = 0; piece < 3; ++piece)
                        w[(((size_t)(ot * 3 + piece) * NB + b) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
                }
    return o;
}

// fp32 A-fragment image -> two-piece f16 image of node_linear16_kernel / node_chain16_kernel:
//   [(((ot * 2 + piece) * NB + b) * 64 + lane) * 4 + q] u32
size_t pack_linear16_image(Image &im, size_t src, int rows, int K) {
    const int ntk = K / 16, NB = K / 32, nto = rows / 16;
    const size_t o = im.alloc((size_t)nto * 2 * NB * 256);
    uint32_t *w = reinterpret_cast<uint32_t *>(&im.d[o]);
    for (int ot = 0; ot < nto; ++ot)
        for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 4; ++q) {
                    uint16_t pc[2][2];
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * q + e, t = 2 * b + (j >> 2), r = j & 3;
                        split2_host(im.d[src + ((size_t)(ot * ntk + t) * 64 + lane) * 4 + r], pc[e]);
                    }
                    for (int piece = 0; piece < 2; ++piece)
                        w[(((size_t)(ot * 2 + piece) * NB + b) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
                }
    return o;
}

// one edge MLP -> the producer part of the streaming kernels (StreamMap<H, .>::P_*, sm_edge_stream.h): RBF block of the first
// Linear as three bf16 pieces (whole A fragments of the K = 32 step: centres 0..5 of lane group g in words 0..2, word 3 zero), gamma, beta, b2 and,
// for the heads-wide value MLP of h2x, the second Linear (rows = heads in natural order, padded to 16)
template <int H>
size_t pack_stream_part(Image &im, const Mlp &m, int kv_in, bool h2x_value) {
    constexpr int NT = H / 16, NB = NT / 2;
    using MX = StreamMap<H, false>; using MH = StreamMap<H, true>;
    const int total = h2x_value ? MH::PART_V : MX::PART_K;
    const size_t o = im.alloc(total);
    uint32_t *d = reinterpret_cast<uint32_t *>(&im.d[o]);
    std::memset(d, 0, (size_t)total * 4);
    for (int t = 0; t < NT; ++t)
        for (int lane = 0; lane < 64; ++lane)
            for (int q = 0; q < 3; ++q) {
                uint16_t pc[2][3];
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * q + e;
                    split3_host(j < 5 ? m.l1.w[(size_t)(16 * t + (lane & 15)) * kv_in + 4 * j + (lane >> 4)] : 0.f, pc[e]);
                }
                for (int piece = 0; piece < 3; ++piece)
                    d[MX::P_W1 + ((size_t)(piece * NT + t) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);      // (word 3 stays zero)
            }
    float *pp = &im.d[o];
    std::memcpy(pp + MX::P_G, m.g, H * sizeof(float));
    std::memcpy(pp + MX::P_B, m.be, H * sizeof(float));
    std::memcpy(pp + MX::P_B2, m.l2.b, std::min(m.l2.out, H) * sizeof(float));
    if (h2x_value) {
        for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = lane & 15, g = lane >> 4;
                for (int q = 0; q < 4; ++q) {
                    uint16_t pc[2][3];
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * q + e;
                        const int col = 16 * (2 * b + (j >> 2)) + 4 * g + (j & 3);
                        split3_host(row < m.l2.out ? m.l2.w[(size_t)row * H + col] : 0.f, pc[e]);
                    }
                    for (int piece = 0; piece < 3; ++piece)
                        d[MH::P_W2 + (((size_t)piece * NB + b) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
                }
            }
    }
    return o;
}

// second Linear [H][H] of an edge MLP as three bf16 pieces [3][NT][NB][64][4] u32, element order of gemm_bf16x6 (rows natural)
template <int H>
size_t pack_stream_w2(Image &im, const Mlp &m) {
    constexpr int NT = H / 16, NB = NT / 2;
    const size_t o = im.alloc((size_t)3 * NT * NB * 256);
    uint32_t *w2 = reinterpret_cast<uint32_t *>(&im.d[o]);
    for (int t2 = 0; t2 < NT; ++t2)
        for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = 16 * t2 + (lane & 15), g = lane >> 4;
                for (int q = 0; q < 4; ++q) {
                    uint16_t pc[2][3];
                    for (int e = 0; e < 2; ++e) {
                        const int j = 2 * q + e;
                        const int col = 16 * (2 * b + (j >> 2)) + 4 * g + (j & 3);
                        split3_host(m.l2.w[(size_t)row * H + col], pc[e]);
                    }
                    for (int piece = 0; piece < 3; ++piece)
                        w2[(((size_t)(piece * NT + t2) * NB + b) * 64 + lane) * 4 + q] = (uint32_t)pc[0][piece] | ((uint32_t)pc[1][piece] << 16);
                }
            }
    return o;
}

template <int H>
int build_layer_image(const shapemol_config &c, const HostLayer &L, Image &im, DevLayer &D, float &hid_max) {
    const int G = c.num_r_gaussian, SL = c.shape_latent_dim, S = c.shape_dim, hd = c.n_heads;
    const int kv = G + 2 * H + SL, cin = 1 + hd + S, NT = H / 16;
    bool contiguous = true;
    auto put_pre = [&](const Mlp &k, const Mlp &v) {      // 4 images of [H][H]: k_i, k_j, v_i, v_j column blocks
        const Mlp *src[4] = {&k, &k, &v, &v};
        size_t first = 0;
        for (int blk = 0; blk < 4; ++blk) {
            const size_t o = pack_image(im, src[blk]->l1.w, H, H, H, kv, G + (blk & 1) * H);
            if (blk == 0) first = o;
            else if (o != first + (size_t)blk * H * H) contiguous = false;     // images must be contiguous
        }
        return first;
    };
    auto put_scols = [&](const Mlp &m) {
        const size_t o = im.alloc((size_t)H * SL);
        for (int f = 0; f < H; ++f) std::memcpy(&im.d[o + (size_t)f * SL], m.l1.w + (size_t)f * kv + G + 2 * H, SL * sizeof(float));
        return o;
    };
    D.pre_x2h = put_pre(L.hk, L.hv); D.pre_h2x = put_pre(L.xk, L.xv);
    if (!contiguous) return fail("shapemol_create: packed first-layer images are not contiguous (H * H must be a multiple of 64)");
    D.sk_x2h = put_scols(L.hk); D.sv_x2h = put_scols(L.hv); D.sk_h2x = put_scols(L.xk); D.sv_h2x = put_scols(L.xv);
    D.bk_x2h = im.put(L.hk.l1.b, H); D.bv_x2h = im.put(L.hv.l1.b, H);
    D.bk_h2x = im.put(L.xk.l1.b, H); D.bv_h2x = im.put(L.xv.l1.b, H);
    D.q_x2h = put_mlp_img(im, L.hq); D.q_h2x = put_mlp_img(im, L.xq); D.no = put_mlp_img(im, L.no);
    {
        using B = EdgeBlob<H, false>;
        const size_t o = im.alloc(B::TOTAL); D.blob_x2h = o; float *b = &im.d[o];
        pack_edge_mlp(L.hk, H, kv, false, b + B::K_WR, b + B::K_W2, b + B::K_G, b + B::K_B, b + B::K_B2);
        pack_edge_mlp(L.hv, H, kv, false, b + B::V_WR, b + B::V_W2, b + B::V_G, b + B::V_B, b + B::V_B2);
    }
    {
        using B = EdgeBlob<H, true>;
        const size_t o = im.alloc(B::TOTAL); D.blob_h2x = o; float *b = &im.d[o];
        pack_edge_mlp(L.xk, H, kv, false, b + B::K_WR, b + B::K_W2, b + B::K_G, b + B::K_B, b + B::K_B2);
        pack_edge_mlp(L.xv, H, kv, true, b + B::V_WR, b + B::V_W2, b + B::V_G, b + B::V_B, b + B::V_B2);
    }
    D.img_kx = pack_phase_image<H>(im, L.hk, kv, false); D.img_vx = pack_phase_image<H>(im, L.hv, kv, false);
    D.img_kh = pack_phase_image<H>(im, L.xk, kv, false); D.img_vh = pack_phase_image<H>(im, L.xv, kv, true);
    hid_max = std::max(hid_max, pack_image16<H>(im, L.hk, kv, false, D.i16_kx));
    hid_max = std::max(hid_max, pack_image16<H>(im, L.hv, kv, false, D.i16_vx));
    hid_max = std::max(hid_max, pack_image16<H>(im, L.xk, kv, false, D.i16_kh));
    hid_max = std::max(hid_max, pack_image16<H>(im, L.xv, kv, true, D.i16_vh));
    D.st_kx = pack_stream_part<H>(im, L.hk, kv, false); D.st_vx = pack_stream_part<H>(im, L.hv, kv, false);
    D.st_kh = pack_stream_part<H>(im, L.xk, kv, false); D.st_vh = pack_stream_part<H>(im, L.xv, kv, true);
    D.sw2_kx = pack_stream_w2<H>(im, L.hk); D.sw2_vx = pack_stream_w2<H>(im, L.hv); D.sw2_kh = pack_stream_w2<H>(im, L.xk);
    D.sb2_vx = im.put(L.hv.l2.b, H);
    D.vn_f = im.put(L.vn_f, (size_t)hd * cin); D.vn_d = im.put(L.vn_d, (size_t)hd * cin);
    D.bn_g = im.put(L.bn_g, hd); D.bn_b = im.put(L.bn_b, hd);
    D.wf_x = im.alloc(hd); D.wd_x = im.alloc(hd); D.wf_o = im.alloc((size_t)hd * 16); D.wd_o = im.alloc((size_t)hd * 16);
    for (int ch = 0; ch < hd; ++ch) {
        im.d[D.wf_x + ch] = L.vn_f[(size_t)ch * cin];
        im.d[D.wd_x + ch] = L.vn_d[(size_t)ch * cin];
        for (int m = 0; m < 16; ++m) {
            const int hh = head_of_row(m, NT);
            im.d[D.wf_o + ch * 16 + m] = hh < 0 ? 0.f : L.vn_f[(size_t)ch * cin + 1 + hh];
            im.d[D.wd_o + ch * 16 + m] = hh < 0 ? 0.f : L.vn_d[(size_t)ch * cin + 1 + hh];
        }
    }
    return 0;
}

struct ProfRec { const char *name; hipEvent_t e0, e1; };

}  // namespace

struct shapemol_ctx {
    shapemol_config cfg{};
    int device = 0;
    int KP = 8;
    float *d_img = nullptr;
    float *ttab = nullptr;      // [T][D] time-embedding table (built once)
    float *etab = nullptr;      // [T][C][H] atom embedding of every (timestep, atom type) pair (built once)
    DevModel dm;
    // workspace
    int64_t capN = 0, capB = 0;
    std::vector<void *> allocs;
    int *mol_of = nullptr, *mol_off = nullptr, *t_mol = nullptr, *nbr = nullptr, *steps = nullptr;
    float *temb = nullptr, *inv = nullptr, *add0 = nullptr, *addp = nullptr, *ps = nullptr, *ew = nullptr;
    float *h_a = nullptr, *h_b = nullptr, *pre0 = nullptr, *preAB = nullptr, *q_x = nullptr, *q_h = nullptr, *att = nullptr, *o3 = nullptr, *pd = nullptr;
    ShapeTermArgs *prep_terms = nullptr; VnShapeArgs *prep_vn = nullptr; int n_prep_terms = 0;   // argument blocks of run_prep's two batched launches
    int2 *mol_span = nullptr;   // [N] molecule span of every atom
    float *xsum = nullptr;      // [N][3] per-atom sum of the h2x attention rows (folded coordinate update)
    float *part_rows = nullptr, *part_ms = nullptr;   // k > 16: rows [2N][H] and softmax state [2N][heads][2] of the half-atom tiles
    float *alpha = nullptr;     // [N*KP][2][NT] attention weights handed from the key phase to the value phase
    float *x_a = nullptr, *x_b = nullptr, *x_state = nullptr, *pred_pos = nullptr, *pred_v = nullptr;
    int64_t *v_state = nullptr;
    unsigned long long *stamps = nullptr;   // [1024][2] diagnostic clock stamps
    unsigned long long *kstamps = nullptr;  // per-wave phase stamps of ONE selected kernel launch (SM_STAMPS build)
    int kstamp_sel = -1;                    // which launch: 0 node_pre, 1 edge_x2h, 2 edge_h2x (layer 0)
    int stamp_on = 0;
    double *bn_acc = nullptr;
    int *status = nullptr;                  // [8] sticky error flags (StatusFlag), cleared at the start of _score/_sample
    ChainParams *chain_params = nullptr;    // per-chain parameters read by the posterior-step kernel
    // last evaluation (debug_read)
    int64_t lastN = 0, lastB = 0;
    const float *last_h = nullptr, *last_x = nullptr;
    // options
    int stop_layer = -1, edge_threads = 0 /* 0 = chosen per launch */, lin_waves = 16, edge_bf16 = 2, lin_bf16 = 1, chain_bf16 = 1, vn_fuse = 2;
    int vn_fold = 1;            // coordinate update of layer l in the prologue of the x2h kernel of layer l + 1 (needs max_mol_atoms)
    int max_mol_atoms = 0;      // largest molecule of the batches to come (option; 0 = unknown: no fold)
    int stream_whole_rounds = 0;   // streaming edge kernels: 1 = tiles per workgroup rounded up to whole rounds.  0 (default): as many workgroups as
                                   // the tiles give; the last round of an odd chunk has one tile (B = 256: 252 workgroups of 11 tiles instead of
                                   // 231 of 12 -- x2h 26.7 -> 25.6 us, h2x 24.7 -> 23.9; B = 1024, 43 instead of 44: unchanged)
    int lin_fuse = 0;           // 1: per-node products of the next attentions inside node_chain16_kernel instead of a node_linear
    int bn_eval = 0;            // 1: evaluation-mode batch-norm (running statistics, shapemol_set_bn_running) instead of the batch's
    float *bn_run = nullptr;    // [2][L][heads] running mean | running variance (device)
    double *bn_eval_acc = nullptr;   // [L][kBnReplicas][2][heads] sums that reproduce them for the current batch size
    int ddpm_fold = 1;          // 1: the last layer's coordinate update inside the DDPM kernel (chains without guidance)
    DdpmFold ddpm_vf{};         // ... handed from run_score to run_ddpm
    int graph_fuse = 1;         // 1: kNN graph + edge weights in one launch (graph_kernel) when max_mol_atoms <= kGraphCap is known
    int x2h_chain = 1;          // 1: x2h attention and the node stage of a layer in one launch (x2h_chain16_kernel) when every wave has one job
                                // launch (measured: 28.5 us against 15.3 + 11.1 us, eight dependent weight blocks per wave)
    int node_f16 = 0;           // 1: node kernels on two-piece f16 operands (sm_node16.h) instead of exactly split bf16 (sm_node.h) [default 0]
    int feat_f16 = 0;           // 1: "f16 features" -- matrix products on the leading f16 piece only (one product per term instead of
                                // three; accumulation, LayerNorm, softmax, coordinates fp32).  Reduced precision, NOT a parity mode
    int edge_tiles = -1;        // f16 edge kernels when the waves have several jobs: 0 = sliced launches of the one-job kernel,
                                // 1 = one looping launch (eight waves per workgroup, next job's rows prefetched), -1 = automatic (= 1) [default]
    float hid_max = 0.f;        // bound of the edge MLPs' hidden activations (LayerNorm outputs): must fit fp16 for edge_bf16 = 3
    int num_cu = 256;
    // point-cloud shape guidance (shapemol_set_guidance)
    double *g_cloud = nullptr; int64_t g_points = 0; double g_radius = 0.0; int g_grad_step = 0; const double *g_draws = nullptr;
    int first_step = 0;         // option "first_step": the next chains start at reverse step first_step (t = T-1-first_step)
    // diagnostic: neighbour lists pinned at given (reverse step, atom) pairs (shapemol_set_knn_pins)
    int *pin_off = nullptr, *pin_atom = nullptr, *pin_nbr = nullptr; int64_t n_pins = 0; int pin_steps = 0, pin_k = 0;
    // profiling
    bool prof_on = false;
    std::vector<ProfRec> prof;
    hipEvent_t cur_e0 = nullptr, cur_e1 = nullptr;     // event pair of the launch being issued
    // graph cache
    hipGraphExec_t gexec = nullptr, gexec_u = nullptr;     // one step / kGraphUnroll steps
    int64_t n_captures = 0;                                // graph captures so far (debug_read "captures")
    // the captured step depends on the batch geometry only: seed, noise and trajectory pointers live in chain_params
    // what a captured step depends on besides the options (which drop the graphs when set): sizes, guidance, and the two
    // launch decisions taken from the max_mol_atoms hint (folded coordinate update, fused graph kernel)
    struct GraphKey { int64_t N = 0, B = 0; int guided = 0, fold = 0, gfuse = 0;
                      bool operator==(const GraphKey &o) const { return N == o.N && B == o.B && guided == o.guided && fold == o.fold && gfuse == o.gfuse; } } gkey{};
    hipStream_t gstream = nullptr; bool gstream_set = false;     // the stream the executables were last launched on
    void drop_graphs() {      // a replay may still be in flight: drain it before destroying the executables (only the stream the
        if (!gexec && !gexec_u) return;       // graphs ran on: another context's chain may be running beside, and must not be waited for)
        if (!gstream_set || hipStreamSynchronize(gstream) != hipSuccess) hipDeviceSynchronize();
        if (gexec) { hipGraphExecDestroy(gexec); gexec = nullptr; }
        if (gexec_u) { hipGraphExecDestroy(gexec_u); gexec_u = nullptr; }
    }
    const float *P(size_t off) const { return d_img + off; }
};

namespace {

// Profiling mode (shapemol_profile_begin): every kernel is launched with hipExtLaunchKernelGGL and a start/stop event
// pair that takes the begin/end timestamps of the DISPATCH ITSELF (what rocprofv3's kernel trace reports), not of
// marker packets around it.
#define SMK(kern, grid, block, shm, stream, ...)                                                                    \
    do {                                                                                                             \
        if (c->prof_on) hipExtLaunchKernelGGL(kern, grid, block, shm, stream, c->cur_e0, c->cur_e1, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, shm, stream, __VA_ARGS__);                                        \
    } while (0)

template <typename F>
int launch(shapemol_ctx *c, const char *name, hipStream_t s, F &&f) {
    (void)s;
    if (c->prof_on) {
        ProfRec r{name, nullptr, nullptr};
        HIPCHK(hipEventCreate(&r.e0)); HIPCHK(hipEventCreate(&r.e1));
        c->cur_e0 = r.e0; c->cur_e1 = r.e1;
        f();
        c->prof.push_back(r);
    } else {
        f();
    }
    HIPCHK(hipGetLastError());
    return 0;
}
#define LAUNCH(name, ...) do { if (launch(c, name, s, [&]() { __VA_ARGS__; })) return 1; } while (0)

int ensure_workspace(shapemol_ctx *c, int64_t N, int64_t B) {
    if (N <= c->capN && B <= c->capB) return 0;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    for (void *p : c->allocs) hipFree(p);
    c->allocs.clear();
    c->drop_graphs();
    const int64_t capN = std::max<int64_t>(N, c->capN), capB = std::max<int64_t>(B, c->capB);
    const shapemol_config &g = c->cfg;
    const int H = g.hidden_dim, L = g.num_layers, hd = g.n_heads;
    auto A = [&](auto **p, size_t n) -> int {
        void *q = nullptr;
        HIPCHK(hipMalloc(&q, n * sizeof(**p) + 256));
        HIPCHK(hipMemset(q, 0, n * sizeof(**p) + 256));
        c->allocs.push_back(q);
        *p = reinterpret_cast<std::remove_reference_t<decltype(*p)>>(q);
        return 0;
    };
    if (A(&c->mol_of, capN) || A(&c->mol_off, capB + 1) || A(&c->t_mol, capB) || A(&c->nbr, capN * c->KP) ||
        A(&c->steps, 4) || A(&c->temb, capB * g.time_emb_dim) || A(&c->inv, capB * g.shape_latent_dim) ||
        A(&c->add0, (size_t)capB * 4 * H) || A(&c->addp, (size_t)L * capB * 8 * H) || A(&c->ps, (size_t)L * capB * 2 * hd * 3) || A(&c->ew, capN * c->KP) ||
        A(&c->h_a, capN * H) || A(&c->h_b, capN * H) || A(&c->pre0, capN * 4 * H) || A(&c->preAB, capN * 8 * H) || A(&c->q_x, capN * H) || A(&c->q_h, capN * H) ||
        A(&c->att, capN * H) || A(&c->o3, capN * 48) || A(&c->xsum, capN * 3) || A(&c->mol_span, capN) || A(&c->alpha, capN * c->KP * 2 * (H / 16)) || A(&c->pd, capN * hd * 6) || A(&c->x_a, capN * 3) ||
        A(&c->x_b, capN * 3) || A(&c->x_state, capN * 3) || A(&c->pred_pos, capN * 3) ||
        A(&c->pred_v, capN * g.num_classes) || A(&c->v_state, capN) || A(&c->stamps, 2048) || A(&c->kstamps, 8 * 16 * 4096) || A(&c->bn_acc, (size_t)L * kBnReplicas * 2 * hd + L + 1) ||
        (c->KP > 16 && (A(&c->part_rows, (size_t)2 * capN * H) || A(&c->part_ms, (size_t)2 * capN * hd * 2))) ||
        A(&c->status, 8) || A(&c->chain_params, 1) || A(&c->prep_terms, 2 * L + 1) || A(&c->prep_vn, L))
        return 1;
    c->capN = capN; c->capB = capB;
    {   // argument blocks of the batched prep launches (they point into the workspace just allocated)
        const int SL = g.shape_latent_dim, S = g.shape_dim;
        std::vector<ShapeTermArgs> terms;
        std::vector<VnShapeArgs> vns;
        const DevLayer &D0 = c->dm.layer[0];
        terms.push_back(ShapeTermArgs{c->inv, c->P(D0.sk_x2h), c->P(D0.bk_x2h), c->P(D0.sv_x2h), c->P(D0.bv_x2h), c->add0, SL, H, SL, 4 * H});
        for (int l = 0; l < L; ++l) {
            const DevLayer &D = c->dm.layer[l];
            float *addp = c->addp + (size_t)l * c->capB * 8 * H;      // [B][8H]: this layer's h2x | the next layer's x2h
            terms.push_back(ShapeTermArgs{c->inv, c->P(D.sk_h2x), c->P(D.bk_h2x), c->P(D.sv_h2x), c->P(D.bv_h2x), addp, SL, H, SL, 8 * H});
            if (l + 1 < L) {
                const DevLayer &Dn = c->dm.layer[l + 1];
                terms.push_back(ShapeTermArgs{c->inv, c->P(Dn.sk_x2h), c->P(Dn.bk_x2h), c->P(Dn.sv_x2h), c->P(Dn.bv_x2h), addp + 4 * H, SL, H, SL, 8 * H});
            }
            vns.push_back(VnShapeArgs{nullptr, c->P(D.vn_f), c->P(D.vn_d), c->ps + (size_t)l * c->capB * 2 * hd * 3, S, hd});
        }
        c->n_prep_terms = (int)terms.size();
        HIPCHK(hipMemcpy(c->prep_terms, terms.data(), terms.size() * sizeof(ShapeTermArgs), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->prep_vn, vns.data(), vns.size() * sizeof(VnShapeArgs), hipMemcpyHostToDevice));
    }
    return 0;
}

static int stream_chunk(const shapemol_ctx *c, int n_atoms);
constexpr int kVnFoldBytes = kVnFoldCap * 3 * 4 + 32 * 8;      // LDS of the folded coordinate update: table + batch sums

// can the coordinate update of a layer be folded into the next x2h kernel?  f16 one-job (or sliced) edge kernels, the
// VN-linear + statistics epilogue in h2x, and every workgroup's molecule span inside the LDS table
bool vn_fold_ok(const shapemol_ctx *c, int n_atoms) {
    if (!c->vn_fold || (c->edge_bf16 != 3 && c->edge_bf16 != 2) || c->KP > 16 || c->vn_fuse != 2 || c->max_mol_atoms <= 0) return false;
    if (c->edge_bf16 == 2) return stream_chunk(c, n_atoms) * (16 / c->KP) + 2 * (c->max_mol_atoms - 1) <= kVnFoldCap;      // (KP <= 16 here)
    const int apj = 16 / c->KP, njobs = (n_atoms + apj - 1) / apj;
    const int waves = std::max(4, std::min(12, (njobs + c->num_cu - 1) / c->num_cu));
    const int grid = std::max(1, std::min(c->num_cu, (njobs + waves - 1) / waves));
    if (c->edge_threads > 0) return false;                                 // wave-count sweeps: keep the plain path
    const int tiles_mode = c->edge_tiles >= 0 ? c->edge_tiles : 1;        // (as launch_edge16)
    if (njobs > grid * waves && tiles_mode == 1) {                        // looping launches: a workgroup owns `chunk` consecutive jobs
        const int lwv = 8;
        const int lgrid = std::max(1, std::min(c->num_cu, (njobs + lwv - 1) / lwv)), chunk = (njobs + lgrid - 1) / lgrid;
        return chunk * apj + 2 * (c->max_mol_atoms - 1) <= kVnFoldCap;
    }
    const int waves_max = std::max(waves, c->cfg.hidden_dim / 16);          // the fused x2h + node-stage launch never uses fewer
    return waves_max * apj + 2 * (c->max_mol_atoms - 1) <= kVnFoldCap;
}

template <int H>
int set_edge_attr(int KP) {
    const int b0 = EdgeBlob<H, false>::TOTAL * 4, b1 = EdgeBlob<H, true>::TOTAL * 4;
#define SETATTR(K)                                                                                                    \
    HIPCHK(hipFuncSetAttribute((const void *)edge_attention_kernel<H, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, b0)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge_attention_kernel<H, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, b1));
#define SETATTR3(K)                                                                                                   \
    HIPCHK(hipFuncSetAttribute((const void *)edge_fused_kernel<H, K, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, EdgePhaseImage<H, H / 16>::TOTAL * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge_fused_kernel<H, K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, EdgePhaseImage<H, H / 16>::TOTAL * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge_fused_kernel<H, K, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgePhaseImage<H, H / 16>::TOTAL + EdgePhaseImage<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge_fused_kernel<H, K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgePhaseImage<H, H / 16>::TOTAL + EdgePhaseImage<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4));
    HIPCHK(hipFuncSetAttribute((const void *)node_prologue6_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * Chain6Lds<H>::FRAG * 16 + Chain6Lds<H>::PRE * 4)));
    HIPCHK(hipFuncSetAttribute((const void *)node_chain6_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Chain6Lds<H>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)node_linear6_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, kLin6Chunk * 3 * H * 32));
    HIPCHK(hipFuncSetAttribute((const void *)node_prologue16_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * Chain16Lds<H>::FRAG * 16 + Chain16Lds<H>::PRE * 4)));
    HIPCHK(hipFuncSetAttribute((const void *)node_chain16_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Chain16Lds<H>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)node_linear16_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, kLin16Chunk * 2 * H * 32));
    HIPCHK(hipFuncSetAttribute((const void *)node_prologue16_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * Chain16Lds<H>::FRAG * 16 + Chain16Lds<H>::PRE * 4)));
    HIPCHK(hipFuncSetAttribute((const void *)node_chain16_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Chain16Lds<H>::BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)node_linear16_kernel<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLin16Chunk * 2 * H * 32));
#define SETATTR4(K)                                                                                                   \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4 + kVnFoldBytes)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4 + kVnFoldBytes)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4 + kVnFoldBytes)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4 + kVnFoldBytes)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)x2h_chain16_kernel<H, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4 + kVnFoldBytes)); \
    HIPCHK(hipFuncSetAttribute((const void *)x2h_chain16_kernel<H, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4 + kVnFoldBytes));
#define SETATTR6(K)                                                                                                   \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EdgeImage16<H, H / 16>::TOTAL * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_kernel<H, K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge16_loop_kernel<H, K, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, 1>::TOTAL) * 4 + vn_red_doubles(12, H / 8) * 8 + 12 * 96 * 4));
#define SETATTR7(K)                                                                                                   \
    HIPCHK(hipFuncSetAttribute((const void *)edge_stream_kernel<H, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, StreamMap<H, false>::O_TAIL * 4 + kVnFoldBytes)); \
    HIPCHK(hipFuncSetAttribute((const void *)edge_stream_kernel<H, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, StreamMap<H, true>::O_TAIL * 4 + (H / 16 + kStreamProducers) * 64 * 2 * 8));
    if (KP == 8) { SETATTR(8) SETATTR3(8) SETATTR4(8) SETATTR7(8) } else if (KP == 16) { SETATTR(16) SETATTR3(16) SETATTR4(16) SETATTR7(16) } else { SETATTR6(32) SETATTR7(32) }
#undef SETATTR7
#undef SETATTR6
#undef SETATTR4
#undef SETATTR3
#undef SETATTR
    return 0;
}

// Waves per workgroup of the single-tile edge kernels: one job per wave while the jobs fit (the launch then lasts
// one job latency), on as many CUs as possible: ceil(jobs / CUs) waves, at least 4 and at most 12 (the 168-VGPR budget).
static int edge_waves_for(const shapemol_ctx *c, int njobs) {
    if (c->edge_threads > 0) return c->edge_threads / 64;
    return std::max(4, std::min(12, (njobs + c->num_cu - 1) / c->num_cu));
}

template <int H, bool H2X>
int launch_edge(shapemol_ctx *c, hipStream_t s, const EdgeArgs &a) {     // fp32-MFMA edge kernel (option edge_bf16 = 0; k > 16)
    const int KP = c->KP;
    const int apj = KP >= 16 ? 1 : 16 / KP;
    const int njobs = (a.n_atoms + apj - 1) / apj;
    const int waves = c->edge_threads > 0 ? c->edge_threads / 64 : 12;
    const int grid = std::max(1, std::min(c->num_cu, njobs));
    const size_t shm = EdgeBlob<H, H2X>::TOTAL * sizeof(float);
    const char *nm = H2X ? "edge_h2x" : "edge_x2h";
    if (KP == 8) LAUNCH(nm, SMK((edge_attention_kernel<H, 8, H2X>), dim3(grid), dim3(waves * 64), shm, s, a));
    else if (KP == 16) LAUNCH(nm, SMK((edge_attention_kernel<H, 16, H2X>), dim3(grid), dim3(waves * 64), shm, s, a));
    else return fail("k > 16 runs on the two-piece f16 edge kernels only (option edge_bf16 = 3)");
    return 0;
}

template <int H, bool H2X>
int launch_fused(shapemol_ctx *c, hipStream_t s, const EdgeFusedArgs &a) {
    const int KP = c->KP, apj = 16 / KP;
    const int njobs = (a.n_atoms + apj - 1) / apj, waves = edge_waves_for(c, njobs);
    const int grid = std::max(1, std::min(c->num_cu, (njobs + waves - 1) / waves));
    const size_t shm = (EdgePhaseImage<H, H / 16>::TOTAL + (H2X ? EdgePhaseImage<H, 1>::TOTAL : 0)) * sizeof(float)
                       + (H2X ? (size_t)vn_red_doubles(waves, H / 8) * 8 + (size_t)waves * apj * 48 * 4 : 0);   // + reduction scratch and attention rows of the fused coordinate update
    const char *nm = H2X ? "edge_h2x" : "edge_x2h";
    const bool one = njobs <= grid * waves;      // every wave has at most one job: straight-line instantiation
    if (KP == 8) {
        if (one) LAUNCH(nm, SMK((edge_fused_kernel<H, 8, H2X, true>), dim3(grid), dim3(waves * 64), shm, s, a));
        else LAUNCH(nm, SMK((edge_fused_kernel<H, 8, H2X, false>), dim3(grid), dim3(waves * 64), shm, s, a));
    } else {
        if (one) LAUNCH(nm, SMK((edge_fused_kernel<H, 16, H2X, true>), dim3(grid), dim3(waves * 64), shm, s, a));
        else LAUNCH(nm, SMK((edge_fused_kernel<H, 16, H2X, false>), dim3(grid), dim3(waves * 64), shm, s, a));
    }
    return 0;
}

template <int H, bool H2X>
int launch_edge16(shapemol_ctx *c, hipStream_t s, const Edge16Args &a) {
    const int KP = c->KP;
    const char *nm = H2X ? "edge_h2x" : "edge_x2h";
    const int tiles_mode = c->edge_tiles >= 0 ? c->edge_tiles : 1;      // -1 = automatic: looping launches
#define EDGE_DISPATCH(KERNEL, ...)                                                                  \
    do {                                                                                            \
        if (c->feat_f16) {                                                                          \
            if (KP == 8) LAUNCH(nm, SMK((KERNEL<H, 8, H2X, true>), __VA_ARGS__));                   \
            else if (KP == 16) LAUNCH(nm, SMK((KERNEL<H, 16, H2X, true>), __VA_ARGS__));            \
            else LAUNCH(nm, SMK((KERNEL<H, 32, H2X, true>), __VA_ARGS__));                          \
        } else if (KP == 8) LAUNCH(nm, SMK((KERNEL<H, 8, H2X>), __VA_ARGS__));                      \
        else if (KP == 16) LAUNCH(nm, SMK((KERNEL<H, 16, H2X>), __VA_ARGS__));                      \
        else LAUNCH(nm, SMK((KERNEL<H, 32, H2X>), __VA_ARGS__));                                    \
    } while (0)
    {
        // a job = one 16-slot tile: 16 / KP centre atoms (k <= 16) or half an atom (k > 16: two tiles per atom, merged afterwards)
        const int apj = KP >= 16 ? 1 : 16 / KP;
        const int njobs = KP > 16 ? 2 * a.n_atoms : (a.n_atoms + apj - 1) / apj, waves = edge_waves_for(c, njobs);
        const int grid = std::max(1, std::min(c->num_cu, (njobs + waves - 1) / waves));
        const bool one = njobs <= grid * waves;      // every wave has at most one job: straight-line instantiation
        auto shm_for = [&](int w) {
            return (EdgeImage16<H, H / 16>::TOTAL + EdgeImage16<H, (H2X ? 1 : H / 16)>::TOTAL) * sizeof(float)
                   + (H2X ? (size_t)vn_red_doubles(w, H / 8) * 8 + (size_t)w * apj * 48 * 4 : (a.vf.enable ? kVnFoldBytes : 0));
        };
        if (!one && tiles_mode == 1) {
            // looping launch: eight waves per workgroup (two per SIMD, 256 VGPRs: a job's state plus the next job's gathered rows
            // without spilling), every workgroup owns `chunk` consecutive jobs
            const int lw = c->edge_threads > 0 ? std::min(8, c->edge_threads / 64) : 8;
            const int lgrid = std::max(1, std::min(c->num_cu, (njobs + lw - 1) / lw));
            Edge16Args b = a;
            b.job_base = 0; b.job_end = njobs; b.nwave = lw; b.chunk = (njobs + lgrid - 1) / lgrid;
            const int g2 = (njobs + b.chunk - 1) / b.chunk;
            EDGE_DISPATCH(edge16_loop_kernel, dim3(g2), dim3(lw * 64), shm_for(lw), s, b);
            return 0;
        }
        // one launch of the straight-line instantiation, or (larger batches, edge_tiles = 0) several, each over a slice of
        // grid x waves jobs (no spills, full overlap inside a launch; the image fill is paid per slice)
        // ... of equal size: the jobs are spread evenly over the fewest launches that can hold them (a last slice that
        // is nearly empty costs a full launch floor)
        int ws = waves, gs = grid;
        if (!one && c->edge_threads == 0) {      // (an explicit edge_waves option keeps its wave count)
            const int nsl = (njobs + c->num_cu * 12 - 1) / (c->num_cu * 12), target = (njobs + nsl - 1) / nsl;
            ws = std::max(4, std::min(12, (target + c->num_cu - 1) / c->num_cu));
            gs = std::max(1, std::min(c->num_cu, (target + ws - 1) / ws));
        }
        const int per = gs * ws;
        for (int base = 0; base < njobs; base += per) {
            Edge16Args b = a;
            b.job_base = base; b.job_end = std::min(njobs, base + per); b.nwave = ws;
            const int g2 = std::max(1, std::min(gs, (b.job_end - base + ws - 1) / ws));
            EDGE_DISPATCH(edge16_kernel, dim3(g2), dim3(ws * 64), shm_for(ws), s, b);
        }
    }
#undef EDGE_DISPATCH
    return 0;
}

// streaming edge kernels (sm_edge_stream.h; option edge_bf16 = 2): consecutive tiles per workgroup -- every CU one workgroup;
// half-atom tiles (k > 16) in pairs
static int stream_jobs(const shapemol_ctx *c, int n_atoms) {      // a job = one 16-slot tile: 16 / KP atoms, or half an atom (k > 16)
    return c->KP > 16 ? 2 * n_atoms : (n_atoms + 16 / c->KP - 1) / (16 / c->KP);
}
static int stream_chunk(const shapemol_ctx *c, int n_atoms) {
    const int njobs = stream_jobs(c, n_atoms);
    const int per_cu = (njobs + c->num_cu - 1) / c->num_cu;
    if (!c->stream_whole_rounds && c->KP <= 16) return std::max(1, per_cu);
    return std::max(kStreamTPR, (per_cu + kStreamTPR - 1) / kStreamTPR * kStreamTPR);
}

template <int H, bool H2X>
int launch_stream(shapemol_ctx *c, hipStream_t s, EdgeStreamArgs a) {
    const int KP = c->KP, njobs = stream_jobs(c, a.n_atoms);
    a.chunk = stream_chunk(c, a.n_atoms);
    const int grid = (njobs + a.chunk - 1) / a.chunk;
    using M = StreamMap<H, H2X>;
    constexpr int NWAVE = H / 16 + kStreamProducers;
    const size_t shm = (size_t)M::O_TAIL * 4 + (H2X ? (size_t)NWAVE * 64 * 2 * 8 : (a.vf.enable ? (size_t)kVnFoldBytes : 0));
    const char *nm = H2X ? "edge_h2x" : "edge_x2h";
    if (KP == 8) LAUNCH(nm, SMK((edge_stream_kernel<H, 8, H2X>), dim3(grid), dim3(NWAVE * 64), shm, s, a));
    else if (KP == 16) LAUNCH(nm, SMK((edge_stream_kernel<H, 16, H2X>), dim3(grid), dim3(NWAVE * 64), shm, s, a));
    else LAUNCH(nm, SMK((edge_stream_kernel<H, 32, H2X>), dim3(grid), dim3(NWAVE * 64), shm, s, a));
    return 0;
}

// k > 16: merge the two half-atom tiles of every atom (combine32_kernel) into `out` ([N][H] or [N][48])
template <bool H2X>
int launch_combine32(shapemol_ctx *c, hipStream_t s, float *out, int n_atoms) {
    Combine32Args ca{c->part_rows, c->part_ms, out, n_atoms, c->cfg.n_heads, c->cfg.hidden_dim / 16};
    const int items = n_atoms * (H2X ? 48 : c->cfg.hidden_dim / 4);
    LAUNCH("edge_combine", SMK(combine32_kernel<H2X>, dim3((items + 255) / 256), dim3(256), 0, s, ca));
    return 0;
}

// x2h attention + node stage in one launch: possible when the f16 kernels are in use, every wave gets one job in a single
// launch, and a workgroup has at least H / 16 waves (the node stage's output blocks)
template <int H>
bool x2h_chain_ok(const shapemol_ctx *c, int n_atoms) {
    if (!c->x2h_chain || c->edge_bf16 != 3 || c->KP > 16 || !c->chain_bf16 || !c->node_f16 || c->lin_fuse || c->edge_threads > 0) return false;
    const int apj = 16 / c->KP, njobs = (n_atoms + apj - 1) / apj;
    const int waves = std::max(H / 16, edge_waves_for(c, njobs));
    const int grid = std::max(1, std::min(c->num_cu, (njobs + waves - 1) / waves));
    // one launch only: in slices (larger batches) the fused kernel loses to the separate ones, whose node stage then has
    // enough atoms per launch to run at its throughput (B = 1024: 523 against 538 molecules/s)
    return waves <= 12 && waves * apj <= CHAIN_COLS * 16 && njobs <= grid * waves && Chain16Lds<H>::BYTES <= 2 * EdgeImage16<H, H / 16>::TOTAL * 4;
}

template <int H>
int launch_x2h_chain(shapemol_ctx *c, hipStream_t s, const Edge16Args &a, const NodeChainArgs &na) {
    const int KP = c->KP, apj = 16 / KP;
    const int njobs = (a.n_atoms + apj - 1) / apj, waves = std::max(H / 16, edge_waves_for(c, njobs));
    const int grid = std::max(1, std::min(c->num_cu, (njobs + waves - 1) / waves));
    const size_t shm = 2 * EdgeImage16<H, H / 16>::TOTAL * sizeof(float) + (a.vf.enable ? kVnFoldBytes : 0);
    Edge16Args b = a;
    b.job_base = 0; b.job_end = njobs; b.nwave = waves;
    if (c->feat_f16) {
        if (KP == 8) LAUNCH("edge_x2h_chain", SMK((x2h_chain16_kernel<H, 8, true>), dim3(grid), dim3(waves * 64), shm, s, b, na, c->status + ST_RANGE));
        else LAUNCH("edge_x2h_chain", SMK((x2h_chain16_kernel<H, 16, true>), dim3(grid), dim3(waves * 64), shm, s, b, na, c->status + ST_RANGE));
    } else if (KP == 8) LAUNCH("edge_x2h_chain", SMK((x2h_chain16_kernel<H, 8>), dim3(grid), dim3(waves * 64), shm, s, b, na, c->status + ST_RANGE));
    else LAUNCH("edge_x2h_chain", SMK((x2h_chain16_kernel<H, 16>), dim3(grid), dim3(waves * 64), shm, s, b, na, c->status + ST_RANGE));
    return 0;
}

template <int H, int KT>
int launch_mlp2(shapemol_ctx *c, hipStream_t s, const char *name, const DevMlpImg &m, const float *in0, const float *in1,
                int mode, const float *resid, float *out, int ld_out, int n_store, int n_atoms) {
    NodeMlpArgs a{};
    a.in0 = in0; a.in1 = in1; a.w1img = c->P(m.w1img); a.b1 = c->P(m.b1); a.ln_g = c->P(m.g); a.ln_b = c->P(m.be);
    a.w2img = c->P(m.w2img); a.b2 = c->P(m.b2); a.resid = resid; a.out = out; a.ld_out = ld_out; a.n_store = n_store;
    a.mode = mode; a.nt2 = m.nt2; a.n_atoms = n_atoms;
    const int n_ct = (n_atoms + 15) / 16, teams = NodeMlpLds<H>::TEAMS;
    LAUNCH(name, SMK((node_mlp2_kernel<H, KT>), dim3((n_ct + teams - 1) / teams), dim3(kNodeThreads), 0, s, a));
    return 0;
}

template <int H>
int launch_linear(shapemol_ctx *c, hipStream_t s, const char *name, const float *in, const float *wimg, const float *wimg6, const float *wimg16, const float *add_mol,
                  int ld_add, float *out, int ld_out, int n_out_tiles, int n_atoms, unsigned long long *stamps) {
    const int nwave = c->lin_waves;
    const int n_ct = (n_atoms + 15) / 16, ogroups = (n_out_tiles + nwave - 1) / nwave;
    const int want_groups = std::max(1, c->num_cu / ogroups);
    const int tpg = std::max(1, (n_ct + want_groups - 1) / want_groups);
    const int agroups = (n_ct + tpg - 1) / tpg;
    const bool f16 = c->lin_bf16 && c->node_f16;
    NodeLinArgs a{in, f16 ? wimg16 : (c->lin_bf16 ? wimg6 : wimg), add_mol, c->mol_of, out, n_atoms, n_out_tiles, tpg, ld_add, ld_out, stamps, nwave};
    if (f16) {
        const size_t shm = (size_t)std::min(tpg, kLin16Chunk) * 2 * H * 32;
        if (c->feat_f16) LAUNCH(name, SMK((node_linear16_kernel<H, true>), dim3(ogroups * agroups), dim3(nwave * 64), shm, s, a, c->status + ST_RANGE));
        else LAUNCH(name, SMK(node_linear16_kernel<H>, dim3(ogroups * agroups), dim3(nwave * 64), shm, s, a, c->status + ST_RANGE));
    } else if (c->lin_bf16) {
        const size_t shm = (size_t)std::min(tpg, kLin6Chunk) * 3 * H * 32;
        LAUNCH(name, SMK(node_linear6_kernel<H>, dim3(ogroups * agroups), dim3(nwave * 64), shm, s, a));
    } else {
        LAUNCH(name, SMK(node_linear_kernel<H>, dim3(ogroups * agroups), dim3(nwave * 64), 0, s, a));
    }
    return 0;
}

NodeFollow follow_of(const shapemol_ctx *c, const DevMlpImg &m, int mode, float *out, int ld_out, int n_store) {
    NodeFollow f{};
    f.w1img = c->P(m.w1img); f.b1 = c->P(m.b1); f.ln_g = c->P(m.g); f.ln_b = c->P(m.be);
    f.w1img6 = c->P(c->node_f16 ? m.w1img16 : m.w1img6); f.w2img6 = c->P(c->node_f16 ? m.w2img16 : m.w2img6);
    f.w2img = c->P(m.w2img); f.b2 = c->P(m.b2); f.out = out; f.ld_out = ld_out; f.n_store = n_store; f.mode = mode; f.nt2 = m.nt2;
    return f;
}

// Step-invariant per-batch quantities (molecule index, invariant shape embedding, shape terms)
// Diagnostic (shapemol_set_knn_pins): overwrite the neighbour rows of the atoms pinned at the current reverse step with the
// given lists.  off [steps + 1] CSR offsets per reverse step, atom [n], pnbr [n][k].
__global__ void knn_pin_kernel(const int *off, int n_steps, const int *atom, const int *pnbr, int k, int kp, const int *step_cur, int n_atoms, int *nbr) {
    const int step = *step_cur;
    if (step < 0 || step >= n_steps) return;
    const int lo = off[step], hi = off[step + 1];
    for (int e = lo + blockIdx.x * blockDim.x + threadIdx.x; e < hi; e += gridDim.x * blockDim.x) {
        const int i = atom[e];
        if (i < 0 || i >= n_atoms) continue;
        for (int sl = 0; sl < kp; ++sl) nbr[(size_t)i * kp + sl] = sl < k ? pnbr[(size_t)e * k + sl] : -1;
    }
}

template <int H>
int run_prep(shapemol_ctx *c, hipStream_t s, const int64_t *d_batch, int64_t N, int64_t B, const float *d_shape) {
    const shapemol_config &g = c->cfg;
    const int L = g.num_layers, hd = g.n_heads, SL = g.shape_latent_dim, S = g.shape_dim;
    LAUNCH("prep", SMK(mol_index_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_batch, (int)N, (int)B, c->mol_of, c->mol_off, c->status));
    LAUNCH("prep", SMK(mol_span_kernel, dim3((N + 255) / 256), dim3(256), 0, s, c->mol_of, c->mol_off, (int)N, c->mol_span));
    ShapeInvArgs si{d_shape, c->P(c->dm.inv.w1), c->P(c->dm.inv.b1), c->P(c->dm.inv.g), c->P(c->dm.inv.be),
                    c->P(c->dm.inv.w2), c->P(c->dm.inv.b2), c->inv, S, SL};
    LAUNCH("prep", SMK(shape_invariant_kernel, dim3(B), dim3(64), 0, s, si));
    // every step-invariant shape term (layer-0 x2h; per layer h2x | next x2h) and the shape part of every VN-linear: two launches
    if (SL == 32) LAUNCH("prep", SMK(shape_term_multi_tiled_kernel<32>, dim3((unsigned)((B + kShapeTermMols - 1) / kShapeTermMols), (unsigned)c->n_prep_terms), dim3(256), 0, s, c->prep_terms, (int)B));
    else LAUNCH("prep", SMK(shape_term_multi_kernel, dim3((unsigned)B, (unsigned)c->n_prep_terms), dim3(256), 0, s, c->prep_terms));
    LAUNCH("prep", SMK(vn_shape_multi_kernel, dim3((unsigned)B, (unsigned)L), dim3(128), 0, s, d_shape, c->prep_vn));
    (void)SL; (void)S; (void)hd;
    return 0;
}

// One score evaluation on prepared batch data.  x_in/v_in: current state; outputs as given.
template <int H>
int run_score(shapemol_ctx *c, hipStream_t s, const float *x_in, const int64_t *v_in, int64_t N, int64_t B,
              bool sampling, int t_first, float *out_pos, float *out_h, float *out_v) {
    const shapemol_config &g = c->cfg;
    const int L = g.num_layers, hd = g.n_heads, C = g.num_classes, D = g.time_emb_dim, KP = c->KP;
    const int n = (int)N;
    AtomEmbArgs ae{c->P(c->dm.embw), c->P(c->dm.embb), v_in, c->mol_of, c->ttab, c->t_mol, sampling ? c->steps : nullptr,
                   c->steps + 1, c->bn_acc, c->h_a, n, H, C, D, t_first, L * kBnReplicas * 2 * hd + L};   // + the grid-barrier counters
    const int nlay = c->stop_layer >= 0 ? std::min(c->stop_layer, L) : L;
    const bool fused_prologue = c->chain_bf16 && c->lin_bf16;   // embedding + first queries + first per-node products in one launch
    if (fused_prologue) {
        const DevLayer &D0 = c->dm.layer[0];
        NodePrologueArgs pa{};
        pa.emb_wT = c->P(c->dm.embwT); pa.emb_b = ae.b; pa.v = v_in; pa.mol_of = c->mol_of; pa.ttab = c->ttab; pa.etab = c->etab; pa.t_mol = c->t_mol;
        pa.step_ptr = ae.step_ptr; pa.step_cur = ae.step_cur; pa.bn_acc = c->bn_acc; pa.h_out = c->h_a;
        pa.q = follow_of(c, D0.q_x2h, NODE_LN_RELU, c->q_x, H, H);
        pa.lin_img6 = c->P(c->node_f16 ? D0.pre16_x2h : D0.pre6_x2h); pa.add_mol = c->add0; pa.pre_out = c->pre0;
        pa.n_lin_tiles = nlay > 0 ? 4 * (H / 16) : 0; pa.ld_add = 4 * H; pa.ld_out = 4 * H;
        pa.n_atoms = n; pa.C = C; pa.D = D; pa.t_first = t_first; pa.bn_acc_len = ae.bn_acc_len;
        pa.stamps = c->kstamp_sel == 5 ? c->kstamps : nullptr;
        const int n_ct = (n + 15) / 16;
        if (c->node_f16 && c->feat_f16) LAUNCH("node_prologue", SMK((node_prologue16_kernel<H, true>), dim3((n_ct + CHAIN_COLS - 1) / CHAIN_COLS), dim3(H * 4),
                                                   2 * Chain16Lds<H>::FRAG * 16 + Chain16Lds<H>::PRE * 4, s, pa, c->status + ST_RANGE));
        else if (c->node_f16) LAUNCH("node_prologue", SMK(node_prologue16_kernel<H>, dim3((n_ct + CHAIN_COLS - 1) / CHAIN_COLS), dim3(H * 4),
                                                   2 * Chain16Lds<H>::FRAG * 16 + Chain16Lds<H>::PRE * 4, s, pa, c->status + ST_RANGE));
        else LAUNCH("node_prologue", SMK(node_prologue6_kernel<H>, dim3((n_ct + CHAIN_COLS - 1) / CHAIN_COLS), dim3(H * 4),
                                                   2 * Chain6Lds<H>::FRAG * 16 + Chain6Lds<H>::PRE * 4, s, pa));
    } else {
        LAUNCH("embed", SMK(atom_embed_kernel, dim3((N * H + 255) / 256), dim3(256), 0, s, ae));
    }
    // (chains only: the hint is set for the batch of a chain; a score evaluation on other data must not trust a stale one)
    const bool graph_fused = sampling && c->graph_fuse && c->max_mol_atoms > 0 && c->max_mol_atoms <= kGraphCap && KP <= 32 && c->n_pins == 0;
    if (graph_fused) {
        GraphArgs ga{x_in, c->mol_span, n, g.knn, KP, c->nbr, c->P(c->dm.ew.w1), c->P(c->dm.ew.b1), c->P(c->dm.ew.g), c->P(c->dm.ew.be),
                     c->P(c->dm.ew.w2), c->P(c->dm.ew.b2), c->ew, c->status + ST_SPAN, c->kstamp_sel == 4 ? c->kstamps : nullptr};
        const int apb = kGraphWaves * (KP >= 16 ? 1 : 16 / KP);      // atoms per workgroup
        if (KP == 8) LAUNCH("graph", SMK((graph_kernel<H, 8>), dim3((n + apb - 1) / apb), dim3(kGraphWaves * 64), 0, s, ga));
        else if (KP == 16) LAUNCH("graph", SMK((graph_kernel<H, 16>), dim3((n + apb - 1) / apb), dim3(kGraphWaves * 64), 0, s, ga));
        else LAUNCH("graph", SMK((graph_kernel<H, 32>), dim3((n + apb - 1) / apb), dim3(kGraphWaves * 64), 0, s, ga));
    } else LAUNCH("knn", SMK(knn_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x_in, c->mol_of, c->mol_off, n, g.knn, KP, c->nbr));
    if (sampling && c->n_pins > 0)      // diagnostic: the pinned atoms of this reverse step take the given neighbour lists
        LAUNCH("knn", SMK(knn_pin_kernel, dim3(32), dim3(256), 0, s, c->pin_off, c->pin_steps, c->pin_atom, c->pin_nbr, c->pin_k, KP, c->steps + 1, n, c->nbr));
    EdgeWeightArgs ea{x_in, c->nbr, c->P(c->dm.ew.w1), c->P(c->dm.ew.b1), c->P(c->dm.ew.g), c->P(c->dm.ew.be),
                      c->P(c->dm.ew.w2), c->P(c->dm.ew.b2), c->ew, n * KP, KP};
    if (!graph_fused) {
        const int tiles = (n * KP + 15) / 16;
        LAUNCH("edge_weight", SMK(edge_weight_kernel<H>, dim3((tiles + 3) / 4), dim3(256), 0, s, ea));
    }
    const float *cur_x = x_in;
    float *cur_h = c->h_a;
    constexpr int NT = H / 16;
    bool v_done = false;
    if (nlay > 0 && !fused_prologue) {   // prologue: per-node products and queries of the first x2h attention
        const DevLayer &D0 = c->dm.layer[0];
        if (launch_linear<H>(c, s, "node_pre", cur_h, c->P(D0.pre_x2h), c->P(D0.pre6_x2h), c->P(D0.pre16_x2h), c->add0, 4 * H, c->pre0, 4 * H, 4 * NT, n,
                             c->kstamp_sel == 0 ? c->kstamps : nullptr)) return 1;
        if (launch_mlp2<H, 1>(c, s, "node_q", D0.q_x2h, cur_h, nullptr, NODE_LN_RELU, nullptr, c->q_x, H, H, n)) return 1;
    }
    const bool fold = sampling && vn_fold_ok(c, n);       // chains only: coordinate update of layer l inside the x2h kernel of layer l + 1
    // evaluation-mode batch-norm: the consumers of the batch sums read sums that reproduce the running statistics
    double *stat_acc = c->bn_acc;
    if (c->bn_eval) {
        if (!c->bn_run) return fail("bn_eval = 1 needs the running statistics (shapemol_set_bn_running)");
        if (c->vn_fuse == 1) return fail("bn_eval = 1 is not available with vn_fuse = 1 (coordinate update behind a grid barrier)");
        const int tot = L * kBnReplicas * 2 * hd;
        LAUNCH("prep", SMK(bn_eval_fill_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, c->bn_run, c->bn_run + (size_t)L * hd, L, hd, n, c->bn_eval_acc));
        stat_acc = c->bn_eval_acc;
    }
    VnFold pending{};                         // ... which then receives this
    for (int l = 0; l < nlay; ++l) {
        const DevLayer &Dl = c->dm.layer[l];
        const bool last = (l == nlay - 1), has_next = !last;
        const bool phases = c->edge_bf16 && KP <= 16;
        const bool f16 = c->edge_bf16 == 3;     // two-piece f16 operands (sm_edge16.h), the default
        const bool stream = c->edge_bf16 == 2;      // exactly split bf16 operands, streaming kernels (sm_edge_stream.h)
        const bool xc_fused = f16 && x2h_chain_ok<H>(c, n);
        const bool half_tiles = (f16 || c->edge_bf16 == 2) && KP > 16;          // k > 16: two 16-slot tiles per atom + combine
        Edge16Args xea{};
        if (f16) {   // x2h attention: both MLP images resident, one barrier
            Edge16Args &ea = xea;
            ea.image_k = c->P(Dl.i16_kx); ea.image_v = c->P(Dl.i16_vx);
            ea.pre = l == 0 ? c->pre0 : c->preAB + 4 * H; ea.q = c->q_x; ea.x = cur_x; ea.nbr = c->nbr; ea.ew = c->ew; ea.out = c->att;
            ea.n_atoms = n; ea.ld_pre = l == 0 ? 4 * H : 8 * H; ea.stamps = (c->kstamp_sel == 1 && l == 0) ? c->kstamps : nullptr;
            ea.vf = pending; pending = VnFold{};
            if (half_tiles) { ea.out = c->part_rows; ea.part_ms = c->part_ms; }       // k > 16: per-tile rows, merged below
            if (!xc_fused && launch_edge16<H, false>(c, s, ea)) return 1;     // (fused: launched with the node stage below)
            if (half_tiles && launch_combine32<false>(c, s, c->att, n)) return 1;
        } else if (stream) {   // x2h attention, exactly split bf16 operands, producer / consumer waves (sm_edge_stream.h)
            EdgeStreamArgs sa{};
            sa.part_k = c->P(Dl.st_kx); sa.part_v = c->P(Dl.st_vx);
            sa.w2k = reinterpret_cast<const unsigned *>(c->P(Dl.sw2_kx)); sa.w2v = reinterpret_cast<const unsigned *>(c->P(Dl.sw2_vx));
            sa.b2v = c->P(Dl.sb2_vx);
            sa.pre = l == 0 ? c->pre0 : c->preAB + 4 * H; sa.q = c->q_x; sa.x = cur_x; sa.nbr = c->nbr; sa.ew = c->ew; sa.out = c->att;
            sa.n_atoms = n; sa.ld_pre = l == 0 ? 4 * H : 8 * H; sa.stamps = (c->kstamp_sel == 1 && l == 0) ? c->kstamps : nullptr;
            sa.vf = pending; pending = VnFold{};
            if (half_tiles) { sa.out = c->part_rows; sa.part_ms = c->part_ms; }       // k > 16: per-tile rows, merged below
            if (launch_stream<H, false>(c, s, sa)) return 1;
            if (half_tiles && launch_combine32<false>(c, s, c->att, n)) return 1;
        } else if (phases && c->edge_bf16 == 1) {   // x2h attention, key and value phase in one launch
            EdgeFusedArgs fa{c->P(Dl.img_kx), c->P(Dl.img_vx), l == 0 ? c->pre0 : c->preAB + 4 * H, c->q_x, cur_x, c->nbr, c->ew,
                             c->alpha, c->att, n, l == 0 ? 4 * H : 8 * H, (c->kstamp_sel == 1 && l == 0) ? c->kstamps : nullptr};
            if (launch_fused<H, false>(c, s, fa)) return 1;
        } else {   // x2h attention (fp32 MFMA kernels)
            EdgeArgs e{c->P(Dl.blob_x2h), l == 0 ? c->pre0 : c->preAB + 4 * H, c->q_x, cur_x, c->nbr, c->ew, c->att, n,
                       l == 0 ? 4 * H : 8 * H, (c->kstamp_sel == 1 && l == 0) ? c->kstamps : nullptr};
            if (launch_edge<H, false>(c, s, e)) return 1;
        }
        {   // node side: h' = h + MLP([att | h]); queries of h2x (this layer) and x2h (next layer) or the v head
            float *dst = (last && out_h) ? out_h : (cur_h == c->h_a ? c->h_b : c->h_a);
            NodeChainArgs na{};
            na.att = c->att; na.h = cur_h; na.h_out = dst; na.n_atoms = n;
            na.stamps = (c->kstamp_sel == 3 && l == 0) ? c->kstamps : nullptr;
            na.w1img = c->P(Dl.no.w1img); na.b1 = c->P(Dl.no.b1); na.ln_g = c->P(Dl.no.g); na.ln_b = c->P(Dl.no.be);
            na.w2img = c->P(Dl.no.w2img); na.b2 = c->P(Dl.no.b2);
            na.w1img6 = c->P(c->node_f16 ? Dl.no.w1img16 : Dl.no.w1img6); na.w2img6 = c->P(c->node_f16 ? Dl.no.w2img16 : Dl.no.w2img6);
            na.f[0] = follow_of(c, Dl.q_h2x, NODE_LN_RELU, c->q_h, H, H);
            na.n_follow = 1;
            if (has_next) { na.f[1] = follow_of(c, c->dm.layer[l + 1].q_x2h, NODE_LN_RELU, c->q_x, H, H); na.n_follow = 2; }
            else if (out_v) { na.f[1] = follow_of(c, c->dm.vhead, NODE_SSP, out_v, C, C); na.n_follow = 2; v_done = true; }
            const int n_ct = (n + 15) / 16;
            const int lin_tiles = (has_next && l + 1 < L ? 8 : 4) * NT;
            const bool lin_fused = c->chain_bf16 && c->lin_bf16 && c->node_f16 && c->lin_fuse;
            if (lin_fused) {
                na.lin_img16 = c->P(Dl.lin16_img); na.add_mol = c->addp + (size_t)l * c->capB * 8 * H; na.mol_of = c->mol_of;
                na.pre_out = c->preAB; na.n_lin_tiles = lin_tiles; na.ld_add = 8 * H; na.ld_out = 8 * H;
            }
            if (xc_fused) { if (launch_x2h_chain<H>(c, s, xea, na)) return 1; }
            else if (c->chain_bf16 && c->node_f16 && c->feat_f16) LAUNCH("node_chain", SMK((node_chain16_kernel<H, true>), dim3((n_ct + CHAIN_COLS - 1) / CHAIN_COLS), dim3(H * 4), Chain16Lds<H>::BYTES, s, na, c->status + ST_RANGE));
            else if (c->chain_bf16 && c->node_f16) LAUNCH("node_chain", SMK(node_chain16_kernel<H>, dim3((n_ct + CHAIN_COLS - 1) / CHAIN_COLS), dim3(H * 4), Chain16Lds<H>::BYTES, s, na, c->status + ST_RANGE));
            else if (c->chain_bf16) LAUNCH("node_chain", SMK(node_chain6_kernel<H>, dim3((n_ct + CHAIN_COLS - 1) / CHAIN_COLS), dim3(H * 4), Chain6Lds<H>::BYTES, s, na));
            else LAUNCH("node_chain", SMK(node_chain_kernel<H>, dim3((n_ct + CHAIN_COLS - 1) / CHAIN_COLS), dim3(H * 4), 0, s, na));
            cur_h = dst;
            // per-node halves of the edge MLPs' first Linear: h2x of this layer | x2h of the next one
            if (!lin_fused && launch_linear<H>(c, s, "node_pre", cur_h, c->P(Dl.lin_img), c->P(Dl.lin6_img), c->P(Dl.lin16_img), c->addp + (size_t)l * c->capB * 8 * H, 8 * H,
                                 c->preAB, 8 * H, lin_tiles, n, (c->kstamp_sel == 0 && l == 0) ? c->kstamps : nullptr)) return 1;
        }
        float *x_next = (last && out_pos) ? out_pos : ((cur_x == c->x_a) ? c->x_b : c->x_a);
        bool vn_done = false, stats_done = false;
        if (f16 && c->vn_fuse != 1) {   // h2x attention (+ VN-linear and batch statistics when vn_fuse = 2)
            Edge16Args ea{};
            ea.image_k = c->P(Dl.i16_kh); ea.image_v = c->P(Dl.i16_vh);
            ea.pre = c->preAB; ea.q = c->q_h; ea.x = cur_x; ea.nbr = c->nbr; ea.ew = c->ew; ea.out = c->o3;
            ea.n_atoms = n; ea.ld_pre = 8 * H; ea.stamps = (c->kstamp_sel == 2 && l == 0) ? c->kstamps : nullptr;
            if (half_tiles) { ea.out = c->part_rows; ea.part_ms = c->part_ms; }       // (no fused VN-linear: vn_stats / vn_apply below)
            if (c->vn_fuse && !half_tiles) {
                ea.vn = {c->ps + (size_t)l * c->capB * 2 * hd * 3, c->P(Dl.wf_x), c->P(Dl.wd_x), c->P(Dl.wf_o), c->P(Dl.wd_o),
                         c->P(Dl.bn_g), c->P(Dl.bn_b), c->mol_of, c->pd, c->bn_acc + (size_t)l * kBnReplicas * 2 * hd,
                         nullptr, c->status + ST_VN_BARRIER, x_next, 2};
                stats_done = true;
                if (fold && has_next) {      // no vn_apply launch: the next x2h kernel finishes the update
                    ea.xsum = c->xsum;
                    pending = VnFold{c->pd, stat_acc + (size_t)l * kBnReplicas * 2 * hd, c->P(Dl.bn_g), c->P(Dl.bn_b), c->xsum, cur_x, x_next,
                                     c->mol_span, c->status + ST_SPAN, 1};
                    vn_done = true;
                } else if (fold && last && l == L - 1 && out_pos && c->ddpm_fold && c->g_points == 0 && C <= 16 && hd <= 16) {
                    // ... or, for the last layer of a chain step, the DDPM kernel
                    ea.xsum = c->xsum;
                    c->ddpm_vf = DdpmFold{c->pd, stat_acc + (size_t)l * kBnReplicas * 2 * hd, c->P(Dl.bn_g), c->P(Dl.bn_b), c->xsum, cur_x, out_pos, hd, 1};
                    vn_done = true;
                }
            }
            if (launch_edge16<H, true>(c, s, ea)) return 1;
            if (half_tiles && launch_combine32<true>(c, s, c->o3, n)) return 1;
        } else if (stream && c->vn_fuse != 1) {   // h2x attention (+ VN-linear and batch statistics when vn_fuse = 2), streaming kernel
            EdgeStreamArgs sa{};
            sa.part_k = c->P(Dl.st_kh); sa.part_v = c->P(Dl.st_vh);
            sa.w2k = reinterpret_cast<const unsigned *>(c->P(Dl.sw2_kh));
            sa.pre = c->preAB; sa.q = c->q_h; sa.x = cur_x; sa.nbr = c->nbr; sa.ew = c->ew; sa.out = c->o3;
            sa.n_atoms = n; sa.ld_pre = 8 * H; sa.stamps = (c->kstamp_sel == 2 && l == 0) ? c->kstamps : nullptr;
            if (half_tiles) { sa.out = c->part_rows; sa.part_ms = c->part_ms; }       // (no fused VN-linear: vn_stats / vn_apply below)
            if (c->vn_fuse && !half_tiles) {
                sa.vn = {c->ps + (size_t)l * c->capB * 2 * hd * 3, c->P(Dl.wf_x), c->P(Dl.wd_x), c->P(Dl.wf_o), c->P(Dl.wd_o),
                         c->P(Dl.bn_g), c->P(Dl.bn_b), c->mol_of, c->pd, c->bn_acc + (size_t)l * kBnReplicas * 2 * hd,
                         nullptr, c->status + ST_VN_BARRIER, x_next, 2};
                stats_done = true;
                if (fold && has_next) {      // no vn_apply launch: the next x2h kernel finishes the update
                    sa.xsum = c->xsum;
                    pending = VnFold{c->pd, stat_acc + (size_t)l * kBnReplicas * 2 * hd, c->P(Dl.bn_g), c->P(Dl.bn_b), c->xsum, cur_x, x_next,
                                     c->mol_span, c->status + ST_SPAN, 1};
                    vn_done = true;
                } else if (fold && last && l == L - 1 && out_pos && c->ddpm_fold && c->g_points == 0 && C <= 16 && hd <= 16) {
                    sa.xsum = c->xsum;
                    c->ddpm_vf = DdpmFold{c->pd, stat_acc + (size_t)l * kBnReplicas * 2 * hd, c->P(Dl.bn_g), c->P(Dl.bn_b), c->xsum, cur_x, out_pos, hd, 1};
                    vn_done = true;
                }
            }
            if (launch_stream<H, true>(c, s, sa)) return 1;
            if (half_tiles && launch_combine32<true>(c, s, c->o3, n)) return 1;
        } else if (phases) {   // h2x attention, both images resident in LDS (exactly split bf16 operands)
            EdgeFusedArgs fa{c->P(Dl.img_kh), c->P(Dl.img_vh), c->preAB, c->q_h, cur_x, c->nbr, c->ew, c->alpha, c->o3, n, 8 * H,
                             (c->kstamp_sel == 2 && l == 0) ? c->kstamps : nullptr};
            if (c->vn_fuse) {   // VN-linear + batch statistics (2) or the whole coordinate update (1: grid barrier inside) behind the attention
                double *tail = c->bn_acc + (size_t)L * kBnReplicas * 2 * hd;
                fa.vn = {c->ps + (size_t)l * c->capB * 2 * hd * 3, c->P(Dl.wf_x), c->P(Dl.wd_x), c->P(Dl.wf_o), c->P(Dl.wd_o),
                         c->P(Dl.bn_g), c->P(Dl.bn_b), c->mol_of, c->pd, c->bn_acc + (size_t)l * kBnReplicas * 2 * hd,
                         reinterpret_cast<unsigned *>(tail + l), c->status + ST_VN_BARRIER, x_next, c->vn_fuse == 1 ? 1 : 2};
                if (c->vn_fuse == 1) vn_done = true; else stats_done = true;
            }
            if (launch_fused<H, true>(c, s, fa)) return 1;
        } else {   // h2x attention (fp32 MFMA kernels)
            EdgeArgs e{c->P(Dl.blob_h2x), c->preAB, c->q_h, cur_x, c->nbr, c->ew, c->o3, n, 8 * H,
                       (c->kstamp_sel == 2 && l == 0) ? c->kstamps : nullptr};
            if (launch_edge<H, true>(c, s, e)) return 1;
        }
        VnArgs va{cur_x, c->o3, c->ps + (size_t)l * c->capB * 2 * hd * 3, c->P(Dl.wf_x), c->P(Dl.wd_x), c->P(Dl.wf_o),
                  c->P(Dl.wd_o), c->P(Dl.bn_g), c->P(Dl.bn_b), c->mol_of, c->pd, c->bn_acc + (size_t)l * kBnReplicas * 2 * hd, x_next, n, hd};
        const int per_blk = 256 / hd;
        if (!vn_done) {
            if (!stats_done) LAUNCH("vn_stats", SMK(vn_stats_kernel, dim3((N + per_blk - 1) / per_blk), dim3(kVnThreads), 0, s, va));
            if (c->bn_eval) va.acc = stat_acc + (size_t)l * kBnReplicas * 2 * hd;      // (the statistics pass wrote the batch's sums elsewhere)
            LAUNCH("vn_apply", SMK(vn_apply_kernel, dim3((N + per_blk - 1) / per_blk), dim3(256), 0, s, va));
        }
        cur_x = x_next;
    }
    c->last_h = cur_h; c->last_x = cur_x;
    if (nlay == 0 && out_pos) HIPCHK(hipMemcpyAsync(out_pos, x_in, N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_v && !v_done) {
        if (launch_mlp2<H, 1>(c, s, "v_head", c->dm.vhead, cur_h, nullptr, NODE_SSP, nullptr, out_v, C, C, n)) return 1;
    }
    return 0;
}

template <int H>
int run_ddpm(shapemol_ctx *c, hipStream_t s, int64_t N) {
    const shapemol_config &g = c->cfg;
    DdpmArgs a{};
    a.pred_pos = c->pred_pos; a.pred_v = c->pred_v; a.x_t = c->x_state; a.v_t = c->v_state; a.mol_of = c->mol_of;
    a.t_first = c->cfg.num_timesteps - 1;
    a.c0 = c->P(c->dm.tab[0]); a.ct = c->P(c->dm.tab[1]); a.logvar = c->P(c->dm.tab[2]); a.log_a = c->P(c->dm.tab[3]);
    a.log_1ma = c->P(c->dm.tab[4]); a.log_abar = c->P(c->dm.tab[5]); a.log_1mabar = c->P(c->dm.tab[6]);
    a.cp = c->chain_params; a.step_cur = c->steps + 1; a.step_ptr = c->steps;
    a.x_next = c->x_state; a.v_next = c->v_state;
    a.n_atoms = (int)N; a.C = g.num_classes;
    a.vf = c->ddpm_vf; c->ddpm_vf = DdpmFold{};
    if (c->stamp_on) LAUNCH("stamp", SMK(clock_stamp_kernel, dim3(1), dim3(64), 0, s, c->stamps, c->steps + 1, 1024));
    if (g.num_classes <= 16) LAUNCH("ddpm", SMK(ddpm_step16_kernel, dim3((N * 16 + 255) / 256), dim3(256), 0, s, a));
    else LAUNCH("ddpm", SMK(ddpm_step_kernel<32>, dim3((N + 127) / 128), dim3(128), 0, s, a));
    return 0;
}

int check_cfg(const shapemol_config &g) {
    if (g.hidden_dim != 128 && g.hidden_dim != 32) return fail("hidden_dim must be 128 (or 32 for the reduced test model)");
    if (g.n_heads * 8 != g.hidden_dim) return fail("hidden_dim / n_heads must be 8");
    if (g.num_r_gaussian != 20) return fail("num_r_gaussian must be 20 (fixed RBF centres)");
    if (g.knn < 1 || g.knn > 32) return fail("knn must be in 1..32");
    if (g.shape_dim < 1 || g.shape_dim > 64 || g.shape_latent_dim < 4 || g.shape_latent_dim > 64 || (g.shape_latent_dim & 3))
        return fail("shape_dim must be 1..64, shape_latent_dim a multiple of 4 in 4..64");
    if (g.time_emb_dim < 4 || g.time_emb_dim > 16 || (g.time_emb_dim & 1)) return fail("time_emb_dim must be even, 4..16");
    if (g.num_classes < 2 || g.num_classes > 32) return fail("num_classes must be 2..32");
    if (g.num_layers < 1 || g.num_timesteps < 1) return fail("num_layers / num_timesteps must be positive");
    return 0;
}

#define DISPATCH_H(c, call128, call32) ((c)->cfg.hidden_dim == 128 ? (call128) : (call32))

}  // namespace

// =================================================================================================
extern "C" {

int shapemol_abi_version(void) { return SHAPEMOL_ABI_VERSION; }
void shapemol_set_error_(const char *msg) { g_err = msg ? msg : ""; }      // other translation units of the library
const char *shapemol_last_error(void) { return g_err.c_str(); }

size_t shapemol_weight_count(const shapemol_config *cfg) { return cfg ? weight_count(*cfg) : 0; }

int shapemol_create(const shapemol_config *cfg, const float *weights, size_t n_weights, int device, shapemol_ctx **out) {
    if (!cfg || !weights || !out) return fail("shapemol_create: null argument");
    if (check_cfg(*cfg)) return 1;
    if (n_weights != weight_count(*cfg)) return fail("shapemol_create: weight count mismatch (got " + std::to_string(n_weights) + ", want " + std::to_string(weight_count(*cfg)) + ")");
    HostModel hm;
    if (!parse_weights(*cfg, weights, n_weights, hm)) return fail("shapemol_create: weight layout mismatch");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("shapemol_create: no such HIP device");
    HIPCHK(hipSetDevice(device));
    auto *c = new shapemol_ctx();
    c->cfg = *cfg; c->device = device;
    c->KP = cfg->knn <= 8 ? 8 : (cfg->knn <= 16 ? 16 : 32);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    const int H = cfg->hidden_dim, C = cfg->num_classes, T = cfg->num_timesteps;
    Image im;
    DevModel &dm = c->dm;
    for (int i = 0; i < 7; ++i) dm.tab[i] = im.put(hm.tab[i], T);
    dm.te1w = im.put(hm.te1.w, (size_t)hm.te1.out * hm.te1.in); dm.te1b = im.put(hm.te1.b, hm.te1.out);
    dm.te2w = im.put(hm.te2.w, (size_t)hm.te2.out * hm.te2.in); dm.te2b = im.put(hm.te2.b, hm.te2.out);
    dm.embw = im.put(hm.emb.w, (size_t)hm.emb.out * hm.emb.in); dm.embb = im.put(hm.emb.b, hm.emb.out);
    dm.embwT = im.alloc((size_t)hm.emb.in * hm.emb.out);          // [C + D][H]
    for (int f = 0; f < hm.emb.out; ++f)
        for (int k = 0; k < hm.emb.in; ++k) im.d[dm.embwT + (size_t)k * hm.emb.out + f] = hm.emb.w[(size_t)f * hm.emb.in + k];
    auto put_mlp = [&](const Mlp &m) {
        DevMlp d;
        d.w1 = im.put(m.l1.w, (size_t)m.l1.out * m.l1.in); d.b1 = im.put(m.l1.b, m.l1.out);
        d.g = im.put(m.g, m.l1.out); d.be = im.put(m.be, m.l1.out);
        d.w2 = im.put(m.l2.w, (size_t)m.l2.out * m.l2.in); d.b2 = im.put(m.l2.b, m.l2.out);
        return d;
    };
    dm.ew = put_mlp(hm.ew);
    dm.inv = put_mlp(hm.inv);
    {
        Mlp vh; vh.l1 = hm.v1; vh.l2 = hm.v2; vh.g = nullptr; vh.be = nullptr;
        dm.vhead = put_mlp_img(im, vh);
    }
    dm.layer.resize(cfg->num_layers);
    for (int l = 0; l < cfg->num_layers; ++l) {
        if (H == 128 ? build_layer_image<128>(*cfg, hm.layer[l], im, dm.layer[l], c->hid_max)
                     : build_layer_image<32>(*cfg, hm.layer[l], im, dm.layer[l], c->hid_max)) { delete c; return 1; }
    }
    for (int l = 0; l < cfg->num_layers; ++l) {      // paired images: pre_h2x(l) | pre_x2h(l + 1)
        const size_t blk = (size_t)4 * H * H;
        const size_t o = im.alloc(2 * blk);
        std::memcpy(&im.d[o], &im.d[dm.layer[l].pre_h2x], blk * sizeof(float));
        if (l + 1 < cfg->num_layers) std::memcpy(&im.d[o + blk], &im.d[dm.layer[l + 1].pre_x2h], blk * sizeof(float));
        dm.layer[l].lin_img = o;
    }
    for (int l = 0; l < cfg->num_layers; ++l) {
        dm.layer[l].lin6_img = pack_linear6_image(im, dm.layer[l].lin_img, 8 * H, H);
        dm.layer[l].pre6_x2h = l == 0 ? pack_linear6_image(im, dm.layer[l].pre_x2h, 4 * H, H) : 0;
        dm.layer[l].lin16_img = pack_linear16_image(im, dm.layer[l].lin_img, 8 * H, H);
        dm.layer[l].pre16_x2h = l == 0 ? pack_linear16_image(im, dm.layer[l].pre_x2h, 4 * H, H) : 0;
    }
    im.alloc(64);
    if (hipMalloc((void **)&c->d_img, im.d.size() * sizeof(float)) != hipSuccess) { delete c; return fail("hipMalloc(weights) failed"); }
    if (hipMemcpy(c->d_img, im.d.data(), im.d.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { hipFree(c->d_img); delete c; return fail("hipMemcpy(weights) failed"); }
    if (H == 128 ? set_edge_attr<128>(c->KP) : set_edge_attr<32>(c->KP)) { hipFree(c->d_img); delete c; return 1; }
    {   // time-embedding table over all timesteps
        if (hipMalloc((void **)&c->ttab, (size_t)T * cfg->time_emb_dim * sizeof(float)) != hipSuccess) { hipFree(c->d_img); delete c; return fail("hipMalloc(time table) failed"); }
        TimeTableArgs ta{c->P(dm.te1w), c->P(dm.te1b), c->P(dm.te2w), c->P(dm.te2b), c->ttab, T, cfg->time_emb_dim};
        hipLaunchKernelGGL(time_table_kernel, dim3((T + 63) / 64), dim3(64), 0, nullptr, ta);
        if (hipDeviceSynchronize() != hipSuccess) { hipFree(c->ttab); hipFree(c->d_img); delete c; return fail("time table kernel failed"); }
        // ... and the atom embedding of every (timestep, atom type) pair
        const int C = cfg->num_classes;
        if (hipMalloc((void **)&c->etab, (size_t)T * C * H * sizeof(float)) != hipSuccess) { hipFree(c->ttab); hipFree(c->d_img); delete c; return fail("hipMalloc(embedding table) failed"); }
        const int items = T * C * (H / 4);
        hipLaunchKernelGGL(emb_table_kernel, dim3((items + 255) / 256), dim3(256), 0, nullptr, c->P(dm.embwT), c->P(dm.embb), c->ttab, c->etab, T, C, cfg->time_emb_dim, H);
        if (hipDeviceSynchronize() != hipSuccess) { hipFree(c->etab); hipFree(c->ttab); hipFree(c->d_img); delete c; return fail("embedding table kernel failed"); }
    }
    // (the default kernels split every operand into bf16 pieces, which have the fp32 exponent range: no bound on the weights.  The
    //  optional two-piece f16 kernels, edge_bf16 = 3, refuse weights whose LayerNorm outputs could leave the fp16 range: hid_max)
    *out = c;
    return 0;
}

void shapemol_destroy(shapemol_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    c->drop_graphs();
    for (auto &r : c->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    for (void *p : c->allocs) hipFree(p);
    hipFree(c->ttab);
    hipFree(c->etab);
    hipFree(c->d_img);
    if (c->g_cloud) hipFree(c->g_cloud);
    if (c->bn_run) hipFree(c->bn_run);
    if (c->bn_eval_acc) hipFree(c->bn_eval_acc);
    if (c->pin_off) { hipFree(c->pin_off); hipFree(c->pin_atom); hipFree(c->pin_nbr); }
    delete c;
}

int shapemol_set_bn_running(shapemol_ctx *c, const float *h_mean, const float *h_var, int64_t count) {
    if (!c || !h_mean || !h_var) return fail("shapemol_set_bn_running: null argument");
    const int L = c->cfg.num_layers, hd = c->cfg.n_heads;
    if (count != (int64_t)L * hd) return fail("shapemol_set_bn_running: count must be num_layers * n_heads");
    for (int64_t i = 0; i < count; ++i)
        if (!(h_var[i] >= 0.f) || !std::isfinite(h_mean[i]) || !std::isfinite(h_var[i])) return fail("shapemol_set_bn_running: statistics must be finite, variance >= 0");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    if (!c->bn_run) {
        HIPCHK(hipMalloc(&c->bn_run, 2 * count * sizeof(float)));
        HIPCHK(hipMalloc(&c->bn_eval_acc, (size_t)L * kBnReplicas * 2 * hd * sizeof(double)));
    }
    HIPCHK(hipMemcpy(c->bn_run, h_mean, count * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->bn_run + count, h_var, count * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int shapemol_reserve(shapemol_ctx *c, int64_t max_atoms, int64_t max_mols) {
    if (!c || max_atoms < 1 || max_mols < 1) return fail("shapemol_reserve: bad argument");
    return ensure_workspace(c, max_atoms, max_mols);
}

int shapemol_score(shapemol_ctx *c, const float *d_pos, const int64_t *d_v, const int64_t *d_batch, int64_t N,
                   int64_t B, const float *d_shape, const int64_t *d_t, float *out_pos, float *out_h, float *out_v,
                   void *stream) {
    if (!c || !d_pos || !d_v || !d_batch || !d_shape || !d_t || !out_pos || !out_v) return fail("shapemol_score: null argument");
    if (N < 1 || B < 1 || N > (1 << 27)) return fail("shapemol_score: n_atoms / n_mols out of range");
    HIPCHK(hipSetDevice(c->device));
    if (ensure_workspace(c, N, B)) return 1;
    hipStream_t s = (hipStream_t)stream;
    c->lastN = N; c->lastB = B;
    HIPCHK(hipMemsetAsync(c->status, 0, 8 * sizeof(int), s));
    if (DISPATCH_H(c, run_prep<128>(c, s, d_batch, N, B, d_shape), run_prep<32>(c, s, d_batch, N, B, d_shape))) return 1;
    LAUNCH("prep", SMK(t_convert_kernel, dim3((B + 255) / 256), dim3(256), 0, s, d_t, (int)B, c->cfg.num_timesteps, c->t_mol, c->status));
    LAUNCH("prep", SMK(v_check_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_v, (int)N, c->cfg.num_classes, c->status));
    return DISPATCH_H(c, run_score<128>(c, s, d_pos, d_v, N, B, false, 0, out_pos, out_h, out_v),
                      run_score<32>(c, s, d_pos, d_v, N, B, false, 0, out_pos, out_h, out_v));
}

int shapemol_sample(shapemol_ctx *c, const float *d_init_pos, const int64_t *d_init_v, const int64_t *d_batch,
                    int64_t N, int64_t B, const float *d_shape, int32_t num_steps, const float *d_eps, const float *d_u,
                    uint64_t seed, const shapemol_traj *traj, float *out_pos, int64_t *out_v, int32_t use_graph, void *stream) {
    if (!c || !d_init_pos || !d_init_v || !d_batch || !d_shape || !out_pos || !out_v) return fail("shapemol_sample: null argument");
    if (N < 1 || B < 1 || N > (1 << 27)) return fail("shapemol_sample: n_atoms / n_mols out of range");
    if (num_steps < 1 || c->first_step + num_steps > c->cfg.num_timesteps) return fail("shapemol_sample: num_steps out of range");
    if ((d_eps == nullptr) != (d_u == nullptr)) return fail("shapemol_sample: d_eps and d_u must be given together");
    HIPCHK(hipSetDevice(c->device));
    if (ensure_workspace(c, N, B)) return 1;
    hipStream_t s = (hipStream_t)stream;
    c->lastN = N; c->lastB = B;
    const int t_first = c->cfg.num_timesteps - 1;
    HIPCHK(hipMemsetAsync(c->status, 0, 8 * sizeof(int), s));
    if (DISPATCH_H(c, run_prep<128>(c, s, d_batch, N, B, d_shape), run_prep<32>(c, s, d_batch, N, B, d_shape))) return 1;
    LAUNCH("prep", SMK(v_check_kernel, dim3((N + 255) / 256), dim3(256), 0, s, d_init_v, (int)N, c->cfg.num_classes, c->status));
    {
        ChainParams cp{};
        cp.seed = seed; cp.eps = d_eps; cp.u = d_u; cp.step_base = c->first_step; cp.guide_draws = c->g_draws;
        if (traj) { cp.tr_pos = traj->pos_traj; cp.tr_v = traj->v_traj; cp.tr_v0 = traj->v0_traj; cp.tr_vt = traj->vt_traj;
                    cp.tr_pos_cond = traj->pos_cond_traj; cp.tr_v_cond = traj->v_cond_traj; }
        LAUNCH("prep", SMK(set_chain_params_kernel, dim3(1), dim3(1), 0, s, c->chain_params, cp, c->steps));
    }
    HIPCHK(hipMemcpyAsync(c->x_state, d_init_pos, N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->v_state, d_init_v, N * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    auto one_step = [&]() -> int {
        if (DISPATCH_H(c, run_score<128>(c, s, c->x_state, c->v_state, N, B, true, t_first, c->pred_pos, nullptr, c->pred_v),
                       run_score<32>(c, s, c->x_state, c->v_state, N, B, true, t_first, c->pred_pos, nullptr, c->pred_v))) return 1;
        if (c->g_points > 0) {     // point-cloud shape guidance of the predicted x0 (steps with t > grad_step)
            PcGuideArgs ga{c->pred_pos, c->g_cloud, c->chain_params, c->steps + 1, (int)N, (int)c->g_points, t_first, c->g_grad_step, c->g_radius, 0.2};
            LAUNCH("pc_guidance", SMK(pc_guidance_kernel, dim3((N * 16 + 255) / 256), dim3(256), (size_t)c->g_points * 24, s, ga));
        }
        return DISPATCH_H(c, run_ddpm<128>(c, s, N), run_ddpm<32>(c, s, N));
    };
    if (use_graph && !c->prof_on) {
        shapemol_ctx::GraphKey key{};
        key.N = N; key.B = B; key.guided = c->g_points > 0; key.fold = vn_fold_ok(c, (int)N);
        key.gfuse = c->graph_fuse && c->max_mol_atoms > 0 && c->max_mol_atoms <= kGraphCap && c->n_pins == 0;
        // two executables: one reverse step, and kGraphUnroll steps back to back (the gap between two graph launches,
        // ~8 us, is then paid once per kGraphUnroll steps); every step reads its index from the device-side counter
        auto capture = [&](int n_steps, hipGraphExec_t *exec) -> int {
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            int rc = 0;
            for (int i = 0; i < n_steps && !rc; ++i) rc = one_step();
            const hipError_t ce = hipStreamEndCapture(s, &graph);
            if (rc) { if (graph) hipGraphDestroy(graph); return 1; }
            if (ce != hipSuccess) return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
            const hipError_t ie = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (ie != hipSuccess) { *exec = nullptr; return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
            ++c->n_captures;
            hipGraphUpload(*exec, s);      // stage the executable on the device now, not at its first launch inside a timed chain
            return 0;
        };
        if (!c->gexec || !(key == c->gkey)) {
            c->drop_graphs();
            if (capture(1, &c->gexec)) return 1;
            c->gkey = key;
        }
        // both executables are built at the first capture, whatever this chain's length: a short warm-up chain then leaves
        // nothing to capture inside a later, timed chain
        if (!c->gexec_u && capture(kGraphUnroll, &c->gexec_u)) return 1;
        int st = 0;
        c->gstream = s; c->gstream_set = true;
        for (; st + kGraphUnroll <= num_steps; st += kGraphUnroll) HIPCHK(hipGraphLaunch(c->gexec_u, s));
        for (; st < num_steps; ++st) HIPCHK(hipGraphLaunch(c->gexec, s));
    } else {
        for (int st = 0; st < num_steps; ++st) if (one_step()) return 1;
    }
    HIPCHK(hipMemcpyAsync(out_pos, c->x_state, N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(out_v, c->v_state, N * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    return 0;
}

int shapemol_log_sample_categorical(shapemol_ctx *c, const float *d_logits, const float *d_u, int64_t n_rows,
                                    int32_t n_classes, uint64_t seed, int64_t *out_index, void *stream) {
    if (!d_logits || !out_index || n_rows < 1 || n_classes < 1) return fail("shapemol_log_sample_categorical: bad argument");
    if (c) HIPCHK(hipSetDevice(c->device));      // ctx may be NULL: the current device is used
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gumbel_argmax_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, d_logits, d_u, (int)n_rows, (int)n_classes, seed, out_index);
    HIPCHK(hipGetLastError());
    return 0;
}

int shapemol_set_option(shapemol_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return fail("shapemol_set_option: null argument");
    const std::string k(name);
    if (k == "max_mol_atoms") {   // hint for the folded coordinate update; the captured graph is keyed on the resulting decision
        if (value < 0) return fail("max_mol_atoms must be >= 0");
        c->max_mol_atoms = (int)std::min<int64_t>(value, 1 << 20);
        return 0;
    }
    if (k == "first_step") {      // does not touch the captured graph: the step counter lives in device memory
        if (value < 0 || value >= c->cfg.num_timesteps) return fail("first_step must be in [0, num_timesteps)");
        c->first_step = (int)value;
        return 0;
    }
    if (k == "stop_layer") c->stop_layer = (int)value;
    else if (k == "edge_bf16") {
        if (value < 0 || value > 3) return fail("edge_bf16 must be 0 (fp32 MFMA), 1 (exactly split bf16, phase kernels), 2 (exactly split bf16, streaming kernels) or 3 (two-piece f16)");
        if (value != 3 && value != 2 && c->KP > 16) return fail("k > 16 runs on the streaming (edge_bf16 = 2) or the two-piece f16 (edge_bf16 = 3) edge kernels");
        if (value == 3 && c->hid_max > 6.0e4f) return fail("edge_bf16 = 3: the edge MLPs' LayerNorm outputs may exceed the fp16 range for these weights");
        c->edge_bf16 = (int)value;
    }
    else if (k == "edge_tiles") { if (value < -1 || value > 1) return fail("edge_tiles must be -1 (automatic), 0 (sliced one-job launches) or 1 (looping launch)"); c->edge_tiles = (int)value; }
    else if (k == "node_f16") c->node_f16 = value != 0;
    else if (k == "feat_f16") {
        if (value && (c->edge_bf16 != 3 || !c->node_f16)) return fail("feat_f16 = 1 needs the f16 kernels (edge_bf16 = 3, node_f16 = 1)");
        c->feat_f16 = value != 0;
    }
    else if (k == "lin_fuse") c->lin_fuse = value != 0;
    else if (k == "stream_whole_rounds") c->stream_whole_rounds = value != 0;
    else if (k == "x2h_chain") c->x2h_chain = value != 0;
    else if (k == "graph_fuse") c->graph_fuse = value != 0;
    else if (k == "bn_eval") c->bn_eval = value != 0;
    else if (k == "ddpm_fold") c->ddpm_fold = value != 0;
    else if (k == "vn_fold") c->vn_fold = value != 0;
    else if (k == "lin_bf16") c->lin_bf16 = (int)value;
    else if (k == "chain_bf16") c->chain_bf16 = (int)value;
    else if (k == "vn_fuse") c->vn_fuse = (int)value;
    else if (k == "lin_waves") { if (value < 1 || value > 16) return fail("lin_waves must be 1..16"); c->lin_waves = (int)value; }
    else if (k == "stamps") c->stamp_on = (int)value;
    else if (k == "kstamp_sel") c->kstamp_sel = (int)value;
    else if (k == "edge_waves") { if (value < 0 || value > 12) return fail("edge_waves must be 0 (automatic) .. 12"); c->edge_threads = (int)value * 64; }   // 0 = automatic
    else return fail("unknown option " + k);
    hipSetDevice(c->device);
    c->drop_graphs();
    return 0;
}

int64_t shapemol_debug_read(shapemol_ctx *c, const char *name, void *dst, size_t max_bytes) {
    if (!c || !name || !dst) { fail("shapemol_debug_read: null argument"); return -1; }
    const std::string k(name);
    const shapemol_config &g = c->cfg;
    const int64_t N = c->lastN;
    const void *src = nullptr; size_t bytes = 0;
    int64_t dims[8] = {N, c->lastB, c->KP, g.hidden_dim, g.n_heads, g.num_layers, c->capN, c->capB};
    if (k == "dims") { if (max_bytes < sizeof(dims)) return -1; std::memcpy(dst, dims, sizeof(dims)); return sizeof(dims); }
    if (k == "captures") { if (max_bytes < 8) return -1; std::memcpy(dst, &c->n_captures, 8); return 8; }
    if (k == "nbr") { src = c->nbr; bytes = N * c->KP * 4; }
    else if (k == "ew") { src = c->ew; bytes = N * c->KP * 4; }
    else if (k == "h") { src = c->last_h; bytes = N * g.hidden_dim * 4; }
    else if (k == "x") { src = c->last_x; bytes = N * 3 * 4; }
    else if (k == "pre") { src = c->preAB; bytes = N * 8 * g.hidden_dim * 4; }
    else if (k == "q") { src = c->q_h; bytes = N * g.hidden_dim * 4; }
    else if (k == "att") { src = c->att; bytes = N * g.hidden_dim * 4; }
    else if (k == "o3") { src = c->o3; bytes = N * 48 * 4; }
    else if (k == "stamps") { src = c->stamps; bytes = 2048 * 8; }
    else if (k == "kstamps") { src = c->kstamps; bytes = (size_t)8 * 16 * 4096 * 8; }
    else if (k == "vn_err") { src = c->status + ST_VN_BARRIER; bytes = 4; }
    else if (k == "bnstat") { src = c->bn_acc; bytes = (size_t)g.num_layers * kBnReplicas * 2 * g.n_heads * 8; }
    else { fail("shapemol_debug_read: unknown buffer " + k); return -1; }
    if (!src || bytes > max_bytes) { fail("shapemol_debug_read: buffer unavailable or destination too small"); return -1; }
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fail("shapemol_debug_read: copy failed"); return -1; }
    return (int64_t)bytes;
}

/* Diagnostic: the exact three-way bf16 split the packers apply to every weight (and the kernels, with the same arithmetic, to every
 * activation): pieces[0..2] = hi, mid, lo bit patterns, x == float(hi) + float(mid) + float(lo) exactly for |x| >= 2^-110. */
void shapemol_debug_split_exact(float x, uint16_t *pieces) {
    uint16_t p[3];
    split3_host(x, p);
    pieces[0] = p[0]; pieces[1] = p[1]; pieces[2] = p[2];
}

int shapemol_set_knn_pins(shapemol_ctx *c, const int32_t *h_off, int32_t n_steps, const int32_t *h_atom, const int32_t *h_nbr, int64_t n_pins, int32_t k) {
    if (!c) return fail("shapemol_set_knn_pins: null context");
    HIPCHK(hipSetDevice(c->device));
    c->drop_graphs();
    HIPCHK(hipDeviceSynchronize());
    if (c->pin_off) { hipFree(c->pin_off); hipFree(c->pin_atom); hipFree(c->pin_nbr); c->pin_off = c->pin_atom = c->pin_nbr = nullptr; }
    c->n_pins = 0; c->pin_steps = 0; c->pin_k = 0;
    if (n_pins <= 0) return 0;
    if (!h_off || !h_atom || !h_nbr || n_steps < 1 || k < 1 || k > c->KP) return fail("shapemol_set_knn_pins: bad arguments");
    if (h_off[0] != 0 || h_off[n_steps] != n_pins) return fail("shapemol_set_knn_pins: offsets do not cover the pins");
    for (int i = 0; i < n_steps; ++i) if (h_off[i + 1] < h_off[i]) return fail("shapemol_set_knn_pins: offsets must not decrease");
    HIPCHK(hipMalloc(&c->pin_off, (size_t)(n_steps + 1) * 4)); HIPCHK(hipMalloc(&c->pin_atom, (size_t)n_pins * 4)); HIPCHK(hipMalloc(&c->pin_nbr, (size_t)n_pins * k * 4));
    HIPCHK(hipMemcpy(c->pin_off, h_off, (size_t)(n_steps + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->pin_atom, h_atom, (size_t)n_pins * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->pin_nbr, h_nbr, (size_t)n_pins * k * 4, hipMemcpyHostToDevice));
    c->n_pins = n_pins; c->pin_steps = n_steps; c->pin_k = k;
    return 0;
}

int shapemol_set_guidance(shapemol_ctx *c, const double *h_cloud, int64_t n_points, double radius, int32_t grad_step, const double *d_draws) {
    if (!c) return fail("shapemol_set_guidance: null ctx");
    if (n_points < 0 || n_points > 2048 || (n_points > 0 && n_points < 3)) return fail("shapemol_set_guidance: the cloud needs 3 .. 2048 points (it is staged in LDS)");
    if (n_points > 0 && (!h_cloud || !(radius > 0.0))) return fail("shapemol_set_guidance: cloud / radius missing");
    HIPCHK(hipSetDevice(c->device));
    c->drop_graphs();                            // also drains the device: the old cloud may still be in use
    if (c->g_cloud) { hipFree(c->g_cloud); c->g_cloud = nullptr; }
    c->g_points = 0; c->g_draws = nullptr;
    if (n_points == 0) return 0;
    HIPCHK(hipMalloc((void **)&c->g_cloud, (size_t)n_points * 3 * sizeof(double)));
    HIPCHK(hipMemcpy(c->g_cloud, h_cloud, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice));
    c->g_points = n_points; c->g_radius = radius; c->g_grad_step = grad_step; c->g_draws = d_draws;
    return 0;
}

int shapemol_guide_points(shapemol_ctx *c, float *d_pos, int64_t N, const double *d_draws, uint64_t seed, void *stream) {
    if (!c || !d_pos || N < 1) return fail("shapemol_guide_points: bad argument");
    if (c->g_points <= 0) return fail("shapemol_guide_points: no cloud set (shapemol_set_guidance)");
    HIPCHK(hipSetDevice(c->device));
    if (ensure_workspace(c, std::max<int64_t>(N, 1), 1)) return 1;
    hipStream_t s = (hipStream_t)stream;
    ChainParams cp{};
    cp.seed = seed; cp.guide_draws = d_draws; cp.step_base = 0;
    LAUNCH("prep", SMK(set_chain_params_kernel, dim3(1), dim3(1), 0, s, c->chain_params, cp, c->steps));
    PcGuideArgs ga{d_pos, c->g_cloud, c->chain_params, nullptr, (int)N, (int)c->g_points, c->g_grad_step + 1, c->g_grad_step, c->g_radius, 0.2};
    LAUNCH("pc_guidance", SMK(pc_guidance_kernel, dim3((N * 16 + 255) / 256), dim3(256), (size_t)c->g_points * 24, s, ga));
    return 0;
}

int shapemol_pointcloud_guidance(const double *h_cloud, int64_t n_points, double radius, double ratio, float *d_pos, int64_t N,
                                 const double *d_draws, uint64_t seed, void *stream) {
    if (!h_cloud || !d_pos || N < 1) return fail("shapemol_pointcloud_guidance: bad argument");
    if (n_points < 3 || n_points > 2048) return fail("shapemol_pointcloud_guidance: the cloud needs 3 .. 2048 points (it is staged in LDS)");
    if (!(radius > 0.0) || !(ratio >= 0.0 && ratio < 0.8)) return fail("shapemol_pointcloud_guidance: radius must be > 0, ratio in [0, 0.8)");
    hipStream_t s = (hipStream_t)stream;
    const size_t cloud_bytes = (size_t)n_points * 3 * sizeof(double);
    unsigned char *blk = nullptr;                       // [cloud | ChainParams | step counter]
    HIPCHK(hipMalloc((void **)&blk, cloud_bytes + sizeof(ChainParams) + 16));
    double *d_cloud = reinterpret_cast<double *>(blk);
    ChainParams *d_cp = reinterpret_cast<ChainParams *>(blk + cloud_bytes);
    int *d_step = reinterpret_cast<int *>(blk + cloud_bytes + sizeof(ChainParams));
    hipError_t e = hipMemcpyAsync(d_cloud, h_cloud, cloud_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        ChainParams cp{};
        cp.seed = seed; cp.guide_draws = d_draws; cp.step_base = 0;
        hipLaunchKernelGGL(set_chain_params_kernel, dim3(1), dim3(1), 0, s, d_cp, cp, d_step);
        PcGuideArgs ga{d_pos, d_cloud, d_cp, nullptr, (int)N, (int)n_points, 1, 0, radius, ratio};      // t_first - 0 > grad_step: always guided
        hipLaunchKernelGGL(pc_guidance_kernel, dim3((N * 16 + 255) / 256), dim3(256), (size_t)n_points * 24, s, ga);
        e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize(s);       // the block is freed below; the reference's function is synchronous too
    hipFree(blk);
    if (e != hipSuccess) return fail(std::string("shapemol_pointcloud_guidance: ") + hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(std::string("shapemol_pointcloud_guidance: ") + hipGetErrorString(e2));
    return 0;
}

static int status_message(const int32_t (&f)[8]) {
    if (f[ST_BATCH]) return fail("batch vector is not sorted ascending or names a molecule >= n_mols; results are invalid");
    if (f[ST_ATOM_TYPE]) return fail("an atom type is outside [0, num_classes); results are invalid");
    if (f[ST_TIME]) return fail("a time step is outside [0, num_timesteps); results are invalid");
    if (f[ST_SPAN]) return fail("a molecule is larger than the max_mol_atoms hint says (folded coordinate update / fused graph kernel); results are invalid");
    if (f[ST_RANGE]) return fail("an activation left the fp16 range of the two-piece f16 node kernels (|x| >= 6e4 or NaN); results are invalid: set option node_f16 = 0 (exactly split bf16 kernels)");
    if (f[ST_VN_BARRIER]) return fail("grid barrier of the fused coordinate update timed out (workgroups not co-resident); results are invalid");
    return 0;
}

int shapemol_status(shapemol_ctx *c, int32_t *flags_out) {
    if (!c) return fail("shapemol_status: null ctx");
    int32_t f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    if (c->status) HIPCHK(hipMemcpy(f, c->status, sizeof(f), hipMemcpyDeviceToHost));
    if (flags_out) std::memcpy(flags_out, f, sizeof(f));
    return status_message(f);
}

int shapemol_status_stream(shapemol_ctx *c, int32_t *flags_out, void *stream) {
    if (!c) return fail("shapemol_status_stream: null ctx");
    int32_t f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (c->status) HIPCHK(hipMemcpyAsync(f, c->status, sizeof(f), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flags_out) std::memcpy(flags_out, f, sizeof(f));
    return status_message(f);
}

int shapemol_profile_begin(shapemol_ctx *c) {
    if (!c) return fail("shapemol_profile_begin: null ctx");
    for (auto &r : c->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    c->prof.clear();
    c->prof_on = true;
    return 0;
}

int shapemol_profile_end(shapemol_ctx *c, char (*names)[32], double *total_ms, int64_t *launches, int cap) {
    if (!c) { fail("shapemol_profile_end: null ctx"); return -1; }
    c->prof_on = false;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    std::vector<std::string> order;
    std::map<std::string, std::pair<double, int64_t>> acc;
    for (auto &r : c->prof) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        if (!acc.count(r.name)) order.push_back(r.name);
        acc[r.name].first += ms; acc[r.name].second += 1;
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    c->prof.clear();
    int n = 0;
    for (auto &nm : order) {
        if (n >= cap) break;
        std::snprintf(names[n], 32, "%s", nm.c_str());
        total_ms[n] = acc[nm].first; launches[n] = acc[nm].second; ++n;
    }
    return n;
}

}  // extern "C"
