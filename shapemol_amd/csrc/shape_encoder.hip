// C ABI of the device shape encoder (include/shapemol_hip.h, shapemol_se_*): VN_DGCNN_Encoder.forward of the reference
// (models/shape_pointcloud_modelAE.py:207-255).  Kernels: sm_shape.h.
#include "../../include/shapemol_hip.h"
#include "sm_shape.h"

#include <cstring>
#include <string>
#include <vector>

extern "C" void shapemol_set_error_(const char *msg);     // shapemol_hip.hip: stores the thread's last error

namespace {
int se_fail(const std::string &m) { shapemol_set_error_(m.c_str()); return 1; }
#define SECHK(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return se_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
constexpr int kC = 128;           // hidden_dim of the shipped shape auto-encoder (se_model.pt: config.model.hidden_dim)
}  // namespace

struct shapemol_se_ctx {
    int C = kC, LAT = 32, L = 4, device = 0;
    float *d_w = nullptr;
    size_t o_pos_wf = 0, o_pos_g = 0, o_pos_b = 0, o_pos_wd = 0, o_c_wf = 0, o_c_g = 0, o_c_b = 0, o_c_wd = 0;
    std::vector<size_t> o_img, o_g, o_b;
    // workspace
    int64_t capP = 0;
    float *h0 = nullptr, *hcat = nullptr, *y = nullptr, *xx = nullptr, *pd = nullptr;
    int *idx = nullptr;
    double *acc = nullptr;
};

extern "C" {

size_t shapemol_se_weight_count(int32_t hidden_dim, int32_t latent_dim, int32_t layer_num) {
    const size_t C = hidden_dim, LAT = latent_dim, L = layer_num;
    return (2 * C + 2 * C + 2 * C) + L * (2 * C * C + 2 * C + 2 * C * C) + (LAT * L * C + 2 * LAT + L * C);
}

int shapemol_se_create(int32_t hidden_dim, int32_t latent_dim, int32_t layer_num, int32_t num_k, const float *w, size_t n_weights,
                       int device, shapemol_se_ctx **out) {
    if (!w || !out) return se_fail("shapemol_se_create: null argument");
    if (hidden_dim != kC) return se_fail("shapemol_se_create: hidden_dim must be 128");
    if (num_k != kSeK) return se_fail("shapemol_se_create: num_k must be 20");
    if (latent_dim < 1 || latent_dim > 256 || layer_num < 1 || layer_num > 8) return se_fail("shapemol_se_create: latent_dim / layer_num out of range");
    if (n_weights != shapemol_se_weight_count(hidden_dim, latent_dim, layer_num)) return se_fail("shapemol_se_create: weight count mismatch");
    int ndev = 0;
    SECHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return se_fail("shapemol_se_create: no such HIP device");
    SECHK(hipSetDevice(device));
    auto *c = new shapemol_se_ctx();
    c->LAT = latent_dim; c->L = layer_num; c->device = device;
    const int C = kC;
    std::vector<float> img;
    auto put = [&](const float *src, size_t n) { const size_t o = (img.size() + 63) & ~size_t(63); img.resize(o + n); std::memcpy(&img[o], src, n * 4); return o; };
    const float *p = w;
    c->o_pos_wf = put(p, 2 * C); p += 2 * C;
    c->o_pos_g = put(p, C); p += C;
    c->o_pos_b = put(p, C); p += C;
    c->o_pos_wd = put(p, 2 * C); p += 2 * C;
    for (int l = 0; l < layer_num; ++l) {
        const float *wf = p; p += (size_t)C * 2 * C;
        c->o_g.push_back(put(p, C)); p += C;
        c->o_b.push_back(put(p, C)); p += C;
        const float *wd = p; p += (size_t)C * 2 * C;
        // W' = [Wf1 ; Wf2 - Wf1 ; Wd1 ; Wd2 - Wd1]  (4C x C), as A fragments of se_point_linear_kernel
        std::vector<float> Wp((size_t)4 * C * C);
        for (int m = 0; m < C; ++m)
            for (int k = 0; k < C; ++k) {
                Wp[(size_t)(0 * C + m) * C + k] = wf[(size_t)m * 2 * C + k];
                Wp[(size_t)(1 * C + m) * C + k] = wf[(size_t)m * 2 * C + C + k] - wf[(size_t)m * 2 * C + k];
                Wp[(size_t)(2 * C + m) * C + k] = wd[(size_t)m * 2 * C + k];
                Wp[(size_t)(3 * C + m) * C + k] = wd[(size_t)m * 2 * C + C + k] - wd[(size_t)m * 2 * C + k];
            }
        std::vector<float> im((size_t)4 * C * C);
        for (int t = 0; t < 4 * C / 16; ++t)
            for (int s4 = 0; s4 < C / 16; ++s4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r)
                        im[((size_t)(t * (C / 16) + s4) * 64 + lane) * 4 + r] = Wp[(size_t)(16 * t + (lane & 15)) * C + 4 * (4 * s4 + r) + (lane >> 4)];
        c->o_img.push_back(put(im.data(), im.size()));
    }
    c->o_c_wf = put(p, (size_t)latent_dim * layer_num * C); p += (size_t)latent_dim * layer_num * C;
    c->o_c_g = put(p, latent_dim); p += latent_dim;
    c->o_c_b = put(p, latent_dim); p += latent_dim;
    c->o_c_wd = put(p, (size_t)layer_num * C); p += (size_t)layer_num * C;
    if (hipMalloc((void **)&c->d_w, img.size() * 4) != hipSuccess || hipMemcpy(c->d_w, img.data(), img.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        delete c; return se_fail("shapemol_se_create: device allocation failed");
    }
    *out = c;
    return 0;
}

void shapemol_se_destroy(shapemol_se_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (void *q : {(void *)c->h0, (void *)c->hcat, (void *)c->y, (void *)c->xx, (void *)c->pd, (void *)c->idx, (void *)c->acc, (void *)c->d_w}) if (q) hipFree(q);
    delete c;
}

int shapemol_se_encode(shapemol_se_ctx *c, const float *d_points, int64_t B, int64_t N, float *d_out, void *stream) {
    if (!c || !d_points || !d_out) return se_fail("shapemol_se_encode: null argument");
    if (B < 1 || N < kSeK || (N % 16) != 0 || N > 8192 || B * N > (1 << 24)) return se_fail("shapemol_se_encode: need B >= 1, N a multiple of 16 in [32, 8192]");
    SECHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int C = c->C, L = c->L, LAT = c->LAT;
    const int64_t P = B * N;
    if (P > c->capP) {
        SECHK(hipDeviceSynchronize());
        for (void *q : {(void *)c->h0, (void *)c->hcat, (void *)c->y, (void *)c->xx, (void *)c->pd, (void *)c->idx, (void *)c->acc}) if (q) hipFree(q);
        SECHK(hipMalloc((void **)&c->h0, P * C * 3 * 4)); SECHK(hipMalloc((void **)&c->hcat, P * L * C * 3 * 4));
        SECHK(hipMalloc((void **)&c->y, P * 4 * C * 3 * 4)); SECHK(hipMalloc((void **)&c->xx, P * 4));
        SECHK(hipMalloc((void **)&c->pd, P * (LAT + 1) * 3 * 4)); SECHK(hipMalloc((void **)&c->idx, P * kSeK * 4));
        SECHK(hipMalloc((void **)&c->acc, (size_t)kSeReplicas * 2 * 256 * 8));
        c->capP = P;
    }
    const size_t knn_lds = (size_t)16 * N * sizeof(float);
    SECHK(hipFuncSetAttribute((const void *)se_knn_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_lds));
    SECHK(hipFuncSetAttribute((const void *)se_knn_kernel<3 * kC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_lds));
    const float *W = c->d_w;
    const dim3 knn_grid((unsigned)(N / 16), (unsigned)B);
    // conv_pos on the raw points
    hipLaunchKernelGGL(se_knn_kernel<3>, knn_grid, dim3(256), knn_lds, s, d_points, 3, nullptr, (int)N, c->idx);
    SeEdgeArgs e{};
    e.x = d_points; e.w0f = W + c->o_pos_wf; e.w0d = W + c->o_pos_wd; e.idx = c->idx; e.bn_g = W + c->o_pos_g; e.bn_b = W + c->o_pos_b;
    e.acc = c->acc; e.h_out = c->h0; e.n_total = (int)P; e.N = (int)N; e.C = C; e.h_ld = 3 * C; e.h_off = 0;
    SECHK(hipMemsetAsync(c->acc, 0, (size_t)kSeReplicas * 2 * 256 * 8, s));
    hipLaunchKernelGGL(se_edge_stats_kernel<true>, dim3((unsigned)P), dim3(C), 0, s, e);
    hipLaunchKernelGGL(se_edge_apply_kernel<true>, dim3((unsigned)P), dim3(C), 0, s, e);
    // DGCNN blocks
    for (int l = 0; l < L; ++l) {
        const float *hin = l == 0 ? c->h0 : c->hcat + (size_t)(l - 1) * 3 * C;
        const int ld = l == 0 ? 3 * C : L * 3 * C;
        hipLaunchKernelGGL(se_sqnorm_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, s, hin, (int)P, 3 * C, ld, c->xx);
        hipLaunchKernelGGL(se_knn_kernel<3 * kC>, knn_grid, dim3(256), knn_lds, s, hin, ld, c->xx, (int)N, c->idx);
        hipLaunchKernelGGL(se_point_linear_kernel<kC>, dim3((unsigned)((P * 3 + 15) / 16)), dim3(256), 0, s, hin, ld, W + c->o_img[l], (int)(P * 3), 4 * C, c->y);
        SeEdgeArgs b{};
        b.y = c->y; b.idx = c->idx; b.bn_g = W + c->o_g[l]; b.bn_b = W + c->o_b[l]; b.acc = c->acc; b.h_out = c->hcat;
        b.n_total = (int)P; b.N = (int)N; b.C = C; b.h_ld = L * 3 * C; b.h_off = l * 3 * C;
        SECHK(hipMemsetAsync(c->acc, 0, (size_t)kSeReplicas * 2 * 256 * 8, s));
        hipLaunchKernelGGL(se_edge_stats_kernel<false>, dim3((unsigned)P), dim3(C), 0, s, b);
        hipLaunchKernelGGL(se_edge_apply_kernel<false>, dim3((unsigned)P), dim3(C), 0, s, b);
    }
    // conv_c + mean over the points
    SeHeadArgs ha{c->hcat, W + c->o_c_wf, W + c->o_c_wd, W + c->o_c_g, W + c->o_c_b, c->pd, c->acc, d_out, (int)P, (int)N, L * C, LAT};
    SECHK(hipMemsetAsync(c->acc, 0, (size_t)kSeReplicas * 2 * 256 * 8, s));
    hipLaunchKernelGGL(se_head_linear_kernel, dim3((unsigned)((P * (LAT + 1) + 3) / 4)), dim3(256), 0, s, ha);
    hipLaunchKernelGGL(se_head_apply_kernel, dim3((unsigned)(B * LAT)), dim3(256), 0, s, ha);
    SECHK(hipGetLastError());
    return 0;
}

}  // extern "C"
