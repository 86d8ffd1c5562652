// Edge attention with the second Linear of the edge MLPs on the bf16 matrix cores (round 1's kernels; option edge_bf16 = 1).
//
// Same semantics and formulation as sm_edge.h (reference: models/uni_transformer.py:48-81 for x2h,
// :121-151 for h2x).  Differences:
//   * the H x H (or heads x H) second Linear of each edge MLP is evaluated as six bf16 MFMA products of
//     exactly split operands (sm_device.h, gemm_bf16x6 / tile_bf16x6): fp32-level accuracy at ~2.7x the
//     fp32-MFMA rate, on the matrix pipe, where vector work can issue beside it;
//   * the split weights take 6 bytes per element, so only ONE full-width MLP fits in LDS: the key path and
//     the value path are separate PHASES that hand the attention weights over through a small global
//     buffer  alpha[N * KP][2][NT]  (16 floats per edge slot at H = 128):
//         key    : logits + per-atom softmax over the neighbour slots           -> alpha
//         value x: x2h values, sum_j alpha * e_w * v_ij                          -> att [N][H]
//         value h: h2x values (one per head), sum_j alpha * e_w * v_ij * rel_x   -> o3  [N][16][3]
//   * edge_fused_kernel runs both phases of an attention in one launch (image swap in LDS for x2h, both images
//     resident for h2x) and, for h2x, the VN-linear + batch-norm statistics of the coordinate update in its epilogue.
//     (Round 2 made the two-piece f16 kernels of sm_edge16.h the default, round 4 the streaming kernels of sm_edge_stream.h, which
//     use this file's exact three-piece arithmetic; the phase kernels here remain as option edge_bf16 = 1.)
// One job = the KP <= 16 neighbour slots of 16 / KP centre atoms = one 16-column tile, one job per wave where
// the jobs fit; loads of a job are issued before the weight image is copied to LDS.
#pragma once
#include "sm_device.h"
#ifndef SM_ABLATE
#define SM_ABLATE 0          // diagnostic builds only (build.sh --ablate MASK): compile-time mask of parts to drop
#endif
#ifndef SM_ABL
#define SM_ABL(bit) (((SM_ABLATE) >> (bit)) & 1)
#endif

// LDS image of ONE edge MLP, in 32-bit words
template <int H, int NT2>
struct EdgePhaseImage {
    static constexpr int NT = H / 16, NB = NT / 2;
    static constexpr int G4 = (NT + 3) / 4;             // groups of four output tiles
    static constexpr bool BF1 = NT2 > 1;                // first Linear's RBF block also on the bf16 matrix cores (the
                                                        // heads-wide value MLP of h2x keeps fp32: both images share LDS)
    static constexpr int O_WR = 0;                      // BF1: 3 pieces x [NT][64][4] u32, K = 20 padded to 32: element j of
                                                        //      lane (m, g) = piece(W1[16t + m][4j + g]) for j < 5, else 0
                                                        // else [5][G4][64][4] fp32 A fragments of W1[:, 0:20]: one 16-byte
                                                        //      LDS read feeds k-step s of tiles 4h .. 4h+3
    static constexpr int O_G = O_WR + (BF1 ? 3 * NT * 256 : 5 * G4 * 256);     // gamma[H]
    static constexpr int O_B = O_G + H;                 // beta[H]
    static constexpr int O_B2 = O_B + H;                // b2[NT2 * 16]
    static constexpr int O_W2 = O_B2 + NT2 * 16;        // 3 pieces x [NT2][NB][64][4] u32 (two bf16 each)
    static constexpr int TOTAL = O_W2 + 3 * NT2 * NB * 256;
};

// acc[t] += W1[:, 0:20] rbf for all NT tiles (fp32 MFMA, K = 20 in five steps).  The A fragments of a k-step
// are requested one step ahead: an MFMA that waits for an LDS read issued just before it pays the LDS latency
// (~100+ cycles under load) instead of its 32 issue cycles.
template <int NT>
SM_DEV void first_linear_rbf(const float *wr, const float (&rb)[5], f32x4 (&acc)[NT], int lane) {
    constexpr int G4 = (NT + 3) / 4;
    float4 cur[G4], nxt[G4];
#pragma unroll
    for (int h = 0; h < G4; ++h) cur[h] = ldg4(wr + (h * 64 + lane) * 4);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (s + 1 < 5) {
#pragma unroll
            for (int h = 0; h < G4; ++h) nxt[h] = ldg4(wr + (((s + 1) * G4 + h) * 64 + lane) * 4);
        }
#pragma unroll
        for (int h = 0; h < G4; ++h) {
            const float w4[4] = {cur[h].x, cur[h].y, cur[h].z, cur[h].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * h + q < NT) acc[4 * h + q] = mfma16(w4[q], rb[s], acc[4 * h + q]);
        }
#pragma unroll
        for (int h = 0; h < G4; ++h) cur[h] = nxt[h];
    }
}

// The same product with exactly split operands on the bf16 matrix cores: one K = 32 step (20 centres + padding),
// six piece products per tile; overlaps with vector work, which the fp32 MFMA does not.
template <int NT>
SM_DEV void first_linear_rbf_bf16(const float *wr, const float (&rb)[5], f32x4 (&acc)[NT], int lane) {
    const unsigned *w = reinterpret_cast<const unsigned *>(wr);
    u32x4 bh = {0u, 0u, 0u, 0u}, bm = {0u, 0u, 0u, 0u}, bl = {0u, 0u, 0u, 0u};
    {
        unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2;
        split3_pair(rb[0], rb[1], h0, m0, l0);
        split3_pair(rb[2], rb[3], h1, m1, l1);
        split3_pair(rb[4], 0.f, h2, m2, l2);
        bh[0] = h0; bh[1] = h1; bh[2] = h2; bm[0] = m0; bm[1] = m1; bm[2] = m2; bl[0] = l0; bl[1] = l1; bl[2] = l2;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const u32x4 ah = *reinterpret_cast<const u32x4 *>(w + ((0 * NT + t) * 64 + lane) * 4);
        const u32x4 am = *reinterpret_cast<const u32x4 *>(w + ((1 * NT + t) * 64 + lane) * 4);
        const u32x4 al = *reinterpret_cast<const u32x4 *>(w + ((2 * NT + t) * 64 + lane) * 4);
        f32x4 c = acc[t];
        c = mfma_bf16(al, bh, c);
        c = mfma_bf16(am, bm, c);
        c = mfma_bf16(ah, bl, c);
        c = mfma_bf16(am, bh, c);
        c = mfma_bf16(ah, bm, c);
        c = mfma_bf16(ah, bh, c);
        acc[t] = c;
    }
}
template <class IM, int NT>
SM_DEV void first_linear_of(const float *img, const float (&rb)[5], f32x4 (&acc)[NT], int lane) {
    if constexpr (IM::BF1) first_linear_rbf_bf16<NT>(img + IM::O_WR, rb, acc, lane);
    else first_linear_rbf<NT>(img + IM::O_WR, rb, acc, lane);
}

// -------------------------------------------------------------------------------------------------
// Fused variant: key phase and value phase of one attention in ONE launch.  alpha never crosses
// workgroups (the wave that produced the weights of a job consumes them), so the hand-over only needs
// the workgroup barrier that the weight swap in LDS needs anyway:
//   x2h : [copy K image] barrier | key phase of all jobs | barrier [copy V image over it] barrier | value phase
//   h2x : the value image is small (heads x H): both images are resident, no swap
// -------------------------------------------------------------------------------------------------
struct EdgeFusedArgs {
    const float *image_k, *image_v;
    const float *pre;       // node pre-products [N][ld_pre]: A_k | B_k | A_v | B_v at column offsets 0, H, 2H, 3H
    const float *q;         // [N][H]
    const float *x;         // [N][3]
    const int *nbr;         // [N][KP]
    const float *ew;        // [N][KP]
    float *alpha;           // [N*KP][2][NT] scratch
    float *out;             // x2h: [N][H]; h2x: [N][16][3]
    int n_atoms, ld_pre;
    unsigned long long *stamps;   // diagnostic build only
    // h2x only: the coordinate update that follows the attention (VN-linear + train-mode batch-norm + VN-leaky-ReLU,
    // shape_vn_layers.py:41-61,95-110; uni_transformer.py:153-162) fused behind it.  enable = 0 leaves it to
    // vn_stats_kernel / vn_apply_kernel (sm_misc.h), whose arithmetic this follows line by line.
    struct VnFuse {
        const float *ps;            // [B][2][heads][3]
        const float *wf_x, *wd_x;   // [heads]
        const float *wf_o, *wd_o;   // [heads][16]
        const float *bn_g, *bn_b;   // [heads]
        const int *mol_of;
        float *pd;                  // [N][heads][6]
        double *acc;                // [kVnReplicas][2][heads], zeroed at the start of the evaluation
        unsigned *arrive;           // grid-barrier counter of this launch, zeroed likewise
        int *err;                   // set to 1 if the grid barrier timed out (workgroups not co-resident)
        float *x_out;               // [N][3]
        int enable;                 // 0 off, 1 whole update (grid barrier inside), 2 VN-linear + statistics only
    } vn;
};
constexpr int kVnReplicas = 16;
// doubles of reduction scratch behind the LDS images of the h2x kernel (host and device use the same formula)
__host__ __device__ constexpr int vn_red_doubles(int nwave, int hd) {
    return nwave * 64 > (2 + 2 * kVnReplicas) * hd ? nwave * 64 : (2 + 2 * kVnReplicas) * hd;
}

// ONE = true: the launch gives every wave at most one job (jobs <= workgroups x waves; the common case up to ~6k atoms):
// the job loops become straight-line code and no next job's rows are kept in flight (fewer live registers).
// amdgpu_waves_per_eu(3, 3): the launch never has more than three waves per SIMD (<= 12 per workgroup, one workgroup per
// CU), so the register allocator may use the whole 168-VGPR budget instead of aiming at four waves (128): the
// single-job instantiation then takes 138 registers and schedules its LDS reads further ahead (+4 % on the step).
template <int H, int KP, bool H2X, bool ONE = false>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
edge_fused_kernel(EdgeFusedArgs a) {
    static_assert(KP == 8 || KP == 16, "single-tile jobs");
    constexpr int NT = H / 16;
    constexpr int NT2V = H2X ? 1 : NT;
    using IMK = EdgePhaseImage<H, NT>;
    using IMV = EdgePhaseImage<H, NT2V>;
    constexpr int V_BASE = H2X ? IMK::TOTAL : 0;       // h2x: value image behind the key image; x2h: swapped in
    constexpr int APJ = 16 / KP, SEGW = KP;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    float cen[5];
    rbf_centres(g, cen);
    const int njobs = (a.n_atoms + APJ - 1) / APJ;
    const int jstride = gridDim.x * nwave;
    const int job0 = blockIdx.x * nwave + wave;

    int atom = 0, jn = 0, edge = 0;
    bool atom_ok = false, ok = false;
    float xi[3], xj[3];
    float4 ga[NT], gb[NT];

    auto issue_loads = [&](int jb, int col_a, int col_b) {
        const int atom_raw = jb * APJ + n / SEGW;
        atom_ok = atom_raw < a.n_atoms;
        atom = atom_ok ? atom_raw : a.n_atoms - 1;
        edge = atom * KP + n % SEGW;
        const int jraw = a.nbr[edge];
        ok = atom_ok && jraw >= 0;
        jn = ok ? jraw : atom;
#pragma unroll
        for (int k = 0; k < 3; ++k) { xi[k] = a.x[atom * 3 + k]; xj[k] = a.x[jn * 3 + k]; }
        const float *pi = a.pre + (size_t)atom * a.ld_pre + col_a, *pj = a.pre + (size_t)jn * a.ld_pre + col_b;
#pragma unroll
        for (int t = 0; t < NT; ++t) { ga[t] = ldg4(pi + 16 * t + 4 * g); gb[t] = ldg4(pj + 16 * t + 4 * g); }
    };
    // hidden = ReLU(LN(A_i + B_j + W_r rbf)) with the image at `img`
    auto hidden = [&](auto im_tag, const float *img, const float (&rb)[5], float (&hid)[NT * 4], bool tick) {
        using IM = decltype(im_tag);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = f32x4{ga[t].x + gb[t].x, ga[t].y + gb[t].y, ga[t].z + gb[t].z, ga[t].w + gb[t].w};
        if (tick) SM_TICK(a.stamps, 2);
        first_linear_of<IM, NT>(img, rb, acc, lane);
        if (tick) SM_TICK(a.stamps, 3);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            hid[4 * t + 0] = acc[t][0]; hid[4 * t + 1] = acc[t][1];
            hid[4 * t + 2] = acc[t][2]; hid[4 * t + 3] = acc[t][3];
        }
        ln_relu_dlayout<NT>(hid, img + IM::O_G, img + IM::O_B, g);
    };

    int job = job0;
    bool have = job < njobs;
    SM_TICK(a.stamps, 0);
    if (have) issue_loads(job, 0, H);
    // (LDS-DMA since round 2: the unrolled copy loop degenerated to seven dependent rounds per image, see sm_device.h)
    if (!SM_ABL(8)) image_to_lds(lds, a.image_k, IMK::TOTAL / 4, threadIdx.x, wave, nwave, lane);
    if constexpr (H2X) { if (!SM_ABL(9)) image_to_lds(lds + V_BASE, a.image_v, IMV::TOTAL / 4, threadIdx.x, wave, nwave, lane); }
    __syncthreads();
    SM_TICK(a.stamps, 1);
    if (SM_ABL(10)) have = false;

    // ---- key phase ---------------------------------------------------------------------------------
    bool first = true;
    while (have) {   // (a single pass when ONE)
        if constexpr (!ONE) asm volatile("" ::: "memory");   // keep the loop-invariant LDS weight reads inside the loop
        if (!first) issue_loads(job, 0, H);          // later jobs: rows requested here, not a job ahead (68 registers less across the body)
        first = false;
        const float rel[3] = {xi[0] - xj[0], xi[1] - xj[1], xi[2] - xj[2]};
        float rb[5];
        rbf_dlayout(sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]), cen, rb);
        float hid[NT * 4];
        hidden(IMK{}, lds, rb, hid, true);
        SM_TICK(a.stamps, 4);
        // second Linear one pair of head blocks (t, t + NT/2) at a time, software-pipelined: the logits and
        // softmax of a finished pair are independent vector work placed under the next pair's MFMAs
        float alpha[NT / 2];
        {
            u32x4 bh[NT / 2], bm[NT / 2], bl[NT / 2];
            split_act<NT>(hid, bh, bm, bl);
            SM_TICK(a.stamps, 5);
            const unsigned *w2 = reinterpret_cast<const unsigned *>(lds) + IMK::O_W2;
            auto pair_of = [&](int t, f32x4 &ka, f32x4 &kb) {
                const float4 ba = ldg4(lds + IMK::O_B2 + 16 * t + 4 * g), bb = ldg4(lds + IMK::O_B2 + 16 * (t + NT / 2) + 4 * g);
                ka = tile_bf16x6<NT, NT>(w2, t, bh, bm, bl, f32x4{ba.x, ba.y, ba.z, ba.w}, lane);
                kb = tile_bf16x6<NT, NT>(w2, t + NT / 2, bh, bm, bl, f32x4{bb.x, bb.y, bb.z, bb.w}, lane);
            };
            const float *qrow = a.q + (size_t)atom * H + 4 * g;       // query rows (L2-hot), requested a pair ahead
            float4 qa = ldg4(qrow), qb = ldg4(qrow + 16 * (NT / 2));
            f32x4 ka, kb;
            pair_of(0, ka, kb);
#pragma unroll
            for (int t = 0; t < NT / 2; ++t) {
                f32x4 na = ka, nb = kb;
                float4 nqa = qa, nqb = qb;
                if (t + 1 < NT / 2) {
                    nqa = ldg4(qrow + 16 * (t + 1)); nqb = ldg4(qrow + 16 * (t + 1 + NT / 2));
                    pair_of(t + 1, na, nb);
                }
                alpha[t] = attention_weight_pair<NT, SEGW>(qa, qb, ka, kb, ok);
                ka = na; kb = nb; qa = nqa; qb = nqb;
            }
        }
        if (atom_ok) {
            float *ap = a.alpha + (size_t)edge * 2 * NT + (g >> 1) * NT + (NT / 2) * (g & 1);
            if constexpr (NT % 8 == 0) {
#pragma unroll
                for (int i = 0; i < NT / 8; ++i)
                    stg4(ap + 4 * i, float4{alpha[4 * i], alpha[4 * i + 1], alpha[4 * i + 2], alpha[4 * i + 3]});
            } else {
#pragma unroll
                for (int t = 0; t < NT / 2; ++t) ap[t] = alpha[t];
            }
        }
        SM_TICK(a.stamps, 6);
        if constexpr (ONE) break;
        job += jstride;
        have = job < njobs;
    }

    // ---- hand-over: value-phase loads of the first job fly across the weight swap ------------------
    job = job0;
    have = job < njobs;
    if (have) issue_loads(job, 2 * H, 3 * H);
    __syncthreads();                         // key weights no longer needed; alpha stores of this wave drained
    if constexpr (!H2X) {
        if (!SM_ABL(9)) image_to_lds(lds, a.image_v, IMV::TOTAL / 4, threadIdx.x, wave, nwave, lane);
        __syncthreads();
    }
    SM_TICK(a.stamps, 7);
    if (SM_ABL(11)) have = false;

    // ---- value phase -------------------------------------------------------------------------------
    const float *imv = lds + V_BASE;
    constexpr int HD = H / 8;                                     // heads = VN channels
    // fused coordinate update (h2x): lane = (atom of the job, channel)
    double *vn_red = reinterpret_cast<double *>(lds + V_BASE + IMV::TOTAL);   // [nwave][32][2] behind the images; the grid-barrier
                                                                              // epilogue stages [2 + 2 * kVnReplicas][HD] doubles there
    float *vn_o = reinterpret_cast<float *>(vn_red + vn_red_doubles(nwave, HD)) + wave * (APJ * 48);   // this wave's attention rows [APJ][16][3]
    const bool one_job = njobs <= jstride;                                     // every wave has at most one job
    first = true;
    while (have) {
        if constexpr (!ONE) asm volatile("" ::: "memory");
        if (!first) issue_loads(job, 2 * H, 3 * H);
        first = false;
        const float rel[3] = {xi[0] - xj[0], xi[1] - xj[1], xi[2] - xj[2]};
        float rb[5];
        rbf_dlayout(sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]), cen, rb);
        const float w = ok ? a.ew[edge] : 0.f;
        const float *ap = a.alpha + (size_t)edge * 2 * NT + (g >> 1) * NT;
        float al[NT];
        if constexpr (!H2X) {
#pragma unroll
            for (int t = 0; t < NT; ++t) al[t] = ap[t];
        } else {
#pragma unroll
            for (int r = 0; r < NT; ++r) al[r] = r < NT / 2 ? ap[(NT / 2) * (g & 1) + r] : 0.f;
        }
        float hid[NT * 4];
        hidden(IMV{}, imv, rb, hid, false);
        u32x4 bh[NT / 2], bm[NT / 2], bl[NT / 2];
        split_act<NT>(hid, bh, bm, bl);
        const unsigned *w2v = reinterpret_cast<const unsigned *>(imv) + IMV::O_W2;
        const int cur_atom = atom;
        const bool store = atom_ok && (n % SEGW) == 0;
        if constexpr (!H2X) {
            // one output tile at a time: the weighted neighbour sum of tile t issues under the MFMAs of tile t + 1
            float *op = a.out + (size_t)cur_atom * H;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 b2 = ldg4(imv + IMV::O_B2 + 16 * t + 4 * g);
                const f32x4 v = tile_bf16x6<NT, NT>(w2v, t, bh, bm, bl, f32x4{b2.x, b2.y, b2.z, b2.w}, lane);
                const float aw = al[t] * w;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = seg_sum<SEGW>(aw * v[r]);
                if (store) stg4(op + 16 * t + 4 * g, float4{o[0], o[1], o[2], o[3]});
            }
        } else {
            f32x4 vacc[1];
            {
                const float4 b2 = ldg4(imv + IMV::O_B2 + 4 * g);
                vacc[0] = tile_bf16x6<NT, 1>(w2v, 0, bh, bm, bl, f32x4{b2.x, b2.y, b2.z, b2.w}, lane);
            }
            // value row 4g + r belongs to head 2*((NT/2)*(g&1) + r) + (g>>1); rows with r >= NT/2 are padding
            float o[12];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float av = r < NT / 2 ? al[r % NT] * w * vacc[0][r] : 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) o[3 * r + k] = seg_sum<SEGW>(av * rel[k]);
            }
            if (store) {
                float *op = a.out + (size_t)cur_atom * 48 + 12 * g;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    stg4(op + 4 * i, float4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]});
                if (a.vn.enable && one_job) {           // keep the rows at hand for the VN-linear below (no L2 round trip)
                    float *ol = vn_o + (n / SEGW) * 48 + 12 * g;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        stg4(ol + 4 * i, float4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]});
                }
            }
        }
        if constexpr (ONE) break;
        job += jstride;
        have = job < njobs;
    }

    if constexpr (H2X) {
        if (!a.vn.enable) return;
        const int v_al = lane >> 4, v_c = lane & 15;               // lane = (atom of the job, channel)
        const bool v_lane = v_al < APJ && v_c < HD;
        double v_s1 = 0.0, v_s2 = 0.0;
        // ---- VN-linear of this wave's atoms: p, d per channel from the 16 attention rows (+ x, + shape term);
        //      lane = (atom of the job, channel).  The rows were stored by this wave above (same CU, write-through).
        if (!one_job) __syncthreads();               // rows come back through L2: drain this workgroup's stores first
        for (int jb = job0; jb < njobs; jb += jstride) {
            const int va = jb * APJ + v_al;
            if (v_lane && va < a.n_atoms) {
                float orow[48];
                if (one_job) {
                    const float *ov = vn_o + v_al * 48;
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const float4 t = ldg4(ov + 4 * i);
                        orow[4 * i] = t.x; orow[4 * i + 1] = t.y; orow[4 * i + 2] = t.z; orow[4 * i + 3] = t.w;
                    }
                } else {
                    const float *ov = a.out + (size_t)va * 48;
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const float4 t = ldg4(ov + 4 * i);
                        orow[4 * i] = t.x; orow[4 * i + 1] = t.y; orow[4 * i + 2] = t.z; orow[4 * i + 3] = t.w;
                    }
                }
                float wf[16], wd[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 t = ldg4(a.vn.wf_o + v_c * 16 + 4 * i), u = ldg4(a.vn.wd_o + v_c * 16 + 4 * i);
                    wf[4 * i] = t.x; wf[4 * i + 1] = t.y; wf[4 * i + 2] = t.z; wf[4 * i + 3] = t.w;
                    wd[4 * i] = u.x; wd[4 * i + 1] = u.y; wd[4 * i + 2] = u.z; wd[4 * i + 3] = u.w;
                }
                const float *psf = a.vn.ps + ((size_t)a.vn.mol_of[va] * 2 * HD + v_c) * 3;
                const float *psd = psf + HD * 3;
                const float wfx = a.vn.wf_x[v_c], wdx = a.vn.wd_x[v_c];
                float p[3], d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float xk = a.x[va * 3 + k];
                    float pp = wfx * xk, dd = wdx * xk;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        pp += wf[r] * orow[r * 3 + k];
                        dd += wd[r] * orow[r * 3 + k];
                    }
                    p[k] = pp + psf[k];
                    d[k] = dd + psd[k];
                }
                float *out = a.vn.pd + ((size_t)va * HD + v_c) * 6;
                out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
                const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
                v_s1 += (double)nrm;
                v_s2 += (double)nrm * (double)nrm;
            }
        }
        // ---- batch statistics of ||p|| over ALL atoms of the batch: workgroup sums -> replicated double atomics ----
        if (lane < 32) { vn_red[(wave * 32 + lane) * 2] = v_s1; vn_red[(wave * 32 + lane) * 2 + 1] = v_s2; }
        __syncthreads();
        if (threadIdx.x < HD) {
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < nwave; ++w)
                for (int al = 0; al < APJ; ++al) {
                    s1 += vn_red[(w * 32 + al * 16 + threadIdx.x) * 2];
                    s2 += vn_red[(w * 32 + al * 16 + threadIdx.x) * 2 + 1];
                }
            double *acc = a.vn.acc + (size_t)(blockIdx.x % kVnReplicas) * 2 * HD;
            atomicAdd(acc + threadIdx.x, s1);
            atomicAdd(acc + HD + threadIdx.x, s2);
        }
        if (a.vn.enable == 2) return;        // statistics only: vn_apply_kernel follows as its own launch
        // ---- grid barrier: every workgroup is resident (grid <= CUs, one workgroup per CU), so arrival
        //      counting cannot deadlock; a bounded wait turns a violated assumption into an error flag.
        //      Only device-scope atomics cross workgroups here (the sums and the counter): no cache fence is
        //      needed, just the order "sums acknowledged (vmcnt drained by the barrier) -> arrive". ----
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(a.vn.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(a.vn.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 4000000) { *a.vn.err = 1; break; }
            }
        }
        __syncthreads();
        double *stat = vn_red;                                    // [2][HD] totals, behind the staged replicas
        for (int i = threadIdx.x; i < kVnReplicas * 2 * HD; i += blockDim.x)      // one coherent load per thread
            vn_red[2 * HD + i] = __hip_atomic_load(a.vn.acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x < 2 * HD) {
            double t = 0.0;
            for (int r = 0; r < kVnReplicas; ++r) t += vn_red[2 * HD + r * 2 * HD + threadIdx.x];
            stat[threadIdx.x] = t;
        }
        __syncthreads();
        // ---- normalise, VN-leaky-ReLU, mean over channels, x update (vn_apply_kernel) -------------------
        float meanf = 0.f, rstd = 0.f, bng = 0.f, bnb = 0.f;
        if (v_lane) {
            const double cnt = (double)a.n_atoms;
            const double mean = stat[v_c] / cnt;
            double var = stat[HD + v_c] / cnt - mean * mean;
            var = var > 0.0 ? var : 0.0;
            meanf = (float)mean;
            rstd = 1.0f / sqrtf((float)var + 1e-5f);
            bng = a.vn.bn_g[v_c]; bnb = a.vn.bn_b[v_c];
        }
        for (int jb = job0; jb < njobs; jb += jstride) {
            const int va = jb * APJ + v_al;
            const bool on = v_lane && va < a.n_atoms;
            float o[3] = {0.f, 0.f, 0.f};
            if (on) {
                const float *pd = a.vn.pd + ((size_t)va * HD + v_c) * 6;
                float p[3] = {pd[0], pd[1], pd[2]};
                const float d[3] = {pd[3], pd[4], pd[5]};
                const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
                const float nbn = (nrm - meanf) * rstd * bng + bnb;
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
            float res[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) res[k] = seg_sum<HD>(o[k]);           // over the channels of the atom
            if (on && v_c < 3) {
                const float r = v_c == 0 ? res[0] : (v_c == 1 ? res[1] : res[2]);
                const float *ob = a.out + (size_t)va * 48;
                float att = 0.f;
                for (int rr = 0; rr < 16; ++rr) att += ob[rr * 3 + v_c];       // padding rows are zero
                a.vn.x_out[va * 3 + v_c] = a.x[va * 3 + v_c] + (att / HD + r / HD);
            }
        }
    }
}
