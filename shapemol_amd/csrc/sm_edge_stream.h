// Edge attention as a producer / consumer pipeline inside the workgroup (round 4): the reference-precision kernels.
// Same semantics and formulation as sm_edge_bf16.h / sm_edge16.h (reference: models/uni_transformer.py:48-81 for x2h,
// :121-151 for h2x; MLP block models/common.py:47-67): per edge the key and value MLPs on [rbf | h_i | h_j | s_i] with the
// first Linear factorised into gathered per-node products plus an RBF block, per-head softmax over the neighbour slots of
// the centre atom, weighted neighbour sums.
//
// Arithmetic: every matrix operand is split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits = the 24 of fp32, by
// truncation, full fp32 exponent range: exact for every value of magnitude >= 2^-110, no scaling) and a product is the six piece products of total
// order <= 2 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (sm_device.h); each dropped term (mid x lo, lo x mid) is below 2^-24 |x w|.
//
// Structure (what is new): weights stationary in REGISTERS, activations streaming through LDS.
//   * wave t2 < H / 16 is a CONSUMER: it holds row block t2 (16 output features = heads 2 t2, 2 t2 + 1) of the key and the
//     value MLP's second Linear as A fragments in registers (3 pieces x H / 32 k-steps x 4 VGPRs per MLP) for the whole
//     launch and applies them to every 16-column edge tile of the workgroup.  Because a head's 8 dimensions live inside one
//     row block, the wave owns its two heads end to end: logits (q . k), softmax over the centre atom's neighbour slots
//     (DPP row segments), value rows, weighted neighbour sums, store -- no exchange between waves, no permutes.
//   * four PRODUCER waves (one per SIMD) turn an edge tile into the consumers' B operand: gather the per-node products of
//     the centre atom and the neighbour, RBF block of the first Linear (six products, K = 32), LayerNorm + ReLU in
//     registers, exact three-way split, one ds_write_b128 per fragment.  A producer owns one MLP (key or value) of one of
//     the two tiles of a round; the next round's rows are requested as soon as the current ones are consumed.
//   * rounds of two tiles, double-buffered in LDS, one workgroup barrier per round.  The vector-bound half of the work
//     (LayerNorm, split, softmax) and the matrix-bound half (second Linears) run in DIFFERENT waves of the same SIMD, so the
//     two pipes overlap by construction instead of alternating inside one wave's instruction stream (sm_edge16.h: VALU 40 %,
//     MFMA 22 % busy, 30 % of the matrix cycles overlapped), and a weight fragment is never re-read: the LDS traffic is
//     the activations' (one 16-byte read per lane feeds two matrix instructions).
//   * h2x: the heads-wide value MLP (16 x H second Linear) is finished by its producer (weights in LDS, 12 KB), which hands
//     the per-head values times edge weight and the relative positions to the consumers through the tile slot.
// The same launch serves any batch size: a workgroup owns `chunk` consecutive tiles and loops over them.
#pragma once
#include "sm_edge16.h"
#include <type_traits>

struct EdgeStreamArgs {
    const float *part_k, *part_v;     // producer parts of the key / value MLP (StreamMap::P_*), copied to LDS
    const unsigned *w2k, *w2v;        // second Linears as three bf16 pieces [3][NT2][NB][64][4] u32 (consumers' registers; x2h: both)
    const float *b2v;                 // x2h: bias of the value MLP's second Linear [H]
    const float *pre;                 // node pre-products [N][ld_pre]: A_k | B_k | A_v | B_v at column offsets 0, H, 2H, 3H
    const float *q;                   // [N][H]
    const float *x;                   // [N][3]
    const int *nbr;                   // [N][KP]
    const float *ew;                  // [N][KP]
    float *out;                       // x2h: [N][H]; h2x: [N][16][3] (rows in the order of head_of_row)
    int n_atoms, ld_pre;
    int chunk;                        // consecutive tiles per workgroup
    unsigned long long *stamps;       // diagnostic build only
    EdgeFusedArgs::VnFuse vn;         // h2x: VN-linear + batch statistics behind the attention (enable = 0 or 2)
    float *xsum;                      // h2x with vn.enable: [N][3] sum of the attention rows per atom (for a following fold), or nullptr
    VnFold vf;                        // x2h: coordinate update of the previous layer in the prologue
    float *part_ms;                   // KP = 32: [2 N][heads][2] running max and sum of every half-atom tile's softmax (sm_edge16.h)
};

constexpr int kStreamProducers = 4, kStreamTPR = 2;      // producer waves; tiles per round
#ifndef SM_STREAM_ABL
#define SM_STREAM_ABL 0          // timing attribution builds only (bit 0: consumers skip their matrix products, 1: producers skip the
#endif                           // first Linear's products, 2: producers skip LayerNorm, 3: producers skip the split, 4: no s_setprio for producers,
                                 // 5: LayerNorm gain / shift not read from LDS, 6: first Linear's weight fragments not read from LDS, 7: every gather reads
                                 // row 0, 8: consumers do not read the hidden fragments, 9: consumers skip softmax and sums, 11 / 12: no A[i] / B[j]
                                 // gathers.  Ablations change the numbers a chain produces: compare per-kernel times, not step times)
#define SM_SABL(bit) (((SM_STREAM_ABL) >> (bit)) & 1)
// Attribution build (tools/kprof_stream.py): every wave accumulates where its time goes -- the work of its rounds, waiting for its
// gathers, waiting at the round barriers -- and writes the sums once at its end (stamps[row of the wave][0..7], 100 MHz ticks).
#ifdef SM_STREAM_PROF
#define SM_PCLK(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define SM_PROF(x) x
#else
#define SM_PROF(x)
#endif

// dynamic LDS map in 32-bit words
template <int H, bool H2X>
struct StreamMap {
    static constexpr int NT = H / 16, NB = NT / 2, NM = H2X ? 1 : 2, TPR = kStreamTPR;
    // producer part of one MLP (the global image is copied verbatim)
    static constexpr int P_W1 = 0;                               // RBF block of the first Linear: [3 pieces][NT][64][4] u32 (K = 20 -> 32: whole A
                                                                 // fragments, the fourth word zero: one 16-byte LDS read per fragment
                                                                 // instead of an 8- and a 4-byte one -- 24 fewer LDS instructions per unit)
    static constexpr int P_G = 3 * NT * 256, P_B = P_G + H, P_B2 = P_B + H;      // gamma[H] | beta[H] | b2 (h2x value: [16])
    static constexpr int P_W2 = P_B2 + H;                        // h2x value only: [3 pieces][NB][64][4] u32, rows = heads
    static constexpr int PART_K = (P_W2 + 255) / 256 * 256;
    static constexpr int PART_V = (P_W2 + (H2X ? 3 * NB * 256 : 0) + 255) / 256 * 256;
    static constexpr int O_K = 0, O_V = PART_K;
    static constexpr int HID_MLP = 3 * NB * 256, HID_TILE = NM * HID_MLP;        // B fragments [piece][NB][64] u32x4 per MLP
    static constexpr int O_HID = O_V + PART_V;                   // [2 buffers][TPR][NM]
    static constexpr int Q_TILE = 2 * H;                         // query rows of the tile's centre atoms
    static constexpr int O_Q = O_HID + 2 * TPR * HID_TILE;
    static constexpr int O_W = O_Q + 2 * TPR * Q_TILE;           // [2][TPR][16] edge weight of the column, -1 = no edge
    static constexpr int VT_TILE = H2X ? 16 * 16 + 64 : 0;       // h2x: value x edge weight [16 heads][16 columns] | rel [3][16]
    static constexpr int O_VT = O_W + 2 * TPR * 16;
    static constexpr int O_TAIL = O_VT + 2 * TPR * VT_TILE;      // x2h: folded coordinate table; h2x: reduction scratch
};

// o3 row (sm_edge16.h / head_of_row on the host) that holds head `head`
template <int NT>
SM_DEV int stream_row_of_head(int head) {
    constexpr int HB = NT / 2 > 0 ? NT / 2 : 1;
    const int blk = head >> 1, gp = 2 * (head & 1) + blk / HB;
    return 4 * gp + blk % HB;
}

SM_DEV f32x4 mfma_bf16x6(const u32x4 &ah, const u32x4 &am, const u32x4 &al, const u32x4 &xh, const u32x4 &xm, const u32x4 &xl,
                         f32x4 &small, f32x4 big) {
    if (SM_SABL(0)) { big[0] += __builtin_bit_cast(float, ah[0] ^ am[1] ^ al[2] ^ xh[3] ^ xm[0] ^ xl[1]); return big; }
    small = mfma_bf16(al, xh, small);      // the three terms of order 2 in one chain ...
    small = mfma_bf16(am, xm, small);
    small = mfma_bf16(ah, xl, small);
    big = mfma_bf16(am, xh, big);          // ... the leading ones in another: two independent accumulation chains
    big = mfma_bf16(ah, xm, big);
    big = mfma_bf16(ah, xh, big);
    return big;
}

// LayerNorm (eps 1e-5, biased variance, two-pass) + ReLU of one D-layout column vector, as ln_relu_dlayout (sm_device.h), written
// for a wave that has the SIMD's vector pipe to itself (a producer): the two reductions run as four independent partial sums
// (eight-term chains instead of 32-term ones -- a lone wave pays the full latency of every dependent instruction), gamma and beta
// come from LDS.
template <int NT>
SM_DEV void ln_relu_stream(float (&v)[NT * 4], const float *gamma, const float *beta, int g) {
    constexpr int H = NT * 16, W = NT >= 4 ? 4 : NT;
    float ps[W];
#pragma unroll
    for (int w = 0; w < W; ++w) ps[w] = 0.f;
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) ps[i % W] += v[i];
    float s = ps[0];
#pragma unroll
    for (int w = 1; w < W; ++w) s += ps[w];
    const float mean = sum_groups(s) * (1.0f / H);
    float pq[W];
#pragma unroll
    for (int w = 0; w < W; ++w) pq[w] = 0.f;
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) { const float d = v[i] - mean; pq[i % W] += d * d; }
    float q = pq[0];
#pragma unroll
    for (int w = 1; w < W; ++w) q += pq[w];
    const float var = sum_groups(q) * (1.0f / H);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 ga = SM_SABL(5) ? float4{1.f + g, 1.5f, 2.f + t, 0.5f} : ldg4(gamma + 16 * t + 4 * g);
        const float4 be = SM_SABL(5) ? float4{0.1f, 0.2f + g, 0.3f, 0.4f + t} : ldg4(beta + 16 * t + 4 * g);
        v[4 * t + 0] = fmaxf((v[4 * t + 0] - mean) * rstd * ga.x + be.x, 0.f);
        v[4 * t + 1] = fmaxf((v[4 * t + 1] - mean) * rstd * ga.y + be.y, 0.f);
        v[4 * t + 2] = fmaxf((v[4 * t + 2] - mean) * rstd * ga.z + be.z, 0.f);
        v[4 * t + 3] = fmaxf((v[4 * t + 3] - mean) * rstd * ga.w + be.w, 0.f);
    }
}

// geometry of a launch, shared by the roles
template <int H, int KP, bool H2X>
struct StreamGeo {
    static constexpr int NT = H / 16, NB = NT / 2, TPR = kStreamTPR, NCONS = NT, NWAVE = NT + kStreamProducers;
    static constexpr bool HALF = KP == 32;                       // k > 16: a tile is half an atom (sm_edge16.h), merged by combine32_kernel
    static constexpr int SEGW = KP >= 16 ? 16 : KP, APJ = 16 / SEGW, HD = H / 8;
    // tile geometry: column n of tile t is neighbour slot slot_of(t, n) of centre atom atom_of(t, n)
    SM_DEV static int atom_of(int tile, int n) { return HALF ? (tile >> 1) : tile * APJ + n / SEGW; }
    SM_DEV static int slot_of(int tile, int n) { return HALF ? 16 * (tile & 1) + n : n % SEGW; }
    SM_DEV static int first_atom_of(int tile) { return HALF ? (tile >> 1) : tile * APJ; }
    int lane, wave, n, g, wg_first, wg_end, rounds;
    SM_DEV StreamGeo(const EdgeStreamArgs &a) {
        lane = threadIdx.x & 63; wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        n = lane & 15; g = lane >> 4;
        const int njobs = HALF ? 2 * a.n_atoms : (a.n_atoms + APJ - 1) / APJ;
        wg_first = blockIdx.x * a.chunk; wg_end = min(njobs, wg_first + a.chunk);
        rounds = (wg_end - wg_first + TPR - 1) / TPR;
    }
};

// Prologue every wave runs (after its own first requests): the producer parts by LDS-DMA and, for x2h, the coordinate update
// of the PREVIOUS layer folded in (sm_edge16.h: VnFold): the workgroup recomputes the new coordinates of every atom of the
// molecules its tiles touch into an LDS table and writes those of its own atoms to global memory.  Ends with the barrier
// that publishes both.
template <int H, int KP, bool H2X>
SM_DEV void stream_prologue(const EdgeStreamArgs &a, const StreamGeo<H, KP, H2X> &G, float *lds, int &span0, int &span_n) {
    using M = StreamMap<H, H2X>;
    using GE = StreamGeo<H, KP, H2X>;
    constexpr int HD = GE::HD, APJ = GE::APJ, NWAVE = GE::NWAVE;
    const int lane = G.lane;
    dma_to_lds(lds + M::O_K, a.part_k, M::PART_K / 4, G.wave, NWAVE, lane);
    dma_to_lds(lds + M::O_V, a.part_v, M::PART_V / 4, G.wave, NWAVE, lane);
    span0 = 0; span_n = 1;
    if constexpr (!H2X) {
        if (a.vf.enable) {
            float *xt = lds + M::O_TAIL;                       // [kVnFoldCap][3] new coordinates of the touched molecules
            double *sred = reinterpret_cast<double *>(xt + 3 * kVnFoldCap);          // [2][16] batch sums
            const int first_atom = G.wg_first * APJ;
            const int last_atom = min(a.n_atoms, G.wg_end * APJ) - 1;
            if (first_atom <= last_atom) {
                span0 = a.vf.mol_span[first_atom].x;
                const int span = a.vf.mol_span[last_atom].y - span0;
                if (span > kVnFoldCap && threadIdx.x == 0) *a.vf.span_flag = 1;      // the max_mol_atoms hint was too small
                span_n = min(span, kVnFoldCap);
            }
            if (threadIdx.x < 32) {
                const int c = threadIdx.x & 15, which = threadIdx.x >> 4;
                double t = 0.0;
                if (c < HD) for (int r = 0; r < kVnReplicas; ++r) t += a.vf.acc[(size_t)r * 2 * HD + which * HD + c];
                sred[threadIdx.x] = t;
            }
            __syncthreads();
            const int c = lane & 15;
            float meanf = 0.f, rstd = 0.f, bng = 0.f, bnb = 0.f;
            if (c < HD) {
                const double cnt = (double)a.n_atoms;
                const double mean = sred[c] / cnt;
                double var = sred[16 + c] / cnt - mean * mean;
                var = var > 0.0 ? var : 0.0;
                meanf = (float)mean;
                rstd = 1.0f / sqrtf((float)var + 1e-5f);
                bng = a.vf.bn_g[c]; bnb = a.vf.bn_b[c];
            }
            for (int it = threadIdx.x; it < span_n * 16; it += NWAVE * 64) {       // item = (atom of the span, channel)
                const int va = span0 + (it >> 4);
                float o[3] = {0.f, 0.f, 0.f};
                if (c < HD) {
                    const float *pdp = a.vf.pd + ((size_t)va * HD + c) * 6;
                    float p[3] = {pdp[0], pdp[1], pdp[2]};
                    const float d[3] = {pdp[3], pdp[4], pdp[5]};
                    const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
                    const float nbn = (nrm - meanf) * rstd * bng + bnb;
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
                    const float xn = a.vf.x_old[va * 3 + c] + (a.vf.xsum[va * 3 + c] / HD + r / HD);
                    xt[(it >> 4) * 3 + c] = xn;
                    if (va >= first_atom && va <= last_atom) a.vf.x_new[va * 3 + c] = xn;
                }
            }
        }
    }
    __syncthreads();                       // the images have landed, the coordinate table is complete
}

// h2x epilogue every wave runs: VN-linear of the workgroup's atoms -- p, d per channel from the 16 attention rows (+ x, + shape
// term) -- and the batch sums of ||p|| (shape_vn_layers.py:41-61,95-110): lane = (atom, channel); the rows come back through
// L2 (the last round's barrier drained this workgroup's stores).
template <int H, int KP>
SM_DEV void stream_vn_epilogue(const EdgeStreamArgs &a, const StreamGeo<H, KP, true> &G, float *lds) {
    using M = StreamMap<H, true>;
    using GE = StreamGeo<H, KP, true>;
    constexpr int HD = GE::HD, APJ = GE::APJ, NWAVE = GE::NWAVE;
    if (!a.vn.enable) return;
    const int lane = G.lane, wave = G.wave;
    double *vn_red = reinterpret_cast<double *>(lds + M::O_TAIL);          // [NWAVE][64][2]
    const int v_al = lane >> 4, v_c = lane & 15;
    const int first_atom = G.wg_first * APJ, end_atom = min(a.n_atoms, G.wg_end * APJ);
    double v_s1 = 0.0, v_s2 = 0.0;
    for (int va = first_atom + wave * 4 + v_al; va < end_atom; va += NWAVE * 4) {
        if (v_c >= HD) continue;
        float orow[48];
        const float *ov = a.out + (size_t)va * 48;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float4 t = ldg4(ov + 4 * i);
            orow[4 * i] = t.x; orow[4 * i + 1] = t.y; orow[4 * i + 2] = t.z; orow[4 * i + 3] = t.w;
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
            for (int rr = 0; rr < 16; ++rr) {
                pp += wf[rr] * orow[rr * 3 + k];
                dd += wd[rr] * orow[rr * 3 + k];
            }
            p[k] = pp + psf[k];
            d[k] = dd + psd[k];
        }
        float *out = a.vn.pd + ((size_t)va * HD + v_c) * 6;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
        if (a.xsum && v_c < 3) {                     // sum over the attention rows (padding rows are zero), r ascending as vn_apply_kernel
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { a0 += orow[rr * 3]; a1 += orow[rr * 3 + 1]; a2 += orow[rr * 3 + 2]; }
            a.xsum[va * 3 + v_c] = v_c == 0 ? a0 : (v_c == 1 ? a1 : a2);
        }
        const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + 1e-6f;
        v_s1 += (double)nrm;
        v_s2 += (double)nrm * (double)nrm;
    }
    vn_red[(wave * 64 + lane) * 2] = v_s1; vn_red[(wave * 64 + lane) * 2 + 1] = v_s2;
    __syncthreads();
    if (threadIdx.x < HD) {
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < NWAVE; ++w)
            for (int al = 0; al < 4; ++al) {
                s1 += vn_red[(w * 64 + al * 16 + threadIdx.x) * 2];
                s2 += vn_red[(w * 64 + al * 16 + threadIdx.x) * 2 + 1];
            }
        double *acc = a.vn.acc + (size_t)(blockIdx.x % kVnReplicas) * 2 * HD;
        atomicAdd(acc + threadIdx.x, s1);
        atomicAdd(acc + HD + threadIdx.x, s2);
    }
}

// ---- producer wave: MLP = 0 key, 1 value; my_slot = which tile of a round ------------------------------------------------------
template <int H, int KP, bool H2X, int MLP>
SM_DEV void stream_producer(const EdgeStreamArgs &a, const StreamGeo<H, KP, H2X> &G, float *lds, int my_slot) {
    using M = StreamMap<H, H2X>;
    using GE = StreamGeo<H, KP, H2X>;
    constexpr int NT = GE::NT, NB = GE::NB, TPR = GE::TPR, SEGW = GE::SEGW, APJ = GE::APJ;
    unsigned *ldsu = reinterpret_cast<unsigned *>(lds);
    const int lane = G.lane, n = G.n, g = G.g, wg_first = G.wg_first, wg_end = G.wg_end, rounds = G.rounds;
    bool fold = false;
    if constexpr (!H2X) fold = a.vf.enable != 0;

    // the rows of the unit in flight
    float4 ga[NT], gb[NT];                 // A[i], B[j] of this producer's MLP
    float4 qrow = {0.f, 0.f, 0.f, 0.f};    // key producer: this lane's 16 bytes of the tile's query rows
    float xi[3] = {0.f, 0.f, 0.f}, xj[3] = {0.f, 0.f, 0.f}, wgt = 0.f;
    int atom = 0, jn = 0, jraw_nx = -1;
    bool ok = false;
    auto tile_of = [&](int r) { return wg_first + TPR * r + my_slot; };
    auto peek = [&](int tile) { return a.nbr[min(GE::atom_of(tile, n), a.n_atoms - 1) * KP + GE::slot_of(tile, n)]; };
    auto request = [&](int tile, int jraw) {
        const int atom_raw = GE::atom_of(tile, n);
        const bool atom_ok = atom_raw < a.n_atoms;
        atom = atom_ok ? atom_raw : a.n_atoms - 1;
        ok = atom_ok && jraw >= 0;
        jn = ok ? jraw : atom;
        if (MLP == 0 && 4 * lane < APJ * H) {      // (first: whatever the address costs is paid before the long gathers are in flight)
            const int qa = GE::first_atom_of(tile) + (4 * lane) / H;
            int opq = 0;                             // (an offset the compiler cannot see through: the 64-bit lane address is formed
            asm volatile("" : "+v"(opq));            //  here, not hoisted out of the round loop and spilled)
            qrow = qa < a.n_atoms ? ldg4(a.q + (size_t)GE::first_atom_of(tile) * H + 4 * (lane + opq)) : float4{0.f, 0.f, 0.f, 0.f};
        }
        const float *pi = a.pre + (size_t)(SM_SABL(7) ? 0 : atom) * a.ld_pre + 2 * H * MLP + 4 * g;
        const float *pj = a.pre + (size_t)(SM_SABL(7) ? 0 : jn) * a.ld_pre + 2 * H * MLP + H + 4 * g;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            ga[t] = SM_SABL(11) ? float4{0.1f, 0.2f, 0.3f, 0.4f} : ldg4(pi + 16 * t);
            gb[t] = SM_SABL(12) ? float4{0.1f, 0.2f, 0.3f, 0.4f} : ldg4(pj + 16 * t);
        }
        if (!fold) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { xi[k] = a.x[atom * 3 + k]; xj[k] = a.x[jn * 3 + k]; }
        }
        wgt = a.ew[atom * KP + GE::slot_of(tile, n)];
    };

    SM_TICK(a.stamps, 0);
    SM_PROF(unsigned long long p_t0; unsigned long long p_ta; unsigned long long p_tb; unsigned long long p_tc; unsigned long long p_cmp = 0; unsigned long long p_gw = 0;
            unsigned long long p_bar = 0; unsigned long long p_pro = 0; unsigned long long p_first = 0; SM_PCLK(p_t0);)
    if (!SM_SABL(4)) __builtin_amdgcn_s_setprio(3);      // the producers are the critical path of a round: their instructions go first
    if (tile_of(0) < wg_end) {             // the first unit's rows: the oldest memory operations of the wave
        const int j0 = peek(tile_of(0));
        if (tile_of(1) < wg_end) jraw_nx = peek(tile_of(1));
        request(tile_of(0), j0);
    }
    int span0, span_n;
    stream_prologue<H, KP, H2X>(a, G, lds, span0, span_n);
    SM_TICK(a.stamps, 1);
    const float *xt = lds + M::O_TAIL;
    const float *part = lds + (MLP ? M::O_V : M::O_K);

    // one unit: (tile of round r, this wave's MLP) -> B fragments in slot (r & 1, my_slot)
    auto produce = [&](int r) {
        const int tile = tile_of(r);
        if (tile >= wg_end) return;
        const int slot = (r & 1) * TPR + my_slot;
        if (r == 1) SM_TICK(a.stamps, 3);      // (diagnostic build: phases of the unit of round 1 -- 3 start, 4 rows summed, 5 first Linear,
        const bool c_ok = ok;                  //  6 LayerNorm, 7 split + writes issued)
        const float c_wgt = wgt;
        const float4 c_q = qrow;
        if (fold) {      // coordinates from the table the prologue built
            const int ia = min(max(atom - span0, 0), span_n - 1) * 3, ja = min(max(jn - span0, 0), span_n - 1) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { xi[k] = xt[ia + k]; xj[k] = xt[ja + k]; }
        }
        const float rel[3] = {xi[0] - xj[0], xi[1] - xj[1], xi[2] - xj[2]};
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = f32x4{ga[t].x + gb[t].x, ga[t].y + gb[t].y, ga[t].z + gb[t].z, ga[t].w + gb[t].w};
        if (r == 1) SM_TICK(a.stamps, 4);
        auto request_next = [&]() {
            if (tile_of(r + 1) < wg_end) {
                const int jr = jraw_nx;
                if (tile_of(r + 2) < wg_end) jraw_nx = peek(tile_of(r + 2));
                request(tile_of(r + 1), jr);
            }
        };
        // RBF block of the first Linear: one K = 32 step, six piece products per output tile
        {
            float cen[5], rb[5];
            rbf_centres(g, cen);
            rbf_dlayout(sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]), cen, rb);
            u32x4 rh = {0u, 0u, 0u, 0u}, rm = {0u, 0u, 0u, 0u}, rl = {0u, 0u, 0u, 0u};
            unsigned h_, m_, l_;
            split3_pair(rb[0], rb[1], h_, m_, l_); rh[0] = h_; rm[0] = m_; rl[0] = l_;
            split3_pair(rb[2], rb[3], h_, m_, l_); rh[1] = h_; rm[1] = m_; rl[1] = l_;
            split3_pair(rb[4], 0.f, h_, m_, l_); rh[2] = h_; rm[2] = m_; rl[2] = l_;
            const u32x4 *w1 = reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned *>(part) + M::P_W1) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                u32x4 ah, am, al;
                if (!SM_SABL(6)) { ah = w1[(0 * NT + t) * 64]; am = w1[(1 * NT + t) * 64]; al = w1[(2 * NT + t) * 64]; }
                else { ah = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u + t, 0x3f803f80u}; am = ah + 0x00010001u; al = am + 0x00010001u; }
                f32x4 c = acc[t];
                if (!SM_SABL(1)) {
                c = mfma_bf16(al, rh, c);      // smallest terms first
                c = mfma_bf16(am, rm, c);
                c = mfma_bf16(ah, rl, c);
                c = mfma_bf16(am, rh, c);
                c = mfma_bf16(ah, rm, c);
                c = mfma_bf16(ah, rh, c);
                } else c[0] += __builtin_bit_cast(float, al[0] ^ am[1] ^ ah[2]) * __builtin_bit_cast(float, rh[0] ^ rm[1] ^ rl[2]);
                acc[t] = c;
            }
        }
        // the rows have been consumed and the first Linear's fragments are dead: request the next unit's rows (they fly under
        // the LayerNorm, the split and the barrier; the h2x value producer, which still has a matrix product ahead and needs
        // the registers for it, requests them behind that product)
        if (r == 1) SM_TICK(a.stamps, 5);
        if constexpr (!(H2X && MLP == 1)) request_next();
        float hid[NT * 4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            hid[4 * t + 0] = acc[t][0]; hid[4 * t + 1] = acc[t][1]; hid[4 * t + 2] = acc[t][2]; hid[4 * t + 3] = acc[t][3];
        }
        if (!SM_SABL(2)) ln_relu_stream<NT>(hid, part + M::P_G, part + M::P_B, g);
        if (r == 1) SM_TICK(a.stamps, 6);
        if constexpr (H2X && MLP == 1) {
            // h2x value MLP: the heads-wide second Linear here, from the LDS image; rows = heads, row 4 g + r in this lane
            const float4 bb = ldg4(part + M::P_B2 + 4 * g);
            f32x4 small = {0.f, 0.f, 0.f, 0.f}, big = {bb.x, bb.y, bb.z, bb.w};
            const u32x4 *w2 = reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned *>(part) + M::P_W2) + lane;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = hid[8 * b + j];
                u32x4 xh, xm, xl;
                split3_bf16(v, xh, xm, xl);
                big = mfma_bf16x6(w2[(0 * NB + b) * 64], w2[(1 * NB + b) * 64], w2[(2 * NB + b) * 64], xh, xm, xl, small, big);
            }
            const float w = c_ok ? c_wgt : 0.f;
            float *vt = lds + M::O_VT + slot * M::VT_TILE;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) vt[(4 * g + rr) * 16 + n] = (big[rr] + small[rr]) * w;
            if (g == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) vt[256 + 16 * k + n] = rel[k];
            }
            request_next();
        } else {
            u32x4 *hb = reinterpret_cast<u32x4 *>(ldsu + M::O_HID + slot * M::HID_TILE + (H2X ? 0 : MLP) * M::HID_MLP) + lane;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = hid[8 * b + j];
                u32x4 xh, xm, xl;
                if (!SM_SABL(3)) split3_bf16(v, xh, xm, xl);
                else { xh = u32x4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
                       xm = u32x4{__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]), __builtin_bit_cast(unsigned, v[7])}; xl = xh; }
                hb[(0 * NB + b) * 64] = xh; hb[(1 * NB + b) * 64] = xm; hb[(2 * NB + b) * 64] = xl;
            }
        }
        if constexpr (MLP == 0) {      // the key producer also hands over the tile's query rows and the columns' edge weights
            if (4 * lane < APJ * H) *reinterpret_cast<float4 *>(lds + M::O_Q + slot * M::Q_TILE + 4 * lane) = c_q;
            if (g == 0) lds[M::O_W + slot * 16 + n] = c_ok ? c_wgt : -1.f;
        }
        if (r == 1) SM_TICK(a.stamps, 7);
    };

    SM_PROF(SM_PCLK(p_ta); p_pro = p_ta - p_t0;)
    produce(0);
    SM_PROF(SM_PCLK(p_tb); p_first = p_tb - p_ta; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SM_PCLK(p_tc); p_gw += p_tc - p_tb;)
    __syncthreads();
    SM_PROF(SM_PCLK(p_ta); p_bar += p_ta - p_tc;)
    SM_TICK(a.stamps, 2);
    for (int r = 0; r < rounds; ++r) {
        produce(r + 1);
        SM_PROF(SM_PCLK(p_tb); p_cmp += p_tb - p_ta; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SM_PCLK(p_tc); p_gw += p_tc - p_tb;)
        __syncthreads();
        SM_PROF(SM_PCLK(p_ta); p_bar += p_ta - p_tc;)
    }
    SM_PROF(if (a.stamps != nullptr && lane == 0) {
        unsigned long long *row = a.stamps + ((size_t)blockIdx.x * GE::NWAVE + G.wave) * 8;
        row[0] = p_t0; row[1] = p_pro; row[2] = p_first; row[3] = p_cmp; row[4] = p_gw; row[5] = p_bar; row[6] = p_ta - p_t0; row[7] = (unsigned long long)rounds;
    })
}

// ---- consumer wave t2 = G.wave ------------------------------------------------------------------------------------------------
template <int H, int KP, bool H2X>
SM_DEV void stream_consumer(const EdgeStreamArgs &a, const StreamGeo<H, KP, H2X> &G, float *lds) {
    using M = StreamMap<H, H2X>;
    using GE = StreamGeo<H, KP, H2X>;
    constexpr int NT = GE::NT, NB = GE::NB, TPR = GE::TPR, SEGW = GE::SEGW, APJ = GE::APJ;
    const unsigned *ldsu = reinterpret_cast<const unsigned *>(lds);
    const int lane = G.lane, wave = G.wave, n = G.n, g = G.g, wg_first = G.wg_first, wg_end = G.wg_end, rounds = G.rounds;
    SM_TICK(a.stamps, 0);
    SM_PROF(unsigned long long p_t0; unsigned long long p_ta; unsigned long long p_tb; unsigned long long p_cmp = 0; unsigned long long p_bar = 0; unsigned long long p_pro = 0;
            unsigned long long p_first = 0; SM_PCLK(p_t0);)
    int span0, span_n;
    stream_prologue<H, KP, H2X>(a, G, lds, span0, span_n);
    SM_PROF(SM_PCLK(p_ta); p_pro = p_ta - p_t0;)
    SM_TICK(a.stamps, 1);
    // this wave's row block of the second Linears, in registers for the whole launch (loaded while the producers work on the
    // first round)
    u32x4 wk[3][NB], wv[3][NB];
    float4 b2v = {0.f, 0.f, 0.f, 0.f};
    {
        const u32x4 *pk = reinterpret_cast<const u32x4 *>(a.w2k) + lane;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int b = 0; b < NB; ++b) wk[pc][b] = pk[((pc * NT + wave) * NB + b) * 64];
        if constexpr (!H2X) {
            const u32x4 *pv = reinterpret_cast<const u32x4 *>(a.w2v) + lane;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int b = 0; b < NB; ++b) wv[pc][b] = pv[((pc * NT + wave) * NB + b) * 64];
            b2v = ldg4(a.b2v + 16 * wave + 4 * g);
        }
    }
    SM_PROF(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SM_PCLK(p_tb); p_first = p_tb - p_ta;)
    __syncthreads();
    SM_PROF(SM_PCLK(p_ta); p_bar += p_ta - p_tb;)
    SM_TICK(a.stamps, 2);
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int s = 0; s < TPR; ++s) {
            const int tile = wg_first + TPR * r + s;
            if (tile >= wg_end) break;
            const int slot = (r & 1) * TPR + s;
            const u32x4 *hk = reinterpret_cast<const u32x4 *>(ldsu + M::O_HID + slot * M::HID_TILE) + lane;
            f32x4 ksm = {0.f, 0.f, 0.f, 0.f}, kbg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < NB; ++b)
                kbg = SM_SABL(8) ? mfma_bf16x6(wk[0][b], wk[1][b], wk[2][b], wk[0][b], wk[1][b], wk[2][b], ksm, kbg)
                                 : mfma_bf16x6(wk[0][b], wk[1][b], wk[2][b], hk[(0 * NB + b) * 64], hk[(1 * NB + b) * 64], hk[(2 * NB + b) * 64], ksm, kbg);
            const int atom_raw = GE::atom_of(tile, n);
            const bool atom_ok = atom_raw < a.n_atoms;
            const int orow = GE::HALF ? tile : atom_raw;           // k > 16: every half-atom tile stores its own (tile-normalised) rows
            const float wc = lds[M::O_W + slot * 16 + n];
            const bool okc = wc >= 0.f;
            // logits of head 2 wave + (g >> 1): four of its dimensions in this lane group, four in the partner group g ^ 1.
            // The bias of the key MLP's second Linear adds the same q_i . b2 to every neighbour's logit and cancels in the softmax.
            const float4 qv = *reinterpret_cast<const float4 *>(lds + M::O_Q + slot * M::Q_TILE + (n / SEGW) * H + 16 * wave + 4 * g);
            float pp = qv.x * (kbg[0] + ksm[0]) + qv.y * (kbg[1] + ksm[1]) + qv.z * (kbg[2] + ksm[2]) + qv.w * (kbg[3] + ksm[3]);
            float mx = 0.f, ssum = 1.f, alpha = pp;
            if (!SM_SABL(9)) {
                pp = sum_xor16(pp);
                pp = okc ? pp * 0.35355339059327373f : -INFINITY;
                mx = seg_max<SEGW>(pp);
                const float e = okc ? fast_exp(pp - mx) : 0.f;
                ssum = seg_sum<SEGW>(e);
                alpha = ssum > 0.f ? e * __builtin_amdgcn_rcpf(ssum) : 0.f;
            }
            const bool store = atom_ok && (n % SEGW) == 0;
            if constexpr (GE::HALF) {        // this tile's softmax state of head 2 wave + (g >> 1), for the combine (online-softmax identity)
                if (store && (g & 1) == 0)
                    *reinterpret_cast<float2 *>(a.part_ms + ((size_t)tile * GE::HD + 2 * wave + (g >> 1)) * 2) = float2{mx, ssum};
            }
            if constexpr (!H2X) {
                const u32x4 *hv = hk + M::HID_MLP / 4;
                f32x4 vsm = {0.f, 0.f, 0.f, 0.f}, vbg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    vbg = SM_SABL(8) ? mfma_bf16x6(wv[0][b], wv[1][b], wv[2][b], wv[0][b], wv[1][b], wv[2][b], vsm, vbg)
                                     : mfma_bf16x6(wv[0][b], wv[1][b], wv[2][b], hv[(0 * NB + b) * 64], hv[(1 * NB + b) * 64], hv[(2 * NB + b) * 64], vsm, vbg);
                // sum_j a_ij (W2 hid_ij + b2) = sum_j a_ij W2 hid_ij + b2 sum_j a_ij
                const float aw = okc ? alpha * wc : 0.f;
                const float sw = SM_SABL(9) ? aw : seg_sum<SEGW>(aw);
                float o[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) o[rr] = SM_SABL(9) ? aw * (vbg[rr] + vsm[rr]) : seg_sum<SEGW>(aw * (vbg[rr] + vsm[rr]));
                if (store) stg4(a.out + (size_t)orow * H + 16 * wave + 4 * g,
                                float4{o[0] + sw * b2v.x, o[1] + sw * b2v.y, o[2] + sw * b2v.z, o[3] + sw * b2v.w});
            } else {
                const float *vt = lds + M::O_VT + slot * M::VT_TILE;
                const int head = 2 * wave + (g >> 1);
                const float av = alpha * vt[head * 16 + n];          // (value + bias) x edge weight, 0 where there is no edge
                float o[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) o[k] = SM_SABL(9) ? av * vt[256 + 16 * k + n] : seg_sum<SEGW>(av * vt[256 + 16 * k + n]);
                if (store && (g & 1) == 0) {
                    float *op = a.out + (size_t)orow * 48 + stream_row_of_head<NT>(head) * 3;
                    op[0] = o[0]; op[1] = o[1]; op[2] = o[2];
                }
            }
        }
        if (r == 0) SM_TICK(a.stamps, 3);      // (diagnostic build: this wave's work of rounds 0, 1, 2 done; 6: last barrier passed)
        if (r == 1) SM_TICK(a.stamps, 4);
        if (r == 2) SM_TICK(a.stamps, 5);
        SM_PROF(SM_PCLK(p_tb); p_cmp += p_tb - p_ta;)
        __syncthreads();
        SM_PROF(SM_PCLK(p_ta); p_bar += p_ta - p_tb;)
    }
    SM_TICK(a.stamps, 6);
    SM_PROF(if (a.stamps != nullptr && lane == 0) {
        unsigned long long *row = a.stamps + ((size_t)blockIdx.x * GE::NWAVE + wave) * 8;
        row[0] = p_t0; row[1] = p_pro; row[2] = p_first; row[3] = p_cmp; row[4] = 0; row[5] = p_bar; row[6] = p_ta - p_t0; row[7] = (unsigned long long)rounds;
    })
}

// One workgroup barrier per round; every role runs its own copy of the prologue, the loop and the barriers (a barrier counts
// arriving waves, whichever s_barrier instruction they execute), so that no register of one role is live on another role's path.
template <int H, int KP, bool H2X>
__global__ void __launch_bounds__((H / 16 + kStreamProducers) * 64) __attribute__((amdgpu_waves_per_eu(3, 3)))
edge_stream_kernel(EdgeStreamArgs a) {
    static_assert(KP == 8 || KP == 16 || KP == 32, "16-slot tiles");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const StreamGeo<H, KP, H2X> G(a);
    if (G.wave < G.NCONS) stream_consumer<H, KP, H2X>(a, G, lds);
    else if (((G.wave - G.NCONS) & 1) == 0) stream_producer<H, KP, H2X, 0>(a, G, lds, (G.wave - G.NCONS) >> 1);
    else stream_producer<H, KP, H2X, 1>(a, G, lds, (G.wave - G.NCONS) >> 1);
    if constexpr (H2X && KP <= 16) stream_vn_epilogue<H, KP>(a, G, lds);      // (k > 16: vn_stats / vn_apply follow the combine)
}
