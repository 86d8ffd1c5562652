// Edge attention with both Linears of the edge MLPs as THREE f16 MFMA products of two-piece operands (option edge_bf16 = 3: the
// default of rounds 2-3, since round 4 the optional faster mode -- the default is the exactly split form of sm_edge_stream.h).  Same semantics and formulation as sm_edge_bf16.h (reference: models/uni_transformer.py:48-81 for x2h,
// :121-151 for h2x); what changes is the arithmetic of the matrix products and, through it, the kernel's structure:
//
//   * a float is carried as hi + lo with hi = f16(x), lo = f16(x - hi), both round-to-nearest: 22 significand bits ONLY where
//     the low piece is a normal fp16 number -- |x - hi - lo| <= 2^-22 |x| for |x| >= 2^-3, an ABSOLUTE 2^-25 below (the residual
//     of a smaller x is an fp16 subnormal: weights below 1/8, most of them, carry 19-21 bits; round 3's "22 bits" overstated
//     the mode, which is why it is no longer the default) --, and x * w = hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with
//     fp32 accumulation (the dropped lo*lo term is < 2^-22 |x w|).  That is three matrix instructions per product where
//     the exactly split bf16 pieces (8 + 8 + 8 bits) need six, at an error per term that stays below the rounding of
//     the fp32 accumulation of a 128-term sum.  fp16 subnormal pieces are honoured by the matrix cores
//     (profiles/r02/mfma_f16_probe.txt), so small activations keep an absolute error of 2^-25; the operands are bounded
//     by construction: Gaussian smearing values lie in [0, 1], hidden activations are LayerNorm outputs
//     (|y| <= |gamma| sqrt(H - 1) + |beta|, checked against the fp16 range when the context is created), weights are
//     below 1.
//   * two-piece weights take 4 bytes per element: the LDS images of BOTH edge MLPs of an attention fit together
//     (2 x 78 KB at H = 128 for x2h; the RBF block of the first Linear is stored without the zero word of its K = 20 -> 32
//     padding, which is what leaves room for the LayerNorm parameters and biases).  The images arrive by LDS-DMA
//     (global_load_lds_dwordx4, no register staging): no weight swap, one workgroup barrier, and the attention weights go
//     from the key phase to the value phase in registers (v_permlane16_swap) instead of through a global scratch array.
//
// One job = the KP <= 16 neighbour slots of 16 / KP centre atoms = one 16-column tile; a wave runs the key MLP and
// then the value MLP of its job.  Layouts as in sm_device.h (D layout; the accumulator of one product is the B operand
// of the next).
#pragma once
#include "sm_device.h"
#include "sm_edge_bf16.h"      // EdgeFusedArgs::VnFuse, kVnReplicas, vn_red_doubles

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

SM_DEV f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two floats -> {hi pair, lo pair}, each one u32 of two f16 (low half = x0)
// lo = f16(x - hi) by v_fma_mix{lo,hi}_f16: the product-sum hi * (-1) + x is formed in fp32 (exact: hi is x rounded to 11
// bits) from the f16 half of `hi` directly and rounded to nearest into one half of `lo` -- three instructions per pair
// where convert-back, subtract, convert take six; the same values bit for bit.
SM_DEV void split2_pair(float x0, float x1, unsigned &hi, unsigned &lo) {
    const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    lo = l;
}

// D-layout activations -> B fragments of the NT / 2 k-steps (two pieces each)
template <int NT>
SM_DEV void split_act16(const float (&act)[NT * 4], u32x4 (&bh)[NT / 2], u32x4 (&bl)[NT / 2]) {
#pragma unroll
    for (int b = 0; b < NT / 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned h, l;
            split2_pair(act[8 * b + 2 * q], act[8 * b + 2 * q + 1], h, l);
            bh[b][q] = h; bl[b][q] = l;
        }
}

// LDS image of ONE edge MLP, in 32-bit words
//   W1 (RBF block of the first Linear, K = 20 padded to one K = 32 step): 2 pieces x [NT][3][64] u32: word q (two f16) of
//       lane (m, g) = pieces of W1[16t + m][4j + g], j = 2q, 2q + 1 (j < 5, else 0); the fourth word of the A fragment
//       (j = 6, 7) is always zero and not stored
//   W2: 2 pieces x [NT2][NB][64][4] u32; element j of lane (m, g) at k-step b = piece(W2[16 t2 + m][16 (2b + (j >> 2)) + 4g + (j & 3)])
//   gamma[H] | beta[H] | b2[NT2 * 16] (fp32), padded to a whole 1 KB DMA piece
template <int H, int NT2>
struct EdgeImage16 {
    static constexpr int NT = H / 16, NB = NT / 2;
    static constexpr int O_W1 = 0;
    static constexpr int O_W2 = O_W1 + 2 * NT * 192;
    static constexpr int O_G = O_W2 + 2 * NT2 * NB * 256;
    static constexpr int O_B = O_G + H;
    static constexpr int O_B2 = O_B + H;
    static constexpr int TOTAL = (O_B2 + NT2 * 16 + 255) / 256 * 256;
};

// acc[t] += W1[:, 0:20] rbf for all NT tiles: one K = 32 step, three products per tile
// P1 (option feat_f16, "f16 features"): only the leading f16 piece of both operands -- one product per term instead of three,
// 11 significand bits in the matrix operands (accumulation, LayerNorm, softmax, coordinates stay fp32).  Not a parity mode.
template <int NT, bool P1 = false>
SM_DEV void first_linear16(const unsigned *w1, u32x4 rh, u32x4 rl, f32x4 (&acc)[NT], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const unsigned *ph = w1 + (0 * NT + t) * 192 + lane, *pl = w1 + (1 * NT + t) * 192 + lane;
        const u32x4 ah = {ph[0], ph[64], ph[128], 0u};
        f32x4 c = acc[t];
        if constexpr (!P1) {
            const u32x4 al = {pl[0], pl[64], pl[128], 0u};
            c = mfma_f16(al, rh, c);      // smallest terms first
            c = mfma_f16(ah, rl, c);
        }
        c = mfma_f16(ah, rh, c);
        acc[t] = c;
    }
}

// one output tile of the second Linear
template <int NT, int NT2, bool P1 = false>
SM_DEV f32x4 tile_f16x3(const unsigned *w2, int t2, const u32x4 (&bh)[NT / 2], const u32x4 (&bl)[NT / 2], f32x4 c, int lane) {
    constexpr int NB = NT / 2;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const u32x4 ah = *reinterpret_cast<const u32x4 *>(w2 + (((0 * NT2 + t2) * NB + b) * 64 + lane) * 4);
        if constexpr (!P1) {
            const u32x4 al = *reinterpret_cast<const u32x4 *>(w2 + (((1 * NT2 + t2) * NB + b) * 64 + lane) * 4);
            c = mfma_f16(al, bh[b], c);
            c = mfma_f16(ah, bl[b], c);
        }
        c = mfma_f16(ah, bh[b], c);
    }
    return c;
}

// x2h only: the coordinate update of the PREVIOUS layer (vn_apply_kernel: batch-norm of ||p||, VN-leaky-ReLU, mean over the
// channels, x += ...; uni_transformer.py:153-162, shape_vn_layers.py:41-61,95-110) folded into this kernel's prologue: the
// workgroup recomputes the new coordinates of every atom of the molecules its jobs touch (their neighbours live there) into
// an LDS table, from the p / d vectors and the batch sums the h2x kernel left behind, and writes those of its own atoms to
// global memory for the kernels that follow.  Saves the vn_apply launch (5 us) of every layer but the last.
constexpr int kVnFoldCap = 256;       // atoms of the LDS coordinate table (the host enables the fold only if every span fits)
struct VnFold {
    const float *pd;          // [N][heads][6]
    const double *acc;        // [kVnReplicas][2][heads] batch sums of the previous layer
    const float *bn_g, *bn_b; // [heads]
    const float *xsum;        // [N][3] sum over the attention rows, written by the h2x epilogue
    const float *x_old;       // [N][3]
    float *x_new;             // [N][3]
    const int2 *mol_span;     // [N] first / one-past-last atom of each atom's molecule
    int *span_flag;           // status flag raised if a workgroup's molecule span exceeds the table
    int enable;
};

struct Edge16Args {
    const float *image_k, *image_v;   // EdgeImage16 of the key / value MLP
    const float *pre;       // node pre-products [N][ld_pre]: A_k | B_k | A_v | B_v at column offsets 0, H, 2H, 3H
    const float *q;         // [N][H]
    const float *x;         // [N][3]
    const int *nbr;         // [N][KP]
    const float *ew;        // [N][KP]
    float *out;             // x2h: [N][H]; h2x: [N][16][3]
    int n_atoms, ld_pre;
    int job_base, job_end;            // edge16_kernel: this launch covers jobs [job_base, job_end) (job_end = 0: all of them)
    int nwave;                        // waves per workgroup (= blockDim.x / 64; as an argument because reading blockDim costs
                                      // two dependent loads from the implicit kernel arguments at the head of every launch)
    int chunk;                        // looping launches (edge16_loop_kernel): consecutive jobs per workgroup; its waves take
                                      // them round robin (job = first + wave, + nwave, ...)
    unsigned long long *stamps;       // diagnostic build only
    EdgeFusedArgs::VnFuse vn;         // h2x: VN-linear + batch statistics behind the attention (enable = 0 or 2)
    float *xsum;                      // h2x with vn.enable: [N][3] sum of the attention rows per atom (for a following VnFold), or nullptr
    VnFold vf;                        // x2h: coordinate update of the previous layer in the prologue
    float *part_ms;                   // KP = 32: [2 N][heads][2] running max and sum of every half-atom tile's softmax (see below)
};

// ONE = true: every wave has at most one job (jobs <= workgroups x waves; the common case up to ~6k atoms).
// ONE = false (edge16_loop_kernel, two waves per SIMD): a workgroup owns a.chunk consecutive jobs, its waves loop over them
// round robin and request the NEXT job's key rows under the current job's value MLP (the 256-VGPR budget of two waves per
// SIMD holds a job's state and the next job's 64 gathered registers without spilling; with three waves per SIMD -- 168
// VGPRs -- the looping form spilled 54-75 registers and lost to sliced launches of the one-job form).
// KEEP (x2h, ONE): the attention rows are not stored; the storing lanes (n % SEGW == 0: lane (n, g) owns features
// 16 t + 4 g .. + 3 of centre atom n / SEGW of the wave's job) return them in keep[t] for a node stage fused behind
// (x2h_chain16_kernel, sm_node16.h); all other lanes return zeros.
//
// KP = 32 (k > 16): an atom's 32 neighbour slots are TWO jobs, one 16-slot tile each (jobs 2 a and 2 a + 1), run like any
// other tile with the softmax taken over the tile's own slots; each tile stores its rows (normalised inside the tile) to
// a.out[job] and the running maximum m and sum s of every head's softmax to a.part_ms[job], and combine32_kernel merges the
// two tiles of an atom with the weights s_t exp(m_t - M) / sum_t s_t exp(m_t - M) (the "online softmax" identity).  No wave
// ever holds two tiles, so k = 32 runs in the same registers as k = 8 (the two-tile kernel of round 2 needed 256 VGPRs and
// spilled 40-75); the VN-linear epilogue is not fused for k > 16 (vn_stats_kernel / vn_apply_kernel follow the combine).
template <int H, int KP, bool H2X, bool ONE, bool KEEP, bool P1 = false>
SM_DEV void edge16_body(const Edge16Args &a, float4 (&keep)[H / 16]) {
    static_assert(KP == 8 || KP == 16 || KP == 32, "16-slot tiles");
    static_assert(!KEEP || (ONE && !H2X && KP <= 16), "KEEP: one-job x2h only");
    constexpr int NT = H / 16;
    constexpr int NT2V = H2X ? 1 : NT;
    using IMK = EdgeImage16<H, NT>;
    using IMV = EdgeImage16<H, NT2V>;
    constexpr int V_BASE = IMK::TOTAL;
    constexpr bool HALF = KP == 32;                               // a job is half an atom
    constexpr int SEGW = KP >= 16 ? 16 : KP, APJ = 16 / SEGW;
    constexpr int HD = H / 8;                                     // heads = VN channels
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = a.nwave;
    const int n = lane & 15, g = lane >> 4;
    float cen[5];
    rbf_centres(g, cen);
    const int njobs_all = HALF ? 2 * a.n_atoms : (a.n_atoms + APJ - 1) / APJ;
    const int njobs = a.job_end > 0 ? min(a.job_end, njobs_all) : njobs_all;       // a launch may cover a slice of the jobs
    const int chunk = ONE ? nwave : a.chunk;                                       // consecutive jobs of this workgroup
    const int wg_first = a.job_base + blockIdx.x * chunk, wg_end = min(njobs, wg_first + chunk);
    const int jstride = nwave;
    const int job0 = wg_first + wave;

    int atom = 0, jn = 0, edge = 0;
    bool atom_ok = false, ok = false;
    float xi[3], xj[3];
    bool fold = false;
    if constexpr (!H2X) fold = a.vf.enable != 0;
    float4 ga[NT], gb[NT], gav[NT], gbv[NT];      // gathered rows: A_k[i], B_k[j], A_v[i], B_v[j]
    float4 qv[NT];                                // query row of the centre atom
    float wgt = 0.f;

    // the key MLP's rows are requested up front: the centre atom's first (they do not wait for the neighbour index),
    // then the neighbour's
    auto atom_of_job = [&](int jb) { return HALF ? (jb >> 1) : jb * APJ + n / SEGW; };
    auto slot_of_job = [&](int jb) { return HALF ? 16 * (jb & 1) + n : n % SEGW; };
    auto locate = [&](int jb) {      // -> the job's neighbour index (a load: issue it before anything else of the job)
        const int atom_raw = atom_of_job(jb);
        atom_ok = atom_raw < a.n_atoms;
        atom = atom_ok ? atom_raw : a.n_atoms - 1;
        edge = atom * KP + slot_of_job(jb);
        return a.nbr[edge];
    };
    auto peek = [&](int jb) {        // the same load without adopting the job (the next job's index, requested a phase early)
        return a.nbr[min(atom_of_job(jb), a.n_atoms - 1) * KP + slot_of_job(jb)];
    };
    auto adopt = [&](int jb) {       // the bookkeeping of locate() for an index that peek() has already fetched
        const int atom_raw = atom_of_job(jb);
        atom_ok = atom_raw < a.n_atoms;
        atom = atom_ok ? atom_raw : a.n_atoms - 1;
        edge = atom * KP + slot_of_job(jb);
    };
    auto request = [&](int jraw) {
        const float *pi = a.pre + (size_t)(SM_ABL(17) ? 0 : atom) * a.ld_pre;
#pragma unroll
        for (int t = 0; t < NT; ++t) ga[t] = ldg4(pi + 16 * t + 4 * g);
        if (!fold) {
#pragma unroll
            for (int k = 0; k < 3; ++k) xi[k] = a.x[atom * 3 + k];
        }
        ok = atom_ok && jraw >= 0;
        jn = ok ? jraw : atom;
        const float *pj = a.pre + (size_t)(SM_ABL(17) ? 0 : jn) * a.ld_pre + H;
#pragma unroll
        for (int t = 0; t < NT; ++t) gb[t] = ldg4(pj + 16 * t + 4 * g);
        if (!fold) {
#pragma unroll
            for (int k = 0; k < 3; ++k) xj[k] = a.x[jn * 3 + k];
        }
        wgt = a.ew[edge];
    };
    // the value MLP's rows and the query row are requested once the key rows have been consumed (register budget): they
    // fly under the key MLP's hidden layer (all of whose other operands come from LDS)
    auto request_2 = [&](int atom, int jn) {
        const float *pi = a.pre + (size_t)(SM_ABL(17) ? 0 : atom) * a.ld_pre + 2 * H, *pj = a.pre + (size_t)(SM_ABL(17) ? 0 : jn) * a.ld_pre + 3 * H;
#pragma unroll
        for (int t = 0; t < NT; ++t) { gav[t] = ldg4(pi + 16 * t + 4 * g); gbv[t] = ldg4(pj + 16 * t + 4 * g); }
    };
    // the query row: requested once the value rows have been consumed (it is first used by the key phase, after the value
    // MLP's hidden layer), so that it does not hold 32 registers beside them
    auto request_q = [&](int atom) {
        const float *qrow = a.q + (size_t)atom * H + 4 * g;
#pragma unroll
        for (int t = 0; t < NT; ++t) qv[t] = ldg4(qrow + 16 * t);
    };
    // hidden = ReLU(LN(A_i + B_j + W_r rbf)) -> B fragments
    auto hidden = [&](const float *img, auto im_tag, const float4 (&ra)[NT], const float4 (&rbw)[NT], u32x4 rh, u32x4 rl,
                      u32x4 (&bh)[NT / 2], u32x4 (&bl)[NT / 2], auto after_rows) {
        using IM = decltype(im_tag);
        float hid[NT * 4];
        {
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = f32x4{ra[t].x + rbw[t].x, ra[t].y + rbw[t].y, ra[t].z + rbw[t].z, ra[t].w + rbw[t].w};
            after_rows();
            first_linear16<NT, P1>(reinterpret_cast<const unsigned *>(img) + IM::O_W1, rh, rl, acc, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                hid[4 * t + 0] = acc[t][0]; hid[4 * t + 1] = acc[t][1];
                hid[4 * t + 2] = acc[t][2]; hid[4 * t + 3] = acc[t][3];
            }
        }
        ln_relu_dlayout<NT>(hid, img + IM::O_G, img + IM::O_B, g);
        split_act16<NT>(hid, bh, bl);
    };

    const bool have0 = job0 < wg_end;
    if constexpr (KEEP) {
#pragma unroll
        for (int t = 0; t < NT; ++t) keep[t] = float4{0.f, 0.f, 0.f, 0.f};
    }
    SM_TICK(a.stamps, 0);
    // both weight images by LDS-DMA (asynchronous, no register staging); the job's row gathers fly beside them
    // Order of the wave's memory operations (they complete in order, and the compiler cannot count the DMA pieces of a
    // loop: it waits for ALL outstanding operations before the first dependent address): neighbour index -> the job's
    // row gathers (the long pole, ~3 us from the Infinity Cache) -> image DMA (1.7 us, lands meanwhile).
    // (SM_ABL(bit): timing-attribution builds only, build.sh --ablate MASK: 16 image DMA, 17 row gathers, 18 hidden layers,
    //  19 key second Linear + softmax, 20 value second Linear)
    // ---- folded coordinate update of the previous layer (x2h): issued FIRST, its small loads are the oldest operations ----
    float *xt = lds + V_BASE + IMV::TOTAL;                     // [kVnFoldCap][3] new coordinates of the touched molecules
    int span0 = 0, span_n = 1;
    if constexpr (!H2X) {
        if (fold) {
            double *sred = reinterpret_cast<double *>(xt + 3 * kVnFoldCap);          // [2][16] batch sums
            const int first_atom = wg_first * APJ;
            const int last_atom = min(a.n_atoms, wg_end * APJ) - 1;
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
            for (int it = threadIdx.x; it < span_n * 16; it += nwave * 64) {       // item = (atom of the span, channel)
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
    if (have0) request(locate(job0));
    if (!SM_ABL(16)) {
        dma_to_lds(lds, a.image_k, IMK::TOTAL / 4, wave, nwave, lane);
        dma_to_lds(lds + V_BASE, a.image_v, IMV::TOTAL / 4, wave, nwave, lane);
    }
    __syncthreads();
    SM_TICK(a.stamps, 1);

    // fused coordinate update (h2x): scratch behind the images
    double *vn_red = reinterpret_cast<double *>(lds + V_BASE + IMV::TOTAL);
    auto vn_rows = [&]() { return reinterpret_cast<float *>(vn_red + vn_red_doubles(nwave, HD)) + wave * (APJ * 48); };   // this wave's attention rows [APJ][16][3] (formed where it is used: one register less across the MLPs)
    const bool one_job = ONE;                                 // (looping launches read their attention rows back through L2)

    for (int job = job0; job < wg_end; job += jstride) {
        // Looping launches: the image spans more than the 64 KB a DS instruction's offset field reaches, so the compiler forms
        // some thirty LDS base addresses -- and, in a loop, hoists all of them out of it as loop invariants and SPILLS them
        // (54-75 registers at the 168-VGPR budget; the reloads wait behind the row gathers).  An offset the compiler cannot
        // see through (always 0) keeps the address arithmetic inside the iteration, next to its use, as in the one-job form;
        // the memory clobber keeps the (loop-invariant) weight reads themselves inside the loop.
        int lds_opq = 0;
        if constexpr (!ONE) asm volatile("" : "+v"(lds_opq) : : "memory");
        if constexpr (!ONE) SM_TICK(a.stamps, 6);
        const float *imk = lds + lds_opq, *imv = lds + V_BASE + lds_opq;
        const unsigned *w2k = reinterpret_cast<const unsigned *>(imk) + IMK::O_W2;
        const unsigned *w2v = reinterpret_cast<const unsigned *>(imv) + IMV::O_W2;
        if constexpr (!ONE) rbf_centres(g + lds_opq, cen);     // (five more loop invariants the allocator would rather spill)
        // the job at hand (the request variables are handed to the next job below, under this job's value MLP)
        bool have_next = false;
        int jraw_next = 0;
        const int c_atom = atom, c_jn = jn;
        const bool c_ok = ok, c_atom_ok = atom_ok;
        const float c_wgt = wgt;
        if constexpr (!ONE) {
            have_next = job + jstride < wg_end;
            if (have_next) jraw_next = peek(job + jstride);  // old by the time its gathers are issued: nothing waits for it
        }
        if (fold) {      // coordinates from the table the prologue built (the barrier above made it visible)
            const int ia = min(max(c_atom - span0, 0), span_n - 1) * 3, ja = min(max(c_jn - span0, 0), span_n - 1) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { xi[k] = xt[ia + k]; xj[k] = xt[ja + k]; }
        }
        const float rel[3] = {xi[0] - xj[0], xi[1] - xj[1], xi[2] - xj[2]};
        u32x4 rh = {0u, 0u, 0u, 0u}, rl = {0u, 0u, 0u, 0u};
        {
            float rb[5];
            rbf_dlayout(sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]), cen, rb);
            unsigned h, l;
            split2_pair(rb[0], rb[1], h, l); rh[0] = h; rl[0] = l;
            split2_pair(rb[2], rb[3], h, l); rh[1] = h; rl[1] = l;
            split2_pair(rb[4], 0.f, h, l); rh[2] = h; rl[2] = l;
        }
        // ---- hidden activations of both MLPs, then the two second Linears back to back from registers ----------------
        u32x4 kh[NT / 2], kl[NT / 2], vh[NT / 2], vl[NT / 2];
        auto after_value_rows = [&]() { request_q(c_atom); };      // the value rows have been consumed: the query row
        if (!SM_ABL(18)) {
            hidden(imk, IMK{}, ga, gb, rh, rl, kh, kl, [&]() { request_2(c_atom, c_jn); });
            SM_TICK(a.stamps, 2);
            hidden(imv, IMV{}, gav, gbv, rh, rl, vh, vl, after_value_rows);
        } else {
            request_2(c_atom, c_jn); after_value_rows();
#pragma unroll
            for (int b = 0; b < NT / 2; ++b) {
                kh[b] = u32x4{__builtin_bit_cast(unsigned, ga[b].x + gb[b].x), rh[1], rh[2], rl[0]}; kl[b] = rl; vh[b] = kh[b];
                vl[b] = u32x4{__builtin_bit_cast(unsigned, gav[b].x + gbv[b].y), rl[1], rl[2], rh[0]};
            }
        }
        SM_TICK(a.stamps, 3);
        // ---- key phase: k -> logits -> softmax over the atom's neighbour slots.  The bias of the key MLP's second Linear
        //      adds the same q_i . b2 to every neighbour's logit of a head and cancels in the softmax: it is not applied.
        float alpha[NT / 2];
        if (SM_ABL(19)) {
#pragma unroll
            for (int t = 0; t < NT / 2; ++t) alpha[t] = qv[t].x * __builtin_bit_cast(float, kh[t][0] & 0x3fffffffu);
        } else {
            f32x4 ka = tile_f16x3<NT, NT, P1>(w2k, 0, kh, kl, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
            f32x4 kb = tile_f16x3<NT, NT, P1>(w2k, NT / 2, kh, kl, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
#pragma unroll
            for (int t = 0; t < NT / 2; ++t) {
                f32x4 na = ka, nb = kb;
                if (t + 1 < NT / 2) {
                    na = tile_f16x3<NT, NT, P1>(w2k, t + 1, kh, kl, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
                    nb = tile_f16x3<NT, NT, P1>(w2k, t + 1 + NT / 2, kh, kl, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
                }
                if constexpr (HALF) {        // this tile's softmax state of head 2 (t + (NT/2)(g & 1)) + (g >> 1), for the combine
                    float mx, ssum;
                    alpha[t] = attention_weight_pair<NT, SEGW>(qv[t], qv[t + NT / 2], ka, kb, c_ok, mx, ssum);
                    if (n == 0 && c_atom_ok)
                        *reinterpret_cast<float2 *>(a.part_ms + ((size_t)job * HD + 2 * (t + (NT / 2) * (g & 1)) + (g >> 1)) * 2) = float2{mx, ssum};
                } else alpha[t] = attention_weight_pair<NT, SEGW>(qv[t], qv[t + NT / 2], ka, kb, c_ok);
                ka = na; kb = nb;
            }
        }
        SM_TICK(a.stamps, 4);
        // looping launches: the NEXT job's key rows are requested here, once the key MLP's fragments and the query row are dead,
        // and fly under the value MLP's second Linear (and the partner wave's work)
        if constexpr (!ONE) {
            if (have_next) { adopt(job + jstride); request(jraw_next); }
        }
        // ---- value phase ----------------------------------------------------------------------------------------------
        const float w = c_ok ? c_wgt : 0.f;
        const bool store = c_atom_ok && (n % SEGW) == 0;
        {
            const float *b2 = imv + IMV::O_B2;
            if constexpr (!H2X) {
                // alpha[t] of lane group g belongs to head block t + (NT/2)(g & 1); output tile T needs head block T of the
                // lane's group pair: one v_permlane16_swap hands every lane both halves
                float al[NT];
#pragma unroll
                for (int t = 0; t < NT / 2; ++t) {
                    float lo = alpha[t], hi = alpha[t];
                    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                    al[t] = lo * w; al[t + NT / 2] = hi * w;
                }
                // sum_j a_ij (W2 hid_ij + b2) = sum_j a_ij W2 hid_ij + b2 sum_j a_ij: the bias enters through the per-head sum
                // of the weights, so the accumulators start at zero and b2 is only needed by the storing lanes at the end
                float *op = a.out + (size_t)(HALF ? job : c_atom) * H;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 v = SM_ABL(20) ? f32x4{__builtin_bit_cast(float, vh[t % (NT / 2)][0] & 0x3fffffffu), 1.f, 2.f, 3.f}
                                               : tile_f16x3<NT, NT, P1>(w2v, t, vh, vl, f32x4{0.f, 0.f, 0.f, 0.f}, lane);
                    const float4 bb = ldg4(b2 + 16 * t + 4 * g);
                    const float aw = al[t];
                    const float sw = seg_sum<SEGW>(aw);
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = seg_sum<SEGW>(aw * v[r]);
                    const float4 row = {o[0] + sw * bb.x, o[1] + sw * bb.y, o[2] + sw * bb.z, o[3] + sw * bb.w};
                    if constexpr (KEEP) keep[t] = store ? row : float4{0.f, 0.f, 0.f, 0.f};
                    else if (store) stg4(op + 16 * t + 4 * g, row);
                }
            } else {
                const float4 bb = ldg4(b2 + 4 * g);
                const f32x4 vacc = tile_f16x3<NT, 1, P1>(w2v, 0, vh, vl, f32x4{bb.x, bb.y, bb.z, bb.w}, lane);
                // value row 4g + r belongs to head 2*((NT/2)*(g&1) + r) + (g>>1) = the head of the lane's own alpha[r]
                float o[12];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float av = r < NT / 2 ? alpha[r % (NT / 2)] * w * vacc[r] : 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) o[3 * r + k] = seg_sum<SEGW>(av * rel[k]);
                }
                if (store) {
                    float *op = a.out + (size_t)(HALF ? job : c_atom) * 48 + 12 * g;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        stg4(op + 4 * i, float4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]});
                    if (a.vn.enable && one_job) {           // keep the rows at hand for the VN-linear below (no L2 round trip)
                        float *ol = vn_rows() + (n / SEGW) * 48 + 12 * g;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            stg4(ol + 4 * i, float4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]});
                    }
                }
            }
        }
        SM_TICK(a.stamps, 5);
        if constexpr (ONE) break;
    }

    if constexpr (H2X && !HALF) {
        if (!a.vn.enable) return;
        // ---- VN-linear of this wave's atoms: p, d per channel from the 16 attention rows (+ x, + shape term) and the
        //      batch sums of ||p|| (shape_vn_layers.py:41-61,95-110): lane = (atom of the job, channel) ----
        const int v_al = lane >> 4, v_c = lane & 15;
        const bool v_lane = v_al < APJ && v_c < HD;
        double v_s1 = 0.0, v_s2 = 0.0;
        if (!one_job) __syncthreads();               // rows come back through L2: drain this workgroup's stores first
        for (int jb = job0; jb < wg_end; jb += jstride) {
            const int va = jb * APJ + v_al;
            if (v_lane && va < a.n_atoms) {
                float orow[48];
                const float *ov = one_job ? vn_rows() + v_al * 48 : a.out + (size_t)va * 48;
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
                    for (int r = 0; r < 16; ++r) {
                        pp += wf[r] * orow[r * 3 + k];
                        dd += wd[r] * orow[r * 3 + k];
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
        }
        SM_TICK(a.stamps, 6);
        if (lane < 32) { vn_red[(wave * 32 + lane) * 2] = v_s1; vn_red[(wave * 32 + lane) * 2 + 1] = v_s2; }
        __syncthreads();
        SM_TICK(a.stamps, 7);
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
    }
}

template <int H, int KP, bool H2X, bool P1 = false>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
edge16_kernel(Edge16Args a) {          // one job per wave, up to twelve waves per workgroup
    float4 keep[H / 16];
    edge16_body<H, KP, H2X, true, false, P1>(a, keep);
}

template <int H, int KP, bool H2X, bool P1 = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
edge16_loop_kernel(Edge16Args a) {     // eight waves per workgroup looping over a.chunk consecutive jobs
    float4 keep[H / 16];
    edge16_body<H, KP, H2X, false, false, P1>(a, keep);
}

// k > 16: merge the two half-atom tiles of every atom.  part [2 N][W] rows (W = H for x2h, 48 for h2x: [16 rows][3] with row
// 4 g + r belonging to head_of_row), ms [2 N][heads][2] = (max, sum) of each tile's softmax per head.
struct Combine32Args { const float *part, *ms; float *out; int n_atoms, heads, nt; };
template <bool H2X>
__global__ void __launch_bounds__(256) combine32_kernel(Combine32Args a) {
    const int W = H2X ? 48 : a.heads * 8;
    const int per = H2X ? 48 : W / 4;                              // items per atom: elements (h2x) or float4 chunks (x2h)
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.n_atoms * per) return;
    const int atom = gid / per, it = gid % per;
    int head;
    if constexpr (H2X) {
        const int m = it / 3, g = m >> 2, r = m & 3;               // value row m = 4 g + r (see pack_image16 / head_of_row)
        head = r < a.nt / 2 ? 2 * ((a.nt / 2) * (g & 1) + r) + (g >> 1) : -1;
    } else head = it / 2;                                          // features 4 it .. 4 it + 3 of head (4 it) / 8
    float c0 = 0.f, c1 = 0.f;
    if (head >= 0) {
        const float2 s0 = *reinterpret_cast<const float2 *>(a.ms + ((size_t)(2 * atom) * a.heads + head) * 2);
        const float2 s1 = *reinterpret_cast<const float2 *>(a.ms + ((size_t)(2 * atom + 1) * a.heads + head) * 2);
        const float M = fmaxf(s0.x, s1.x);
        if (M > -INFINITY) {
            const float w0 = s0.y > 0.f ? s0.y * fast_exp(s0.x - M) : 0.f, w1 = s1.y > 0.f ? s1.y * fast_exp(s1.x - M) : 0.f;
            const float inv = 1.0f / (w0 + w1);
            c0 = w0 * inv; c1 = w1 * inv;
        }
    }
    const float *p0 = a.part + (size_t)(2 * atom) * W, *p1 = p0 + W;
    if constexpr (H2X) {
        a.out[(size_t)atom * 48 + it] = c0 * p0[it] + c1 * p1[it];
    } else {
        const float4 r0 = ldg4(p0 + 4 * it), r1 = ldg4(p1 + 4 * it);
        stg4(a.out + (size_t)atom * W + 4 * it, float4{c0 * r0.x + c1 * r1.x, c0 * r0.y + c1 * r1.y, c0 * r0.z + c1 * r1.z, c0 * r0.w + c1 * r1.w});
    }
}
