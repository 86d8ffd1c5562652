// Node kernels on two-piece f16 operands: the arithmetic of sm_edge16.h (x = hi + lo, both f16 round-to-nearest, three
// v_mfma_f32_16x16x32_f16 products per term, fp32 accumulation) in the structure of the bf16x6 node kernels of sm_node.h
// (node_chain6_kernel / node_linear6_kernel / node_prologue6_kernel, whose comments describe the stages).  Per element the
// weights take 4 bytes instead of 6 and a product three matrix instructions instead of six: these kernels stream their
// weights from L2 once per workgroup, so both the bytes and the instruction count matter.
// Unlike the edge MLPs' operands (LayerNorm outputs, bounded by construction), the inputs here are the residual stream h,
// the attention output and shifted-softplus activations: every split checks |x| < 6e4 and raises the context's status
// flag ST_RANGE otherwise (shapemol_status reports it; option node_f16 = 0 selects the exactly split bf16 kernels).
#pragma once
#include "sm_node.h"
#include "sm_edge16.h"

SM_DEV void split2_8(const float (&v)[8], u32x4 &hi, u32x4 &lo, int *range_flag) {
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned h, l;
        split2_pair(v[2 * q], v[2 * q + 1], h, l);
        hi[q] = h; lo[q] = l;
        m = fmaxf(m, fmaxf(fabsf(v[2 * q]), fabsf(v[2 * q + 1])));
    }
    if (!(m < 6.0e4f)) *range_flag = 1;          // also catches NaN
}

// split f16 image of a Linear: wimg[(((ot * 2 + piece) * NB + b) * 64 + lane) * 4 + q] u32, element order of gemm_bf16x6
constexpr int kLin16Chunk = 12;   // column tiles staged at a time: 2 * H * 32 bytes each (8 KB at H = 128)

template <int H, bool P1 = false>
__global__ void __launch_bounds__(kNodeThreads)
node_linear16_kernel(NodeLinArgs a, int *range_flag) {
    constexpr int NB = H / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lin16_lds[];
    u32x4 *frag = reinterpret_cast<u32x4 *>(lin16_lds);             // [tile][piece][NB][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nwave = a.nwave;
    const int ogroups = (a.n_out_tiles + nwave - 1) / nwave;
    const int ot_raw = (blockIdx.x % ogroups) * nwave + wave;
    const bool ot_ok = ot_raw < a.n_out_tiles;
    const int ot = ot_ok ? ot_raw : a.n_out_tiles - 1;
    const int ag = blockIdx.x / ogroups;
    const int n_ct = (a.n_atoms + 15) / 16;
    const int ct0 = ag * a.tiles_per_group, ct1 = min(ct0 + a.tiles_per_group, n_ct);

    SM_TICK(a.stamps, 0);
    u32x4 w[2][NB];
    {
        const u32x4 *wi = reinterpret_cast<const u32x4 *>(a.wimg) + (size_t)ot * 2 * NB * 64 + lane;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int b = 0; b < NB; ++b) w[p][b] = wi[(p * NB + b) * 64];
    }
    for (int cb = ct0; cb < ct1; cb += kLin16Chunk) {
        const int nc = min(kLin16Chunk, ct1 - cb);
        __syncthreads();                                            // previous chunk fully consumed
        for (int idx = threadIdx.x; idx < nc * NB * 64; idx += nwave * 64) {
            const int sl = idx & 63, sb = (idx >> 6) % NB, sc = idx / (64 * NB);
            const int at = min((cb + sc) * 16 + (sl & 15), a.n_atoms - 1);
            const float *src = a.in + (size_t)at * H + 32 * sb + 4 * (sl >> 4);
            const float4 v0 = ldg4(src), v1 = ldg4(src + 16);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            u32x4 hi, lo;
            split2_8(v, hi, lo, range_flag);
            u32x4 *dst = frag + ((size_t)(sc * 2) * NB + sb) * 64 + sl;
            dst[0] = hi; dst[NB * 64] = lo;
        }
        SM_TICK(a.stamps, 1);
        __syncthreads();
        SM_TICK(a.stamps, 2);
        for (int c = 0; c < nc; c += 2) {                           // two tiles at a time: independent MFMA chains
            const bool two = c + 1 < nc;
            const int atom0 = (cb + c) * 16 + n, atom1 = atom0 + 16;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (a.add_mol) {
                const float4 t0 = ldg4(a.add_mol + (size_t)a.mol_of[min(atom0, a.n_atoms - 1)] * a.ld_add + 16 * ot + 4 * g);
                const float4 t1 = ldg4(a.add_mol + (size_t)a.mol_of[min(atom1, a.n_atoms - 1)] * a.ld_add + 16 * ot + 4 * g);
                acc0 = f32x4{t0.x, t0.y, t0.z, t0.w}; acc1 = f32x4{t1.x, t1.y, t1.z, t1.w};
            }
            const u32x4 *f0 = frag + (size_t)(c * 2) * NB * 64 + lane;
            const u32x4 *f1 = frag + (size_t)((two ? c + 1 : c) * 2) * NB * 64 + lane;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const u32x4 h0 = f0[b * 64], l0 = f0[(NB + b) * 64];
                const u32x4 h1 = f1[b * 64], l1 = f1[(NB + b) * 64];
                if constexpr (!P1) {
                    acc0 = mfma_f16(w[1][b], h0, acc0); acc1 = mfma_f16(w[1][b], h1, acc1);      // smallest terms first
                    acc0 = mfma_f16(w[0][b], l0, acc0); acc1 = mfma_f16(w[0][b], l1, acc1);
                }
                acc0 = mfma_f16(w[0][b], h0, acc0); acc1 = mfma_f16(w[0][b], h1, acc1);
            }
            if (ot_ok && atom0 < a.n_atoms) stg4(a.out + (size_t)atom0 * a.ld_out + 16 * ot + 4 * g, float4{acc0[0], acc0[1], acc0[2], acc0[3]});
            if (ot_ok && two && atom1 < a.n_atoms) stg4(a.out + (size_t)atom1 * a.ld_out + 16 * ot + 4 * g, float4{acc1[0], acc1[1], acc1[2], acc1[3]});
        }
        SM_TICK(a.stamps, 3);
    }
    SM_STAMP(a.stamps, 4);
}

template <int H>
struct Chain16Lds {
    static constexpr int NB = H / 32, CC = CHAIN_COLS;
    static constexpr int FRAG = 2 * NB * CC * 64;          // u32x4 per fragment buffer of K = H
    static constexpr int XS = H + 4;
    static constexpr int PRE = CC * 16 * XS;
    static constexpr size_t BYTES = (size_t)3 * FRAG * 16 + (size_t)2 * PRE * 4;
};

// shared device pieces of the chain / prologue kernels
template <int H, bool P1 = false>
struct Node16 {
    static constexpr int NB = H / 32, CC = CHAIN_COLS, XS = Chain16Lds<H>::XS, LPC = NB * 4;
    // w[2][KB]: this wave's block of a split image
    template <int KB>
    SM_DEV static void load_w(const float *img, int tile, int lane, u32x4 (&w)[2][KB]) {
        const u32x4 *wi = reinterpret_cast<const u32x4 *>(img) + (size_t)tile * 2 * KB * 64 + lane;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int b = 0; b < KB; ++b) w[p][b] = wi[(p * KB + b) * 64];
    }
    // acc[c] += W * X_c over KB k-steps; fragments at f[((piece * KB + b) * CC + c) * 64 + slot]
    template <int KB>
    SM_DEV static void gemm(const u32x4 (&w)[2][KB], const u32x4 *f, f32x4 (&acc)[CC], int lane) {
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            u32x4 xh[CC], xl[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                xh[c] = f[((0 * KB + b) * CC + c) * 64 + frag_slot(b, lane)];
                xl[c] = f[((1 * KB + b) * CC + c) * 64 + frag_slot(b, lane)];
            }
            if constexpr (!P1) {
#pragma unroll
                for (int c = 0; c < CC; ++c) acc[c] = mfma_f16(w[1][b], xh[c], acc[c]);      // smallest terms first
#pragma unroll
                for (int c = 0; c < CC; ++c) acc[c] = mfma_f16(w[0][b], xl[c], acc[c]);
            }
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = mfma_f16(w[0][b], xh[c], acc[c]);
        }
    }
    SM_DEV static void store_pre(float *pre, const f32x4 (&acc)[CC], int n, int f0) {
#pragma unroll
        for (int c = 0; c < CC; ++c) stg4(pre + (c * 16 + n) * XS + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    }
    // activation of a pre-activation buffer -> fragments: LPC lanes per column, 8 features per lane
    SM_DEV static void normalise(const float *pre, int mode, const float *gam, const float *bet, u32x4 *fo, int ot, int lane, int *range_flag) {
        const int col = ot * (64 / LPC) + lane / LPC, ln = lane % LPC;
        const int b = ln >> 2, gg = ln & 3;
        const int fa = 32 * b + 4 * gg;
        const float4 p0 = ldg4(pre + col * XS + fa), p1 = ldg4(pre + col * XS + fa + 16);
        float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        if (mode == NODE_LN_RELU) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
            const float mean = seg_sum<LPC>(s) * (1.0f / H);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
            const float var = seg_sum<LPC>(q) * (1.0f / H);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            const float4 g0 = ldg4(gam + fa), g1 = ldg4(gam + fa + 16), b0 = ldg4(bet + fa), b1 = ldg4(bet + fa + 16);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf((v[i] - mean) * rstd * ga[i] + be[i], 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (v[i] > 20.f ? v[i] : log1pf(expf(v[i]))) - 0.6931471805599453f;
        }
        u32x4 hi, lo;
        split2_8(v, hi, lo, range_flag);
        u32x4 *dst = fo + (b * CC + (col >> 4)) * 64 + frag_slot(b, gg * 16 + (col & 15));
        dst[0] = hi; dst[NB * CC * 64] = lo;
    }
};

// The node stage of a layer on the column block [first_atom, first_atom + ncols) (ncols <= 16 CC), waves 0 .. NT - 1 owning
// one 16-row block of output features each.
//   FUSED = false: the body of node_chain16_kernel (a workgroup of NT waves; [att | h] rows staged from global memory).
//   FUSED = true:  the tail of x2h_chain16_kernel: the workgroup's waves (>= NT of them) have just produced the attention rows
//                  of these atoms (keep[], storing lanes only: edge16_body<KEEP>); the LDS they read the edge images from
//                  becomes the fragment buffers, so a barrier separates the two uses.  Weights and h rows are requested
//                  BEFORE that barrier (a wave that has finished its edge job has its registers free while the slower
//                  waves finish theirs).  Waves >= NT help with the staging and retire.
template <int H, bool FUSED, int SEGW, bool P1 = false>
SM_DEV void chain16_body(const NodeChainArgs &a, int *range_flag, int first_atom, int ncols, const float4 (&keep)[H / 16], int nthreads) {
    using L = Chain16Lds<H>;
    using N16 = Node16<H, P1>;
    constexpr int NT = H / 16, NB = H / 32, CC = CHAIN_COLS;
    static_assert(CC == 2, "the normalise pass covers exactly 32 columns");
    extern __shared__ __attribute__((aligned(16))) unsigned char chain16_lds[];
    u32x4 *fin = reinterpret_cast<u32x4 *>(chain16_lds);   // [att | h] fragments (K = 2H); later the two hidden tiles
    u32x4 *fhid0 = fin, *fhid1 = fin + L::FRAG;
    u32x4 *fh = fin + 2 * L::FRAG;                         // fragments of the new h
    float *pre0 = reinterpret_cast<float *>(fin + 3 * L::FRAG), *pre1 = pre0 + L::PRE;
    const int lane = threadIdx.x & 63, ot = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int f0 = 16 * (ot < NT ? ot : 0) + 4 * g;
    auto atom_of = [&](int c) { return min(first_atom + c * 16 + n, a.n_atoms - 1); };
    auto atom_ok = [&](int c) { return c * 16 + n < ncols && first_atom + c * 16 + n < a.n_atoms; };

    SM_TICK(a.stamps, 0);
    // ---- stage 0: weights of the output MLP; [att | h] tiles -> fragments ----------------------------
    u32x4 w1[2][2 * NB], w2[2][NB];
    float4 hres[CC];
    float4 b1, b2;
    if constexpr (!FUSED) {
        N16::template load_w<2 * NB>(a.w1img6, ot, lane, w1);
        for (int idx = threadIdx.x; idx < CC * 2 * NB * 64; idx += NT * 64) {
            const int sl = idx & 63, sb = (idx >> 6) % (2 * NB), sc = idx / (64 * 2 * NB);
            const int at = min(first_atom + sc * 16 + (sl & 15), a.n_atoms - 1);
            const float *src = (sb < NB ? a.att + (size_t)at * H + 32 * sb : a.h + (size_t)at * H + 32 * (sb - NB)) + 4 * (sl >> 4);
            const float4 v0 = ldg4(src), v1 = ldg4(src + 16);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            u32x4 hi, lo;
            split2_8(v, hi, lo, range_flag);
            u32x4 *dst = fin + (sb * CC + sc) * 64 + frag_slot(sb, sl);
            dst[0] = hi; dst[2 * NB * CC * 64] = lo;
        }
        N16::template load_w<NB>(a.w2img6, ot, lane, w2);
#pragma unroll
        for (int c = 0; c < CC; ++c) hres[c] = ldg4(a.h + (size_t)atom_of(c) * H + f0);    // residual
        b1 = ldg4(a.b1 + f0); b2 = ldg4(a.b2 + f0);
        __syncthreads();
    } else {
        // requests first (they fly while the other waves finish their edge jobs) ...
        const bool worker = ot < NT;
        if (worker) {
            N16::template load_w<2 * NB>(a.w1img6, ot, lane, w1);
            N16::template load_w<NB>(a.w2img6, ot, lane, w2);
#pragma unroll
            for (int c = 0; c < CC; ++c) hres[c] = ldg4(a.h + (size_t)atom_of(c) * H + f0);
            b1 = ldg4(a.b1 + f0); b2 = ldg4(a.b2 + f0);
        }
        // h rows: CC * NB * 64 fragment items, one per thread of the first 8 waves' worth of threads
        const int hidx = threadIdx.x;
        const bool hstage = hidx < CC * NB * 64;
        const int hsl = hidx & 63, hsb = (hidx >> 6) % NB, hsc = hidx / (64 * NB);
        float4 hv0 = {0.f, 0.f, 0.f, 0.f}, hv1 = hv0;
        if (hstage) {
            const int at = min(first_atom + hsc * 16 + (hsl & 15), a.n_atoms - 1);
            const float *src = a.h + (size_t)at * H + 32 * hsb + 4 * (hsl >> 4);
            hv0 = ldg4(src); hv1 = ldg4(src + 16);
        }
        __syncthreads();                                     // every wave is done with the edge images
        // ... the attention rows this wave holds -> fragments of its columns (the storing lanes)
        if ((n % SEGW) == 0) {
            const int col = ot * (16 / SEGW) + n / SEGW;
            if (col < CC * 16) {
#pragma unroll
                for (int sb = 0; sb < NB; ++sb) {
                    const float v[8] = {keep[2 * sb].x, keep[2 * sb].y, keep[2 * sb].z, keep[2 * sb].w,
                                        keep[2 * sb + 1].x, keep[2 * sb + 1].y, keep[2 * sb + 1].z, keep[2 * sb + 1].w};
                    u32x4 hi, lo;
                    split2_8(v, hi, lo, range_flag);
                    u32x4 *dst = fin + (sb * CC + (col >> 4)) * 64 + frag_slot(sb, g * 16 + (col & 15));
                    dst[0] = hi; dst[2 * NB * CC * 64] = lo;
                }
            }
        }
        // columns beyond the workgroup's atoms: zero attention fragments
        for (int idx = threadIdx.x; idx < CC * NB * 64; idx += nthreads) {
            const int sl = idx & 63, sb = (idx >> 6) % NB, sc = idx / (64 * NB);
            if (sc * 16 + (sl & 15) >= ncols) {
                u32x4 *dst = fin + (sb * CC + sc) * 64 + frag_slot(sb, sl);
                dst[0] = u32x4{0u, 0u, 0u, 0u}; dst[2 * NB * CC * 64] = u32x4{0u, 0u, 0u, 0u};
            }
        }
        if (hstage) {
            const float v[8] = {hv0.x, hv0.y, hv0.z, hv0.w, hv1.x, hv1.y, hv1.z, hv1.w};
            u32x4 hi, lo;
            split2_8(v, hi, lo, range_flag);
            u32x4 *dst = fin + ((NB + hsb) * CC + hsc) * 64 + frag_slot(NB + hsb, hsl);
            dst[0] = hi; dst[2 * NB * CC * 64] = lo;
        }
        __syncthreads();
        if (!worker) return;
    }
    SM_TICK(a.stamps, 1);

    // ---- stage 1: h' = h + W2 relu(LN(W1 [att | h] + b1)) + b2 ---------------------------------------
    {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b1.x, b1.y, b1.z, b1.w};
        N16::template gemm<2 * NB>(w1, fin, acc, lane);
        N16::store_pre(pre0, acc, n, f0);
    }
    u32x4 wf0[2][NB], wf1[2][NB];                                    // first Linears of the follow-up MLPs
    if (a.n_follow > 0) N16::template load_w<NB>(a.f[0].w1img6, ot, lane, wf0);
    if (a.n_follow > 1) N16::template load_w<NB>(a.f[1].w1img6, ot, lane, wf1);
    __syncthreads();
    SM_TICK(a.stamps, 2);
    N16::normalise(pre0, NODE_LN_RELU, a.ln_g, a.ln_b, fhid0, ot, lane, range_flag);
    __syncthreads();
    SM_TICK(a.stamps, 3);
    {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b2.x, b2.y, b2.z, b2.w};
        N16::template gemm<NB>(w2, fhid0, acc, lane);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float4 hn = {acc[c][0] + hres[c].x, acc[c][1] + hres[c].y, acc[c][2] + hres[c].z, acc[c][3] + hres[c].w};
            if (atom_ok(c)) stg4(a.h_out + (size_t)atom_of(c) * H + f0, hn);
            // this lane's four features are half of fragment (k-step ot / 2, lane) of column tile c
            unsigned ph[2], pl[2];
            split2_pair(hn.x, hn.y, ph[0], pl[0]);
            split2_pair(hn.z, hn.w, ph[1], pl[1]);
            if (!(fmaxf(fmaxf(fabsf(hn.x), fabsf(hn.y)), fmaxf(fabsf(hn.z), fabsf(hn.w))) < 6.0e4f)) *range_flag = 1;
            uint2 *dst = reinterpret_cast<uint2 *>(fh + ((ot >> 1) * CC + c) * 64 + frag_slot(ot >> 1, lane)) + (ot & 1);
            dst[0] = uint2{ph[0], ph[1]};
            dst[NB * CC * 64 * 2] = uint2{pl[0], pl[1]};
        }
    }
    if (a.n_follow == 0) return;
    const bool on0 = ot < a.f[0].nt2, on1 = a.n_follow > 1 && ot < a.f[1].nt2;
    u32x4 wg0[2][NB], wg1[2][NB];                                    // second Linears of the follow-up MLPs
    if (on0) N16::template load_w<NB>(a.f[0].w2img6, ot, lane, wg0);
    __syncthreads();
    SM_TICK(a.stamps, 4);

    // ---- stage 2: follow-up MLPs on the new h ---------------------------------------------------------
    {
        const float4 ba = ldg4(a.f[0].b1 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{ba.x, ba.y, ba.z, ba.w};
        N16::template gemm<NB>(wf0, fh, acc, lane);
        N16::store_pre(pre0, acc, n, f0);
    }
    if (a.n_follow > 1) {
        const float4 bb = ldg4(a.f[1].b1 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{bb.x, bb.y, bb.z, bb.w};
        N16::template gemm<NB>(wf1, fh, acc, lane);
        N16::store_pre(pre1, acc, n, f0);
    }
    if (on1) N16::template load_w<NB>(a.f[1].w2img6, ot, lane, wg1);
    // per-node products of the next attentions: tiles ot, ot + NT, ... of [n_lin_tiles * 16][H], weights alternating between
    // two register sets so that the next block is in flight during the current product (as node_prologue16_kernel)
    if (!FUSED && a.n_lin_tiles > 0) {
        u32x4 wl[2][NB];
        N16::template load_w<NB>(a.lin_img16, ot, lane, wl);
        auto lin_tile = [&](int tile, const u32x4 (&w)[2][NB]) {
            f32x4 acc[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float4 t = a.add_mol ? ldg4(a.add_mol + (size_t)a.mol_of[atom_of(c)] * a.ld_add + 16 * tile + 4 * g)
                                           : float4{0.f, 0.f, 0.f, 0.f};
                acc[c] = f32x4{t.x, t.y, t.z, t.w};
            }
            N16::template gemm<NB>(w, fh, acc, lane);
#pragma unroll
            for (int c = 0; c < CC; ++c)
                if (atom_ok(c)) stg4(a.pre_out + (size_t)atom_of(c) * a.ld_out + 16 * tile + 4 * g, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
        };
        for (int tile = ot; tile < a.n_lin_tiles; tile += 2 * NT) {
            u32x4 wn[2][NB];
            const bool more1 = tile + NT < a.n_lin_tiles, more2 = tile + 2 * NT < a.n_lin_tiles;
            if (more1) N16::template load_w<NB>(a.lin_img16, tile + NT, lane, wn);
            lin_tile(tile, wl);
            if (more2) N16::template load_w<NB>(a.lin_img16, tile + 2 * NT, lane, wl);
            if (more1) lin_tile(tile + NT, wn);
        }
    }
    __syncthreads();
    SM_TICK(a.stamps, 5);
    N16::normalise(pre0, a.f[0].mode, a.f[0].ln_g, a.f[0].ln_b, fhid0, ot, lane, range_flag);
    if (a.n_follow > 1) N16::normalise(pre1, a.f[1].mode, a.f[1].ln_g, a.f[1].ln_b, fhid1, ot, lane, range_flag);
    __syncthreads();
    SM_TICK(a.stamps, 6);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!(k == 0 ? on0 : on1)) continue;
        const NodeFollow &F = a.f[k];
        const float4 b = ldg4(F.b2 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b.x, b.y, b.z, b.w};
        if (k == 0) N16::template gemm<NB>(wg0, fhid0, acc, lane); else N16::template gemm<NB>(wg1, fhid1, acc, lane);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            if (!atom_ok(c)) continue;
            const int atom = atom_of(c);
            if (f0 + 4 <= F.n_store && (F.ld_out & 3) == 0) {
                stg4(F.out + (size_t)atom * F.ld_out + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f0 + r < F.n_store) F.out[(size_t)atom * F.ld_out + f0 + r] = acc[c][r];
            }
        }
    }
    SM_STAMP(a.stamps, 7);
}

template <int H, bool P1 = false>
__global__ void __launch_bounds__(H * 4)
node_chain16_kernel(NodeChainArgs a, int *range_flag) {
    float4 keep[H / 16];
    chain16_body<H, false, 8, P1>(a, range_flag, blockIdx.x * CHAIN_COLS * 16, CHAIN_COLS * 16, keep, H * 4);
}

// x2h attention and the node stage of the same layer in one launch (one job per wave, >= H / 16 waves per workgroup): the
// attention rows of a workgroup's atoms never leave the CU, the node stage's first weights are requested while the slower
// waves still finish their edge jobs, and one launch (with its ramp, drain and cache write-back) per layer disappears.
template <int H, int KP, bool P1 = false>
__global__ void __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
x2h_chain16_kernel(Edge16Args e, NodeChainArgs a, int *range_flag) {
    float4 keep[H / 16];
    edge16_body<H, KP, false, true, true, P1>(e, keep);
    const int nwave = e.nwave, apj = 16 / KP;
    const int first_atom = (e.job_base + blockIdx.x * nwave) * apj;
    chain16_body<H, true, KP, P1>(a, range_flag, first_atom, nwave * apj, keep, nwave * 64);
}

template <int H, bool P1 = false>
__global__ void __launch_bounds__(H * 4)
node_prologue16_kernel(NodePrologueArgs a, int *range_flag) {
    SM_TICK(a.stamps, 0);
    using L = Chain16Lds<H>;
    using N16 = Node16<H, P1>;
    constexpr int NT = H / 16, NB = H / 32, CC = CHAIN_COLS, XS = L::XS;
    extern __shared__ __attribute__((aligned(16))) unsigned char chain16_lds[];
    u32x4 *fh = reinterpret_cast<u32x4 *>(chain16_lds);      // fragments of h0
    u32x4 *fhid = fh + L::FRAG;                             // hidden tile of the query MLP
    float *pre0 = reinterpret_cast<float *>(fh + 2 * L::FRAG);
    const int lane = threadIdx.x & 63, ot = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int ct0 = blockIdx.x * CC;
    const int f0 = 16 * ot + 4 * g;
    auto atom_of = [&](int c) { return min((ct0 + c) * 16 + n, a.n_atoms - 1); };
    auto atom_ok = [&](int c) { return (ct0 + c) * 16 + n < a.n_atoms; };

    // ---- bookkeeping of the evaluation ------------------------------------------------------------------
    const int step = a.step_ptr ? *a.step_ptr : 0;
    {
        const int gid = blockIdx.x * (H * 4) + threadIdx.x;           // (the launch uses H * 4 threads: no blockDim read)
        if (gid == 0 && a.step_ptr) *a.step_cur = step;
        for (int i = gid; i < a.bn_acc_len; i += gridDim.x * (H * 4)) a.bn_acc[i] = 0.0;
    }
    // ---- stage 0: embedding of the workgroup's atoms -> global h0 and LDS fragments ----------------------
    u32x4 wq1[2][NB], wl[2][NB];
    N16::template load_w<NB>(a.q.w1img6, ot, lane, wq1);
    for (int idx = threadIdx.x; idx < CC * NB * 64; idx += NT * 64) {
        const int sl = idx & 63, sb = (idx >> 6) % NB, sc = idx / (64 * NB);
        const int at_raw = (ct0 + sc) * 16 + (sl & 15);
        const int at = min(at_raw, a.n_atoms - 1);
        const int t = a.step_ptr ? a.t_first - step : a.t_mol[a.mol_of[at]];
        const float *te = a.ttab + (size_t)t * a.D;
        const int vi = min(max((int)a.v[at], 0), a.C - 1);       // out-of-range types are flagged by v_check_kernel
        const int fa = 32 * sb + 4 * (sl >> 4);
        float vv[8];
        if (a.etab) {                                         // the precomputed row of (t, v): the same sum, done once
            const float *row = a.etab + ((size_t)t * a.C + vi) * H + fa;
            const float4 r0 = ldg4(row), r1 = ldg4(row + 16);
            vv[0] = r0.x; vv[1] = r0.y; vv[2] = r0.z; vv[3] = r0.w; vv[4] = r1.x; vv[5] = r1.y; vv[6] = r1.z; vv[7] = r1.w;
        } else
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                      // y = (b + W[:, v]) + sum_k W[:, C + k] te[k], k ascending
            const int f = fa + 16 * hf;
            const float4 bb = ldg4(a.emb_b + f), wv = ldg4(a.emb_wT + (size_t)vi * H + f);
            float y[4] = {bb.x + wv.x, bb.y + wv.y, bb.z + wv.z, bb.w + wv.w};
            for (int k = 0; k < a.D; ++k) {
                const float4 wk = ldg4(a.emb_wT + (size_t)(a.C + k) * H + f);
                const float tk = te[k];
                y[0] += wk.x * tk; y[1] += wk.y * tk; y[2] += wk.z * tk; y[3] += wk.w * tk;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[4 * hf + j] = y[j];
        }
        if (at_raw < a.n_atoms) {
            stg4(a.h_out + (size_t)at * H + fa, float4{vv[0], vv[1], vv[2], vv[3]});
            stg4(a.h_out + (size_t)at * H + fa + 16, float4{vv[4], vv[5], vv[6], vv[7]});
        }
        u32x4 hi, lo;
        split2_8(vv, hi, lo, range_flag);
        u32x4 *dst = fh + (sb * CC + sc) * 64 + frag_slot(sb, sl);
        dst[0] = hi; dst[NB * CC * 64] = lo;
    }
    if (a.n_lin_tiles > 0) N16::template load_w<NB>(a.lin_img6, ot, lane, wl);
    const float4 b1 = ldg4(a.q.b1 + f0);
    SM_TICK(a.stamps, 1);
    __syncthreads();
    SM_TICK(a.stamps, 2);

    // ---- stage 1: first Linear of the query MLP; the per-node linear outputs of the edge MLPs ------------
    {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b1.x, b1.y, b1.z, b1.w};
        N16::template gemm<NB>(wq1, fh, acc, lane);
        N16::store_pre(pre0, acc, n, f0);
    }
    const bool onq = ot < a.q.nt2;
    if (onq) N16::template load_w<NB>(a.q.w2img6, ot, lane, wq1);   // second Linear of the query MLP (reuses the registers)
    auto lin_tile = [&](int tile, const u32x4 (&w)[2][NB]) {
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float4 t = a.add_mol ? ldg4(a.add_mol + (size_t)a.mol_of[atom_of(c)] * a.ld_add + 16 * tile + 4 * g)
                                       : float4{0.f, 0.f, 0.f, 0.f};
            acc[c] = f32x4{t.x, t.y, t.z, t.w};
        }
        N16::template gemm<NB>(w, fh, acc, lane);
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (atom_ok(c)) stg4(a.pre_out + (size_t)atom_of(c) * a.ld_out + 16 * tile + 4 * g, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    };
    for (int tile = ot; tile < a.n_lin_tiles; tile += 2 * NT) {
        u32x4 wn[2][NB];
        const bool more1 = tile + NT < a.n_lin_tiles, more2 = tile + 2 * NT < a.n_lin_tiles;
        if (more1) N16::template load_w<NB>(a.lin_img6, tile + NT, lane, wn);
        lin_tile(tile, wl);
        if (more2) N16::template load_w<NB>(a.lin_img6, tile + 2 * NT, lane, wl);
        if (more1) lin_tile(tile + NT, wn);
    }
    SM_TICK(a.stamps, 3);
    __syncthreads();
    SM_TICK(a.stamps, 4);
    N16::normalise(pre0, NODE_LN_RELU, a.q.ln_g, a.q.ln_b, fhid, ot, lane, range_flag);
    __syncthreads();
    SM_TICK(a.stamps, 5);
    if (onq) {
        const float4 b = ldg4(a.q.b2 + f0);
        f32x4 acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = f32x4{b.x, b.y, b.z, b.w};
        N16::template gemm<NB>(wq1, fhid, acc, lane);
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (atom_ok(c)) stg4(a.q.out + (size_t)atom_of(c) * a.q.ld_out + f0, float4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]});
    }
    SM_STAMP(a.stamps, 6);
}
