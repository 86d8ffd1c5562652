#!/usr/bin/env python3
"""Developer tool (GPU box): randomised shapes through the training nodes (HipMLP, HipEdgeMLP, HipSegAttention, HipVN) against
torch autograd of the reference formulation in float64.  Exercises the GEMM's float4 / scalar dispatch (unaligned column blocks,
K / M / N tails), split reductions with few and many partials, ragged graphs.
    python tools/fuzz_training_ops.py [--cases 60] [--seed 0]"""
import argparse, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shapemol_amd.training import EdgeGraph, HipEdgeMLP, HipMLP, HipSegAttention, HipVN

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
DEV = "cuda:0"
rng = np.random.RandomState(a.seed)
g = torch.Generator().manual_seed(a.seed)
mk = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(DEV)  # noqa: E731
# relative to the tensor's largest entry, with a floor: a batch-normalised VN block is scale-invariant in its weights, so with one
# input row the exact weight gradients are ~0 and only rounding noise is left to compare
rel = lambda p, q: float((p.detach().double() - q.detach()).abs().max() / q.detach().abs().max().clamp(min=1e-3)) if q.numel() else 0.0  # noqa: E731
worst, noisy, fails = {}, {}, 0


def check(name, desc, outs, refs, refs32, tol_out=2e-5, tol_grad=2e-4):
    """Error against float64 autograd, judged against what torch's own float32 autograd of the same formulation loses on the same
    inputs (a ReLU or leaky-ReLU mask that flips at a pre-activation within rounding of 0 moves single entries by 1e-3 of the
    largest one in either implementation): fail = beyond the tolerance AND more than 4 x torch-float32's own error."""
    global fails
    e = [rel(p, q) for p, q in zip(outs, refs)]
    e32 = [rel(p, q) for p, q in zip(refs32, refs)]
    bad = [i for i in range(len(e)) if e[i] > (tol_out if i == 0 else tol_grad) and e[i] > 4 * e32[i]]
    worst[name] = max(worst.get(name, 0.0), max(e))
    noisy[name] = noisy.get(name, 0) + int(any(e[i] > (tol_out if i == 0 else tol_grad) for i in range(len(e))) and not bad)
    if bad or not all(torch.isfinite(p).all() for p in outs):
        fails += 1
        print("FAIL", name, desc, ["%.2e" % x for x in e], "torch fp32:", ["%.2e" % x for x in e32], flush=True)


def graph(n, max_deg):
    deg = torch.from_numpy(rng.randint(0, max_deg + 1, n))
    ptr = torch.zeros(n + 1, dtype=torch.int64); ptr[1:] = torch.cumsum(deg, 0)
    E = int(ptr[-1])
    dst = torch.repeat_interleave(torch.arange(n), deg)
    src = torch.from_numpy(rng.randint(0, n, max(E, 1)))[:E]
    return EdgeGraph(src.to(DEV), dst.to(DEV), ptr.to(DEV)), E


def clear_of_zero(x_rows, ins, o):
    """A ReLU input within float32 rounding of 0 makes the mask, and with it whole gradient rows, a matter of rounding (seen: one
    row of 47 k with a pre-activation of -1.5e-7 moved dz by 0.09): shift beta until no pre-activation is that close."""
    w1, b1, ga, be = ins[o:o + 4]
    for _ in range(8):
        pre = F.layer_norm(F.linear(x_rows.double(), w1.double(), b1.double()), (w1.shape[0],), ga.double(), be.double(), 1e-5)
        close = (pre.abs() < 2e-5).any(0)
        if not bool(close.any()):
            return
        be += close.float() * 1.7e-3


def mlp_ref(x, w1, b1, ga, be, w2, b2):
    return F.linear(torch.relu(F.layer_norm(F.linear(x, w1, b1), (w1.shape[0],), ga, be, 1e-5)), w2, b2)


for case in range(a.cases):
    hidden = int(rng.choice([16, 32, 64, 128, 256]))
    n_out = int(rng.choice([1, 3, 16, 20, 128, 200]))
    # ---- HipMLP
    rows, k_in = int(rng.choice([1, 5, 63, 64, 65, 300, 2000, 9000])), int(rng.choice([1, 7, 20, 32, 100, 128, 308]))
    ins = [mk(rows, k_in), mk(hidden, k_in, sc=k_in ** -0.5), mk(hidden, sc=0.3), 1 + mk(hidden, sc=0.2), mk(hidden, sc=0.3), mk(n_out, hidden, sc=hidden ** -0.5), mk(n_out, sc=0.3)]
    dy = mk(rows, n_out)
    clear_of_zero(ins[0], ins, 1)
    A = [t.clone().requires_grad_(True) for t in ins]; y = HipMLP.apply(*A); y.backward(dy)
    R = []
    for dt in (torch.float64, torch.float32):
        Bd = [t.to(dt).clone().requires_grad_(True) for t in ins]; yr = mlp_ref(*Bd); yr.backward(dy.to(dt))
        R.append([yr.detach()] + [t.grad for t in Bd])
    check("HipMLP", (rows, k_in, hidden, n_out), [y] + [t.grad for t in A], R[0], R[1])
    # ---- HipEdgeMLP
    n = int(rng.choice([1, 2, 17, 300, 3000]))
    G, E = graph(n, int(rng.choice([1, 8, 32])))
    if E > 0:
        kr, kn, ks = int(rng.choice([3, 20, 24])), int(rng.choice([8, 30, 128])), int(rng.choice([0, 5, 32]))
        K1 = kr + 2 * kn + ks
        ins = [mk(E, kr), mk(n, kn), mk(n, ks), mk(hidden, K1, sc=K1 ** -0.5), mk(hidden, sc=0.3), 1 + mk(hidden, sc=0.2), mk(hidden, sc=0.3), mk(n_out, hidden, sc=hidden ** -0.5), mk(n_out, sc=0.3)]
        dy = mk(E, n_out)
        clear_of_zero(torch.cat([ins[0], ins[1][G.dst], ins[1][G.src], ins[2][G.dst]], -1), ins, 3)
        A = [t.clone().requires_grad_(True) for t in ins]; y = HipEdgeMLP.apply(A[0], A[1], A[2], G, *A[3:]); y.backward(dy)
        ga, R = [t.grad for t in A], []
        if ks == 0: ga[2] = torch.zeros(1)
        for dt in (torch.float64, torch.float32):
            Bd = [t.to(dt).clone().requires_grad_(True) for t in ins]
            yr = mlp_ref(torch.cat([Bd[0], Bd[1][G.dst], Bd[1][G.src], Bd[2][G.dst]], -1), *Bd[3:]); yr.backward(dy.to(dt))
            gb = [t.grad for t in Bd]
            if ks == 0: gb[2] = torch.zeros(1)
            R.append([yr.detach()] + gb)
        check("HipEdgeMLP", (E, n, kr, kn, ks, hidden, n_out), [y] + ga, R[0], R[1])
        # ---- HipSegAttention
        heads, dh, W = int(rng.choice([1, 4, 16])), int(rng.choice([2, 4, 8])), int(rng.choice([1, 3, 8]))
        ins = [mk(n, heads * dh) * 2, mk(E, heads * dh) * 2, mk(E, heads, W)]
        do = mk(n, heads, W)
        A = [t.clone().requires_grad_(True) for t in ins]; o = HipSegAttention.apply(A[0], A[1], A[2], G.ptr, heads); o.backward(do)
        R = []
        for dt in (torch.float64, torch.float32):
            Bd = [t.to(dt).clone().requires_grad_(True) for t in ins]
            logit = (Bd[0][G.dst].view(-1, heads, dh) * Bd[1].view(-1, heads, dh) / np.sqrt(dh)).sum(-1)
            idx = G.dst.view(-1, 1).expand_as(logit)
            mx = torch.full((n, heads), float("-inf"), device=DEV, dtype=dt).scatter_reduce(0, idx, logit.detach(), "amax")
            ex = torch.exp(logit - mx[G.dst])
            al = ex / torch.zeros((n, heads), device=DEV, dtype=dt).index_add(0, G.dst, ex)[G.dst]
            orf = torch.zeros((n, heads, W), device=DEV, dtype=dt).index_add(0, G.dst, al.unsqueeze(-1) * Bd[2]); orf.backward(do.to(dt))
            R.append([orf.detach()] + [t.grad for t in Bd])
        check("HipSegAttention", (n, E, heads, dh, W), [o] + [t.grad for t in A], R[0], R[1])
    # ---- HipVN
    nm = int(rng.choice([1, 3, 40, 400]))
    counts = torch.from_numpy(rng.randint(1, 40, nm))
    batch = torch.repeat_interleave(torch.arange(nm), counts).to(DEV)
    na, ro, rs, ch = int(counts.sum()), int(rng.choice([0, 3, 16])), int(rng.choice([0, 4, 32])), int(rng.choice([1, 8, 16, 32]))
    if ro + rs == 0: rs = 4      # (one input row: the block is scale-invariant in both weights, their exact gradients are ~0)
    training = bool(rng.randint(2)) and na > 1
    cin = 1 + ro + rs
    x, o3, shape = mk(na, 3), mk(na, ro, 3, sc=0.5), mk(nm, rs, 3)
    ws = [mk(ch, cin, sc=cin ** -0.5), mk(ch, cin, sc=cin ** -0.5), 1 + mk(ch, sc=0.2), mk(ch, sc=0.3)]
    rm0, rv0 = mk(ch, sc=0.1) + 1.0, torch.rand(ch, generator=g).to(DEV) + 0.5
    go = mk(na, 3)
    A = [t.clone().requires_grad_(True) for t in [x, o3] + ws]
    out = HipVN.apply(A[0], A[1], shape, batch, A[2], A[3], A[4], A[5], rm0.clone(), rv0.clone(), training); out.backward(go)
    ga, R = [t.grad for t in A], []
    if ro == 0: ga[1] = torch.zeros(1)
    for dt in (torch.float64, torch.float32):
        Bd = [t.to(dt).clone().requires_grad_(True) for t in [x, o3] + ws]
        z = torch.cat((Bd[0].unsqueeze(1), Bd[1], shape.to(dt)[batch]), dim=1)
        pf = torch.einsum("oc,ncd->nod", Bd[2], z); nrm = torch.sqrt((pf * pf).sum(2)) + 1e-6
        mean, var = (nrm.mean(0), ((nrm - nrm.mean(0)) ** 2).mean(0)) if training else (rm0.to(dt), rv0.to(dt))
        nbn = (nrm - mean) / torch.sqrt(var + 1e-5) * Bd[4] + Bd[5]
        pf = pf / nrm.unsqueeze(2) * nbn.unsqueeze(2)
        d = torch.einsum("oc,ncd->nod", Bd[3], z); dot = (pf * d).sum(2, keepdim=True); mask = (dot >= 0).to(dt)
        ref = (0.2 * pf + 0.8 * (mask * pf + (1 - mask) * (pf - (dot / ((d * d).sum(2, keepdim=True) + 1e-6)) * d))).mean(dim=1); ref.backward(go.to(dt))
        gb = [t.grad for t in Bd]
        if ro == 0: gb[1] = torch.zeros(1)
        R.append([ref.detach()] + gb)
    check("HipVN", (na, nm, ro, rs, ch, training), [out] + ga, R[0], R[1], tol_out=5e-5, tol_grad=2e-4)
print(f"{a.cases} cases, {fails} failures; worst relative error per node:", {k: "%.2e" % v for k, v in worst.items()},
      "; cases beyond the tolerance where torch float32 is as far off (mask flips):", noisy)
sys.exit(1 if fails else 0)
