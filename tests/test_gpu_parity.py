"""Parity of the HIP path (through the C ABI) against the golden vectors of the reference and
against the CPU oracle on the same inputs.  Run on the GPU box:  pytest tests -m gpu"""
import json

import numpy as np
import pytest
import torch

from util import O, T, golden, hash_noise, hip_model, maxabs, oracle_model, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL = 2e-5      # one forward, float32, values up to ~4
POS_TOL = 1e-4      # BASELINE.json north_star: coordinates within 1e-4 abs, atom types exact


def run_forward(m, f, tkey, pos="pos", v="v"):
    with torch.no_grad():
        return m(T(f[pos], DEV), T(f[v], DEV), T(f["batch"], DEV), T(f["shape"], DEV), T(f[tkey], DEV))


def test_library_loaded_and_device():
    from shapemol_amd import _lib
    assert _lib.load().shapemol_abi_version() == _lib.ABI_VERSION == 2
    assert torch.cuda.is_available()


def test_graph_stage_neighbours_and_edge_weights():
    """kNN neighbour lists are identical (integer-exact) and e_w within 1e-6 of the reference."""
    m = hip_model()
    f = golden("forward_b4.npz")
    m.set_option("stop_layer", 0)
    try:
        run_forward(m, f, "t999_t")
        n = len(f["batch"])
        nbr = m.debug_read("nbr", (n, 8), np.int32)
        ew = m.debug_read("ew", (n, 8), np.float32)
    finally:
        m.set_option("stop_layer", -1)
    src, dst = f["t999_edge_index"]
    assert np.array_equal(dst, np.repeat(np.arange(n), 8))
    assert np.array_equal(nbr.reshape(-1), src.astype(np.int32))
    ref = 1 / (1 + np.exp(-f["t999_ew_logit"].astype(np.float64).reshape(n, 8)))
    assert np.abs(ew - ref).max() < 1e-6


@pytest.mark.parametrize("nl", [1, 2, 4, 8])
def test_layer_taps(nl):
    """h and x after the first `nl` layers against the reference's per-layer hooks."""
    m = hip_model()
    f = golden("forward_b4.npz")
    m.set_option("stop_layer", nl)
    try:
        out = run_forward(m, f, "t999_t")
    finally:
        m.set_option("stop_layer", -1)
    x_ref = f["pos"].astype(np.float32).copy()
    for l in range(nl):
        x_ref = x_ref + f[f"t999_dx_{l}"]
    assert maxabs(out["pred_ligand_h"], f[f"t999_h_{nl - 1}"]) < FWD_TOL
    assert maxabs(out["pred_ligand_pos"], x_ref) < FWD_TOL


@pytest.mark.parametrize("name", ["t999", "t500", "t0", "tmix"])
def test_forward_b4_golden(name):
    m = hip_model()
    f = golden("forward_b4.npz")
    out = run_forward(m, f, name + "_t")
    for k in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
        assert maxabs(out[k], f[f"{name}_{k}"]) < FWD_TOL, k


@pytest.mark.parametrize("opts", [{"edge_bf16": 0}, {"edge_bf16": 2}, {"lin_bf16": 0, "chain_bf16": 0}, {"vn_fuse": 1}, {"vn_fuse": 0}],
                         ids=["edge_fp32", "edge_phases", "node_fp32", "vn_grid_barrier", "vn_separate"])
def test_forward_alternative_kernels_golden(opts):
    """The optional kernel variants behind shapemol_set_option compute the same forward (ragged batch too)."""
    m = hip_model()
    try:
        for k, v in opts.items():
            m.set_option(k, v)
        f = golden("forward_b4.npz")
        out = run_forward(m, f, "t500_t")
        for k in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
            assert maxabs(out[k], f[f"t500_{k}"]) < FWD_TOL, k
        f = golden("forward_ragged.npz")
        out = run_forward(m, f, "t")
        for k in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
            assert maxabs(out[k], f[k]) < FWD_TOL, k
        assert int(m.debug_read("vn_err", (1,), np.int32)[0]) == 0
    finally:
        for k in opts:
            m.set_option(k, {"edge_bf16": 1, "lin_bf16": 1, "chain_bf16": 1, "vn_fuse": 2}[k])


def test_forward_ragged_golden():
    """1-, 2-, 5-atom molecules: fewer than k neighbours / none at all."""
    m = hip_model()
    f = golden("forward_ragged.npz")
    out = run_forward(m, f, "t")
    for k in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
        assert maxabs(out[k], f[k]) < FWD_TOL, k


@pytest.mark.parametrize("tag", ["small", "k32"])
def test_forward_variants_golden(tag):
    """reduced-width model (H=32, 4 heads, 2 layers) and the k=32 / 40-80 atom stress variant."""
    f = golden(f"forward_{tag}.npz")
    ov = json.loads(str(f["overrides"]))
    m = hip_model(seed=9, **ov)
    out = run_forward(m, f, "t", pos="init_pos", v="init_v")
    for k in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
        assert maxabs(out[k], f[k]) < FWD_TOL, k


@pytest.mark.parametrize("k", [4, 12, 16])
def test_forward_other_k_vs_oracle(k):
    """Neighbour counts other than the configured 8: k = 4 (half-empty 8-slot tiles), k = 12 and 16 (the 16-slot,
    one-atom-per-job instantiation of the edge kernels), with molecules both smaller and larger than k + 1."""
    m = hip_model(seed=5, knn=k)
    sd, dm, _, _ = oracle_model(seed=5, knn=k)
    bb = synth.synthetic_batch(12, seed=77, atoms_range=(6, 30))
    t = (synth.hash_u24(12, 3, 3) % 1000).astype(np.int64)
    ref = O.score(sd, dm, T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), T(t))
    with torch.no_grad():
        out = m(T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV), T(t, DEV))
    for key in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
        assert maxabs(out[key], ref[key]) < FWD_TOL, key


@pytest.mark.parametrize("nmol,rng", [(1, (17, 17)), (1, (1, 1)), (3, (33, 48))], ids=["one_molecule", "one_atom", "large_molecules"])
def test_forward_extreme_batches_vs_oracle(nmol, rng):
    """A batch of a single molecule, of a single atom (no edges at all; batch-norm over one sample), and molecules larger
    than anything in the MOSES prior."""
    m = hip_model()
    sd, dm, _, _ = oracle_model()
    bb = synth.synthetic_batch(nmol, seed=31, atoms_range=rng)
    t = np.full(nmol, 321, np.int64)
    ref = O.score(sd, dm, T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), T(t))
    with torch.no_grad():
        out = m(T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV), T(t, DEV))
    for key in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
        assert maxabs(out[key], ref[key]) < FWD_TOL, key


def test_forward_b256_vs_oracle():
    """BASELINE config-2 size (256 molecules, ~5.5k atoms) against the CPU oracle, one evaluation."""
    m = hip_model()
    sd, dm, _, _ = oracle_model()
    bb = synth.synthetic_batch(256, seed=2021)
    t = (synth.hash_u24(256, 9, 9) % 1000).astype(np.int64)
    ref = O.score(sd, dm, T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), T(t))
    with torch.no_grad():
        out = m(T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV), T(t, DEV))
    for k in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
        assert maxabs(out[k], ref[k]) < FWD_TOL, k
    # the same evaluation with ONE wave per workgroup of the edge kernels: ~11 jobs per wave instead of one, i.e. the
    # multi-job paths (next job's loads in flight, per-job statistics of the fused epilogue) that larger batches take
    try:
        m.set_option("edge_waves", 1)
        with torch.no_grad():
            out2 = m(T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV), T(t, DEV))
    finally:
        m.set_option("edge_waves", 0)
    for k in ("pred_ligand_pos", "pred_ligand_h", "pred_ligand_v"):
        assert maxabs(out2[k], ref[k]) < FWD_TOL, k


def _chain(m, init_pos, init_v, batch, shape, steps, eps, u, **kw):
    return m.sample_diffusion(T(init_pos, DEV), T(init_v, DEV), T(batch, DEV), T(shape, DEV).view(len(shape), -1),
                              num_steps=steps, center_pos_mode="none", noise=(T(eps, DEV), T(u, DEV)), **kw)


@pytest.mark.parametrize("use_graph", [False, True])
def test_chain_b4_s50_golden(use_graph):
    """BASELINE config 1 analogue with the reference's own torch-RNG draws replayed: atom types
    integer-exact at every step, coordinates within 1e-4."""
    m = hip_model()
    c = golden("chain_b4_s50_torchrng.npz")
    r = _chain(m, c["init_pos"], c["init_v"], c["batch"], c["shape"], 50, c["eps"], c["u"], use_graph=use_graph)
    assert np.array_equal(r["v"].cpu().numpy(), c["v"])
    assert np.array_equal(torch.stack(r["v_traj"]).numpy(), c["v_traj"])
    assert maxabs(r["pos"], c["pos"]) < POS_TOL
    assert maxabs(torch.stack(r["pos_traj"]), c["pos_traj"]) < POS_TOL
    assert maxabs(r["v0_traj"][-1], c["v0_last"]) < POS_TOL
    assert maxabs(r["vt_traj"][-1], c["vt_last"]) < POS_TOL
    assert maxabs(r["pos_cond_traj"][-1], c["pos_cond_last"]) < POS_TOL
    assert maxabs(r["v_cond_traj"][-1], c["v_cond_last"]) < POS_TOL
    assert len(r["pos_traj"]) == 50 and r["pos_traj"][0].device.type == "cpu" and r["pos_cond_traj"][0].is_cuda
    assert r["pos_uncond_traj"] == [] and r["v_uncond_traj"] == []


def test_init_v_sampling_matches_reference_draw():
    """log_sample_categorical on the recorded uniforms reproduces the reference's initial atom types."""
    import shapemol_amd
    c = golden("chain_b4_s50_torchrng.npz")
    n = len(c["batch"])
    v = shapemol_amd.log_sample_categorical(torch.zeros(n, 15, device=DEV), u=T(c["init_u"], DEV))
    assert np.array_equal(v.cpu().numpy(), c["init_v"])


@pytest.mark.parametrize("tag", ["b16_s100", "b4_s1000"])
def test_chain_hash_noise_golden(tag):
    """Full 1000-step chain (B=4) and a 100-step B=16 chain against the reference's end state."""
    m = hip_model()
    c = golden(f"chain_{tag}_hash.npz")
    B, S, seed, every = int(c["B"]), int(c["S"]), int(c["seed"]), int(c["every"])
    bb = synth.synthetic_batch(B, seed=seed)
    eps, u = hash_noise(len(bb["batch"]), S, seed)
    r = _chain(m, bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"], S, eps, u)
    assert np.array_equal(r["v"].cpu().numpy(), c["v"])
    assert np.array_equal(torch.stack(r["v_traj"][::every]).numpy(), c["v_traj_sub"])
    assert maxabs(r["pos"], c["pos"]) < POS_TOL
    assert maxabs(torch.stack(r["pos_traj"][::every]), c["pos_traj_sub"]) < POS_TOL


def test_chain_b256_vs_oracle_30_steps():
    """Headline batch size: 256 molecules, first 30 reverse steps, against the CPU oracle."""
    m = hip_model()
    sd, dm, _, _ = oracle_model()
    bb = synth.synthetic_batch(256, seed=2021)
    n, S = len(bb["batch"]), 30
    eps, u = hash_noise(n, S, 2021)
    ref = O.sample_chain(sd, dm, T(bb["init_pos"]), T(bb["init_v"]), T(bb["batch"]), T(bb["shape"]), S,
                         lambda s: (eps[s], u[s]), keep_traj=False)
    r = _chain(m, bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"], S, eps, u, return_traj=False)
    assert np.array_equal(r["v"].cpu().numpy(), ref["v"].numpy())
    assert maxabs(r["pos"], ref["pos"]) < POS_TOL


# ---- size-independent properties at full size ---------------------------------------------
def _rotation(seed):
    q, _ = np.linalg.qr(np.random.RandomState(seed).randn(3, 3))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.astype(np.float32)


def test_so3_equivariance_b256():
    """Rotating x and the shape latent rotates pred_ligand_pos and leaves logits invariant
    (the net is not translation-equivariant)."""
    m = hip_model()
    bb = synth.synthetic_batch(256, seed=5)
    t = T(np.full(256, 700, np.int64), DEV)
    Q = _rotation(1)
    args = (T(bb["init_v"], DEV), T(bb["batch"], DEV))
    with torch.no_grad():
        a = m(T(bb["init_pos"], DEV), args[0], args[1], T(bb["shape"], DEV), t)
        b = m(T(bb["init_pos"] @ Q.T, DEV), args[0], args[1], T(bb["shape"] @ Q.T, DEV), t)
    assert maxabs(a["pred_ligand_pos"].cpu().numpy() @ Q.T, b["pred_ligand_pos"]) < 5e-5
    assert maxabs(a["pred_ligand_v"], b["pred_ligand_v"]) < 5e-5
    assert maxabs(a["pred_ligand_h"], b["pred_ligand_h"]) < 5e-5


def test_determinism_and_graph_equals_eager():
    m = hip_model()
    bb = synth.synthetic_batch(64, seed=8)
    eps, u = hash_noise(len(bb["batch"]), 20, 8)
    r1 = _chain(m, bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"], 20, eps, u, use_graph=True)
    r2 = _chain(m, bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"], 20, eps, u, use_graph=True)
    r3 = _chain(m, bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"], 20, eps, u, use_graph=False)
    assert torch.equal(r1["v"], r2["v"]) and torch.equal(r1["v"], r3["v"])
    assert maxabs(r1["pos"], r2["pos"]) < 1e-6 and maxabs(r1["pos"], r3["pos"]) < 1e-6


def test_batch_coupling_only_through_batchnorm():
    """Train-mode VN batch-norm couples the molecules of a batch: perturbing other
    molecules changes molecule 0, while permuting whole molecules only permutes the result."""
    m = hip_model()
    bb = synth.synthetic_batch(8, seed=4)
    t = T(np.full(8, 300, np.int64), DEV)
    counts = bb["counts"]
    off = np.concatenate([[0], np.cumsum(counts)])
    with torch.no_grad():
        a = m(T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV), t)
        perm = np.array([3, 0, 7, 1, 2, 6, 5, 4])
        idx = np.concatenate([np.arange(off[p], off[p + 1]) for p in perm])
        b = m(T(bb["init_pos"][idx], DEV), T(bb["init_v"][idx], DEV),
              T(np.repeat(np.arange(8), counts[perm]), DEV), T(bb["shape"][perm], DEV), t)
        pos2 = bb["init_pos"].copy()
        pos2[off[1]:] += 0.05
        c = m(T(pos2, DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV), t)
    assert maxabs(a["pred_ligand_pos"][idx], b["pred_ligand_pos"]) < 2e-5
    assert maxabs(a["pred_ligand_pos"][:off[1]], c["pred_ligand_pos"][:off[1]]) > 1e-6


def test_philox_noise_statistics_and_seed_control():
    """Device-noise mode: reproducible for a seed, different across seeds, sane moments."""
    m = hip_model()
    bb = synth.synthetic_batch(32, seed=6)
    args = (T(bb["init_pos"], DEV), T(bb["init_v"], DEV), T(bb["batch"], DEV), T(bb["shape"], DEV).view(32, -1))
    r1 = m.sample_diffusion(*args, num_steps=5, center_pos_mode="none", seed=123)
    r2 = m.sample_diffusion(*args, num_steps=5, center_pos_mode="none", seed=123)
    r3 = m.sample_diffusion(*args, num_steps=5, center_pos_mode="none", seed=124)
    assert torch.equal(r1["v"], r2["v"]) and maxabs(r1["pos"], r2["pos"]) < 1e-6
    assert maxabs(r1["pos"], r3["pos"]) > 1e-3
    # x_{t-1} - (c0 x0_hat + ct x_t) = sigma_t * eps  ->  recover eps of the first step
    sd, dm, _, _ = oracle_model()
    tt = 999
    mean = sd["posterior_mean_c0_coef"][tt] * r1["pos_cond_traj"][0].cpu() + sd["posterior_mean_ct_coef"][tt] * T(bb["init_pos"])
    e = (r1["pos_traj"][0] - mean) / torch.exp(0.5 * sd["posterior_logvar"][tt])
    assert abs(float(e.mean())) < 0.15 and abs(float(e.std()) - 1.0) < 0.15


def test_error_reporting():
    from shapemol_amd import _lib
    m = hip_model()
    with pytest.raises(_lib.ShapeMolLibraryError):
        m.set_option("no_such_option", 1)
    with pytest.raises(NotImplementedError):
        m.sample_diffusion(torch.zeros(3, 3, device=DEV), torch.zeros(3, dtype=torch.long, device=DEV),
                           torch.zeros(3, dtype=torch.long, device=DEV), torch.zeros(1, 96, device=DEV),
                           num_steps=2, center_pos_mode="center")


def test_sampling_driver_layout_and_equivalence():
    """shapemol_amd.sampling.sample_diffusion_ligand (the reference driver's batching / init / unbatching contract):
    result layout, dtypes, and that a batch equals the direct chain on the same initial state and noise seed."""
    from shapemol_amd.sampling import pack_result, sample_diffusion_ligand
    from shapemol_amd import log_sample_categorical
    m = hip_model()
    shape_emb = synth.synthetic_batch(1, seed=5)["shape"][0]
    counts = iter([[9, 12], [7, 15], [11]])
    steps = 12
    torch.manual_seed(1234)
    out = sample_diffusion_ligand(m, shape_emb, num_samples=5, batch_size=2, device=DEV, num_steps=steps,
                                  sample_num_atoms="size", sample_func=lambda n: next(counts), seed=99)
    pos, v, pos_traj, v_traj, v0_traj, vt_traj, times, pos_cond, v_cond = out
    want = [9, 12, 7, 15, 11]
    assert len(times) == 3 and [len(x) for x in (pos, v, pos_traj, v_traj, v0_traj, vt_traj, pos_cond, v_cond)] == [5] * 8
    for k, n in enumerate(want):
        assert pos[k].shape == (n, 3) and pos[k].dtype == np.float64
        assert v[k].shape == (n,) and v[k].dtype == np.int64
        assert pos_traj[k].shape == (steps, n, 3) and pos_traj[k].dtype == np.float64
        assert v_traj[k].shape == (steps, n) and v0_traj[k].shape == (steps, n, 15) and vt_traj[k].shape == (steps, n, 15)
        assert np.array_equal(pos_traj[k][-1], pos[k]) and np.array_equal(v_traj[k][-1], v[k])
    res = pack_result({"id": 0}, out)
    assert set(res) == {"data", "pred_ligand_pos", "pred_ligand_v", "pred_ligand_pos_traj", "pred_ligand_v_traj", "time",
                        "pred_ligand_pos_cond_traj", "pred_ligand_v_cond_traj"}
    # first batch again, by hand, in the driver's order of random draws
    torch.manual_seed(1234)
    batch = torch.repeat_interleave(torch.arange(2), torch.tensor([9, 12])).to(DEV)
    p0 = torch.randn(21, 3).to(DEV)
    v0 = log_sample_categorical(torch.zeros(21, 15, device=DEV))
    sh = torch.as_tensor(shape_emb).reshape(1, -1).repeat(2, 1).to(DEV)
    r = m.sample_diffusion(p0, v0, batch, sh, num_steps=steps, center_pos_mode="none", seed=99)
    ref_pos = r["pos"].cpu().numpy().astype(np.float64)
    assert np.array_equal(ref_pos[:9], pos[0]) and np.array_equal(ref_pos[9:], pos[1])
    assert np.array_equal(r["v"].cpu().numpy()[:9], v[0])
