"""Deterministic synthetic weights, inputs and noise (no torch RNG, no libm).

The trained checkpoint ``trained_models/diff_model.pt`` is not shipped with the reference
(/root/reference/.MISSING_LARGE_BLOBS:2), so parity and benchmark runs use weights that any
machine can regenerate bit-for-bit: a splitmix64 hash of (seed, tensor-tag, element index)
mapped to a uniform value.  Only integer arithmetic and exact float conversions are used, so
the result does not depend on the numpy/torch/libm build.

Noise for long parity chains is produced the same way: uniforms are 24-bit hash fractions,
"Gaussian" draws are the Irwin-Hall sum of 12 such uniforms minus 6 (mean 0, variance 1,
exactly representable sums) -- both implementations under test are fed the *same* numbers,
so the shape of the distribution does not enter the comparison.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u24(n, tag, seed=0):
    """n integers in [0, 2**24), a pure function of (seed, tag, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(tag))
        idx = np.arange(n, dtype=np.uint64)
        return (_splitmix64(idx ^ base) >> np.uint64(40)).astype(np.int64)


def hash_uniform(shape, tag, seed=0):
    """float32 uniforms in [0, 1) on a 2**-24 grid (exact in float32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_u24(n, tag, seed).astype(np.float32) * np.float32(2.0 ** -24)
    return u.reshape(shape)


def hash_normal(shape, tag, seed=0):
    """float32 zero-mean unit-variance draws: Irwin-Hall(12) - 6 (exact integer sums)."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.int64)
    for r in range(12):
        acc += hash_u24(n, tag * 16 + r + 1, seed ^ 0x5A5A5A)
    z = (acc.astype(np.float64) * 2.0 ** -24 - 6.0).astype(np.float32)
    return z.reshape(shape)


def key_tag(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def fill_state_dict(spec, seed=0):
    """spec: ordered {key: (shape, kind)} with kind in
    {'weight','bias','norm_weight','norm_bias','const','running_mean','running_var','counter'}.
    'const' entries (schedules, RBF offsets) are not generated here.
    Returns {key: ndarray}."""
    out = {}
    for key, (shape, kind, fan_in) in spec.items():
        shape = tuple(shape)
        if kind == "const":
            continue
        if kind == "running_mean":
            out[key] = np.zeros(shape, np.float32)
        elif kind == "running_var":
            out[key] = np.ones(shape, np.float32)
        elif kind == "counter":
            out[key] = np.zeros(shape, np.int64)
        else:
            u = hash_uniform(shape, key_tag(key), seed)
            if kind in ("weight", "bias"):
                out[key] = ((2.0 * u - 1.0) * np.float32(1.0 / np.sqrt(fan_in))).astype(np.float32)
            elif kind == "norm_weight":
                out[key] = (1.0 + 0.2 * (u - 0.5)).astype(np.float32)
            elif kind == "norm_bias":
                out[key] = (0.2 * (u - 0.5)).astype(np.float32)
            else:
                raise ValueError(kind)
    return out


# --------------------------------------------------------------------------------------
# synthetic sampling batches
# --------------------------------------------------------------------------------------
# Empirical MOSES atom-count prior, pooled over all voxel-size keys of
# data/MOSES2_training_val_shape_atomnum_dict.pkl (150 000 molecules, 9..27 atoms, mean 21.38).
# Stored as counts (moses_prior.py) so the repo does not need the pickle at run time.
def moses_atom_prior():
    """(atom_numbers, probabilities) of the pooled MOSES prior."""
    from .moses_prior import ATOM_NUMS, ATOM_FREQ
    p = np.asarray(ATOM_FREQ, np.float64)
    return np.asarray(ATOM_NUMS, np.int64), p / p.sum()


def synthetic_batch(num_mols, seed=2021, max_atoms=None, atoms_range=None, shape_points=32):
    """One synthetic sampling batch.

    Returns dict(counts (B,) i64, batch (N,) i64, init_pos (N,3) f32, init_v (N,) i64,
    shape (B,32,3) f32).  Atom counts ~ MOSES prior via RandomState(seed) unless
    `atoms_range=(lo, hi)` asks for the uniform large-molecule stress draw.
    """
    rs = np.random.RandomState(seed)
    if atoms_range is not None:
        counts = rs.randint(atoms_range[0], atoms_range[1] + 1, size=num_mols).astype(np.int64)
    else:
        nums, p = moses_atom_prior()
        counts = rs.choice(nums, size=num_mols, p=p).astype(np.int64)
    if max_atoms is not None:
        counts = np.minimum(counts, max_atoms)
    n = int(counts.sum())
    batch = np.repeat(np.arange(num_mols, dtype=np.int64), counts)
    init_pos = hash_normal((n, 3), tag=101, seed=seed)
    init_v = (hash_u24(n, tag=102, seed=seed) % 15).astype(np.int64)
    shape = hash_normal((num_mols, shape_points, 3), tag=103, seed=seed)
    return dict(counts=counts, batch=batch, init_pos=init_pos, init_v=init_v, shape=shape)


def step_noise(n_atoms, num_classes, step, seed=2021):
    """Host-fed noise of one reverse step, in the reference's draw order
    (molopt_score_model.py:662 randn_like (N,3) then :99 rand_like (N,C))."""
    eps = hash_normal((n_atoms, 3), tag=1000 + 2 * step, seed=seed)
    u = hash_uniform((n_atoms, num_classes), tag=1001 + 2 * step, seed=seed)
    return eps, u


def synthetic_state_dict(model_cfg, seed=7, num_classes=15):
    """Full reference-layout state dict {key: ndarray} for a model config: hash-filled
    weights, the schedule tables and the fixed RBF centres."""
    from .spec import ModelDims, state_dict_spec, RBF_CENTRES
    from .diffusion import build_schedule_tables
    dm = ModelDims(model_cfg, num_classes)
    spec = state_dict_spec(dm)
    out = fill_state_dict(spec, seed=seed)
    out.update(build_schedule_tables(model_cfg))
    for k in spec:
        if k.endswith("distance_expansion.offset"):
            out[k] = np.asarray(RBF_CENTRES, np.float32)
    return {k: out[k] for k in spec}


def running_stats(num_layers, heads, seed=23):
    """Non-trivial batch-norm running statistics {key: ndarray} for the evaluation-mode tests (synthetic_state_dict leaves
    them at 0 / 1, a fresh module's values): means around the typical vector norms, variances in [0.05, 0.55]."""
    out = {}
    for l in range(num_layers):
        p = f"refine_net.base_block.{l}.h2x_layers.0.shape_linear.batchnorm.bn."
        out[p + "running_mean"] = (0.4 + 1.2 * hash_uniform((heads,), key_tag(p + "running_mean"), seed)).astype(np.float32)
        out[p + "running_var"] = (0.05 + 0.5 * hash_uniform((heads,), key_tag(p + "running_var"), seed)).astype(np.float32)
    return out


def shape_encoder_state_dict(hidden=128, latent=32, layers=4, seed=17):
    """Hash-filled weights of the shape encoder under the keys of shapemol_amd.shape_encoder.VN_DGCNN_Encoder
    (conv_pos.*, blocks.{i}.*, conv_c.*; batch-norm running statistics are not part of the forward)."""
    spec = {}

    def vn(prefix, cin, cout, shared=False):
        spec[prefix + ".map_to_feat.weight"] = ((cout, cin), "weight", cin)
        spec[prefix + ".batchnorm.bn.weight"] = ((cout,), "norm_weight", 0)
        spec[prefix + ".batchnorm.bn.bias"] = ((cout,), "norm_bias", 0)
        spec[prefix + ".map_to_dir.weight"] = ((1 if shared else cout, cin), "weight", cin)
    vn("conv_pos", 2, hidden)
    for i in range(layers):
        vn(f"blocks.{i}", 2 * hidden, hidden)
    vn("conv_c", layers * hidden, latent, shared=True)
    return fill_state_dict(spec, seed=seed)
