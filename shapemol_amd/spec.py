"""Model dimensions and the reference state-dict layout, derived from the `model` section of
the training YAML (schema: /root/reference/config/training/*.yml:21-74, consumed at
/root/reference/models/molopt_score_model.py:13-39,176-283).

`state_dict_spec()` lists every entry of the reference checkpoint's ``ckpt['model']`` --
446 entries for the shipped config -- in registration order, so that
``ScorePosNet3D.load_state_dict(strict=True)`` accepts reference checkpoints unchanged
(boundary: /root/reference/scripts/sample_diffusion.py:211-215).
"""
from collections import OrderedDict

from .diffusion import SCHEDULE_KEYS

RBF_CENTRES = (0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10)


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class ModelDims:
    """Checked view of the model config.  Options the shipped configs never enable and the
    reference itself cannot run raise NotImplementedError here."""

    def __init__(self, cfg, ligand_atom_feature_dim=15):
        g = lambda k, d=None: _get(cfg, k, d)  # noqa: E731
        self.H = int(g("hidden_dim"))
        self.heads = int(g("n_heads"))
        self.L = int(g("num_layers"))
        self.k = int(g("knn"))
        self.G = int(g("num_r_gaussian"))
        self.S = int(g("shape_dim"))
        self.S_latent = int(g("shape_latent_dim"))
        self.temb = int(g("time_emb_dim"))
        self.C = int(ligand_atom_feature_dim)
        self.T = int(g("num_diffusion_timesteps"))
        self.v_mode = g("v_mode")
        self.center_pos_mode = g("center_pos_mode")
        self.loss_weight_type = g("loss_weight_type")
        unsupported = []
        if int(g("num_blocks")) != 1: unsupported.append("num_blocks != 1")
        if int(g("edge_feat_dim")) != 0: unsupported.append("edge_feat_dim != 0")
        if g("cutoff_mode") != "knn": unsupported.append(f"cutoff_mode={g('cutoff_mode')}")
        if g("ew_net_type") != "global": unsupported.append(f"ew_net_type={g('ew_net_type')}")
        if g("v_mode") != "uniform": unsupported.append(f"v_mode={g('v_mode')}")
        if g("model_type") != "uni_o2": unsupported.append(f"model_type={g('model_type')}")
        if g("shape_type") != "pointAE_shape": unsupported.append(f"shape_type={g('shape_type')}")
        if g("shape_mode", "attention_residue") != "attention_residue": unsupported.append("shape_mode")
        if "topo" in str(g("topo_emb_type")): unsupported.append(f"topo_emb_type={g('topo_emb_type')}")
        if getattr_default(g, "v_net_type", "mlp") != "mlp": unsupported.append("v_net_type != mlp")
        if int(g("num_x2h", 1)) != 1 or int(g("num_h2x", 1)) != 1: unsupported.append("num_x2h/num_h2x != 1")
        if bool(g("sync_twoup", False)): unsupported.append("sync_twoup")
        if self.temb <= 0 or self.temb % 2: unsupported.append("time_emb_dim must be a positive even number")
        if self.G != len(RBF_CENTRES): unsupported.append("num_r_gaussian != 20 (the reference hard-codes 20 centres)")
        if self.H % self.heads: unsupported.append("hidden_dim % n_heads")
        if unsupported:
            raise NotImplementedError("model config outside the accelerated hot path: " + ", ".join(unsupported))
        self.dh = self.H // self.heads
        self.kv_in = self.G + 2 * self.H + self.S_latent
        self.vn_in = 1 + self.heads + self.S


def getattr_default(g, key, default):
    v = g(key, default)
    return default if v is None else v


def _mlp(out, prefix, d_in, d_hidden, d_out):
    out[prefix + ".net.0.weight"] = ((d_hidden, d_in), "weight", d_in)
    out[prefix + ".net.0.bias"] = ((d_hidden,), "bias", d_in)
    out[prefix + ".net.1.weight"] = ((d_hidden,), "norm_weight", 0)
    out[prefix + ".net.1.bias"] = ((d_hidden,), "norm_bias", 0)
    out[prefix + ".net.3.weight"] = ((d_out, d_hidden), "weight", d_hidden)
    out[prefix + ".net.3.bias"] = ((d_out,), "bias", d_hidden)


def _vn(out, prefix, c_in, c_out):
    out[prefix + ".map_to_feat.weight"] = ((c_out, c_in), "weight", c_in)
    out[prefix + ".batchnorm.bn.weight"] = ((c_out,), "norm_weight", 0)
    out[prefix + ".batchnorm.bn.bias"] = ((c_out,), "norm_bias", 0)
    out[prefix + ".batchnorm.bn.running_mean"] = ((c_out,), "running_mean", 0)
    out[prefix + ".batchnorm.bn.running_var"] = ((c_out,), "running_var", 0)
    out[prefix + ".batchnorm.bn.num_batches_tracked"] = ((), "counter", 0)
    out[prefix + ".map_to_dir.weight"] = ((c_out, c_in), "weight", c_in)


def state_dict_spec(dm):
    """OrderedDict key -> (shape, kind, fan_in); kinds as in synth.fill_state_dict."""
    o = OrderedDict()
    for k in SCHEDULE_KEYS:
        if k == "loss_pos_step_weight" and dm.loss_weight_type != "noise_level":
            continue
        o[k] = ((dm.T,), "const", 0)
    o["time_emb.1.weight"] = ((2 * dm.temb, dm.temb), "weight", dm.temb)
    o["time_emb.1.bias"] = ((2 * dm.temb,), "bias", dm.temb)
    o["time_emb.3.weight"] = ((dm.temb, 2 * dm.temb), "weight", 2 * dm.temb)
    o["time_emb.3.bias"] = ((dm.temb,), "bias", 2 * dm.temb)
    o["ligand_atom_emb.weight"] = ((dm.H, dm.C + dm.temb), "weight", dm.C + dm.temb)
    o["ligand_atom_emb.bias"] = ((dm.H,), "bias", dm.C + dm.temb)
    r = "refine_net."
    o[r + "distance_expansion.offset"] = ((dm.G,), "const", 0)
    _mlp(o, r + "edge_pred_layer", dm.G, dm.H, 1)
    for l in range(dm.L):
        b = f"{r}base_block.{l}."
        o[b + "distance_expansion.offset"] = ((dm.G,), "const", 0)
        _mlp(o, b + "x2h_layers.0.hk_func", dm.kv_in, dm.H, dm.H)
        _mlp(o, b + "x2h_layers.0.hv_func", dm.kv_in, dm.H, dm.H)
        _mlp(o, b + "x2h_layers.0.hq_func", dm.H, dm.H, dm.H)
        _mlp(o, b + "x2h_layers.0.node_output", 2 * dm.H, dm.H, dm.H)
        _mlp(o, b + "h2x_layers.0.xk_func", dm.kv_in, dm.H, dm.H)
        _mlp(o, b + "h2x_layers.0.xv_func", dm.kv_in, dm.H, dm.heads)
        _mlp(o, b + "h2x_layers.0.xq_func", dm.H, dm.H, dm.H)
        _vn(o, b + "h2x_layers.0.shape_linear", dm.vn_in, dm.heads)
    _mlp(o, r + "invariant_shape_layer.hidden_layer", dm.S, dm.S, dm.S_latent)
    # constructed but never called by the reference's forward; kept for strict loading
    _vn(o, r + "equivariant_shape_layer.hidden_layer", dm.S, dm.S_latent // 3)
    o["v_inference.0.weight"] = ((dm.H, dm.H), "weight", dm.H)
    o["v_inference.0.bias"] = ((dm.H,), "bias", dm.H)
    o["v_inference.2.weight"] = ((dm.C, dm.H), "weight", dm.H)
    o["v_inference.2.bias"] = ((dm.C,), "bias", dm.H)
    return o
