"""State dict -> the flat float32 weight array consumed by shapemol_create().

Order (mirrored by parse_weights() in csrc/shapemol_hip.hip); names are reference state-dict keys:
  7 schedule tables of length T: posterior_mean_c0_coef, posterior_mean_ct_coef, posterior_logvar,
      log_alphas_v, log_one_minus_alphas_v, log_alphas_cumprod_v, log_one_minus_alphas_cumprod_v
  time_emb.1.{weight,bias}, time_emb.3.{weight,bias}, ligand_atom_emb.{weight,bias}
  refine_net.edge_pred_layer          (MLP = net.0.{weight,bias}, net.1.{weight,bias}, net.3.{weight,bias})
  for each layer l of refine_net.base_block:
      x2h_layers.0.{hk_func,hv_func,hq_func,node_output} (MLPs), h2x_layers.0.{xk_func,xv_func,xq_func} (MLPs),
      h2x_layers.0.shape_linear.{map_to_feat.weight, batchnorm.bn.weight, batchnorm.bn.bias, map_to_dir.weight}
  refine_net.invariant_shape_layer.hidden_layer (MLP)
  v_inference.0.{weight,bias}, v_inference.2.{weight,bias}
Entries the path never reads (loss weights, running statistics, the dead equivariant_shape_layer,
RBF offsets -- fixed constants) are not packed.
"""
import numpy as np

TABLES = ("posterior_mean_c0_coef", "posterior_mean_ct_coef", "posterior_logvar", "log_alphas_v",
          "log_one_minus_alphas_v", "log_alphas_cumprod_v", "log_one_minus_alphas_cumprod_v")


def _mlp(prefix):
    return [f"{prefix}.net.{i}.{p}" for i in (0, 1, 3) for p in ("weight", "bias")]


def pack_order(num_layers):
    keys = list(TABLES)
    keys += ["time_emb.1.weight", "time_emb.1.bias", "time_emb.3.weight", "time_emb.3.bias",
             "ligand_atom_emb.weight", "ligand_atom_emb.bias"]
    keys += _mlp("refine_net.edge_pred_layer")
    for l in range(num_layers):
        b = f"refine_net.base_block.{l}."
        for m in ("hk_func", "hv_func", "hq_func", "node_output"):
            keys += _mlp(b + "x2h_layers.0." + m)
        for m in ("xk_func", "xv_func", "xq_func"):
            keys += _mlp(b + "h2x_layers.0." + m)
        s = b + "h2x_layers.0.shape_linear."
        keys += [s + "map_to_feat.weight", s + "batchnorm.bn.weight", s + "batchnorm.bn.bias", s + "map_to_dir.weight"]
    keys += _mlp("refine_net.invariant_shape_layer.hidden_layer")
    keys += ["v_inference.0.weight", "v_inference.0.bias", "v_inference.2.weight", "v_inference.2.bias"]
    return keys


def pack_state_dict(sd, num_layers):
    """sd: {key: ndarray or torch tensor}.  Returns a contiguous float32 ndarray."""
    parts = []
    for k in pack_order(num_layers):
        v = sd[k]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        parts.append(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))
    return np.concatenate(parts)
