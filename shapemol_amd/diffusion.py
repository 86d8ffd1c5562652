"""Noise schedules of the DDPM chain (host side, float64 numpy, built once).

Behaviour follows the reference's schedule construction:
  * beta schedules          -> /root/reference/models/diffusion.py:4-35  (get_beta_schedule)
  * cosine alpha-bar        -> /root/reference/models/diffusion.py:38-47 (cosine_beta_schedule)
  * derived position tables -> /root/reference/models/molopt_score_model.py:188-220
  * derived atom-type tables-> /root/reference/models/molopt_score_model.py:222-234
All tables are computed in float64 and rounded to float32 once, exactly where the
reference calls ``.float()`` (molopt_score_model.py:47-50).
"""
import numpy as np

__all__ = ["get_beta_schedule", "cosine_beta_schedule", "build_schedule_tables", "SCHEDULE_KEYS"]

# order = registration order of the 16 non-trainable schedule parameters in the
# reference module (molopt_score_model.py:198-234); `loss_pos_step_weight` exists
# only for loss_weight_type == 'noise_level'.
SCHEDULE_KEYS = (
    "loss_pos_step_weight", "betas", "alphas_cumprod", "alphas_cumprod_prev",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_mean_c0_coef", "posterior_mean_ct_coef",
    "posterior_var", "posterior_logvar", "log_alphas_v", "log_one_minus_alphas_v",
    "log_alphas_cumprod_v", "log_one_minus_alphas_cumprod_v",
)


def cosine_beta_schedule(timesteps, s=0.008):
    grid = np.linspace(0, timesteps + 1, timesteps + 1)
    abar = np.cos((grid / (timesteps + 1) + s) / (1 + s) * np.pi * 0.5) ** 2
    abar = abar / abar[0]
    return np.clip(1 - abar[1:] / abar[:-1], 0, 0.999)


def get_beta_schedule(beta_schedule, num_diffusion_timesteps, **kwargs):
    kw = {k: float(v) for k, v in kwargs.items()}
    T = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(kw["beta_start"] ** 0.5, kw["beta_end"] ** 0.5, T, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(kw["beta_start"], kw["beta_end"], T, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        s = kw.get("s", 3)
        z = np.linspace(-s, s, T)
        betas = 1 / (np.exp(-z) + 1) * (kw["beta_end"] - kw["beta_start"]) + kw["beta_start"]
    elif beta_schedule == "cosine":
        betas = cosine_beta_schedule(T, s=kw.get("s", 0.008))
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (T,)
    return betas


def _log1m_exp(a):
    return np.log(1 - np.exp(a) + 1e-40)


def build_schedule_tables(model_cfg):
    """Return {name: float32 ndarray (T,)} for every schedule vector of the model config."""
    T = int(model_cfg["num_diffusion_timesteps"])
    betas = get_beta_schedule(num_diffusion_timesteps=T, **dict(model_cfg["schedule_pos"]))
    alphas = 1.0 - betas
    abar = np.cumprod(alphas, axis=0)
    abar_prev = np.append(1.0, abar[:-1])
    out = {}
    if model_cfg.get("loss_weight_type") == "noise_level":
        snr = abar / (1 - abar)
        out["loss_pos_step_weight"] = np.clip(
            model_cfg["loss_pos_min_weight"] + snr, None, model_cfg["loss_pos_max_weight"])
    out["betas"] = betas
    out["alphas_cumprod"] = abar
    out["alphas_cumprod_prev"] = abar_prev
    out["sqrt_alphas_cumprod"] = np.sqrt(abar)
    out["sqrt_one_minus_alphas_cumprod"] = np.sqrt(1.0 - abar)
    out["sqrt_recip_alphas_cumprod"] = np.sqrt(1.0 / abar)
    out["sqrt_recipm1_alphas_cumprod"] = np.sqrt(1.0 / abar - 1)
    post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
    out["posterior_mean_c0_coef"] = betas * np.sqrt(abar_prev) / (1.0 - abar)
    out["posterior_mean_ct_coef"] = (1.0 - abar_prev) * np.sqrt(alphas) / (1.0 - abar)
    out["posterior_var"] = post_var
    # The reference takes the log of the *float32-rounded* variance (it indexes the
    # already-converted parameter, molopt_score_model.py:218-220), with entry 0 := entry 1.
    pv32 = post_var.astype(np.float32)
    out["posterior_logvar"] = np.log(np.append(pv32[1], pv32[1:]))
    betas_v = get_beta_schedule(num_diffusion_timesteps=T, **dict(model_cfg["schedule_v"]))
    log_a = np.log(1.0 - betas_v)
    log_abar = np.cumsum(log_a)
    out["log_alphas_v"] = log_a
    out["log_one_minus_alphas_v"] = _log1m_exp(log_a)
    out["log_alphas_cumprod_v"] = log_abar
    out["log_one_minus_alphas_cumprod_v"] = _log1m_exp(log_abar)
    return {k: np.asarray(v).astype(np.float32) for k, v in out.items()}
