"""shapemol_amd -- MI355X-native (gfx950) implementation of ShapeMol's denoising hot path.

Public surface (mirrors the reference's models/molopt_score_model.py):
    ScorePosNet3D, log_sample_categorical, pointcloud_shape_guidance
and, outside it, the frozen shape encoder that produces the conditioning (models/shape_pointcloud_modelAE.py):
    VN_DGCNN_Encoder
plus helpers: synthetic weights/inputs (synth), schedules (diffusion), the C-ABI binding (_lib).
"""
from .molopt_score_model import ScorePosNet3D, log_sample_categorical, pointcloud_shape_guidance  # noqa: F401
from .packing import pack_state_dict  # noqa: F401
from .shape_encoder import VN_DGCNN_Encoder  # noqa: F401

__version__ = "0.3.0"
