"""ladiff_amd - MI355X (gfx950) implementation of the LADiff latent-diffusion sampling hot path.

Host side (this package): parameter containers with the reference's state-dict schema, scheduler tables,
the `LADIFF`-compatible loop owner.  Arithmetic: libladiff_hip.so (ladiff_amd/csrc, C ABI in include/ladiff_hip.h).
"""
from .feats2joints import Feats2Joints                   # noqa: F401
from .modules import LADiffDenoiser, LADiffVae          # noqa: F401
from .pipeline import LADIFF, instantiate_from_config   # noqa: F401
from .schedulers import DDIMScheduler, DDPMScheduler    # noqa: F401
from .text_encoder import MldTextEncoder                # noqa: F401
from .evaluators import (MovementConvEncoder, MotionEncoderBiGRUCo, TextEncoderBiGRUCo,   # noqa: F401
                         TM2TMetrics)

__all__ = ["LADiffDenoiser", "LADiffVae", "LADIFF", "DDIMScheduler", "DDPMScheduler", "instantiate_from_config", "Feats2Joints", "MldTextEncoder", "MovementConvEncoder", "MotionEncoderBiGRUCo", "TextEncoderBiGRUCo",
           "TM2TMetrics"]
