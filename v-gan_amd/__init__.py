"""vgan_amd -- MI355X-native V-GAN training hot path.

Host side (Python, mirroring the reference's Python surface) over ``libvgan_hip.so`` (hand-written
HIP for gfx950 behind the C ABI of ``include/vgan_hip.h``).  There is no CPU compute path in this
package: every operator raises if the HIP library or a GPU is missing.
"""
from . import lib  # noqa: F401
from .modules import (Generator_big, upper_softmax, Encoder, Decoder, Detector, RBF,  # noqa: F401
                      MMDLossConstrained)
from .vgan import VGAN, VGAN_no_kl  # noqa: F401

__all__ = ["VGAN", "VGAN_no_kl", "Generator_big", "upper_softmax", "Encoder", "Decoder", "Detector", "RBF",
           "MMDLossConstrained", "lib"]
