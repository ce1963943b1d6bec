"""Reference module path ``src/models/Detector.py``."""
from vgan_amd.modules import Decoder, Detector, Encoder  # noqa: F401
