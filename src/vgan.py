"""Reference module path ``src/vgan.py``: the two user classes, MI355X-native."""
from vgan_amd.vgan import VGAN, VGAN_no_kl  # noqa: F401

__all__ = ["VGAN", "VGAN_no_kl"]
