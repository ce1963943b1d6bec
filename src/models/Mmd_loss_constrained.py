"""Reference module path ``src/models/Mmd_loss_constrained.py``."""
from vgan_amd.modules import RBF, MMDLossConstrained  # noqa: F401
