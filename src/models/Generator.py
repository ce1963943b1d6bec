"""Reference module path ``src/models/Generator.py``."""
from vgan_amd.modules import Generator_big, upper_softmax  # noqa: F401
