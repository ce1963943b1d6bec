"""Drop-in import path of the reference: ``from src.vgan import VGAN, VGAN_no_kl`` (test.ipynb:16)
and ``from src.models.<Module> import ...`` resolve to the MI355X-native implementation in
``v-gan_amd/`` (loaded under the module name ``vgan_amd``)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import vgan_amd  # noqa: E402,F401
