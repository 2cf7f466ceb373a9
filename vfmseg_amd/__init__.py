"""vfmseg_amd - MI355X-native hot path of tpy001/VFMSeg.

Importing the package registers the model classes under the reference's registry names
(rein/__init__.py does the same by import side effect).
"""
from .precision import compute_dtype, set_compute_dtype  # noqa: F401
from .registry import BACKBONES, MODELS, OPTIM_WRAPPER_CONSTRUCTORS  # noqa: F401
from . import backbones, eva, sam, clip, heads, segmentors, optim, metrics  # noqa: F401,E402

__version__ = "0.1.0"
