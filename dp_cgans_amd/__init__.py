"""MI355X-native DP-CGAN training step behind the reference's API.

``DP_CGAN`` / ``DPCGANSynthesizer`` / ``DataSampler`` / ``DataTransformer`` mirror the reference
classes; every computation of the step runs in csrc/libdpcgan_hip.so (include/dpcgan_hip.h)."""
__version__ = '0.1.0'

from .data_transformer import DataTransformer  # noqa: F401
from .model import DP_CGAN  # noqa: F401
from .synthesizer import DPCGANSynthesizer  # noqa: F401
