"""dp_cgans_amd: MI355X-native hot path for DP-CGAN training (gfx950 HIP kernels behind a C-ABI).

Only the training-step path of sunchang0124/dp_cgans is rebuilt here; the
host code mirrors the reference's DPCGANSynthesizer / DataSampler / DataTransformer interfaces.
"""
__version__ = '0.1.0'
