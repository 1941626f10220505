"""gaot_amd -- MI355X-native GAOT forward/backward hot path (HIP kernels behind the reference's src/model API)."""
__version__ = "0.1.0"
