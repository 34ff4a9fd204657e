"""glare_amd -- MI355X-native hot path of GLARE (low-light enhancement): HIP kernels behind a
C ABI (include/glare_hip.h) plus the Python mirror of the reference's operator surface."""
__version__ = "0.1.0"
