"""xfr_amd -- MI355X-native excitation-backprop saliency engine behind the stresearch/xfr Whitebox API.

Importing the package does not load the HIP library; anything that computes does (xfr_amd._lib) and fails
loudly if `xfr_amd/csrc/libxfr_amd.so` is missing or no HIP device is visible.  There is no CPU fallback.
"""
__version__ = '0.1.0'
