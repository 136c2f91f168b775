"""MI355X-native SuperPoint stereo-VO front end: Python-side tooling and ctypes binding.

The product is the C-ABI library built from `../csrc` (see include/spvo.h) and the
C++ host mirror of the reference's FeatureFrontEnd in `../host`.  This package only
holds (a) the weight/plan packer that replaces the reference's TensorRT engine
generator and (b) a thin ctypes loader used by tests and bench.py.
"""
