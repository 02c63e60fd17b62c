"""joeys2t_amd — MI355X-native hot path of JoeyS2T (speech-to-text), behind the reference's Python API.

Everything numeric runs in hand-written HIP kernels for gfx950 (libjoeys2t_hip.so, C ABI in
include/joeys2t_hip.h).  PyTorch provides device memory, streams, autograd routing and torch.distributed.
"""
__version__ = "0.1.0"
