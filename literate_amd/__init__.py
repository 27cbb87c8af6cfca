"""literate_amd - MI355X-native (gfx950) implementation of LiteRate's RJMCMC birth-death
likelihood path.  Python holds device buffers as torch tensors and calls hand-written HIP
kernels through the C ABI declared in include/literate_hip.h (ctypes, no torch types cross it).

There is no CPU fallback: every compute entry point raises if libliterate_hip.so or a GPU is
missing."""

__version__ = "0.1.0"
