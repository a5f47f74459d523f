"""cufhe_amd -- MI355X-native TFHE gate bootstrapping behind the cuFHE gate API.

The product is libcufhe_amd.so (hand-written HIP for gfx950, C ABI in include/cufhe_amd.h);
this package is the thin Python binding used by tests and bench.py.  Importing it without
the built library raises ImportError: there is no CPU fallback.
"""
from ._lib import lib, LIB_PATH, CufheAmdError, check  # noqa: F401
from .api import *  # noqa: F401,F403
from . import api  # noqa: F401
