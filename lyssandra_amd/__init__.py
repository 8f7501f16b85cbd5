"""lyssandra_amd -- MI355X (gfx950) native engine for the Batch-OMP + dictionary-update hot path of Lyssandra.

Drop-in class API (same names / arguments as the reference's ``lyssa`` package):

    from lyssandra_amd.sparse_coding import sparse_encoder
    from lyssandra_amd.dict_learning import ksvd_coder, online_dictionary_coder

The compute lives in ``liblyssa_hip.so`` (hand-written HIP, C-ABI in include/lyssa_hip.h) and is reached through
ctypes; importing this package does not load the library, the first call does -- and fails loudly if it is missing.
"""
__version__ = "0.1.0"

from .sparse_coding import sparse_encoder  # noqa: F401
