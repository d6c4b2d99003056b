"""
iscc_search_amd -- MI355X-native (gfx950) exact Hamming / NPHD similarity search for ISCC codes.

A ``hip:///`` backend for iscc/iscc-search: ``HipIndexManager`` implements the reference's
``IsccIndexProtocol`` (``iscc_search/protocols/index.py:19-174``) over hand-written HIP kernels
reached through a ctypes C-ABI (``include/isccsearch.h``).  See DESIGN.md and INTEGRATION.md.
"""

__version__ = "0.1.0"

__all__ = ["__version__"]
