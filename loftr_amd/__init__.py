"""loftr_amd -- MI355X-native LoFTR matching path (hand-written HIP kernels behind a C-ABI).

Public surface mirrors ``src.loftr`` of zju3dv/LoFTR (src/loftr/__init__.py:1-2):
``LoFTR`` and ``default_cfg``.
"""
from .config import default_cfg, full_default_cfg, get_cfg       # noqa: F401
from .loftr import LoFTR                                           # noqa: F401
