"""scalce_amd -- MI355X-native SCALCE hot path (LCE tokenize / bucket-reorder / entropy-code).

The product is the C-ABI shared library ``scalce_amd/lib/libscalce_hip.so`` (include/scalce_hip.h);
this package is the thin Python host over it used by tests and bench.py.  There is no CPU path:
importing :mod:`scalce_amd.host` without the built library raises.
"""
from .host import Batch, Context, ScalceError, library_path, qmap_init  # noqa: F401
