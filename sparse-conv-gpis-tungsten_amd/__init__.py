"""MI355X-native sparse-convolution GPIS hot path (gfx950).

The product is the C-ABI shared library built from ``csrc/`` (``libgpis_hip.so``, declared in
``include/gpis.h``) plus the C++ ``Medium``-shaped host adapter in ``host/``.  This Python
package is only glue for tests and ``bench.py``: numpy mirrors of the POD structs and a ctypes
loader.  It never imports anything from ``oracle/`` and has no CPU fallback: if the HIP
library is missing or no GPU is present, the calls fail loudly.
"""
from .bindings import (  # noqa: F401
    GpisLib, Medium, load_library, library_path, libm_eval, sort_pairs_u32,
    PARAMS, MEAN, RAMP, VARIANCE_GRID, variance_grid_desc, FS_STATE, FS_MAX_POINTS, FS_MAX_CTX, RAY_IN, SEG_OUT, COND_COEFF, QUERY, NEE_QUERY, DERIVED, SCENE_S, SURFACE_S, default_surface_s, default_scene_s,
    default_params, params_for_config, as_params, CTX, SCHEME, MEAN_TYPE,
)
from . import dist  # noqa: F401
