"""MI355X-native semi-dense mapping engine (ProbabilityMapping hot path of
atlas-jj/ORB-SLAM-free-space-carving).  The product is the HIP library behind include/sdm_c.h;
this package holds its sources (csrc/, host/), the build driver, a ctypes mirror of the C ABI used
by tests/bench, and the synthetic-scene generator.  Import it through the repo-root shim
``sdm_pkg.load()`` (the directory name is not a valid Python identifier)."""
from . import build as build_mod  # noqa: F401
from .binding import Engine, SdmError, lib_path, load_library  # noqa: F401
from . import synth  # noqa: F401
from . import shard  # noqa: F401
