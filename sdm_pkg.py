"""Import shim: loads the package directory ``orb-slam-free-space-carving_amd/`` (not a valid
Python identifier) as module ``orb_slam_free_space_carving_amd``."""
import importlib.util
import os
import sys

_NAME = "orb_slam_free_space_carving_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "orb-slam-free-space-carving_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
