"""Loader for the hyphenated package directory `voxel-cone-tracing_amd/`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = "voxel_cone_tracing_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(_ROOT, "voxel-cone-tracing_amd")
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        sys.modules.pop(_NAME, None)
        raise
    return mod
