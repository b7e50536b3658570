"""Import alias for the package directory ``attention-interpolation-diffusion_amd/`` (a hyphenated
name cannot be imported directly).  ``import aid_amd`` loads that directory as the package
``aid_amd`` (sub-modules: ``aid_amd.ops``, ``aid_amd.processors`` …)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attention-interpolation-diffusion_amd")
_spec = importlib.util.spec_from_file_location("aid_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["aid_amd"] = _mod
_spec.loader.exec_module(_mod)
