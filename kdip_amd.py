"""Import alias: `import kdip_amd` loads the package that lives in the (non-importable)
directory `k-diffusion-inverse-problems_amd/`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "k-diffusion-inverse-problems_amd")
_spec = importlib.util.spec_from_file_location("kdip_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kdip_amd"] = _mod
_spec.loader.exec_module(_mod)
