"""Import alias: the package directory is `council-gan_amd/` (not a Python identifier), so
`import council_gan_amd` loads it from there and replaces this stub in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "council-gan_amd")
_spec = importlib.util.spec_from_file_location("council_gan_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["council_gan_amd"] = _mod
_spec.loader.exec_module(_mod)
