"""Canonical import name for the package directory `dnn-based-speech-enhancement-in-the-frequency-domain_amd/`
(a hyphenated directory name cannot be written in an `import` statement).  The directory is loaded ONCE, under the
module name `sefd_amd`, so `sefd_amd.optim.Adam` and the package's own relative imports are the same objects."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_ROOT, "dnn-based-speech-enhancement-in-the-frequency-domain_amd")
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_spec = importlib.util.spec_from_file_location("sefd_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["sefd_amd"] = _pkg
_spec.loader.exec_module(_pkg)
