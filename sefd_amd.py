"""Importable alias for the hyphenated package directory `dnn-based-speech-enhancement-in-the-frequency-domain_amd`."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("dnn-based-speech-enhancement-in-the-frequency-domain_amd")
sys.modules[__name__] = _pkg
