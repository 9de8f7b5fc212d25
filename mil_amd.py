"""Import alias: the package directory is ``llm-guided-multimodal-mil_amd/`` (not a valid
Python identifier), so ``import mil_amd`` loads it from that path."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm-guided-multimodal-mil_amd")
_spec = importlib.util.spec_from_file_location(
    "mil_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mil_amd"] = _mod
_spec.loader.exec_module(_mod)
