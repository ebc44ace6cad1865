"""Import alias: loads ``ma-cjd-cooperative-jamming-decision-making-via-marl_amd/`` as package ``macjd_amd``.

The contract-mandated package directory name contains hyphens and cannot be imported by name; this
module replaces itself in ``sys.modules`` with that directory loaded as a regular package."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "ma-cjd-cooperative-jamming-decision-making-via-marl_amd")
_spec = importlib.util.spec_from_file_location(
    "macjd_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["macjd_amd"] = _mod
_spec.loader.exec_module(_mod)
