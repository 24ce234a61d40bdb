"""Import shim: exposes the package directory ``l4dc-mpc-ocd_amd/`` as ``l4dc_mpc_ocd_amd``.

The directory name is fixed by the project layout and contains a hyphen, so it
cannot be imported by name; this module replaces itself in ``sys.modules`` with
the package loaded from that directory.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "l4dc-mpc-ocd_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
