"""Import shim: the package directory is ``kinetic-gan_amd/`` (hyphen, as the layout contract
names it), which Python cannot import by name.  ``import kinetic_gan_amd`` loads that directory
as a regular package under this importable name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "kinetic-gan_amd")
_spec = _ilu.spec_from_file_location(
    "kinetic_gan_amd", _os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["kinetic_gan_amd"] = _mod
_spec.loader.exec_module(_mod)
