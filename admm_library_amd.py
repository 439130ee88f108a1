"""Import shim: the package directory is named ``admm-library_amd/`` (not a
valid Python identifier), so ``import admm_library_amd`` resolves to this file,
which turns itself into that package by pointing ``__path__`` at the directory
and executing its ``__init__.py``."""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "admm-library_amd")
__path__ = [_pkg_dir]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(_pkg_dir, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f
