"""Import alias for the package directory ``dfd-clip_amd/``.

The directory name is fixed by the project layout but is not a valid Python
identifier, so this one-file module turns itself into that package: it points
``__path__`` at the directory and executes its ``__init__.py`` in its own
namespace.  ``import dfd_clip_amd`` / ``from dfd_clip_amd.detector import Detector``
then behave exactly as for an ordinary package.
"""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "dfd-clip_amd")
__path__ = [_pkg_dir]
__package__ = "dfd_clip_amd"
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(_pkg_dir, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg_dir, "__init__.py"), "exec"))
del _f
