"""Importable alias of the package directory ``active-3d-vision-and-touch_amd`` (whose name is not a valid
Python identifier).  ``import a3vt_amd`` executes that package's ``__init__`` under this name."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "active-3d-vision-and-touch_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
