"""Import alias: ``import velocyto_amd`` loads the package that lives in ``velocyto.py_amd/``.

The package directory keeps the name the project layout prescribes (``velocyto.py_amd``), which
is not a legal Python identifier; this one-file shim registers it under ``velocyto_amd`` so that
``import velocyto_amd``, ``from velocyto_amd import estimation`` etc. work.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "velocyto.py_amd")
_spec = importlib.util.spec_from_file_location("velocyto_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["velocyto_amd"] = _mod
_spec.loader.exec_module(_mod)
