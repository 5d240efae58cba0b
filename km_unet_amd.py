"""Import alias: the package lives in the (non-importable) directory name ``km-unet_amd/``.

``import km_unet_amd`` executes this file, which loads that directory as the package
``km_unet_amd`` and replaces itself in sys.modules.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "km-unet_amd")
_spec = importlib.util.spec_from_file_location("km_unet_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["km_unet_amd"] = _pkg
_spec.loader.exec_module(_pkg)
