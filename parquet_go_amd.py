"""Import shim: the package directory is named `parquet-go_amd` (not a valid Python identifier), so
`import parquet_go_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parquet-go_amd")
_spec = importlib.util.spec_from_file_location(
    "parquet_go_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["parquet_go_amd"] = _mod
_spec.loader.exec_module(_mod)
