"""Import alias: `import mt3d_amd` loads the package that lives in the (non-identifier) directory
`multi-task-3d-resencoder-unet_amd/`, so `mt3d_amd.builders.build_network_from_config` etc. resolve
exactly like the reference's `builders.build_network_from_config`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-task-3d-resencoder-unet_amd")
_spec = importlib.util.spec_from_file_location("mt3d_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mt3d_amd"] = _mod
_spec.loader.exec_module(_mod)
