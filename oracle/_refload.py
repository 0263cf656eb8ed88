"""Import the REAL reference (read-only checkout) under alias module names.

Used only by oracle/verify_against_reference.py and oracle/make_golden.py, in the build
container where /root/reference exists.  The GPU box has no reference checkout; nothing in
tests -m gpu / smoke() / bench.py goes through here.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("DIP_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models"))


def load_ref_models():
    """reference `models` package as `ref_models` (relative imports inside keep working)."""
    if "ref_models" in sys.modules:
        return sys.modules["ref_models"]
    path = os.path.join(REF, "models")
    spec = importlib.util.spec_from_file_location(
        "ref_models", os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_models"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_ref_common_utils():
    """reference utils/common_utils.py as `ref_common_utils` (torchvision/matplotlib stubbed:
    utils/common_utils.py:3,10 import them, only the plotting helpers use them)."""
    if "ref_common_utils" in sys.modules:
        return sys.modules["ref_common_utils"]
    for name in ("torchvision", "torchvision.utils", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    path = os.path.join(REF, "utils", "common_utils.py")
    spec = importlib.util.spec_from_file_location("ref_common_utils", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_common_utils"] = mod
    spec.loader.exec_module(mod)
    return mod
