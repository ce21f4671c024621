"""CPU: the C-ABI shared library loads and exports exactly the symbols include/rxunet.h declares
(no compute calls -- there is no GPU here), and the product fails loudly without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "rxunet.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rx_[A-Za-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    import ctypes
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import lib
    so = ctypes.CDLL(lib.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(so, name), f"{name} declared in include/rxunet.h but not exported"
    assert sorted(lib.exported_symbols()) == declared      # the ctypes table covers the whole header
    assert lib.load().rx_abi_version() == 1


def test_no_fallback_without_a_device():
    import torch
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import lib
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(lib.RxError):
        lib.require_device()
