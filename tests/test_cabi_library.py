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


C_CALLER = r"""
#include <stdio.h>
#include "rxunet.h"
/* what a non-Python host (the cgo / JNI / plain C side of INTEGRATION.md) does first: version, a descriptor, the workspace
 * queries (host arithmetic only -- no device is touched), and argument validation that must fail with a status, not a crash */
int main(void) {
  rx_act y = {0};
  y.ptr = (void*)0x1000; y.n = 2; y.z = 16; y.y = 16; y.x = 16; y.c = 32; y.ld = 32; y.cs = 0;
  printf("%d %zu %zu\n", rx_abi_version(), rx_instnorm_stats_workspace(&y), rx_se_workspace(&y));
  rx_act bad = y; bad.c = 0;
  printf("%zu\n", rx_instnorm_stats_workspace(&bad));
  int rc = rx_instnorm_stats_mask(NULL, NULL, 4, NULL);
  printf("%d %s\n", rc, rx_last_error());
  return 0;
}
"""


def test_header_is_plain_c_and_a_c_host_links_against_the_library(tmp_path):
    """include/rxunet.h must be usable from C (the drop-in boundary is a C ABI, not a C++ or Python one): it compiles as strict
    C99, and a C program linked against librxunet.so gets the same answers as the ctypes binding."""
    import subprocess
    import ctypes
    import __graft_entry__
    __graft_entry__.build()
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import lib
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "host.c"
    src.write_text(C_CALLER)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)])
    libdir = os.path.dirname(lib.LIB_PATH)
    exe = tmp_path / "host"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lrxunet",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    so = lib.load()

    class Act(ctypes.Structure):
        _fields_ = [("ptr", ctypes.c_void_p)] + [(k, ctypes.c_int32) for k in ("n", "z", "y", "x", "c", "ld")] + [("cs", ctypes.c_int64)]
    a = Act(0x1000, 2, 16, 16, 16, 32, 32, 0)
    raw = ctypes.CDLL(lib.LIB_PATH)
    for fn in (raw.rx_instnorm_stats_workspace, raw.rx_se_workspace):
        fn.restype, fn.argtypes = ctypes.c_size_t, [ctypes.c_void_p]
    want = f"{so.rx_abi_version()} {raw.rx_instnorm_stats_workspace(ctypes.byref(a))} {raw.rx_se_workspace(ctypes.byref(a))}"
    assert lines[0] == want and int(lines[0].split()[1]) > 0
    assert lines[1] == "0"                                  # a bad descriptor: size 0, no crash
    rc, msg = lines[2].split(" ", 1)
    assert int(rc) < 0 and "rx_instnorm_stats_mask" in msg  # status + message, as every entry point


def test_numbered_event_slots_are_recycled():
    """slots were never returned, so a process that keeps rebuilding plans ran into `out of event slots`.  The table
    is host state only (a hipEvent_t is created at the first record), so this runs without a device."""
    import mt3d_amd  # noqa: F401
    from mt3d_amd.engine import lib
    so = lib.load()
    base = so.rx_event_slots_in_use()
    a = [so.rx_event_new() for _ in range(8)]
    assert len(set(a)) == 8 and so.rx_event_slots_in_use() == base + 8
    for s in a[:5]:
        assert so.rx_event_free(s) == 0
    assert so.rx_event_slots_in_use() == base + 3
    assert so.rx_event_free(a[0]) != 0 and b"not in use" in so.rx_last_error()      # double free is refused
    b = [so.rx_event_new() for _ in range(5)]
    assert sorted(b) == sorted(a[:5])                                                 # the freed numbers come back
    for s in a[5:] + b:
        assert so.rx_event_free(s) == 0
    assert so.rx_event_slots_in_use() == base
    for _ in range(70000):                                                            # more than the table holds, with recycling
        assert so.rx_event_free(so.rx_event_new()) == 0
