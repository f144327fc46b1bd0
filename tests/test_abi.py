"""CPU-side checks of the drop-in boundary: the C-ABI libraries load, export every symbol include/*.h declares, fail loudly without a
GPU (no CPU fallback), and the product never touches oracle/."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header, prefix):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(rf"\b({prefix}[a-z0-9_]+)\s*\(", src))
    return names


@pytest.fixture(scope="module")
def built():
    import __graft_entry__
    __graft_entry__.build()
    from diffsol_amd import _ffi
    return _ffi


def test_device_library_exports_every_declared_symbol(built):
    names = declared_functions("diffsol_hip.h", "dsh_")
    assert len(names) >= 60
    lib = ctypes.CDLL(built.lib_paths()[0])
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/diffsol_hip.h but not exported"
    assert names == set(built.DEVICE_ABI), names ^ set(built.DEVICE_ABI)


def test_host_library_exports_every_declared_symbol(built):
    names = declared_functions("diffsol_hip_solver.h", "dshs_")
    names.discard("dshs_options")
    built.load_device_lib()
    lib = ctypes.CDLL(built.lib_paths()[1])
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/diffsol_hip_solver.h but not exported"
    assert names == set(built.HOST_ABI), names ^ set(built.HOST_ABI)


def test_model_registry_metadata_needs_no_gpu(built):
    L = built.load_device_lib()
    for model, size, n, np_, mass, roots in [(0, 0, 2, 2, 0, 0), (1, 0, 3, 1, 1, 0), (3, 1, 3, 3, 0, 0), (3, 4, 12, 3, 0, 0), (4, 0, 3, 3, 1, 0),
                                              (5, 10, 10, 0, 0, 0), (6, 10, 10, 10, 0, 0), (7, 512, 512, 1, 0, 0), (8, 0, 4, 6, 1, 0), (8, 1, 4, 6, 1, 1),
                                              (9, 0, 2, 2, 0, 1)]:
        a, b, c, d = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int64()
        assert L.dsh_model_info(model, size, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)) == 0
        assert (a.value, b.value, c.value, d.value) == (n, np_, mass, roots)
    assert L.dsh_model_info(99, 0, None, None, None, None) < 0
    assert L.dsh_model_has_fused(3, 1) == 1 and L.dsh_model_has_fused(3, 4) == 0 and L.dsh_model_has_fused(7, 512) == 0


def test_no_cpu_fallback_without_gpu(built):
    """Without a HIP device the product must fail loudly rather than compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import diffsol_amd
    with pytest.raises(diffsol_amd.DiffsolHipError):
        diffsol_amd.Solver("robertson_ode", [0.04, 1e4, 3e7], model_size=1)
    with pytest.raises(diffsol_amd.DiffsolHipError):
        diffsol_amd.HipContext()


def test_product_never_references_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "diffsol_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".cpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"\boracle\b", txt, re.I) and "the oracle" not in txt.lower().replace("cpu oracle", "the oracle"):
                    offenders.append(os.path.join(base, f))
                if re.search(r"(import|from)\s+oracle|#include\s+\"[^\"]*oracle", txt):
                    offenders.append(os.path.join(base, f) + " (imports oracle)")
    for f in os.listdir(os.path.join(ROOT, "include")):
        txt = open(os.path.join(ROOT, "include", f)).read()
        assert "#include \"../oracle" not in txt
    assert not [o for o in offenders if "imports oracle" in o], offenders
