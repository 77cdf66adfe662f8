"""CPU tests of the drop-in boundary: libaztot.so loads without a GPU, exports every symbol include/aztot.h declares,
and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes
import os
import re

import pytest

from aztotmd_amd import api, inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "aztot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aztot_[a-z_0-9]+)\s*\(", text)) - {"aztot_sendrecv_fn", "aztot_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(api.library_path())
    names = declared_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(api.EXPORTS) == names
    assert b"gfx950" in api.lib().aztot_version()


def test_struct_layouts_match_header():
    # sizes computed from the C declarations (natural alignment, x86-64)
    assert ctypes.sizeof(api._Species) == 56 and ctypes.sizeof(api._Vdw) == 64
    assert ctypes.sizeof(api._Options) == 96 and ctypes.sizeof(api._State) == 104
    assert ctypes.sizeof(api._Stats) == 296
    assert ctypes.sizeof(api._BondType) == 56 and ctypes.sizeof(api._AngleType) == 24 and ctypes.sizeof(api._Bonded) == 88


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = api.Model.from_case(inputs.config("F1"))
    with pytest.raises(api.AztotError) as ei:
        api.Engine(m)
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under aztotmd_amd/ may import, link or open it."""
    pkg = os.path.join(ROOT, "aztotmd_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".h", ".hip")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read().lower()
                for needle in ("import oracle", "from oracle", "oracle/", "liboracle", "ref_driver"):
                    assert needle not in text, (dirpath, fn, needle)


def test_rccl_id_size_is_exported():
    assert api.lib().aztot_comm_id_bytes() == 128
