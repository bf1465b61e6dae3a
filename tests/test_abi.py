"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/dibs_hip.h
declares; the ctypes mirror of dibs_config has the C layout."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dibs_hip.h")


@pytest.fixture(scope="module")
def lib():
    from dibs_amd import _lib
    _lib.build()
    return _lib.load()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dibs_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dibs_hip.h but not exported"
    from dibs_amd._lib import EXPORTS
    assert sorted(EXPORTS) == names


def test_abi_version(lib):
    from dibs_amd._abi import ABI_VERSION
    assert lib.dibs_abi_version() == ABI_VERSION


def test_config_struct_layout_matches_header():
    from dibs_amd._abi import DibsConfig
    prog = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "dibs_hip.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu\n", sizeof(dibs_config), offsetof(dibs_config, nn_hidden), offsetof(dibs_config, rank),
             offsetof(dibs_config, alpha_linear), offsetof(dibs_config, nn_sig_param));
      return 0;
    }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    got = [ctypes.sizeof(DibsConfig), DibsConfig.nn_hidden.offset, DibsConfig.rank.offset, DibsConfig.alpha_linear.offset,
           DibsConfig.nn_sig_param.offset]
    assert got == [int(v) for v in out]


def test_engine_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from dibs_amd._abi import make_config
    from dibs_amd.engine import Engine
    from dibs_amd._lib import DibsHipError
    with pytest.raises(DibsHipError):
        Engine(make_config(n_vars=5, n_particles=4, n_observations=10, edges_per_node=1))
