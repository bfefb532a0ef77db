"""The C-ABI library loads and exports every symbol include/sgp_amd.h declares (no GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from sgp_amd import hip


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    return hip.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "sgp_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_functions()
    assert len(names) >= 15
    raw = ctypes.CDLL(hip.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/sgp_amd.h but not exported"


def test_binding_covers_header(lib):
    assert set(hip.SIGNATURES) == set(declared_functions())


def test_version_and_arch(lib):
    assert lib.sgp_abi_version() == 1
    assert lib.sgp_build_arch() == b"gfx950"


def test_argument_errors_do_not_need_a_gpu(lib):
    # null pointers / bad sizes are rejected before anything touches the device
    rc = lib.sgp_spmm_csr_f32(None, None, None, None, 0, 0, None, 0, 0, 0, None, 0, 0,
                              4, 4, 1, 4, None)
    assert rc == -1 and b"null pointer" in lib.sgp_last_error()
    assert lib.sgp_reservoir_workspace_bytes(3, 64) > 0
    assert lib.sgp_reservoir_workspace_bytes(3, 1000) == -1
    assert lib.sgp_spmm_tiled_max_union(64) == 512
    assert lib.sgp_spmm_tiled_max_union(48) == 0


def test_no_cpu_fallback():
    import torch
    import sgp_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    enc = sgp_amd.SGPTemporalEncoder(3, reservoir_size=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.zeros(2, 3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sgp_amd.sgp_spatial_embedding(torch.zeros(2, 3, 4), 3, torch.tensor([[0, 1], [1, 2]]))
