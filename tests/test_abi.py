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
    assert lib.sgp_abi_version() == 3
    assert lib.sgp_build_arch() == b"gfx950"


def test_argument_errors_do_not_need_a_gpu(lib):
    # null pointers / bad sizes are rejected before anything touches the device
    rc = lib.sgp_spmm_csr_f32(None, None, None, None, 0, 0, None, 0, 0, 0, None, 0, 0,
                              4, 4, 1, 4, None, 0, None)
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


ASAN_LIB = os.path.join(ROOT, "sgp_amd", "csrc", "build_asan", "libsgp_amd_asan.so")


def test_host_asan_build():
    """the host halves of the library under AddressSanitizer (`make -C sgp_amd/csrc -f
    Makefile.asan`; GPU ASan is not available on this pool).  Every entry point is driven through its
    argument checks -- null pointers, zero and negative sizes, unsupported shapes -- in a child
    process that preloads the ASan runtime; an ASan report aborts the child."""
    import glob
    import subprocess
    import sys
    # built (or brought up to date: make is incremental) on demand -- about 90 s on 8 cores after a source
    # change; a toolchain that cannot build it skips the test
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "sgp_amd", "csrc"), "-f", "Makefile.asan", "-j8"],
                       capture_output=True, text=True, timeout=1500)
    if r.returncode != 0 or not os.path.exists(ASAN_LIB):
        pytest.skip("ASan build failed here: " + r.stderr[-300:])
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no ASan runtime in this image")
    child = r'''
import ctypes, sys
sys.path.insert(0, %r)
from sgp_amd import hip
lib = ctypes.CDLL(%r)
n_calls = 0
for name, (restype, argtypes) in hip.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = restype, argtypes
    if restype is not ctypes.c_int:
        continue                                   # size queries are exercised below
    for fill in (0, -1, 3):
        scalar = lambda a: isinstance(a, type) and issubclass(a, ctypes._SimpleCData) and \
            a not in (ctypes.c_void_p, ctypes.c_char_p)
        args = [a(fill) if scalar(a) else None for a in argtypes]
        rc = fn(*args)
        assert isinstance(rc, int)
        n_calls += 1
lib.sgp_last_error.restype = ctypes.c_char_p
assert isinstance(lib.sgp_last_error(), bytes)
for f, r, l in ((3, 16, 8), (64, 64, 2), (300, 64, 2), (3, 16, 40)):
    lib.sgp_reservoir_fused_supported(f, r, l); lib.sgp_reservoir_fused_workspace_bytes(f, r, l)
    lib.sgp_reservoir_workspace_bytes(f, r); lib.sgp_gesn_workspace_bytes(100, r, l)
# ---- launch arithmetic with VALID plans: the planners' arrays (host memory) and host operand buffers go
# through every hop entry point; the host halves choose time chunks, grids, LDS sizes and template
# variants from them.  Without a GPU the launch itself fails cleanly (an error code) after that
# arithmetic; with one the host pointers must not reach a kernel, so this part is skipped.
import torch
n_plans = 0
if not torch.cuda.is_available():
    from sgp_amd import graph, synthetic
    for n, k, feat, batch in ((700, 20, 64, 40), (64, 5, 64, 3), (3000, 40, 128, 700)):
        ei, ew, _ = synthetic.knn_graph(n, k, seed=n)
        op = graph.ShiftOperator.from_edges(ei, ew, n)
        cpu = torch.device("cpu")
        plan = op.tile_plan(feat, cpu, tall=False)
        x = torch.zeros(batch, n, feat); y = torch.zeros(batch, n, feat)
        P = lambda t: t.data_ptr()
        rowptr, col, val = op.csr()
        rp32, c32 = rowptr.int(), col.int()
        rc = lib.sgp_spmm_csr_f32(P(rp32), P(c32), P(val), P(x), feat, n * feat, None, 0, 0, 0,
                                  P(y), feat, n * feat, n, n, batch, feat, None, 0, None)
        assert isinstance(rc, int) and rc != 0
        n_plans += 1
        if plan is None or plan.pipe is None:
            continue
        ps = plan.pipe
        common = (P(x), feat, n * feat, None, 0, 0, 0, P(y), feat, n * feat, plan.n_rows, n, batch, feat, None, 0, None)
        for fn in (lib.sgp_spmm_res_f32,):
            rc = fn(P(ps["uptr"]), P(ps["ucol"]), P(ps["usplit"]), P(ps["gptr"]), P(ps["gsup"]), P(ps["gidx"]),
                    P(ps["gw"]), P(ps["rowmap"]), plan.n_tiles, ps["max_union"], ps["max_tile_quads"], *common)
            assert isinstance(rc, int) and rc != 0          # no device: an error, not a crash
            n_plans += 1
        mp = op.mix_plan(feat, cpu, strict=False)
        if mp is not None:
            rc = lib.sgp_spmm_mix_f32(P(mp.uptr), P(mp.ucol), P(mp.usplit), P(mp.gptr), P(mp.gsup), P(mp.gidx),
                                      P(mp.gw), P(mp.rowmap), P(mp.dptr), P(mp.didx), P(mp.dw),
                                      mp.n_tiles, mp.max_union, mp.max_dense, *common)
            assert isinstance(rc, int) and rc != 0
            n_plans += 1
        sp = op.split_plan(cpu)
        if sp is not None and feat %% 16 == 0:
            tab = torch.ones(2, feat)
            rc = lib.sgp_spmm_split_f32(P(sp.hdr), P(sp.rowid), P(sp.ucol), P(sp.afr), P(sp.adr), P(sp.rinv), sp.n_tiles, P(x), feat, n * feat,
                                        None, 0, 0, 0, P(y), feat, n * feat, sp.n_rows, sp.n_cols, batch, feat, P(tab), 0, 0, None, 0, None)
            assert isinstance(rc, int) and rc != 0
            n_plans += 1
    # reservoir layer: every dispatch branch of the launch logic (split-J, exact deal + tail, even deal, stream)
    for N, F, R, T in ((207, 3, 64, 5), (20000, 64, 64, 3), (100000, 64, 64, 2), (131072, 8, 32, 2), (40000, 128, 256, 2)):
        x = torch.zeros(T, N, F); out = torch.zeros(T, N, R)
        w_ih = torch.zeros(R, F); w_hh = torch.zeros(R, R); b = torch.zeros(R)
        ws = torch.zeros(max(1, lib.sgp_reservoir_workspace_bytes(F, R) // 4))
        rc = lib.sgp_reservoir_f32(x.data_ptr(), F, N * F, w_ih.data_ptr(), w_hh.data_ptr(), b.data_ptr(), 0.9, 0,
                                   out.data_ptr(), R, N * R, None, ws.data_ptr(), T, N, F, R, None)
        assert isinstance(rc, int) and rc != 0
        n_plans += 1
    assert n_plans >= 8, n_plans
print("asan-ok", n_calls, n_plans)
''' % (ROOT, ASAN_LIB)
    # SGP_AMD_LIB: hip.load() -- i.e. the native split planner behind op.split_plan() -- uses the ASan build as well
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", SGP_AMD_LIB=ASAN_LIB)
    res = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "asan-ok" in res.stdout, (res.stdout[-500:], res.stderr[-3000:])
