"""ctypes binding of ``libsgp_amd.so`` (C ABI declared in ``include/sgp_amd.h``).

The library is the ONLY compute path of this package: there is no CPU
fallback.  Importing :mod:`sgp_amd` works without it (so host-side logic can be
tested on a CPU box), but the first call that needs a kernel raises
``RuntimeError`` if the shared object is missing or no MI355X is visible.
"""
import ctypes
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# SGP_AMD_LIB: developer override (ablation / experiment builds of tools/build_variant.sh)
LIB_PATH = os.environ.get("SGP_AMD_LIB") or os.path.join(_CSRC, "libsgp_amd.so")

c_i32, c_i64, c_f32, c_f64, c_p = (ctypes.c_int32, ctypes.c_int64,
                                   ctypes.c_float, ctypes.c_double,
                                   ctypes.c_void_p)
c_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/sgp_amd.h one to one
SIGNATURES = {
    "sgp_abi_version": (ctypes.c_int, []),
    "sgp_last_error": (ctypes.c_char_p, []),
    "sgp_build_arch": (ctypes.c_char_p, []),
    "sgp_tune_value": (c_i64, [ctypes.c_char_p, c_i64]),
    "sgp_spmm_csr_f32": (ctypes.c_int, [c_p, c_p, c_p,
                                        c_p, c_i64, c_i64,
                                        c_p, c_i64, c_i64, c_i32,
                                        c_p, c_i64, c_i64,
                                        c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_spmm_tiled_f32": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p,
                                          c_i32, c_i32, c_i32, c_i32,
                                          c_p, c_i64, c_i64,
                                          c_p, c_i64, c_i64, c_i32,
                                          c_p, c_i64, c_i64,
                                          c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_spmm_res_f32": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                        c_i32, c_i32, c_i32,
                                        c_p, c_i64, c_i64,
                                        c_p, c_i64, c_i64, c_i32,
                                        c_p, c_i64, c_i64,
                                        c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_spmm_res_max_union": (c_i32, []),
    "sgp_spmm_res_max_quads": (c_i32, []),
    "sgp_spmm_res_tune": (ctypes.c_int, [c_i32]),
    "sgp_spmm_mix_f32": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                        c_p, c_p, c_p,
                                        c_i32, c_i32, c_i32,
                                        c_p, c_i64, c_i64,
                                        c_p, c_i64, c_i64, c_i32,
                                        c_p, c_i64, c_i64,
                                        c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_spmm_mix_max_union": (c_i32, []),
    "sgp_spmm_mix_max_dense": (c_i32, [c_i32]),
    "sgp_spmm_tiled_max_union": (c_i32, [c_i32]),
    "sgp_spmm_tiled_max_tile_rows": (c_i32, []),
    "sgp_spmm_tiled_max_row_edges": (c_i32, []),
    "sgp_reservoir_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "sgp_reservoir_f32": (ctypes.c_int, [c_p, c_i64, c_i64,
                                         c_p, c_p, c_p,
                                         c_f64, c_i32,
                                         c_p, c_i64, c_i64,
                                         c_p, c_p,
                                         c_i32, c_i32, c_i32, c_i32, c_p]),
    "sgp_reservoir_pieces_f32": (ctypes.c_int, [c_p, c_i64, c_i64,
                                                c_p, c_p, c_p,
                                                c_f64, c_i32,
                                                c_p, c_i64, c_i64,
                                                c_p, c_p,
                                                c_i32, c_i32, c_i32, c_i64, c_i64, c_i32,
                                                c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_reservoir_fused_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgp_reservoir_fused_supported": (c_i32, [c_i32, c_i32, c_i32]),
    "sgp_reservoir_fused_f32": (ctypes.c_int, [c_p, c_i64, c_i64,
                                               c_p, c_p, c_p,
                                               c_p, c_i32,
                                               c_p, c_i64, c_i64,
                                               c_p, c_p,
                                               c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sgp_reservoir_fused_sums_f32": (ctypes.c_int, [c_p, c_i64, c_i64,
                                                    c_p, c_p, c_p,
                                                    c_p, c_i32,
                                                    c_p, c_i64, c_i64,
                                                    c_p, c_p, c_p,
                                                    c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sgp_gesn_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "sgp_gesn_tune": (ctypes.c_int, [c_i32]),
    "sgp_gesn_f32": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i32,
                                    c_p, c_i64, c_i64, c_p, c_p,
                                    c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sgp_gemm_nt_f32": (ctypes.c_int, [c_p, c_i64, c_p, c_i64, c_p, c_p, c_i64,
                                       c_i32, c_i32, c_i32, c_p]),
    "sgp_gesn_update_f32": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_i32,
                                           c_p, c_p, c_i64, c_i32, c_i32, c_p]),
    "sgp_node_mean_bcast_f32": (ctypes.c_int, [c_p, c_i64, c_i64,
                                               c_p, c_i64, c_i64, c_p,
                                               c_i32, c_i32, c_i32, c_p]),
    "sgp_bcast_rows_f32": (ctypes.c_int, [c_p, c_f32, c_p, c_i64, c_i64,
                                          c_i32, c_i32, c_i32, c_p]),
    "sgp_copy_rows_f32": (ctypes.c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_i64,
                                         c_i32, c_i32, c_i32, c_p]),
    "sgp_gather_rows_f32": (ctypes.c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i32,
                                           c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
    "sgp_grouped_linear_packed_floats": (c_i64, [c_i32, c_i32, c_i32]),
    "sgp_grouped_linear_pack_f32": (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p]),
    "sgp_grouped_linear_f32": (ctypes.c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i32,
                                              c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p]),
    "sgp_grouped_linear_fwd_f32": (ctypes.c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i32,
                                                  c_p, c_i64, c_p, c_f64, c_u64,
                                                  c_i32, c_i32, c_i32, c_i32, c_p]),
    "sgp_grouped_linear_dact_f32": (ctypes.c_int, [c_p, c_i64, c_p, c_i32, c_f64, c_u64, c_p, c_i64, c_i32, c_p]),
    "sgp_grouped_linear_transpose_f32": (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p]),
    "sgp_grouped_linear_wgrad_f32": (ctypes.c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p,
                                                    c_i32, c_i32, c_i32, c_i32, c_p]),
    "sgp_abs_max_f32": (ctypes.c_int, [c_p, c_i64, c_i64, c_i32, c_i32, c_i32, c_p, c_p]),
    "sgp_spmm_split_f32": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_p, c_i64, c_i64,
                                          c_p, c_i64, c_i64, c_i32, c_p, c_i64, c_i64,
                                          c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_col_stats_f32": (ctypes.c_int, [c_p, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    "sgp_split_prepare_f32": (ctypes.c_int, [c_p, c_f64, c_f64, c_i32, c_p, c_f32, c_f32, c_i32, c_p, c_p, c_p, c_p]),
    "sgp_spmm_split_wide_f32": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_p, c_i64, c_i64,
                                               c_p, c_i64, c_i64, c_i32, c_p, c_i64, c_i64,
                                               c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_spmm_split_wide_chunks": (c_i32, []),
    "sgp_spmm_split_wide_max_union": (c_i32, []),
    "sgp_spmm_split_wide_waves": (c_i32, []),
    "sgp_spmm_split_wide_rows_per_wave": (c_i32, []),
    "sgp_spmm_split_wide_max_feat": (c_i32, []),
    "sgp_spmm_split_chunks": (c_i32, []),
    "sgp_spmm_split_max_union": (c_i32, []),
    "sgp_spmm_split_waves": (c_i32, []),
    "sgp_spmm_split_rows_per_wave": (c_i32, []),
    "sgp_spmm_split_max_feat": (c_i32, []),
    "sgp_split_plan_deal": (c_i64, [c_p, c_p, c_i64, c_i64, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    "sgp_split_plan_fill": (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64,
                                           c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32]),
    "sgp_spmm_colblock_f32": (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_p, c_i64, c_i64,
                                             c_p, c_i64, c_i64, c_i32, c_p, c_i64, c_i64,
                                             c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    "sgp_spmm_colblock_rows_cap": (c_i32, []),
    "sgp_spmm_colblock_round_pad": (c_i32, []),
    "sgp_event_create": (ctypes.c_int, [ctypes.POINTER(c_p)]),
    "sgp_event_destroy": (ctypes.c_int, [c_p]),
    "sgp_event_record": (ctypes.c_int, [c_p, c_p]),
    "sgp_event_elapsed_ms": (ctypes.c_int, [c_p, c_p, ctypes.POINTER(c_f32)]),
}

ACT_CODES = {"tanh": 0, "relu": 1, "self_norm": 2, "identity": 3,
             "tanh_rel": 4}     # tanh with relative accuracy near zero (layers whose bias is tiny: ReservoirLayer.kernel_activation)

_lib = None


def build(jobs=8, verbose=False, asan=False):
    """Compile every HIP source for gfx950 into ``csrc/libsgp_amd.so`` (in-tree).  ``asan=True`` also
    builds ``csrc/build_asan/libsgp_amd_asan.so`` -- the host halves under AddressSanitizer -- which
    ``tests/test_abi.py::test_host_asan_build`` drives (argument checks of every entry point, planner
    output through the hop kernels' launch arithmetic)."""
    out = subprocess.run(["make", "-C", _CSRC, f"-j{jobs}"], capture_output=True, text=True)
    if asan and out.returncode == 0:
        out = subprocess.run(["make", "-C", _CSRC, "-f", "Makefile.asan", f"-j{jobs}"], capture_output=True, text=True)
    if verbose or out.returncode:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode:
        raise RuntimeError("building libsgp_amd.so failed")
    return LIB_PATH


def load():
    """dlopen the library and attach the prototypes (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C sgp_amd/csrc`). sgp_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.sgp_abi_version() != 3:
        raise RuntimeError("libsgp_amd.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def require_gpu():
    """The product path needs the HIP library AND a device; fail loudly otherwise."""
    lib = load()
    if not torch.cuda.is_available():
        raise RuntimeError("sgp_amd needs an MI355X (torch.cuda.is_available() is "
                           "False) and has no CPU fallback")
    return lib


_masked_streams = {}


def cu_masked_streams(device, n_reserved):
    """Two HIP streams that split the device's compute units: ``(few, rest)`` -- ``few`` may run on ``n_reserved`` CUs only,
    ``rest`` on all the others (hipExtStreamCreateWithCUMask; mask bit i is CU i of the runtime's numbering, which
    interleaves the XCDs: a multiple of 8 reserves the same number in every XCD).  For a latency-bound kernel of a
    few workgroups (the small-graph reservoir: one wave per SIMD) that runs BESIDE bandwidth-bound kernels filling the
    chip: without the split the dispatcher puts waves of both on the same SIMDs and the serial chain pays for every
    issue slot it loses.  Returns None when the runtime refuses (the caller keeps ordinary streams)."""
    dev = torch.device(device)
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(n_reserved))
    if key in _masked_streams:
        return _masked_streams[key]
    pair = None
    if 0 < n_reserved < total:
        try:
            rt = ctypes.CDLL("libamdhip64.so")
            words = (total + 31) // 32
            made = []
            for bits in (range(0, n_reserved), range(n_reserved, total)):
                mask = (ctypes.c_uint32 * words)()
                for b in bits:
                    mask[b // 32] |= 1 << (b % 32)
                h = ctypes.c_void_p()
                with torch.cuda.device(dev):
                    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
                if rc != 0 or not h.value:
                    made = None
                    break
                made.append(torch.cuda.ExternalStream(h.value, device=dev))
            pair = tuple(made) if made else None
        except (OSError, AttributeError):
            pair = None
    _masked_streams[key] = pair
    return pair


def _check(rc, what):
    if rc != 0:
        msg = load().sgp_last_error().decode()
        kind = NotImplementedError if rc == -2 else RuntimeError
        raise kind(f"{what} failed (code {rc}): {msg}")


def _on_device(fn):
    """Run a binding under the device of its first CUDA tensor argument: the launches go to that
    device's current stream, and helper calls inside the library (memsets, copies, event records)
    follow the process's CURRENT device -- a tensor on cuda:1 while cuda:0 is current would
    otherwise mix devices."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        for a in list(args) + list(kw.values()):
            if torch.is_tensor(a) and a.is_cuda:
                if a.device.index == torch.cuda.current_device():
                    break
                with torch.cuda.device(a.device):
                    return fn(*args, **kw)
        return fn(*args, **kw)
    return wrapped


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _view3(t, name):
    """[B, N, D] float32 CUDA view with unit feature stride -> (ptr, row_stride, batch_stride)."""
    if t.dim() != 3 or t.dtype != torch.float32 or not t.is_cuda:
        raise ValueError(f"{name}: expected a 3-D float32 CUDA tensor, got "
                         f"{tuple(t.shape)} {t.dtype} {t.device}")
    if t.shape[2] > 1 and t.stride(2) != 1:
        raise ValueError(f"{name}: feature stride must be 1")
    return t.data_ptr(), t.stride(1), t.stride(0)


MAX_GRID_BATCH = 65535


# ---------------------------------------------------------------- SpMM
@_on_device
def spmm_csr(rowptr, col, val, x, y, halo=None, n_own=None, pred=None):
    """y[b, i, :] = sum_e val[e] x[b, col[e], :] (generic CSR kernel)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    n_rows = rowptr.numel() - 1
    n_cols = x.shape[1] + (halo.shape[1] if halo is not None else 0)
    if halo is not None:
        hp, hrs, hbs = _view3(halo, "halo")
        n_own = x.shape[1] if n_own is None else n_own
    else:
        hp, hrs, hbs, n_own = None, 0, 0, n_cols
    B, D = x.shape[0], x.shape[2]
    # the vector kernel walks 4 batch entries per grid row; feature widths / strides that are not
    # multiples of 4 floats (or unaligned pointers) take the scalar kernel: one entry per grid row
    vec = D % 4 == 0 and all(v % 4 == 0 for v in (xrs, xbs, yrs, ybs, hrs, hbs)) and \
        all((ptr or 0) % 16 == 0 for ptr in (xp, yp, hp))
    step = (4 if vec else 1) * MAX_GRID_BATCH
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        _check(lib.sgp_spmm_csr_f32(
            rowptr.data_ptr(), col.data_ptr(), val.data_ptr(),
            xp + 4 * b0 * xbs, xrs, xbs,
            (hp + 4 * b0 * hbs) if hp else None, hrs, hbs, n_own,
            yp + 4 * b0 * ybs, yrs, ybs,
            n_rows, n_cols, nb, D, *_pred(pred), _stream(x)), "sgp_spmm_csr_f32")


@_on_device
def spmm_tiled(plan, x, y, halo=None, n_own=None, pred=None):
    """Same product through the LDS-staged kernel; ``plan`` from graph.TilePlan.to(device)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    if halo is not None:
        hp, hrs, hbs = _view3(halo, "halo")
        n_own = x.shape[1] if n_own is None else n_own
    else:
        hp, hrs, hbs, n_own = None, 0, 0, 0
    _check(lib.sgp_spmm_tiled_f32(
        plan.trow.data_ptr(), plan.uptr.data_ptr(), plan.ucol.data_ptr(), plan.erow.data_ptr(),
        plan.ecol.data_ptr(), plan.eval.data_ptr(),
        plan.tile_rows, plan.n_tiles, plan.max_union, plan.max_row_edges,
        xp, xrs, xbs, hp, hrs, hbs, n_own, yp, yrs, ybs,
        plan.n_rows, x.shape[1] + (halo.shape[1] if halo is not None else 0),
        x.shape[0], x.shape[2], *_pred(pred), _stream(x)), "sgp_spmm_tiled_f32")


@_on_device
def spmm_res(plan, x, y, halo=None, n_own=None, pred=None):
    """Register-resident two-phase row-group product, exact fp32 (plan: ``TilePlan.pipe``)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    if halo is not None:
        hp, hrs, hbs = _view3(halo, "halo")
        n_own = x.shape[1] if n_own is None else n_own
    else:
        hp, hrs, hbs, n_own = None, 0, 0, 0
    ps = plan.pipe
    _check(lib.sgp_spmm_res_f32(
        ps["uptr"].data_ptr(), ps["ucol"].data_ptr(), ps["usplit"].data_ptr(),
        ps["gptr"].data_ptr(), ps["gsup"].data_ptr(), ps["gidx"].data_ptr(), ps["gw"].data_ptr(),
        ps["rowmap"].data_ptr(),
        plan.n_tiles, ps["max_union"], ps["max_tile_quads"],
        xp, xrs, xbs, hp, hrs, hbs, n_own, yp, yrs, ybs,
        plan.n_rows, x.shape[1] + (halo.shape[1] if halo is not None else 0),
        x.shape[0], x.shape[2], *_pred(pred), _stream(x)), "sgp_spmm_res_f32")


@_on_device
def spmm_mix(plan, x, y, halo=None, n_own=None, pred=None):
    """Mixed dense (16x16x4) / sparse (4x4x1) row-group product (plan: sgp_amd.mixplan.MixPlan on the
    device of ``x``)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    if halo is not None:
        hp, hrs, hbs = _view3(halo, "halo")
        n_own = x.shape[1] if n_own is None else n_own
    else:
        hp, hrs, hbs, n_own = None, 0, 0, 0
    _check(lib.sgp_spmm_mix_f32(
        plan.uptr.data_ptr(), plan.ucol.data_ptr(), plan.usplit.data_ptr(),
        plan.gptr.data_ptr(), plan.gsup.data_ptr(), plan.gidx.data_ptr(), plan.gw.data_ptr(),
        plan.rowmap.data_ptr(), plan.dptr.data_ptr(), plan.didx.data_ptr(), plan.dw.data_ptr(),
        plan.n_tiles, plan.max_union, plan.max_dense,
        xp, xrs, xbs, hp, hrs, hbs, n_own, yp, yrs, ybs,
        plan.n_rows, x.shape[1] + (halo.shape[1] if halo is not None else 0),
        x.shape[0], x.shape[2], *_pred(pred), _stream(x)), "sgp_spmm_mix_f32")


@_on_device
def abs_max(x):
    """max |x| of a [B, N, D] view as a Python float (one device reduction + one 4-byte copy)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _check(lib.sgp_abs_max_f32(xp, xrs, xbs, x.shape[1], x.shape[0], x.shape[2], out.data_ptr(), _stream(x)),
           "sgp_abs_max_f32")
    return float(out.item())


def mark_unit_bounded(state):
    """Record that every entry of ``state`` lies in [-1, 1] NOW (a zeroed state, or one a bounded-activation reservoir of
    this package left): the a-priori bound 1 of the split-fp16 hop may be used for what the recurrence produces from it.
    The mark is tied to the tensor's version counter: any in-place edit by the caller (``state.mul_(5)``) invalidates it
    (the kernels write through raw pointers and leave the counter alone)."""
    state._sgp_unit_bounded = state._version
    return state


def is_unit_bounded(state):
    return state is not None and getattr(state, "_sgp_unit_bounded", None) == state._version


class ColumnBound:
    """Per-column upper bounds of |x| as a DEVICE tensor ``[feat]`` (what ``split_profile`` leaves for the next hop:
    ``bound * ||A||_inf``); a column whose bound is 0 is identically zero."""

    def __init__(self, tensor):
        self.tensor = tensor


class SplitProfile:
    """Device-side decision record of one split-fp16 hop: ``tab[2, feat]`` (per-column scale | inverse), ``flag[1]``
    (1 = the operand meets the kernel's precision contract, 0 = the exact kernels must run), ``bound_out`` (the
    next hop's ``ColumnBound``)."""

    def __init__(self, tab, flag, bound_out):
        self.tab, self.flag, self.bound_out = tab, flag, bound_out


def _pred(pred):
    """``pred`` of the hop bindings: None (unconditional) or ``(flag, run_if)`` -- the launch runs only if the DEVICE
    word ``flag[0] == run_if`` when its kernel starts (include/sgp_amd.h, "Launch predicate")."""
    if pred is None:
        return None, 0
    flag, run_if = pred
    if not (torch.is_tensor(flag) and flag.is_cuda and flag.dtype == torch.int32 and flag.numel() >= 1):
        raise ValueError("pred: expected (int32 CUDA tensor, run_if)")
    return flag.data_ptr(), int(run_if)


@_on_device
def col_stats(x, t_stride=1, stats=None, r_stride=1):
    """Per-column max |x| (float bit patterns) and sum of squares of the steps 0, t_stride, .. and rows 0, r_stride, .. of
    a [B, N, D] view, as a device tensor ``[2, D]``; ``stats`` given = accumulate a second source into it."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    acc = stats is not None
    if stats is None:
        stats = torch.empty(2, x.shape[2], dtype=torch.float32, device=x.device)
    _check(lib.sgp_col_stats_f32(xp, xrs, xbs, x.shape[1], x.shape[0], x.shape[2], int(t_stride), int(r_stride), int(acc),
                                 stats.data_ptr(), _stream(x)), "sgp_col_stats_f32")
    return stats


SPLIT_SAMPLE_STEPS = 8         # steps the admission statistics of a hop read (all of them when there are fewer)
SPLIT_SAMPLE_BYTES = 64 << 20  # ... and at most this many bytes of them (every r-th row beyond that): 0.8 GB -> 51 MB on the target line


@_on_device
def split_profile(x, halo=None, bound=None, norm_inf=1.0, guard=True):
    """Enqueue what the split-fp16 hop needs to know about its operand, all on the device (no host sync):
    per-column statistics of ``x`` (+ ``halo``), then ``sgp_split_prepare_f32`` -> ``SplitProfile``.
    ``bound``: None = measured (one pass over every row), a float = the caller's a-priori bound on |x| (tanh /
    self_norm states: 1), or the ``ColumnBound`` the previous hop left.  With a bound the statistics are read from
    ~``SPLIT_SAMPLE_STEPS`` evenly spaced steps only.  ``guard=False`` skips the statistics altogether (the caller
    vouches; needs a bound)."""
    import math
    lib = require_gpu()
    B, N, D = x.shape
    dev = x.device
    out = torch.empty(3 * D + 4, dtype=torch.float32, device=dev)     # tab[2 D] | bound_out[D] | flag
    tab, bound_out, flag = out[:2 * D].view(2, D), out[2 * D:3 * D], out[3 * D:3 * D + 1].view(torch.int32)
    if isinstance(bound, ColumnBound):
        b_in, b_scalar = bound.tensor, 0.0
        if b_in.numel() != D or b_in.device != dev:
            raise ValueError("ColumnBound does not match the operand")
    elif bound is None:
        b_in, b_scalar = None, 0.0
    else:
        b_in, b_scalar = None, float(bound)
        if not (b_scalar > 0 and math.isfinite(b_scalar)):
            raise ValueError("split_profile needs a finite positive bound on |x| (or None: measured)")
    measured = b_in is None and b_scalar == 0.0
    stats, n_samples, s_eff, full = None, 1.0, 1.0, 0
    if guard or measured:
        t_stride = 1 if measured else max(1, B // SPLIT_SAMPLE_STEPS)
        ns = -(-B // t_stride)
        r_stride = 1 if measured else max(1, -(-(ns * N * D * 4) // SPLIT_SAMPLE_BYTES))
        all_rows = N + (halo.shape[1] if halo is not None else 0)
        sources = [x] + ([halo] if halo is not None and halo.shape[1] > 0 else [])
        if measured or (t_stride == 1 and r_stride == 1):
            for k, src in enumerate(sources):
                stats = col_stats(src, 1, stats if k else None, r_stride=1)
            n_sum = float(B) * all_rows
        else:
            # a strided sample of the steps before the last one ...
            n_sum, first = 0.0, True
            if B > 1:
                ns = -(-(B - 1) // t_stride)
                for src in sources:
                    stats = col_stats(src[:B - 1], t_stride, None if first else stats, r_stride=r_stride)
                    first = False
                    n_sum += float(ns) * -(-src.shape[1] // r_stride)
            # ... and EVERY row of the last step: a non-finite value that entered a recurrence anywhere in this time chunk is
            # still there at its end (and in every hop of it), so the encoders' own operands cannot hide one from the test
            # in an unsampled row or step (round-5 advice; 25 MB on the target line).  Any subset of the rows is a valid
            # sample for the mean-square bound: n_samples and s_eff count what was summed.
            for src in sources:
                stats = col_stats(src[B - 1:], 1, None if first else stats, r_stride=1)
                first = False
                n_sum += float(src.shape[1])
        n_samples, s_eff, full = n_sum, (B * all_rows) / n_sum, int(t_stride == 1 and r_stride == 1)
    _check(lib.sgp_split_prepare_f32(None if stats is None else stats.data_ptr(), n_samples, s_eff, full,
                                     None if b_in is None else b_in.data_ptr(), b_scalar, float(norm_inf), D,
                                     tab.data_ptr(), bound_out.data_ptr(), flag.data_ptr(), _stream(x)),
           "sgp_split_prepare_f32")
    return SplitProfile(tab, flag, ColumnBound(bound_out))


@_on_device
def spmm_split(plan, x, y, profile, t_chunk=0, halo=None, n_own=None, predicated=False):
    """Split-fp16 hop (plan: sgp_amd.splitplan.SplitPlan on the device of ``x``; ``profile``: the ``SplitProfile``
    of this operand, or a float bound on |x| / None for which one is made here without the admission test --
    callers that force the kernel).  ``predicated``: launch under ``profile.flag == 1`` (the caller enqueues the
    exact kernel under ``== 0`` behind it)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    if halo is not None:
        hp, hrs, hbs = _view3(halo, "halo")
        n_own = x.shape[1] if n_own is None else n_own
    else:
        hp, hrs, hbs, n_own = None, 0, 0, 0
    if not isinstance(profile, SplitProfile):
        profile = split_profile(x, halo, profile, guard=False)
    plans = plan if isinstance(plan, (list, tuple)) else [plan]
    pr = _pred((profile.flag, 1) if predicated else None)
    for p in plans:                                   # (several passes: an operator whose long rows were cut into column segments)
        # the plan's geometry names its kernel: 16 waves x 7 chunks (standard) or 8 x 14 (wide: long rows)
        entry = lib.sgp_spmm_split_wide_f32 if p.afr.shape[1] == lib.sgp_spmm_split_wide_waves() and \
            p.afr.shape[2] == lib.sgp_spmm_split_wide_chunks() else lib.sgp_spmm_split_f32
        _check(entry(
            p.hdr.data_ptr(), p.rowid.data_ptr(), p.ucol.data_ptr(), p.afr.data_ptr(), p.adr.data_ptr(),
            p.rinv.data_ptr(), p.n_tiles,
            xp, xrs, xbs, hp, hrs, hbs, n_own, yp, yrs, ybs, p.n_rows, p.n_cols, x.shape[0], x.shape[2],
            profile.tab.data_ptr(), int(p.accumulate), t_chunk, *pr, _stream(x)), "sgp_spmm_split_f32")
    return profile


@_on_device
def spmm_colblock(plan, x, y, halo=None, n_own=None, pred=None):
    """Column-blocked hop for graphs without locality (plan: sgp_amd.colblock.ColBlockPlan on the device
    of ``x``)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    if halo is not None:
        hp, hrs, hbs = _view3(halo, "halo")
        n_own = x.shape[1] if n_own is None else n_own
    else:
        hp, hrs, hbs, n_own = None, 0, 0, 0
    _check(lib.sgp_spmm_colblock_f32(
        plan.entries.data_ptr(), plan.segptr.data_ptr(), plan.wg_row0.data_ptr(), plan.n_wg, plan.n_blocks,
        xp, xrs, xbs, hp, hrs, hbs, n_own, yp, yrs, ybs, plan.n_rows, plan.n_cols, x.shape[0], x.shape[2], *_pred(pred),
        _stream(x)), "sgp_spmm_colblock_f32")


def split_limits(wide=False):
    """Plan limits of the split-fp16 hop: the standard form (16 waves x 7 chunks) or the wide one (8 x 14)."""
    lib = load()
    if wide:
        return dict(waves=lib.sgp_spmm_split_wide_waves(), chunks=lib.sgp_spmm_split_wide_chunks(),
                    max_union=lib.sgp_spmm_split_wide_max_union(), rows_per_wave=lib.sgp_spmm_split_wide_rows_per_wave())
    return dict(waves=lib.sgp_spmm_split_waves(), chunks=lib.sgp_spmm_split_chunks(),
                max_union=lib.sgp_spmm_split_max_union(), rows_per_wave=lib.sgp_spmm_split_rows_per_wave())


def tiled_limits(feat):
    """Plan limits that satisfy every LDS-staged kernel at once (one plan serves them all)."""
    lib = load()
    return dict(max_union=min(lib.sgp_spmm_tiled_max_union(feat), lib.sgp_spmm_res_max_union()),
                max_tile_rows=min(lib.sgp_spmm_tiled_max_tile_rows(), 64),
                max_row_edges=lib.sgp_spmm_tiled_max_row_edges())


def tall_tile_limits(feat):
    """Limits of the VALU kernel alone (sgp_spmm_tiled_f32): tiles of up to 512 rows, 8 per edge group."""
    lib = load()
    return dict(max_union=lib.sgp_spmm_tiled_max_union(feat),
                max_tile_rows=lib.sgp_spmm_tiled_max_tile_rows(),
                max_row_edges=lib.sgp_spmm_tiled_max_row_edges())


# ---------------------------------------------------------------- reservoir
_WORKSPACES = {}


def _workspace(device, nbytes):
    """Device scratch for packed weights, reused per (device, stream): kernels of one stream run
    in order and every call re-packs, so a call never reads what another left behind."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes // 4 + 1, 1024), dtype=torch.float32, device=device)
        _WORKSPACES[key] = ws
    return ws


def reservoir_fused_supported(F, R, L):
    """True when all L layers fit the fused multi-layer kernel (sgp_reservoir_fused_f32)."""
    return bool(load().sgp_reservoir_fused_supported(F, R, L))


@_on_device
def reservoir_stack(x, weights, alphas, activation, out, h_state=None, col_sums=None):
    """All layers of a stacked reservoir in one launch: x[T, N, F] -> out[T, N, L*R] (views
    allowed), ``weights`` = [(w_ih, w_hh, b)] per layer on the device, ``h_state`` [L, N, R].
    ``col_sums`` [T, L*R] (contiguous): also receives the sum over nodes of every step's states (the
    kernel writes per-tile sums from its registers, a small second launch adds the tiles) -- what
    the global_attr block needs, without re-reading the states from HBM."""
    lib = require_gpu()
    xp, xrs, xss = _view3(x, "x")
    op, ors, oss = _view3(out, "out")
    T, N, F = x.shape
    L, R = len(weights), weights[0][1].shape[0]
    for l, (w_ih, w_hh, b) in enumerate(weights):
        for name, w, shape in (("w_ih", w_ih, (R, F if l == 0 else R)), ("w_hh", w_hh, (R, R)), ("b", b, (R,))):
            if tuple(w.shape) != shape or w.dtype != torch.float32 or not w.is_cuda \
                    or not w.is_contiguous():
                raise ValueError(f"layer {l} {name}: expected contiguous float32 CUDA {shape}")
    if out.shape[0] != T or out.shape[1] != N or out.shape[2] != L * R:
        raise ValueError("out: expected [T, N, L*R]")
    if h_state is not None and (tuple(h_state.shape) != (L, N, R) or not h_state.is_contiguous()):
        raise ValueError("h_state: expected contiguous [L, N, R]")
    wsb = lib.sgp_reservoir_fused_workspace_bytes(F, R, L)
    if wsb < 0 or not lib.sgp_reservoir_fused_supported(F, R, L):
        raise NotImplementedError(f"fused reservoir kernel: F={F}, R={R}, L={L} not supported")
    ws = _workspace(x.device, wsb)
    ptrs = lambda k: (ctypes.c_void_p * L)(*[w[k].data_ptr() for w in weights])
    al = (ctypes.c_double * L)(*[float(a) for a in alphas])
    tile_sums = None
    if col_sums is not None:
        if tuple(col_sums.shape) != (T, L * R) or not col_sums.is_contiguous() or col_sums.dtype != torch.float32:
            raise ValueError("col_sums: expected contiguous float32 [T, L*R]")
        tile_sums = torch.empty((N + 15) // 16, T, L * R, dtype=torch.float32, device=x.device)
    _check(lib.sgp_reservoir_fused_sums_f32(
        xp, xrs, xss, ptrs(0), ptrs(1), ptrs(2), al, ACT_CODES[activation], op, ors, oss,
        h_state.data_ptr() if h_state is not None else None, ws.data_ptr(),
        tile_sums.data_ptr() if tile_sums is not None else None,
        T, N, F, R, L, _stream(x)), "sgp_reservoir_fused_sums_f32")
    if tile_sums is not None and T > 0:
        col_sums.copy_(node_sums(tile_sums.permute(1, 0, 2)))       # [T, tiles, D] view: rows = tiles
    return out


@_on_device
def reservoir_layer(x, w_ih, w_hh, b, alpha, activation, out, h_state=None):
    """One leaky-ESN layer over all T steps: x[T, N, F] -> out[T, N, R] (views allowed)."""
    lib = require_gpu()
    xp, xrs, xss = _view3(x, "x")
    op, ors, oss = _view3(out, "out")
    T, N, F = x.shape
    R = w_hh.shape[0]
    for name, w, shape in (("w_ih", w_ih, (R, F)), ("w_hh", w_hh, (R, R)), ("b", b, (R,))):
        if tuple(w.shape) != shape or w.dtype != torch.float32 or not w.is_cuda \
                or not w.is_contiguous():
            raise ValueError(f"{name}: expected contiguous float32 CUDA {shape}")
    if out.shape[0] != T or out.shape[1] != N or out.shape[2] != R:
        raise ValueError("out: expected [T, N, R]")
    wsb = lib.sgp_reservoir_workspace_bytes(F, R)
    if wsb < 0:
        raise NotImplementedError(f"reservoir kernel supports input/hidden sizes <= 256 "
                                  f"(got F={F}, R={R})")
    ws = _workspace(x.device, wsb)
    if h_state is not None and (tuple(h_state.shape) != (N, R) or not h_state.is_contiguous()):
        raise ValueError("h_state: expected contiguous [N, R]")
    _check(lib.sgp_reservoir_f32(
        xp, xrs, xss, w_ih.data_ptr(), w_hh.data_ptr(), b.data_ptr(),
        float(alpha), ACT_CODES[activation], op, ors, oss,
        h_state.data_ptr() if h_state is not None else None, ws.data_ptr(),
        T, N, F, R, _stream(x)), "sgp_reservoir_f32")
    return out


@_on_device
def reservoir_pieces(x, w_ih, w_hh, b, alpha, activation, out, states, t_piece, t_last, x_piece_stride, out_piece_stride,
                     no_store=False, pred=None):
    """``sgp_reservoir_pieces_f32``: ``states[P, N, R]`` (contiguous) holds every piece's initial state and receives its
    final one; piece p reads ``x`` / writes ``out`` at p times the piece strides (in floats) from the given views'
    first element.  ``states[N, R]`` or None with one piece: the sequential layer, optionally under ``pred``."""
    lib = require_gpu()
    xp, xrs, xss = _view3(x, "x")
    op, ors, oss = _view3(out, "out")
    N, F = x.shape[1], x.shape[2]
    R = w_hh.shape[0]
    for name, w, shape in (("w_ih", w_ih, (R, F)), ("w_hh", w_hh, (R, R)), ("b", b, (R,))):
        if tuple(w.shape) != shape or w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous():
            raise ValueError(f"{name}: expected contiguous float32 CUDA {shape}")
    P = 1 if states is None or states.dim() == 2 else states.shape[0]
    if states is not None and (not states.is_contiguous() or tuple(states.shape[-2:]) != (N, R)):
        raise ValueError("states: expected contiguous [P, N, R] (or [N, R])")
    wsb = lib.sgp_reservoir_workspace_bytes(F, R)
    if wsb < 0:
        raise NotImplementedError(f"reservoir kernel supports input/hidden sizes <= 256 (got F={F}, R={R})")
    ws = _workspace(x.device, wsb)
    _check(lib.sgp_reservoir_pieces_f32(
        xp, xrs, xss, w_ih.data_ptr(), w_hh.data_ptr(), b.data_ptr(), float(alpha), ACT_CODES[activation],
        op, ors, oss, states.data_ptr() if states is not None else None, ws.data_ptr(),
        int(t_piece), int(t_last), P, int(x_piece_stride), int(out_piece_stride), int(bool(no_store)),
        N, F, R, *_pred(pred), _stream(x)), "sgp_reservoir_pieces_f32")
    return out


# ---------------------------------------------------------------- DynGESN
@_on_device
def gesn_sequence(rowptr, col, val, x, weights, alphas, activation, out, h_state):
    """Whole DynGESN sequence: x[T, N, F] -> out[T, N, L*R]; ``weights`` is a list of
    (w_ih, w_hh, b) device tensors per layer, ``h_state`` [L, N, R] is updated in place."""
    lib = require_gpu()
    T, n, f = x.shape
    L = len(weights)
    r = weights[0][1].shape[0]
    assert x.stride(2) == 1 and out.stride(2) == 1 and out.shape == (T, n, L * r)
    assert h_state.is_contiguous() and tuple(h_state.shape) == (L, n, r)
    for i, (w_ih, w_hh, b) in enumerate(weights):
        assert w_ih.is_contiguous() and w_hh.is_contiguous() and b.is_contiguous()
        assert tuple(w_ih.shape) == (r, f if i == 0 else r) and tuple(w_hh.shape) == (r, r)
    ptrs = lambda k: (ctypes.c_void_p * L)(*[w[k].data_ptr() for w in weights])
    al = (ctypes.c_double * L)(*[float(a) for a in alphas])
    ws = torch.empty(max(16, lib.sgp_gesn_workspace_bytes(n, r, L)) // 4, dtype=torch.float32,
                     device=x.device)
    _check(lib.sgp_gesn_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(),
                            x.data_ptr(), x.stride(1), x.stride(0), ptrs(0), ptrs(1), ptrs(2), al,
                            ACT_CODES[activation], out.data_ptr(), out.stride(1), out.stride(0),
                            h_state.data_ptr(), ws.data_ptr(), T, n, f, r, L, _stream(x)),
           "sgp_gesn_f32")
    return out


@_on_device
def gemm_nt(a, w, bias, out):
    """out[m, n] = sum_k a[m, k] w[n, k] (+ bias[n]); 2-D float32 CUDA, unit inner strides."""
    lib = require_gpu()
    assert a.dim() == 2 and w.dim() == 2 and out.dim() == 2 and a.shape[1] == w.shape[1]
    assert a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1
    _check(lib.sgp_gemm_nt_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0),
                               bias.data_ptr() if bias is not None else None,
                               out.data_ptr(), out.stride(0), a.shape[0], w.shape[0], a.shape[1],
                               _stream(a)), "sgp_gemm_nt_f32")
    return out


@_on_device
def gesn_update(rowptr, col, val, z, p, h_in, alpha, activation, h_out, out_rows):
    """DynGESN state update for one (step, layer); ``out_rows`` is the [N, R] slot (row stride
    arbitrary) of the embedding that receives the new state."""
    lib = require_gpu()
    n, r = h_in.shape
    for t in (z, p, h_in, h_out):
        assert t.is_contiguous() and tuple(t.shape) == (n, r)
    assert out_rows.shape == (n, r) and out_rows.stride(1) == 1
    _check(lib.sgp_gesn_update_f32(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(),
                                   z.data_ptr(), p.data_ptr(), h_in.data_ptr(), float(alpha),
                                   ACT_CODES[activation], h_out.data_ptr(), out_rows.data_ptr(),
                                   out_rows.stride(0), n, r, _stream(z)), "sgp_gesn_update_f32")
    return h_out


# ---------------------------------------------------------------- helpers
def _batched(fn, B):
    for b0 in range(0, B, MAX_GRID_BATCH):
        fn(b0, min(MAX_GRID_BATCH, B - b0))


@_on_device
def node_mean_bcast(x, y):
    """y[b, i, :] = mean_j x[b, j, :] for every node i (global_attr block)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    B, N, D = x.shape
    scratch = torch.empty(min(B, MAX_GRID_BATCH), D, dtype=torch.float32, device=x.device)

    def run(b0, nb):
        _check(lib.sgp_node_mean_bcast_f32(xp + 4 * b0 * xbs, xrs, xbs, yp + 4 * b0 * ybs,
                                           yrs, ybs, scratch.data_ptr(), N, nb, D,
                                           _stream(x)), "sgp_node_mean_bcast_f32")
    _batched(run, B)
    return y


@_on_device
def node_sums(x):
    """[B, D] un-normalised column sums (multi-GPU: all-reduce these, then bcast_rows)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    B, N, D = x.shape
    out = torch.empty(B, D, dtype=torch.float32, device=x.device)

    def run(b0, nb):
        _check(lib.sgp_node_mean_bcast_f32(xp + 4 * b0 * xbs, xrs, xbs, None, 0, 0,
                                           out.data_ptr() + 4 * b0 * D, N, nb, D,
                                           _stream(x)), "sgp_node_mean_bcast_f32")
    _batched(run, B)
    return out


@_on_device
def bcast_rows(src, scale, y):
    lib = require_gpu()
    yp, yrs, ybs = _view3(y, "y")
    B, N, D = y.shape
    assert src.is_contiguous() and tuple(src.shape) == (B, D)

    def run(b0, nb):
        _check(lib.sgp_bcast_rows_f32(src.data_ptr() + 4 * b0 * D, float(scale),
                                      yp + 4 * b0 * ybs, yrs, ybs, N, nb, D, _stream(y)),
               "sgp_bcast_rows_f32")
    _batched(run, B)
    return y


@_on_device
def copy_rows(x, y):
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    yp, yrs, ybs = _view3(y, "y")
    B, N, D = x.shape

    def run(b0, nb):
        _check(lib.sgp_copy_rows_f32(xp + 4 * b0 * xbs, xrs, xbs, yp + 4 * b0 * ybs, yrs, ybs,
                                     N, nb, D, _stream(x)), "sgp_copy_rows_f32")
    _batched(run, B)
    return y


@_on_device
def gather_nodes(x, node_index, out=None):
    """out[b, k, :] = x[b, node_index[k], :] (halo packing)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    B, _, D = x.shape
    K = node_index.numel()
    if out is None:
        out = torch.empty(B, K, D, dtype=torch.float32, device=x.device)
    op, ors, obs = _view3(out, "out")

    def run(b0, nb):
        _check(lib.sgp_gather_rows_f32(xp + 4 * b0 * xbs, xrs, xbs, None, node_index.data_ptr(),
                                       K, op + 4 * b0 * obs, ors, obs, nb, D, _stream(x)),
               "sgp_gather_rows_f32")
    _batched(run, B)
    return out


@_on_device
def gather_rows(x, step_index, node_index):
    """out[k, :] = x[step_index[k], node_index[k], :] (IID sampling of the embedding)."""
    lib = require_gpu()
    xp, xrs, xbs = _view3(x, "x")
    K, D = node_index.numel(), x.shape[2]
    out = torch.empty(K, D, dtype=torch.float32, device=x.device)
    _check(lib.sgp_gather_rows_f32(xp, xrs, xbs, step_index.data_ptr(), node_index.data_ptr(), K,
                                   out.data_ptr(), D, 0, 1, D, _stream(x)), "sgp_gather_rows_f32")
    return out


GL_ACT_CODES = {None: 0, "linear": 0, "identity": 0, "relu": 1, "silu": 2}


@_on_device
def grouped_linear_pack(weight, groups):
    """Conv1d weight [groups*oc, ic(, 1)] (CUDA) -> MFMA fragment order."""
    lib = require_gpu()
    w = weight.reshape(weight.shape[0], -1).contiguous().float()
    oc, ic = w.shape[0] // groups, w.shape[1]
    packed = torch.empty(lib.sgp_grouped_linear_packed_floats(groups, ic, oc), dtype=torch.float32,
                         device=w.device)
    _check(lib.sgp_grouped_linear_pack_f32(w.data_ptr(), packed.data_ptr(), groups, ic, oc, _stream(w)),
           "sgp_grouped_linear_pack_f32")
    return packed


def _gl_rows(x2, step_index, node_index, source):
    if source is not None:
        xp, xrs, xbs = _view3(source, "source")
        return xp, xrs, xbs, node_index.numel(), step_index.data_ptr(), node_index.data_ptr(), source.device
    if x2.dim() != 2 or x2.stride(1) != 1:
        raise ValueError("grouped_linear: rows must be a 2-D view with unit feature stride")
    return x2.data_ptr(), x2.stride(0), 0, x2.shape[0], None, None, x2.device


@_on_device
def grouped_linear(x2, packed, bias, groups, ic, oc, activation, step_index=None, node_index=None,
                   source=None, want_pre=False, dropout_p=0., seed=0):
    """rows [K, groups*ic] -> [K, groups*oc]; with (step_index, node_index) the rows are gathered
    from ``source[T, N, groups*ic]`` instead of being read from ``x2``.  ``want_pre``: also return the
    pre-activation values (for the backward pass); ``dropout_p`` / ``seed``: Philox dropout mask."""
    lib = require_gpu()
    xp, xrs, xbs, K, sp, np_, dev = _gl_rows(x2, step_index, node_index, source)
    out = torch.empty(K, groups * oc, dtype=torch.float32, device=dev)
    pre = torch.empty(K, groups * oc, dtype=torch.float32, device=dev) if want_pre else None
    _check(lib.sgp_grouped_linear_fwd_f32(xp, xrs, xbs, sp, np_, packed.data_ptr(), bias.data_ptr(),
                                          GL_ACT_CODES[activation], out.data_ptr(), out.stride(0),
                                          pre.data_ptr() if want_pre else None, float(dropout_p), int(seed),
                                          K, groups, ic, oc, _stream(out)), "sgp_grouped_linear_fwd_f32")
    return (out, pre) if want_pre else out


@_on_device
def grouped_linear_dact(dy, pre, activation, dropout_p=0., seed=0):
    """dz = dy * dropout factor * act'(pre) (contiguous [K, width])."""
    lib = require_gpu()
    if dy.dim() != 2 or dy.stride(1) != 1:
        dy = dy.contiguous()
    dz = torch.empty_like(pre)
    _check(lib.sgp_grouped_linear_dact_f32(dy.data_ptr(), dy.stride(0), pre.data_ptr(), GL_ACT_CODES[activation],
                                           float(dropout_p), int(seed), dz.data_ptr(), pre.shape[0], pre.shape[1],
                                           _stream(dz)), "sgp_grouped_linear_dact_f32")
    return dz


@_on_device
def grouped_linear_transpose(weight, groups):
    """Conv1d weight [groups*oc, ic(, 1)] -> the weight [groups*ic, oc] of the transposed grouped layer."""
    lib = require_gpu()
    w = weight.reshape(weight.shape[0], -1).contiguous().float()
    oc, ic = w.shape[0] // groups, w.shape[1]
    wt = torch.empty(groups * ic, oc, dtype=torch.float32, device=w.device)
    _check(lib.sgp_grouped_linear_transpose_f32(w.data_ptr(), wt.data_ptr(), groups, ic, oc, _stream(w)),
           "sgp_grouped_linear_transpose_f32")
    return wt


@_on_device
def grouped_linear_wgrad(x2, dz, groups, ic, oc, step_index=None, node_index=None, source=None):
    """dW[groups*oc, ic] = sum over rows of dz[row, g*oc + o] * x[row, g*ic + i]."""
    lib = require_gpu()
    xp, xrs, xbs, K, sp, np_, dev = _gl_rows(x2, step_index, node_index, source)
    dw = torch.empty(groups * oc, ic, dtype=torch.float32, device=dev)
    _check(lib.sgp_grouped_linear_wgrad_f32(xp, xrs, xbs, sp, np_, dz.data_ptr(), dw.data_ptr(),
                                            K, groups, ic, oc, _stream(dw)), "sgp_grouped_linear_wgrad_f32")
    return dw


class Event:
    """HIP event on the stream the kernels run on (bench.py roofline timing)."""

    def __init__(self):
        self._h = c_p()
        _check(load().sgp_event_create(ctypes.byref(self._h)), "sgp_event_create")

    def record(self, stream=None):
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(load().sgp_event_record(self._h, s), "sgp_event_record")

    def elapsed_ms(self, end):
        ms = c_f32()
        _check(load().sgp_event_elapsed_ms(self._h, end._h, ctypes.byref(ms)),
               "sgp_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            load().sgp_event_destroy(self._h)
        except Exception:
            pass
