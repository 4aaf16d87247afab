"""Helpers shared by the -m gpu tests (they call the product ONLY through the C ABI)."""
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"

# The advanced version's filter bank has two arithmetics the suite runs (include/peaq_amd.h, peaq_ctx_set_fir_mode):
#   "default"  what ships: everything FP64 like the reference (PEAQ_FIR_F64);
#   "f16x3"    the opt-in fast engine: FIR bank on the FP16 matrix instruction with split operands, slopes and
#              upward spreading in FP32 (PEAQ_FIR_F16X3).
# Tests that take the `fir_mode` fixture (tests/conftest.py) run once per mode; every other test runs on the
# default engine.  Tolerances of advanced-version results per mode, stated HERE and nowhere looser -- results of
# the basic version never pass through the filter bank and are held to the "default" column in either mode:
#   movs    MOVs against the real reference's goldens / the oracle (relative; + 1e-9 absolute)
#   odg     DI and ODG against the same (absolute; north star: 0.02)
#   blocks  per-block excitation patterns of the filter bank against the oracle (relative)
#   chunks  one stream cut into launches in different ways (session, broker, batch) against itself (relative)
MODES = ("default", "f16x3")
TOL = {"f16x3": dict(movs=2e-6, odg=1e-6, blocks=1e-4, chunks=1e-9),
       "default": dict(movs=1e-7, odg=1e-7, blocks=1e-9, chunks=1e-10)}
_MODE = "default"
_CTX = {}


def set_mode(mode):
    global _MODE
    assert mode in MODES
    _MODE = mode


def mode():
    return _MODE


def tol(kind, advanced=True):
    return TOL[_MODE if advanced else "default"][kind]


def ctx(mode=None):
    """The shared context of the parity tests in the current FIR mode (or the one asked for)."""
    m = mode or _MODE
    if m == "f64":                                   # the default by its own name
        m = "default"
    if m not in _CTX:
        import gstpeaq_amd
        c = gstpeaq_amd.Context(0)
        assert c.fir_mode() == "f64" and c.fir_fp64() is True, "the reference's FP64 arithmetic must be the engine's default"
        if m == "f16x3":
            c.set_fir_mode("f16x3")
        _CTX[m] = c
    return _CTX[m]


def e2e_records(advanced=None):
    recs = json.loads((GOLD / "ref_e2e.json").read_text())
    if advanced is not None:
        recs = [r for r in recs if r["case"]["advanced"] == advanced]
    return recs


def run_batch(cases_inputs, advanced, channels):
    """cases_inputs: list of (ref, test) float32 [n, channels]; ragged -> one batch call"""
    import torch
    import gstpeaq_amd
    n_pairs = len(cases_inputs)
    stride = max(max(len(r), len(t)) for r, t in cases_inputs)
    stride += stride & 1
    ref = np.zeros((n_pairs, stride, channels), dtype=np.float32)
    test = np.zeros_like(ref)
    n_ref = np.zeros(n_pairs, dtype=np.uint32)
    n_test = np.zeros(n_pairs, dtype=np.uint32)
    for i, (r, t) in enumerate(cases_inputs):
        ref[i, : len(r)] = r
        test[i, : len(t)] = t
        n_ref[i], n_test[i] = len(r), len(t)
    d_ref = torch.from_numpy(ref).cuda()
    d_test = torch.from_numpy(test).cuda()
    return gstpeaq_amd.batch_run(ctx(), advanced, d_ref, d_test, n_ref, n_test)


def compare_result(got, rec, rtol, atol, odg_atol):
    exp = np.array([float(v) for v in rec["movs"]])
    assert got["frames"] == rec["frames"], (got["frames"], rec["frames"])
    assert np.array_equal(np.isnan(got["movs"]), np.isnan(exp)), (got["movs"], exp)
    ok = ~np.isnan(exp)
    np.testing.assert_allclose(got["movs"][ok], exp[ok], rtol=rtol, atol=atol)
    for k, tol in (("di", odg_atol), ("odg", odg_atol), ("totalsnr", 1e-9)):
        e = float(rec[k])
        if np.isnan(e):
            assert np.isnan(got[k]), (k, got[k])
        elif np.isinf(e):
            assert got[k] == e, (k, got[k])
        else:
            assert abs(got[k] - e) <= tol + rtol * abs(e), (k, got[k], e)
