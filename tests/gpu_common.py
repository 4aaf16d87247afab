"""Helpers shared by the -m gpu tests (they call the product ONLY through the C ABI)."""
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"

_CTX = None


def ctx():
    """The shared context of the parity tests.  It runs the filter bank's FIR on the FP64 matrix
    instruction: that is the path held to the oracle's 1e-9 / 1e-7; the engine's default (split-FP16 FIR,
    include/peaq_amd.h peaq_ctx_set_fir_fp64) has its own tests with its own stated tolerances
    (tests/test_gpu_fir_modes.py) and is what the CLI / element / feeder subprocess tests run."""
    global _CTX
    if _CTX is None:
        import gstpeaq_amd
        _CTX = gstpeaq_amd.Context(0)
        _CTX.set_fir_fp64(True)
    return _CTX


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
