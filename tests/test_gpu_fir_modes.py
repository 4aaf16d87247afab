"""The engine's reduced-precision arithmetics for the advanced version: the 40 complex FIR filters of the
filter-bank ear model (fbearmodel.c:399-435) on the FP16 matrix instruction with split operands (the
DEFAULT, PEAQ_FIR_F16X3) and on the FP32 matrix instruction (PEAQ_FIR_F32), everything else FP64
(include/peaq_amd.h, peaq_ctx_set_fir_mode; priced in profiles/r02_precision_ledger.json).
Tolerances of THESE paths, stated here and nowhere looser:
  * per-block excitation patterns vs the oracle: 1e-4 relative (22..24-bit products and FP32 sums over up to
    1456 taps; measured: 2e-5 on noise-like signals, 5e-5 in the bands between the harmonics of a
    sawtooth, where a band's own output is what leaks from its strong neighbours; the FP64 path is
    held to 1e-9 in test_gpu_parity.py),
  * MOVs vs the real reference's goldens: 2e-6 relative (+1e-9 absolute), DI and ODG 1e-6 absolute
    (north star: 0.02),
  * ODG of full-size seeded pairs vs the FP64 path: 1e-6.
Needs an MI355X (`-m gpu`)."""
import json

import numpy as np
import pytest

import cases as case_defs
import oracle_lib as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f16x3", "f32"])
def fp32_ctx(request):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    import gstpeaq_amd
    c = gstpeaq_amd.Context(0)
    assert c.fir_mode() == "f64" and c.fir_fp64() is True, "the reference's FP64 arithmetic must be the default"
    c.set_fir_mode(request.param)
    assert c.fir_mode() == request.param and c.fir_fp64() is False
    yield c
    c.close()


@pytest.mark.parametrize("case", [dict(kind="synth", seed=5, channels=1, n=40000),
                                  dict(kind="synth", seed=6, channels=2, n=30000, test_trim=900),
                                  dict(kind="ats", wave_ref="saw", wave_test="triangle", n=32768, channels=1)],
                         ids=["mono", "stereo-ragged", "saw-triangle"])
def test_filterbank_blocks_fp32_fir(fp32_ctx, case):
    import torch
    import gstpeaq_amd
    ref, test = case_defs.make_inputs(case)
    n_blocks = min(len(ref), len(test)) // 192
    got = gstpeaq_amd.debug_filterbank(fp32_ctx, torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(), n_blocks, 320)
    for c in range(ref.shape[1]):
        for sig, lo in ((ref, 0), (test, 40)):
            exp = orc.fbear(np.ascontiguousarray(sig[:, c]), n_blocks)
            np.testing.assert_allclose(got[:, c, lo:lo + 40], exp["unsmeared"], rtol=1e-4)
            np.testing.assert_allclose(got[:, c, 80 + lo:120 + lo], exp["excitation"], rtol=1e-4)


def test_advanced_goldens_fp32_fir(fp32_ctx, golden_dir):
    """all advanced end-to-end cases of the real reference (27 + 4 playback levels)"""
    import torch
    import gstpeaq_amd
    recs = [r for r in json.loads((golden_dir / "ref_e2e.json").read_text()) if r["case"]["advanced"]]
    recs += [r for r in json.loads((golden_dir / "ref_e2e_level.json").read_text()) if r["case"]["advanced"]]
    worst = 0.0
    for rec in recs:
        case = rec["case"]
        ref, test = case_defs.make_inputs(case)
        n = max(len(ref), len(test), 2)
        n += n & 1
        r = np.zeros((1, n, ref.shape[1]), dtype=np.float32)
        t = np.zeros_like(r)
        r[0, : len(ref)] = ref
        t[0, : len(test)] = test
        got = gstpeaq_amd.batch_run(fp32_ctx, 1, torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda(),
                                    np.array([len(ref)], dtype=np.uint32), np.array([len(test)], dtype=np.uint32),
                                    playback_level=case.get("level", 92.0))[0]
        exp = np.array([float(v) for v in rec["movs"]])
        assert np.array_equal(np.isnan(got["movs"]), np.isnan(exp)), case["name"]
        ok = ~np.isnan(exp)
        np.testing.assert_allclose(got["movs"][ok], exp[ok], rtol=2e-6, atol=1e-9, err_msg=case["name"])
        if not np.isnan(float(rec["odg"])):
            assert abs(got["odg"] - float(rec["odg"])) <= 1e-6 and abs(got["di"] - float(rec["di"])) <= 1e-6, case["name"]
            worst = max(worst, abs(got["odg"] - float(rec["odg"])))
    print(f"{fp32_ctx.fir_mode()} FIR: max |dODG| vs the reference over {len(recs)} advanced cases: {worst:.3e}")


def test_fullsize_pairs_fp32_vs_fp64_fir(fp32_ctx):
    import gstpeaq_amd
    import gpu_common
    ref, test = gstpeaq_amd.synth_fill(fp32_ctx, 1001, 32, 2, 480000)
    a = gstpeaq_amd.batch_run(fp32_ctx, 1, ref, test)
    b = gstpeaq_amd.batch_run(gpu_common.ctx("f64"), 1, ref, test)     # the shared FP64 context
    assert gpu_common.ctx("f64").fir_fp64() is True
    d = max(abs(x["odg"] - y["odg"]) for x, y in zip(a, b))
    assert d <= 1e-6, d
    for x, y in zip(a, b):
        np.testing.assert_allclose(x["movs"], y["movs"], rtol=2e-6, atol=1e-9)
    print(f"{fp32_ctx.fir_mode()} vs FP64 FIR on 32 ten-second pairs: max |dODG| {d:.3e}")


def test_samples_far_beyond_full_scale_are_scaled_not_saturated():
    """Float input far beyond full scale (the WAV float formats allow it; the reference's filter bank is FP64 and
    has no range limit, fbearmodel.c:276-435): the FP16 operands of the split-FP16 FIR hold 30 dB of headroom above
    full scale at the launch's scale; a signal whose filtered peak -- recorded by the high-pass walk -- goes beyond
    it runs at its own power of two instead (peaq_fb.hip, fb_bank_body).  +26 dB (inside the headroom), +60 dB
    and +100 dB all follow the FP64 engine to 1e-6 in ODG and 2e-6 in the MOVs; a burst far beyond full scale
    in an otherwise ordinary signal too."""
    import torch
    import gstpeaq_amd
    import gpu_common
    c = gstpeaq_amd.Context(0)
    c.set_fir_fp64(False)                            # = PEAQ_FIR_F16X3
    assert c.fir_mode() == "f16x3"
    ref, test = case_defs.make_inputs(dict(kind="synth", seed=3, channels=2, n=48000))
    cases = [(ref * np.float32(g), test * np.float32(g)) for g in (20.0, 1000.0, 1e5)]
    burst_r, burst_t = ref.copy(), test.copy()
    burst_r[20000:20400] *= np.float32(3000.)
    burst_t[20000:20400] *= np.float32(3000.)
    cases.append((burst_r, burst_t))
    for r_, t_ in cases:
        r = torch.from_numpy(np.ascontiguousarray(r_[None])).cuda()
        t = torch.from_numpy(np.ascontiguousarray(t_[None])).cuda()
        got = gstpeaq_amd.batch_run(c, 1, r, t)[0]
        exp = gstpeaq_amd.batch_run(gpu_common.ctx("f64"), 1, r, t)[0]
        assert np.isfinite(got["odg"]) and np.all(np.isfinite(got["movs"][:5])), got
        assert abs(got["odg"] - exp["odg"]) <= 1e-6, (float(np.abs(r_).max()), got["odg"], exp["odg"])
        np.testing.assert_allclose(got["movs"][:5], exp["movs"][:5], rtol=2e-6, atol=1e-9)
        # ... and a session fed in pieces (other launches, other peaks per launch) agrees with it
        s = gstpeaq_amd.Session(c, 1, 2)
        for lo in range(0, len(r_), 7000):
            s.push_ref(r_[lo:lo + 7000])
            s.push_test(t_[lo:lo + 7000])
        s.flush()
        live = s.results()
        s.close()
        assert abs(live["odg"] - exp["odg"]) <= 1e-6
    # ... also fed ONE filter-bank block (192 samples) at a time: the window's head then spans eight earlier
    # launches, and the burst (+40 dB over full scale for a moment, 2..7 launches back) must still set the scale --
    # the head's peak is taken from the head itself (fb_hp_kernel), not from "this launch and the one before"
    hot_r, hot_t = ref.copy(), test.copy()
    hot_r[20000:20150] = np.float32(100.) * np.sign(hot_r[20000:20150] + np.float32(1e-9))
    hot_t[20000:20150] = hot_r[20000:20150]
    r = torch.from_numpy(np.ascontiguousarray(hot_r[None])).cuda()
    t = torch.from_numpy(np.ascontiguousarray(hot_t[None])).cuda()
    exp = gstpeaq_amd.batch_run(gpu_common.ctx("f64"), 1, r, t)[0]
    s = gstpeaq_amd.Session(c, 1, 2)
    for lo in range(0, len(hot_r), 192):
        s.push_ref(hot_r[lo:lo + 192])
        s.push_test(hot_t[lo:lo + 192])
    s.flush()
    live = s.results()
    s.close()
    assert np.isfinite(live["odg"]) and abs(live["odg"] - exp["odg"]) <= 1e-6, (live["odg"], exp["odg"])
    np.testing.assert_allclose(live["movs"][:5], exp["movs"][:5], rtol=2e-6, atol=1e-9)
    c.close()
