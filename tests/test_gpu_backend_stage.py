"""Stage-level parity of the HIP pattern layer: the stateful back end
is driven on its own through peaq_debug_backend and compared
  * with the reference's OWN known-answer vectors for the level adapter and the modulation
    processor (testpeaq.c:433-599, test_leveladapt :748-784, test_modulationproc :787-810;
    extracted as data to tests/golden/testpeaq_vectors.json), with the reference's tolerance
    (rel 5e-5 or abs 5e-6, testpeaq.c:33-35,606-621), and
  * frame by frame with the oracle's ear model -> level adapter / modulation processor on two
    golden signal pairs (rtol 1e-9).
Needs an MI355X (`-m gpu`); everything goes through the C ABI."""
import json

import numpy as np
import pytest

import cases as case_defs
import oracle_lib as orc

pytestmark = pytest.mark.gpu

NB = 109


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def tp(golden_dir):
    return json.loads((golden_dir / "testpeaq_vectors.json").read_text())


def assert_testpeaq_close(got, exp, tol):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    err = np.abs(got - exp)
    ok = (err <= tol["rel"] * np.abs(exp)) | (err <= tol["abs"])
    assert ok.all(), (np.flatnonzero(~ok)[:5], got[~ok][:5], exp[~ok][:5])


def hand_records(n_frames, unsm_ref, unsm_test):
    """front-end records carrying the given unsmeared excitation patterns (and their ^0.3, which the
    front end computes for the modulation processor, modpatt.c:235); everything else neutral"""
    rec = np.zeros((n_frames, 1, 576))
    rec[:, 0, 0:NB] = unsm_ref
    rec[:, 0, 112:112 + NB] = unsm_test
    rec[:, 0, 224:224 + NB] = np.asarray(unsm_ref, dtype=np.float64) ** 0.3
    rec[:, 0, 336:336 + NB] = np.asarray(unsm_test, dtype=np.float64) ** 0.3
    rec[:, 0, 448:448 + NB] = 1.0        # noise in bands
    rec[:, 0, 563] = 3.0                 # flags: above threshold + energy (ref)
    rec[:, 0, 564] = 2.0
    return rec


def test_level_adapter_against_testpeaq_vectors(gpu, tp):
    """testpeaq.c:748-784: ref = i + 1, test = 109 - i, two calls.  Fed as unsmeared excitation: with the
    smearing filter starting from zero (fftearmodel.c:319-322) a constant input passes the max() of
    :501-503 unchanged, so the level adapter sees exactly the reference test's input."""
    import gstpeaq_amd.capi as capi
    ref = np.arange(1, NB + 1, dtype=np.float64)
    test = np.arange(NB, 0, -1, dtype=np.float64)
    d, _ = capi.debug_backend(gpu.ctx(), hand_records(2, ref, test))
    # (the kernels exchange E2^(1/4) per band, peaq_device.h: the hand-built E goes through a tenth root and
    # back, so it arrives with a few ulp of rounding, not bit for bit)
    np.testing.assert_allclose(d["exc_ref"][0, 0], ref, rtol=1e-13)
    np.testing.assert_allclose(d["exc_test"][1, 0], test, rtol=1e-13)
    tol = tp["_tolerance"]
    assert_testpeaq_close(d["adapted_ref"][0, 0], tp["spectrally_adapted_ref_patterns1_ref"], tol)
    assert_testpeaq_close(d["adapted_test"][0, 0], tp["spectrally_adapted_test_patterns1_ref"], tol)
    assert_testpeaq_close(d["adapted_ref"][1, 0], tp["spectrally_adapted_ref_patterns2_ref"], tol)
    assert_testpeaq_close(d["adapted_test"][1, 0], tp["spectrally_adapted_test_patterns2_ref"], tol)


def test_modulation_processor_against_testpeaq_vectors(gpu, tp):
    """testpeaq.c:787-810: input i + 1, two calls -> modulation and average loudness"""
    import gstpeaq_amd.capi as capi
    x = np.arange(1, NB + 1, dtype=np.float64)
    d, _ = capi.debug_backend(gpu.ctx(), hand_records(2, x, x))
    tol = tp["_tolerance"]
    for sig in ("ref", "test"):
        assert_testpeaq_close(d[f"mod_{sig}"][0, 0], tp["modulation1_ref"], tol)
        assert_testpeaq_close(d[f"avgloud_{sig}"][0, 0], tp["loudness1_ref"], tol)
        assert_testpeaq_close(d[f"mod_{sig}"][1, 0], tp["modulation2_ref"], tol)
        assert_testpeaq_close(d[f"avgloud_{sig}"][1, 0], tp["loudness2_ref"], tol)


@pytest.mark.parametrize("case", [
    dict(kind="synth", seed=12, channels=1, n=72000),                         # golden case synth_s12_mono
    dict(kind="ats", wave_ref="saw", wave_test="triangle", n=65536, channels=1),   # runtest-1.0.sh pipeline 2
], ids=["synth_s12_mono", "ats_saw_triangle"])
def test_backend_patterns_match_oracle_per_frame(gpu, case):
    """HIP front end -> HIP back end vs the oracle's ear model -> level adapter / modulation processor,
    every frame: excitation (time smearing), adapted patterns, modulation, average loudness, and the
    total loudness of the frames before the loudness gate opens"""
    import torch
    import gstpeaq_amd
    import gstpeaq_amd.capi as capi
    ref, test = case_defs.make_inputs(case)
    n_frames = (len(ref) - 2048) // 1024 + 2           # all full frames + the zero-padded flush frame
    recs = gstpeaq_amd.debug_frontend(gpu.ctx(), NB, torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(),
                                      n_frames)
    d, res = capi.debug_backend(gpu.ctx(), recs)
    pad = np.zeros(2048, dtype=np.float32)             # do_flush pads with zeros (gstpeaq.c:733-738)
    o_ref = orc.fftear(NB, np.concatenate([ref[:, 0], pad]), n_frames, 1024)
    o_test = orc.fftear(NB, np.concatenate([test[:, 0], pad]), n_frames, 1024)
    np.testing.assert_allclose(d["exc_ref"][:, 0], o_ref["excitation"], rtol=1e-9)
    np.testing.assert_allclose(d["exc_test"][:, 0], o_test["excitation"], rtol=1e-9)
    ad_r, ad_t = orc.leveladapt(NB, o_ref["excitation"], o_test["excitation"])
    np.testing.assert_allclose(d["adapted_ref"][:, 0], ad_r, rtol=1e-9)
    np.testing.assert_allclose(d["adapted_test"][:, 0], ad_t, rtol=1e-9)
    for sig, o in (("ref", o_ref), ("test", o_test)):
        mod, loud = orc.modproc(NB, o["unsmeared"])
        # the modulation is driven by |L - L_prev| (modpatt.c:231): for a stationary signal that difference
        # cancels to ~1e-5 of L, and the last-bit differences of the two spectra show at 1e-9 relative
        np.testing.assert_allclose(d[f"mod_{sig}"][:, 0], mod, rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(d[f"avgloud_{sig}"][:, 0], loud, rtol=1e-9)
    # total loudness is only evaluated until the gate opens (gstpeaq.c:841-845)
    gate = np.flatnonzero((o_ref["loudness"] > 0.1) & (o_test["loudness"] > 0.1))
    upto = (gate[0] + 1) if gate.size else n_frames
    np.testing.assert_allclose(d["loudness"][:upto, 0, 0], o_ref["loudness"][:upto], rtol=1e-9)
    np.testing.assert_allclose(d["loudness"][:upto, 0, 1], o_test["loudness"][:upto], rtol=1e-9)
    # and the MOVs after the last frame are those of the whole pair
    exp = orc.run_pair(0, ref, test)
    assert res["frames"] == exp["frames"] == n_frames
    np.testing.assert_allclose(res["movs"], exp["movs"], rtol=1e-7, atol=1e-9)
    assert abs(res["odg"] - exp["odg"]) < 1e-6


@pytest.mark.parametrize("case", [
    dict(kind="synth", seed=5, channels=1, n=72000),
    dict(kind="synth", seed=6, channels=2, n=60000, test_trim=900),
    dict(kind="synth", seed=1, channels=2, n=60000),             # leading digital silence
    dict(kind="synth", seed=9, channels=2, n=40000, atten_shift=9),   # around the detector thresholds
    dict(kind="synth", seed=26, channels=2, n=40000, identical=1),
    dict(kind="ats", wave_ref="saw", wave_test="triangle", n=65536, channels=1),
    dict(kind="synth", seed=41, channels=2, n=60000, gaps=[(20000, 9000), (40000, 3000)]),   # silence in mid-stream
], ids=["mono", "stereo-ragged", "lead-silence", "quiet", "identical", "saw-triangle", "mid-gaps"])
def test_mov_values_match_oracle_per_frame(gpu, case):
    """The MOV layer frame by frame, before the accumulators: modulation differences and their weight
    (movs.c:205-254), noise loudness (:354-371), mean and maximum of the band noise-to-mask ratios (:971-1023),
    detection probability and steps above threshold (:1224-1276) -- every frame, also those the gates of
    gstpeaq.c:871,880-881 keep from the accumulators.  End to end a wrong MOV shows; this says which and where."""
    import torch
    import gstpeaq_amd
    import gstpeaq_amd.capi as capi
    ref, test = case_defs.make_inputs(case)
    n = min(len(ref), len(test))
    ref, test = ref[:n], test[:n]
    n_frames = (n - 2048) // 1024 + 2           # all full frames + the zero-padded flush frame
    recs = gstpeaq_amd.debug_frontend(gpu.ctx(), NB, torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(),
                                      n_frames)
    d, _ = capi.debug_backend(gpu.ctx(), recs)
    exp = orc.mov_trace(ref, test, n_frames)
    ch = ref.shape[1]
    for name in orc.MOV_TRACE:
        got, want = d["mov"][name], exp[name]
        if name in ("p_detect", "steps"):
            got, want = got[:, :1], want[:, :1]
        # the modulation differences inherit the modulation's cancellation (|L - L_prev| of a stationary signal), the
        # noise-to-mask ratios the noise spectrum's (Pr - 2 sqrt(Pr Pt) + Pt of nearly equal spectra): 1e-6 there
        rtol = 1e-6 if name in ("moddiff1", "moddiff2", "nmr_mean", "nmr_max", "noiseloud") else 1e-9
        if case.get("identical") and name in ("nmr_mean", "nmr_max"):
            # identical signals: the noise spectrum is what rounding leaves of Pr - 2 sqrt(Pr Pr) + Pr
            assert np.all(np.abs(got[:, :ch]) < 1e-9) and np.all(np.abs(want[:, :ch]) < 1e-9), name
            continue
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-12, err_msg=name)


ADV_STAGE_CASES = [
    dict(kind="synth", seed=5, channels=1, n=72000),
    dict(kind="synth", seed=6, channels=2, n=60000, test_trim=900),
    dict(kind="synth", seed=1, channels=2, n=60000),             # leading digital silence
    dict(kind="synth", seed=9, channels=2, n=40000, atten_shift=9),   # around the detector thresholds
    dict(kind="synth", seed=26, channels=2, n=40000, identical=1),
    dict(kind="ats", wave_ref="saw", wave_test="triangle", n=65536, channels=1),
    dict(kind="synth", seed=41, channels=2, n=60000, gaps=[(20000, 9000), (40000, 3000)]),   # silence in mid-stream
]


@pytest.mark.parametrize("case", ADV_STAGE_CASES,
                         ids=["mono", "stereo-ragged", "lead-silence", "quiet", "identical", "saw-triangle", "mid-gaps"])
def test_advanced_mov_values_match_oracle_per_block_and_frame(gpu, case, fir_mode):
    """The advanced version's MOV layer block by block and frame by frame, before the accumulators: RmsModDiffA and
    its weight (movs.c:205-254 with the RMS normalisation of :243-244), the noise loudness and the missing-components
    term of RmsNoiseLoudAsymA (movs.c:551-577), AvgLinDistA (movs.c:679-706) from the filter-bank back end -- every
    block, also those the gates of gstpeaq.c:988,996-997 (125 / 13) keep from the accumulators -- the total loudness of
    the blocks before the loudness gate opens, and SegmentalNMRB of every FFT frame (movs.c:1010-1020) from the
    55-band back end.  End to end (the goldens) a wrong MOV shows; this says which, where and in which block.
    HIP filter bank / HIP 55-band front end -> HIP back ends in their debug instantiation, against the oracle's
    ear models -> pattern layer -> MOV functions on the same samples; both arithmetics of the filter bank."""
    import torch
    import gstpeaq_amd
    import gstpeaq_amd.capi as capi
    ref, test = case_defs.make_inputs(case)
    n = min(len(ref), len(test))
    ref, test = ref[:n], test[:n]
    ch = ref.shape[1]
    n_frames = (n - 2048) // 1024 + 2           # all full frames + the zero-padded flush frame
    n_blocks = -(-n // 192)                     # all full blocks + the zero-padded flush block
    ctx = gpu.ctx()
    d_ref, d_test = torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda()
    fb = gstpeaq_amd.debug_filterbank(ctx, d_ref, d_test, n_blocks, 320)
    ff = gstpeaq_amd.debug_frontend(ctx, 55, d_ref, d_test, n_frames)
    blk, frm, res = capi.debug_backend_advanced(ctx, fb, ff)
    eblk, efrm = orc.mov_trace_advanced(ref, test, n_blocks, n_frames)
    loose = gpu.mode() != "default"             # the opt-in engine's filter bank is 1e-4 per block (tests/gpu_common.py)
    for name in orc.MOV_TRACE_ADV_BLOCK:
        got, want = blk[name], eblk[name]
        if name.startswith("loudness"):
            # evaluated while the gate is closed only (gstpeaq.c:841-845): compare where the oracle evaluated it
            m = want != 0.
            np.testing.assert_allclose(got[m], want[m], rtol=1e-4 if loose else 1e-9, err_msg=name)
            continue
        # the modulation difference inherits the modulation's cancellation (|L - L_prev| of a stationary signal:
        # 1e-5 of L), the noise loudness terms the partial loudness's (1 + x)^0.23 - 1 of small x
        rtol = 1e-6 if name in ("rmsmoddiff", "noiseloud", "missing", "lindist") else 1e-9
        if loose:
            rtol = 2e-3
        if case.get("identical") and name in ("rmsmoddiff", "noiseloud", "missing"):
            assert np.all(np.abs(got) < 1e-9) and np.all(np.abs(want) < 1e-9), name   # identical signals: exactly nothing
            continue                                 # (AvgLinDist compares the adapted with the unadapted pattern: not 0)
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-12, err_msg=name)
    for name in orc.MOV_TRACE_ADV_FRAME:
        got, want = frm[name], efrm[name]
        if case.get("identical"):
            assert np.all(got[:, :ch] < -100) or np.all(np.isinf(got)) or np.allclose(got, want, rtol=1e-6, equal_nan=True), name
            continue
        # the noise spectrum's cancellation (Pr - 2 sqrt(Pr Pt) + Pt of nearly equal spectra): 1e-6
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9, err_msg=name)
    # and the result after the last block and frame is that of the whole pair
    exp = orc.run_pair(1, ref, test)
    assert res["frames"] == exp["frames"] == n_frames
    ok = ~np.isnan(exp["movs"][:5])
    assert np.array_equal(np.isnan(res["movs"][:5]), ~ok)
    np.testing.assert_allclose(res["movs"][:5][ok], exp["movs"][:5][ok], rtol=gpu.tol("movs"), atol=1e-9)
    if not np.isnan(exp["odg"]):
        assert abs(res["odg"] - exp["odg"]) < 1e-6
