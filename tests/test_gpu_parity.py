"""Parity of the HIP path (through the C ABI of libpeaq_amd.so) with the CPU
oracle and with the committed outputs of the real reference.  Needs an MI355X:
run with `-m gpu`.  Tolerances are stated per check; the north-star bar is
|dODG| <= 0.02, the FP64 device path is held to ~1e-6 and better."""
import numpy as np
import pytest

import cases as case_defs
import oracle_lib as orc
import synth_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    import gpu_common
    return gpu_common


def test_library_reports_version(gpu):
    import gstpeaq_amd
    assert b"gfx950" in gstpeaq_amd.load_library().peaq_version()


@pytest.mark.parametrize("channels", [1, 2])
def test_device_synth_matches_header(gpu, channels):
    """include/peaq_synth.h is bit-identical in gcc, numpy and HIP"""
    import gstpeaq_amd
    ref, test = gstpeaq_amd.synth_fill(gpu.ctx(), 3, 3, channels, 30000)
    for p in range(3):
        r, t = synth_np.pair(3 + p, channels, 30000)
        assert np.array_equal(ref[p].cpu().numpy(), r)
        assert np.array_equal(test[p].cpu().numpy(), t)


def oracle_records(bands, ref, test, n_frames):
    import ctypes as C
    L = orc.lib()
    ch = ref.shape[1]
    out = np.zeros((n_frames, ch, 576))
    L.orc_flat_frontend_records.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_float), C.c_size_t,
                                            C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    r = np.ascontiguousarray(ref, dtype=np.float32)
    t = np.ascontiguousarray(test, dtype=np.float32)
    L.orc_flat_frontend_records(bands, ch, 92.0, r.ctypes.data_as(C.POINTER(C.c_float)), len(r),
                                t.ctypes.data_as(C.POINTER(C.c_float)), len(t), n_frames,
                                out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


@pytest.mark.parametrize("bands", [109, 55])
@pytest.mark.parametrize("case", [
    dict(kind="synth", seed=5, channels=1, n=20000),
    dict(kind="synth", seed=6, channels=2, n=20000, test_trim=900),
    dict(kind="synth", seed=1, channels=2, n=30000),             # leading digital silence
    dict(kind="synth", seed=9, channels=2, n=20000, atten_shift=9),   # around the detector thresholds
    dict(kind="synth", seed=26, channels=2, n=12000, identical=1),
    dict(kind="ats", wave_ref="saw", wave_test="triangle", n=16384, channels=1),
    dict(kind="synth", seed=41, channels=2, n=30000, gaps=[(8000, 9000)]),   # digital silence in mid-stream
], ids=["mono", "stereo-ragged", "lead-silence", "quiet", "identical", "saw-triangle", "mid-gap"])
def test_frontend_records_match_oracle(gpu, bands, case):
    """stage-level: every field of the per-frame record (unsmeared excitation,
    loudness^0.3, noise in bands, bandwidths, EHS, flags, energies)"""
    import torch
    import gstpeaq_amd
    ref, test = case_defs.make_inputs(case)
    n = min(len(ref), len(test))
    n_frames = (n - 2048) // 1024 + 2           # all full frames + the flush frame
    got = gstpeaq_amd.debug_frontend(gpu.ctx(), bands, torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(),
                                     n_frames)
    exp = oracle_records(bands, ref, test, n_frames)
    # The 55-band kernel is the advanced version's FFT path: that version reads neither the test signal's
    # excitation nor the bandwidths (process_fft_block_advanced, gstpeaq.c:924-959), so the kernel does
    # not compute them and the record carries zeros there.
    fields = (("unsm_ref", 0), ("unsm_test", 112), ("loud_ref", 224), ("loud_test", 336))
    for name, lo in fields:
        if bands == 55 and name.endswith("_test"):
            assert not got[:, :, lo:lo + bands].any(), name
            continue
        np.testing.assert_allclose(got[:, :, lo:lo + bands], exp[:, :, lo:lo + bands], rtol=2e-10, atol=0,
                                   err_msg=name)
    # noise = Pr - 2 sqrt(Pr Pt) + Pt cancels heavily where the signals are close: the rounding of
    # the two DFT implementations (kissfft-like radix-2 in the oracle, 16x16x4 Stockham here) shows
    np.testing.assert_allclose(got[:, :, 448:448 + bands], exp[:, :, 448:448 + bands], rtol=1e-6, atol=0,
                               err_msg="noise")
    if bands == 109:
        assert np.array_equal(got[:, :, 560:562], exp[:, :, 560:562]), "bandwidths"
    else:
        assert not got[:, :, 560:562].any(), "bandwidths (not computed by the advanced version)"
    assert np.array_equal(got[:, :, 563:565], exp[:, :, 563:565]), "flags"
    # EHS: the oracle goes through 512-point FFTs like the reference, the kernel sums directly
    assert np.array_equal(np.isnan(got[:, :, 562]), np.isnan(exp[:, :, 562]))
    np.testing.assert_allclose(got[:, :, 562], exp[:, :, 562], rtol=1e-7, atol=1e-13, err_msg="ehs")
    np.testing.assert_allclose(got[:, :, 565:567], exp[:, :, 565:567], rtol=1e-12, atol=0, err_msg="energies")


@pytest.mark.parametrize("bands", [109, 55])
def test_energy_threshold_at_the_boundary(gpu, bands):
    """fftearmodel.c:508-514: a frame counts for the error harmonic structure when the energy of its second half --
    squares formed in SINGLE precision, summed in double -- reaches 8000 / 32768^2 = 125 * 2^-24.  The kernel sums the
    squares as a tree, the reference in order; frames whose partial sums are all exact in double make the order
    irrelevant and put the threshold itself to the test: 125 samples of 2^-12 give exactly the threshold (flag set);
    one of them an ulp of single precision smaller, 2^-46 below it (flag clear); an ulp larger, 2^-46 above (set).
    (A frame that sits on the threshold to the last bit of a sum that is NOT exact can still come out differently
    from the reference's loop: DESIGN.md 4, "not bit-exact by design".)"""
    import torch
    import gstpeaq_amd
    n_frames = 5
    n = 1024 * (n_frames + 1)
    base = np.float32(2.0 ** -12)
    variants = {
        "exact": (base, True),
        "below": (np.float32(base * np.float32(1 - 2.0 ** -23)), False),
        "above": (np.float32(base * np.float32(1 + 2.0 ** -23)), True),
    }
    for name, (special, expect) in variants.items():
        sig = np.zeros((n, 1), np.float32)
        # frame 2 covers samples 2048 .. 4095; its second half is 3072 .. 4095: 125 scattered samples
        idx = 3072 + (np.arange(125) * 8 + 3)
        sig[idx, 0] = base
        sig[idx[77], 0] = special
        ref = torch.from_numpy(sig).cuda()
        got = gstpeaq_amd.debug_frontend(gpu.ctx(), bands, ref, ref, n_frames)
        exp = oracle_records(bands, sig, sig, n_frames)
        assert np.array_equal(got[:, :, 563:565], exp[:, :, 563:565]), name
        # bit 1 of the flag words = energy threshold reached (ref, test)
        flags = got[:, 0, 563].astype(int)
        assert bool(flags[2] & 2) == expect, (name, flags)
        # the frames before and after hold fewer of the samples: below the threshold
        assert not (flags[0] & 2) and not (flags[4] & 2), (name, flags)


@pytest.mark.parametrize("bpl", [320, 7])
@pytest.mark.parametrize("case", [
    dict(kind="synth", seed=5, channels=1, n=30000),
    dict(kind="synth", seed=6, channels=2, n=25000, test_trim=1000),
    dict(kind="ats", wave_ref="saw", wave_test="triangle", n=20000, channels=1),
], ids=["mono", "stereo-ragged", "saw-triangle"])
def test_filterbank_records_match_oracle(gpu, case, bpl, fir_mode):
    """stage-level, advanced: unsmeared + forward-masked excitation of the 40-band filter bank per
    192-sample block (fbearmodel.c:276-396) and the block boundary flag, for whole-chunk launches
    and for launches of 7 blocks (state, delay line and histories carried between launches)"""
    import torch
    import gstpeaq_amd
    ref, test = case_defs.make_inputs(case)
    ch = ref.shape[1]
    n = min(len(ref), len(test))
    n_blocks = n // 192
    got = gstpeaq_amd.debug_filterbank(gpu.ctx(), torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(),
                                       n_blocks, bpl)
    for c in range(ch):
        for name, sig, lo in (("ref", ref, 0), ("test", test, 40)):
            exp = orc.fbear(np.ascontiguousarray(sig[:, c]), n_blocks)
            np.testing.assert_allclose(got[:, c, lo:lo + 40], exp["unsmeared"], rtol=gpu.tol("blocks"),
                                       err_msg=f"unsmeared {name} ch{c}")
            np.testing.assert_allclose(got[:, c, 80 + lo:120 + lo], exp["excitation"], rtol=gpu.tol("blocks"),
                                       err_msg=f"excitation {name} ch{c}")


@pytest.mark.parametrize("channels", [1, 2])
def test_batch_basic_matches_reference_goldens(gpu, channels):
    """every basic-mode end-to-end case of tests/golden/ref_e2e.json (outputs of the
    REAL reference element), ragged lengths in one batch call"""
    recs = [r for r in gpu.e2e_records(0) if r["case"]["channels"] == channels]
    inputs = [case_defs.make_inputs(r["case"]) for r in recs]
    got = gpu.run_batch(inputs, 0, channels)
    worst = 0.0
    for g, rec in zip(got, recs):
        gpu.compare_result(g, rec, rtol=1e-7, atol=1e-9, odg_atol=1e-6)
        if not np.isnan(float(rec["odg"])):
            worst = max(worst, abs(g["odg"] - float(rec["odg"])))
    print(f"max |dODG| vs reference over {len(recs)} cases: {worst:.3e}")


@pytest.mark.parametrize("channels", [1, 2])
def test_batch_advanced_matches_reference_goldens(gpu, channels, fir_mode):
    """advanced version (55-band FFT model + 40-band filter bank, 5 MOVs) against the outputs
    of the real reference element; pinned by nothing else in the reference but its
    conformance table"""
    recs = [r for r in gpu.e2e_records(1) if r["case"]["channels"] == channels]
    inputs = [case_defs.make_inputs(r["case"]) for r in recs]
    got = gpu.run_batch(inputs, 1, channels)
    worst = 0.0
    for g, rec in zip(got, recs):
        assert g["fb_blocks"] == rec["fb_frames"], rec["case"]["name"]
        gpu.compare_result(g, rec, rtol=gpu.tol("movs"), atol=1e-9, odg_atol=gpu.tol("odg"))
        if not np.isnan(float(rec["odg"])):
            worst = max(worst, abs(g["odg"] - float(rec["odg"])))
    print(f"max |dODG| vs reference over {len(recs)} advanced cases: {worst:.3e}")


def test_reference_odg_regression_strings_on_gpu(gpu):
    # runtest-1.0.sh:18,28
    a = case_defs.make_inputs(dict(kind="ats", wave_ref="sine", wave_test="sine", n=131072, channels=1))
    b = case_defs.make_inputs(dict(kind="ats", wave_ref="saw", wave_test="triangle", n=131072, channels=1))
    got = gpu.run_batch([a, b], 0, 1)
    assert "%.3f" % got[0]["odg"] == "0.171"
    assert "%.3f" % got[1]["odg"] == "-2.007"


def test_batch_matches_oracle_on_seeded_pairs(gpu):
    """64 seeded stereo pairs generated ON the device vs the CPU oracle on the same bits"""
    import gstpeaq_amd
    n, seed0, pairs = 96000, 1000, 64
    ref, test = gstpeaq_amd.synth_fill(gpu.ctx(), seed0, pairs, 2, n)
    got = gstpeaq_amd.batch_run(gpu.ctx(), 0, ref, test)
    worst = 0.0
    for p in range(0, pairs, 4):                      # the oracle needs ~0.1 s per pair
        r, t = synth_np.pair(seed0 + p, 2, n)
        e = orc.run_pair(0, r, t)
        np.testing.assert_allclose(got[p]["movs"], e["movs"], rtol=1e-7, atol=1e-9)
        worst = max(worst, abs(got[p]["odg"] - e["odg"]))
    assert worst < 1e-6
    print(f"max |dODG| vs oracle: {worst:.3e}")


@pytest.mark.parametrize("advanced", [0, 1])
def test_session_streaming_equals_batch(gpu, advanced, fir_mode):
    """pad_chain delivers arbitrary buffer sizes on either pad (gstpeaq.c:614-661)"""
    import gstpeaq_amd
    case = dict(kind="synth", seed=3, channels=2, n=150000, test_trim=1234)
    ref, test = case_defs.make_inputs(case)
    whole = gpu.run_batch([(ref, test)], advanced, 2)[0]
    s = gstpeaq_amd.Session(gpu.ctx(), advanced, 2)
    rng = np.random.default_rng(1)
    pr = pt = 0
    mid = None
    while pr < len(ref) or pt < len(test):
        if pr < len(ref):
            k = int(rng.integers(1, 9000))
            s.push_ref(ref[pr:pr + k])
            pr += k
        if pt < len(test):
            k = int(rng.integers(1, 9000))
            s.push_test(test[pt:pt + k])
            pt += k
        if mid is None and pr > 70000:
            mid = s.results()                          # results are readable mid-stream
    s.flush()
    got = s.results()
    assert mid["frames"] > 0 and mid["frames"] < got["frames"]
    assert got["frames"] == whole["frames"] and got["fb_blocks"] == whole["fb_blocks"]
    # same kernels, same per-frame arithmetic (the basic version: bit for bit); the filter bank walks the
    # stream in different tile alignments, which only moves the rounding of its one recurrence along the
    # stream (the slope filter, FP64 in either mode) and of the FP64 sums of partial tiles
    np.testing.assert_allclose(got["movs"], whole["movs"], rtol=gpu.tol("chunks") if advanced else 0, atol=0)
    assert abs(got["odg"] - whole["odg"]) <= (gpu.tol("chunks") if advanced else 0)
    assert s.results()["odg"] == got["odg"]            # idempotent
    s.close()


def test_session_fed_one_filterbank_block_at_a_time(gpu, fir_mode):
    """Advanced version, a launch per 192-sample block: every tile of the filter bank is shorter than the filters'
    whole-block counts, so the block-sum form (FP64 engine) keeps most of its history from launch to launch
    (peaq_fb.hip, bs_pair: the short-tile path) and re-anchors its running sums every block; the split-FP16
    engine's window head spans eight launches.  Same results as the batch path."""
    import gstpeaq_amd
    ref, test = case_defs.make_inputs(dict(kind="synth", seed=11, channels=1, n=40000))
    whole = gpu.run_batch([(ref, test)], 1, 1)[0]
    s = gstpeaq_amd.Session(gpu.ctx(), 1, 1)
    for lo in range(0, len(ref), 192):
        s.push_ref(ref[lo:lo + 192])
        s.push_test(test[lo:lo + 192])
    s.flush()
    got = s.results()
    s.close()
    assert got["fb_blocks"] == whole["fb_blocks"]
    np.testing.assert_allclose(got["movs"], whole["movs"], rtol=gpu.tol("chunks"), atol=0)
    assert abs(got["odg"] - whole["odg"]) <= gpu.tol("chunks")


@pytest.mark.parametrize("advanced", [0, 1])
def test_empty_and_tiny_pairs_inside_a_batch(gpu, advanced, fir_mode):
    """an element that reaches EOS without data reports NaN (empty accumulators, movaccum.c:438-481);
    one that got a single sample runs exactly the flush frame / block (gstpeaq.c:716-745)"""
    normal = case_defs.make_inputs(dict(kind="synth", seed=5, channels=2, n=50000))
    empty = (np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    one = (normal[0][:1].copy(), normal[1][:1].copy())
    ref_only = (normal[0][:3000].copy(), np.zeros((0, 2), np.float32))     # the test pad never delivered
    got = gpu.run_batch([normal, empty, one, ref_only], advanced, 2)
    alone = gpu.run_batch([normal], advanced, 2)[0]
    np.testing.assert_allclose(got[0]["movs"], alone["movs"], rtol=gpu.tol("chunks") if advanced else 0, atol=0)
    assert got[1]["frames"] == 0 and got[1]["fb_blocks"] == 0 and np.isnan(got[1]["odg"])
    for g, (r, t) in ((got[1], empty), (got[2], one), (got[3], ref_only)):
        e = orc.run_pair(advanced, r, t)
        assert g["frames"] == e["frames"]                # one flush frame if anything was delivered
        assert np.array_equal(np.isnan(g["movs"]), np.isnan(e["movs"][:len(g["movs"])]))
        fin = ~np.isnan(g["movs"])
        np.testing.assert_allclose(g["movs"][fin], e["movs"][:len(g["movs"])][fin], rtol=gpu.tol("movs", advanced), atol=1e-9)


def _threshold_outro_pair(p, n, quiet_blocks, reached):
    """A stereo pair whose last `quiet_blocks` 192-sample blocks are far below the data-boundary threshold, except for
    one block -- a different one, at a different position and in a different channel for every p -- that is silent but
    for a run of five samples of 40 / 32768: five of them sum to 200 / 32768 exactly, the threshold (`reached`), or of
    one float ulp less.  Blocks below the threshold are accumulated tentatively and count only if a later one reaches
    it (gstpeaq.c:971-979, movaccum.c:305-340), so that run decides how much of the quiet end enters the averages."""
    a = np.float32(40.0 / 32768)
    r, t = synth_np.pair(9000 + p, 2, n)
    g = np.random.default_rng(77 + p)
    q = quiet_blocks * 192
    r[n - q:] = (g.uniform(-0.3, 0.3, (q, 2)) * a).astype(np.float32)
    t[n - q:] = (r[n - q:] + g.uniform(-0.05, 0.05, (q, 2)) * a).astype(np.float32)
    blk, at, ch = quiet_blocks - 12 + (5 * p) % 11, 5 + (37 * p) % 180, p & 1
    s = n - q + blk * 192
    r[s: s + 192] = 0
    t[s: s + 192] = 0
    r[s + at: s + at + 5, ch] = a if reached else np.float32(a * np.float32(1 - 2.0 ** -23))
    return r, t


def test_advanced_data_boundary_at_the_threshold_in_full_waves(gpu, fir_mode):
    """The boundary detector of the 192-sample blocks (gstpeaq.c:1081-1099) sits in the high-pass walk, whose
    straight-line blocks need a full wave of 64 signals -- the single-pair stage tests above never give it one.  32
    stereo pairs put the threshold itself to the test (see _threshold_outro_pair), every other one an ulp below it."""
    n, pairs, quiet = 192 * 150, 32, 40
    inputs = [_threshold_outro_pair(p, n, quiet, reached=p % 2 == 0) for p in range(pairs)]
    got = gpu.run_batch(inputs, 1, 2)
    for p in range(pairs):
        e = orc.run_pair(1, *inputs[p])["movs"][:5]
        g = got[p]["movs"][:5]
        assert np.array_equal(np.isnan(g), np.isnan(e)), (p, g, e)
        fin = ~np.isnan(e)
        np.testing.assert_allclose(g[fin], e[fin], rtol=gpu.tol("movs"), atol=1e-9, err_msg=f"pair {p}")
    # the test has teeth: the ulp decides
    e0 = orc.run_pair(1, *_threshold_outro_pair(0, n, quiet, True))["movs"][:5]
    e1 = orc.run_pair(1, *_threshold_outro_pair(0, n, quiet, False))["movs"][:5]
    assert np.isfinite(e0).all() and not np.allclose(e0, e1, rtol=1e-3, atol=0, equal_nan=True)


@pytest.mark.parametrize("advanced", [0, 1])
def test_playback_level_matches_reference_goldens(gpu, advanced, fir_mode):
    """playback_level 60 .. 130 dB SPL through batch run, session and broker vs the real element"""
    import json
    import torch
    import gstpeaq_amd
    recs = [r for r in json.loads((gpu.GOLD / "ref_e2e_level.json").read_text()) if r["case"]["advanced"] == advanced]
    assert len(recs) == 4
    for rec in recs:
        case = rec["case"]
        ref, test = case_defs.make_inputs(case)
        got = gstpeaq_amd.batch_run(gpu.ctx(), advanced, torch.from_numpy(ref[None]).cuda(),
                                    torch.from_numpy(test[None]).cuda(), playback_level=case["level"])[0]
        gpu.compare_result(got, rec, rtol=gpu.tol("movs", advanced), atol=1e-9, odg_atol=gpu.tol("odg", advanced))
        s = gstpeaq_amd.Session(gpu.ctx(), advanced, case["channels"], playback_level=case["level"])
        s.push_ref(ref)
        s.push_test(test)
        s.flush()
        gpu.compare_result(s.results(), rec, rtol=gpu.tol("movs", advanced), atol=1e-9, odg_atol=gpu.tol("odg", advanced))
        s.close()
        b = gstpeaq_amd.Broker(gpu.ctx(), case["channels"], 2, playback_level=case["level"], advanced=advanced)
        sid = b.open()
        b.push(sid, 0, ref)
        b.push(sid, 1, test)
        b.flush(sid)
        gpu.compare_result(b.results(sid), rec, rtol=gpu.tol("movs", advanced), atol=1e-9, odg_atol=gpu.tol("odg", advanced))
        b.close()
    with pytest.raises(gstpeaq_amd.PeaqError):                     # property range 0..130 (gstpeaq.c:275-281)
        gstpeaq_amd.Session(gpu.ctx(), advanced, 2, playback_level=131.0)


def test_run_pair_from_host_memory_equals_the_goldens(gpu, fir_mode):
    """peaq_run_pair (upload + one-pair batch + result): ragged, sub-frame, silent and empty pairs included"""
    import gstpeaq_amd
    for adv in (0, 1):
        for rec in gpu.e2e_records(adv)[:40]:
            ref, test = case_defs.make_inputs(rec["case"])
            got = gstpeaq_amd.run_pair(gpu.ctx(), adv, ref, test)
            gpu.compare_result(got, rec, rtol=gpu.tol("movs", adv), atol=1e-9, odg_atol=gpu.tol("odg", adv))
    empty = np.zeros((0, 2), dtype=np.float32)
    got = gstpeaq_amd.run_pair(gpu.ctx(), 0, empty, empty)
    assert got["frames"] == 0 and np.isnan(got["odg"])


def test_half_hour_mono_pair_is_cut_into_exact_launches(gpu):
    """A long mono file is ONE pair with tens of thousands of frames.  The front end takes a work item apart
    with a 32-bit reciprocal of the frames per launch (peaq_frontend.hip `decode`), exact only up to
    max_frames_per_launch(): peaq_run_pair must cut the stream there.  70 000 frames alone == the same pair
    beside a second one (which forces 64-frame launches), bit for bit, and its last frame is a real one."""
    import torch
    import gstpeaq_amd
    n = 70000 * 1024 + 1024
    frames = gstpeaq_amd.load_library().peaq_frame_count(n, n, 0)     # 70 000 whole frames + the flush frame
    assert frames == 70001
    r1, t1 = synth_np.pair(21, 1, 480000)
    reps = -(-n // len(r1))
    ref = np.tile(r1, (reps, 1))[:n].copy()
    test = np.tile(t1, (reps, 1))[:n].copy()
    test[-3000:] *= 0.5                                 # the tail differs: a dropped last frame would show
    alone = gstpeaq_amd.run_pair(gpu.ctx(), 0, ref, test)
    assert alone["frames"] == frames
    d_ref = torch.from_numpy(np.stack([ref, ref])).cuda()
    d_test = torch.from_numpy(np.stack([test, ref])).cuda()
    both = gstpeaq_amd.batch_run(gpu.ctx(), 0, d_ref, d_test)
    assert both[0]["frames"] == frames
    np.testing.assert_array_equal(alone["movs"], both[0]["movs"])
    assert alone["odg"] == both[0]["odg"] and alone["totalsnr"] == both[0]["totalsnr"]
    # and the stage entry point beyond one launch: the record of the very last frame is written
    recs = gstpeaq_amd.debug_frontend(gpu.ctx(), 109, torch.from_numpy(ref[: 66000 * 1024 + 1024]).cuda(),
                                      torch.from_numpy(test[: 66000 * 1024 + 1024]).cuda(), 66000)
    assert recs[-1, 0, :109].min() > 0. and recs[65535, 0, :109].min() > 0. and recs[65536, 0, :109].min() > 0.
