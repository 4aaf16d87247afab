"""-m gpu: input classes beyond a clean seeded pair (tests/cases.py:input_class_cases) -- digital silence in
MID-stream (the accumulators' NORMAL -> TENTATIVE -> NORMAL recovery, movaccum.c:317-352, driven by the data-boundary
detector on the reference signal, gstpeaq.c:861,1081-1099), hard clipping at full scale, DC offsets, an inverted test
signal, channels 40 dB apart (binaural maxima, movs.c:1224-1276) -- through all three ways into the HIP path
(batch: test_gpu_parity.py's golden tests pick the records up by themselves; here: one session per stream fed in
arbitrary buffer sizes, and all streams of a channel count as concurrent sessions of one broker), each against the
outputs of the REAL reference element (tests/golden/ref_e2e.json, tools/make_golden.py)."""
import numpy as np
import pytest

import cases as case_defs
import gpu_common as gpu

pytestmark = pytest.mark.gpu

NAMES = {c["name"] for c in case_defs.input_class_cases()}


def _records(advanced, channels=None):
    recs = [r for r in gpu.e2e_records(int(advanced)) if r["case"]["name"] in NAMES]
    if channels is not None:
        recs = [r for r in recs if r["case"]["channels"] == channels]
    assert recs
    return recs


def _tols(advanced):
    return dict(rtol=gpu.tol("movs") if advanced else 1e-7, atol=1e-9, odg_atol=gpu.tol("odg") if advanced else 1e-6)


def test_the_goldens_hold_every_class():
    for adv in (0, 1):
        assert {r["case"]["name"] for r in _records(adv)} == NAMES
    # the gaps are long enough to silence whole frames: the reference itself must have seen fewer loud frames
    # than a clean pair (otherwise the case would not reach the TENTATIVE state at all)
    rec = next(r for r in _records(0) if r["case"]["name"] == "gap3_stereo")
    ref, _ = case_defs.make_inputs(rec["case"])
    quiet = [f for f in range(rec["frames"] - 1)
             if not np.abs(ref[f * 1024:f * 1024 + 2048]).any()]
    assert len(quiet) >= 6 and quiet[0] > 10 and quiet[-1] < rec["frames"] - 10, quiet


@pytest.mark.parametrize("advanced", [0, 1])
def test_sessions_fed_in_arbitrary_buffers_match_the_reference(advanced, fir_mode):
    import gstpeaq_amd
    rng = np.random.default_rng(11)
    for rec in _records(advanced):
        case = rec["case"]
        ref, test = case_defs.make_inputs(case)
        s = gstpeaq_amd.Session(gpu.ctx(), advanced, case["channels"])
        pr = pt = 0
        while pr < len(ref) or pt < len(test):
            if pr < len(ref):
                k = int(rng.integers(1, 9000))
                s.push_ref(ref[pr:pr + k])
                pr += k
            if pt < len(test):
                k = int(rng.integers(1, 9000))
                s.push_test(test[pt:pt + k])
                pt += k
        s.flush()
        got = s.results()
        s.close()
        if advanced:
            assert got["fb_blocks"] == rec["fb_frames"], case["name"]
        try:
            gpu.compare_result(got, rec, **_tols(advanced))
        except AssertionError as e:
            raise AssertionError(f"{case['name']} (advanced={advanced}): {e}") from e


@pytest.mark.parametrize("advanced", [0, 1])
@pytest.mark.parametrize("channels", [1, 2])
def test_concurrent_broker_sessions_match_the_reference(channels, advanced, fir_mode):
    import gstpeaq_amd
    recs = _records(advanced, channels)
    streams = [case_defs.make_inputs(r["case"]) for r in recs]
    b = gstpeaq_amd.Broker(gpu.ctx(), channels, max_sessions=16, advanced=bool(advanced))
    sids = [b.open() for _ in recs]
    rng = np.random.default_rng(3)
    pos = [[0, 0] for _ in recs]
    live = set(range(len(recs)))
    rounds = 0
    while live:
        for i in list(live):
            ref, test = streams[i]
            for pad, sig in ((0, ref), (1, test)):
                if pos[i][pad] < len(sig):
                    k = int(rng.integers(1, 5000))
                    b.push(sids[i], pad, sig[pos[i][pad]:pos[i][pad] + k])
                    pos[i][pad] += k
            if pos[i][0] >= len(ref) and pos[i][1] >= len(test):
                b.flush(sids[i])
                live.discard(i)
        rounds += 1
        if rounds % 3 == 0:
            b.tick()                                  # mid-gap ticks included: a launch may hold only silent frames
    for i, rec in enumerate(recs):
        got = b.results(sids[i])
        try:
            gpu.compare_result(got, rec, **_tols(advanced))
        except AssertionError as e:
            raise AssertionError(f"{rec['case']['name']} (advanced={advanced}): {e}") from e
    assert b.stats()["max_active"] == len(recs)
    b.close()
