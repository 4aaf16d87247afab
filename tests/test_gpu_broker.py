"""-m gpu: the live-pipeline broker (BASELINE.json configs[5]).

Many sessions -- each the stream of one hosted `peaq` element, fed in arbitrary buffer sizes on
either pad as pad_chain does (gstpeaq.c:613-661) -- share ONE batched launch per tick.  Every
session must end with exactly the result a whole-pair batch run (and so the reference element,
see test_gpu_parity.py) gives for its stream.
"""
import threading

import numpy as np
import pytest

import cases as case_defs
import gpu_common as gpu

pytestmark = pytest.mark.gpu


def _streams(n, channels):
    out = []
    for i in range(n):
        case = dict(kind="synth", seed=100 + i, channels=channels, n=60000 + 7919 * (i % 11),
                    test_trim=(0, 1234, 5, 2047)[i % 4] if i % 3 else 0)
        out.append(case_defs.make_inputs(case))
    return out


def _same(got, exp, where, rtol=1e-12):
    # advanced: the filter bank walks the stream in other tile alignments and adds its partial sums
    # with LDS atomics, which only moves FP64 rounding -- in BOTH arithmetics of the bank (the reduced-precision
    # engine keeps its one recurrence along the stream, the slope filter, in FP64 for exactly this reason)
    assert got["frames"] == exp["frames"], where
    np.testing.assert_allclose(got["movs"], exp["movs"], rtol=rtol, atol=0, equal_nan=True, err_msg=str(where))
    for k, tol in (("di", 1e3 * rtol), ("odg", 1e3 * rtol), ("totalsnr", 1e-9)):
        if np.isnan(exp[k]):
            assert np.isnan(got[k]), (where, k)
        else:
            assert abs(got[k] - exp[k]) <= tol, (where, k, got[k], exp[k])


def _feed(broker, sid, ref, test, rng, tick_every=None):
    pr = pt = 0
    k_calls = 0
    while pr < len(ref) or pt < len(test):
        if pr < len(ref):
            k = int(rng.integers(1, 6000))
            broker.push(sid, 0, ref[pr:pr + k])
            pr += k
        if pt < len(test):
            k = int(rng.integers(1, 6000))
            broker.push(sid, 1, test[pt:pt + k])
            pt += k
        k_calls += 1
        if tick_every and k_calls % tick_every == 0:
            broker.tick()
    broker.flush(sid)


@pytest.mark.parametrize("advanced", [False, True], ids=["basic", "advanced"])
@pytest.mark.parametrize("channels", [1, 2])
def test_broker_sessions_equal_batch(channels, advanced, fir_mode):
    import gstpeaq_amd
    n = 37
    streams = _streams(n, channels)
    whole = gpu.run_batch(streams, advanced, channels)
    b = gstpeaq_amd.Broker(gpu.ctx(), channels, max_sessions=64, advanced=advanced)
    sids = [b.open() for _ in range(n)]
    assert sorted(sids) == list(range(n))
    rng = np.random.default_rng(5)
    # interleave the sessions: round-robin chunks, a tick now and then
    pos = [[0, 0] for _ in range(n)]
    live = set(range(n))
    rounds = 0
    while live:
        for i in list(live):
            ref, test = streams[i]
            for pad, sig in ((0, ref), (1, test)):
                if pos[i][pad] < len(sig):
                    k = int(rng.integers(1, 7000))
                    b.push(sids[i], pad, sig[pos[i][pad]:pos[i][pad] + k])
                    pos[i][pad] += k
            if pos[i][0] >= len(ref) and pos[i][1] >= len(test):
                b.flush(sids[i])
                live.discard(i)
        rounds += 1
        if rounds % 2 == 0:
            b.tick()
    for i in range(n):
        _same(b.results(sids[i]), whole[i], i, rtol=gpu.tol("chunks") if advanced else 1e-12)
        assert b.results(sids[i])["fb_blocks"] == whole[i]["fb_blocks"]
    st = b.stats()
    assert st["max_active"] > 1 and st["launches"] < sum(w["frames"] for w in whole)   # really batched
    assert st["frames"] == sum(w["frames"] for w in whole)
    b.close()


@pytest.mark.parametrize("advanced", [False, True], ids=["basic", "advanced"])
def test_broker_tick_thread_and_slot_reuse(advanced, fir_mode):
    import gstpeaq_amd
    channels = 2
    streams = _streams(12, channels)
    whole = gpu.run_batch(streams, advanced, channels)
    b = gstpeaq_amd.Broker(gpu.ctx(), channels, max_sessions=4, advanced=advanced)
    b.start(500)
    with pytest.raises(gstpeaq_amd.PeaqError):
        b.start(500)                                    # already running
    results = [None] * len(streams)
    errors = []

    def element(worker):                                # one thread = one pipeline's streaming thread
        try:
            rng = np.random.default_rng(worker)
            for i in range(worker, len(streams), 4):
                sid = b.open()
                _feed(b, sid, *streams[i], rng)
                results[i] = b.results(sid)
                b.close_session(sid)                    # the slot is reused by the next stream
        except Exception as e:                          # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=element, args=(w,)) for w in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    b.stop()
    assert not errors, errors
    for i, got in enumerate(results):
        _same(got, whole[i], i, rtol=gpu.tol("chunks") if advanced else 1e-12)
    assert b.stats()["worker_failed"] == 0
    b.close()


def test_broker_argument_and_state_errors():
    import gstpeaq_amd
    b = gstpeaq_amd.Broker(gpu.ctx(), 1, max_sessions=2)
    s0, s1 = b.open(), b.open()
    with pytest.raises(gstpeaq_amd.PeaqError, match="in use"):
        b.open()
    with pytest.raises(gstpeaq_amd.PeaqError):
        b.push(7, 0, np.zeros(10, np.float32))
    with pytest.raises(gstpeaq_amd.PeaqError):
        b.push(s0, 2, np.zeros(10, np.float32))
    b.close_session(s1)
    with pytest.raises(gstpeaq_amd.PeaqError, match="not open"):
        b.push(s1, 0, np.zeros(10, np.float32))
    # an empty session reports what an element without data reports: no frames, NaN averages
    r = b.results(s0)
    assert r["frames"] == 0
    assert b.tick() == 0
    with pytest.raises(gstpeaq_amd.PeaqError):
        gstpeaq_amd.Broker(gpu.ctx(), 3, max_sessions=2)
    with pytest.raises(gstpeaq_amd.PeaqError):
        gstpeaq_amd.Broker(gpu.ctx(), 1, max_sessions=0)
    b.close()


@pytest.mark.parametrize("advanced", [False, True], ids=["basic", "advanced"])
def test_one_failed_device_of_a_multi_device_broker_does_not_starve_the_others(advanced):
    """peaq_broker_create_multi over {0, 0} (two contexts, two sets of slots on the one GPU of the box) with the second
    device's share stopped as after a device error in one of its ticks (peaq_debug_broker_fail_shard): the tick still
    launches the other device's frames and returns the failed device's own message; sessions on the healthy device run
    to the batch path's results; every call on a session of the failed device reports THAT device's error."""
    import gstpeaq_amd
    channels, n = 2, 8
    streams = _streams(n, channels)
    whole = gpu.run_batch(streams, advanced, channels)
    b = gstpeaq_amd.Broker(gpu.ctx(), channels, max_sessions=8, advanced=advanced, devices=[0, 0])
    assert b.devices() == 2
    sids = [b.open() for _ in range(n)]
    shard_of = [s % 2 for s in sids]                    # session id = shard + 2 x the shard's own id (peaq_broker.hip)
    assert sorted(shard_of) == [0] * 4 + [1] * 4         # dealt out evenly
    rng = np.random.default_rng(9)
    half = [len(r) // 2 for r, _ in streams]
    for i, (ref, test) in enumerate(streams):            # first half of every stream, then a tick of both devices
        b.push(sids[i], 0, ref[:half[i]])
        b.push(sids[i], 1, test[:half[i]])
    assert b.tick() >= 1                                # (both devices launch what their sessions have ready)
    b.fail_shard(1, "injected: device 1 fell off the bus")
    with pytest.raises(gstpeaq_amd.PeaqError, match="device 1 fell off the bus"):
        b.tick()                                         # ... which has ticked device 0 all the same (nothing ready there: fine)
    for i, (ref, test) in enumerate(streams):
        if shard_of[i] == 1:
            with pytest.raises(gstpeaq_amd.PeaqError, match="device 1 fell off the bus"):
                b.push(sids[i], 0, ref[half[i]:])
            with pytest.raises(gstpeaq_amd.PeaqError, match="device 1 fell off the bus"):
                b.flush(sids[i])
            continue
        _feed_rest = [(0, ref[half[i]:]), (1, test[half[i]:])]
        for pad, sig in _feed_rest:
            pos = 0
            while pos < len(sig):
                k = int(rng.integers(1, 6000))
                b.push(sids[i], pad, sig[pos:pos + k])
                pos += k
        b.flush(sids[i])
    with pytest.raises(gstpeaq_amd.PeaqError, match="device 1 fell off the bus"):
        b.tick()                                         # the healthy device's flush frames go out in this very call
    for i in range(n):
        if shard_of[i] == 0:
            _same(b.results(sids[i]), whole[i], i, rtol=gpu.tol("chunks") if advanced else 1e-12)
    b.close()
