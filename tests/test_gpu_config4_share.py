"""One GPU's share of BASELINE.json configs[3] at its STATED size: 262 144 pairs over 8 GPUs =
32 768 ten-second stereo pairs per GPU = 252 GB of input, consumed in waves of 4096 pairs that are
generated on the device into the same two buffers (gstpeaq_amd.parallel.run_waves -- the code
bench.py --gpus 8 runs on every rank).  Checks, on the whole share:
  * framing: every pair reports 468 frame-pairs, no NaN result;
  * wave independence: pairs picked from every wave are bit-identical to the same seeds run as
    one small batch (results do not depend on the wave, the position in it, or its neighbours);
  * parity: those pairs against the CPU oracle (MOVs rtol 1e-7, |dODG| < 1e-6).
Needs an MI355X (`-m gpu`)."""
import numpy as np
import pytest

import oracle_lib as orc
import synth_np

pytestmark = pytest.mark.gpu

SHARE = 32768            # pairs per GPU at configs[3]
WAVE = 4096
N_SAMPLES = 480000
RANK = 5                 # the share of rank 5 of 8: pairs [163840, 196608)


def test_one_gpu_share_of_262144_pairs_in_waves():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    import gpu_common
    import gstpeaq_amd
    from gstpeaq_amd import parallel
    ctx = gpu_common.ctx()
    lo, hi = parallel.shard(8 * SHARE, RANK, 8)
    assert (lo, hi) == (RANK * SHARE, (RANK + 1) * SHARE)
    seed0 = 1 + lo
    ref, test = gstpeaq_amd.synth_fill(ctx, seed0, WAVE, 2, N_SAMPLES)
    results = torch.full((SHARE, 16), float("nan"), dtype=torch.float64, device="cuda")
    timed = parallel.run_waves(ctx, 0, seed0, SHARE, WAVE, ref, test, results)
    rows = results.cpu().numpy()
    assert timed > 0
    assert (rows[:, 14] == 468).all(), "every 10 s pair is 467 full frames + the flush frame"
    assert not np.isnan(rows[:, :13]).any()
    assert len(parallel.waves(SHARE, WAVE)) == 8

    # two pairs out of every wave (every 2048th pair, off the wave boundaries), re-run as ONE small batch
    picks = np.arange(5, SHARE, 2048)
    assert len(picks) == 16
    small_ref = torch.empty((len(picks), N_SAMPLES, 2), dtype=torch.float32, device="cuda")
    small_test = torch.empty_like(small_ref)
    for i, p in enumerate(picks):
        gstpeaq_amd.synth_fill(ctx, seed0 + int(p), 1, 2, N_SAMPLES, out=(small_ref[i:i + 1], small_test[i:i + 1]))
    small = gstpeaq_amd.batch_run(ctx, 0, small_ref, small_test, sync=False)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(small.cpu().numpy(), rows[picks])

    for i, p in enumerate(picks[::2]):          # 8 of them against the oracle (~0.3 s each)
        r, t = synth_np.pair(seed0 + int(p), 2, N_SAMPLES)
        assert np.array_equal(small_ref[2 * i].cpu().numpy(), r)
        exp = orc.run_pair(0, r, t)
        np.testing.assert_allclose(rows[p, :11], exp["movs"], rtol=1e-7, atol=1e-9)
        assert abs(rows[p, 12] - exp["odg"]) < 1e-6 and abs(rows[p, 11] - exp["di"]) < 1e-6
