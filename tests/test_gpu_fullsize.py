"""-m gpu: BASELINE.json's FULL sizes (configs[1]/[2]: 4096 stereo pairs of 10 s on one GPU).

The oracle needs ~0.3 s per 10 s pair, so at this size parity is established through
size-independent properties plus a sparse exact check:
  * position independence: pair k of a batch depends only on its own samples -- the same seeded
    pair gives bit-identical results at a different batch index, in a different batch size (a
    different chunking of the frame axis) and next to different neighbours;
  * a planted pair with test == ref reads as the identical-signal case (NaN->0 EHS, ADB -0.5...)
    whatever surrounds it;
  * frame / block counts of every pair equal the reference element's framing arithmetic;
  * every 512th pair equals the CPU oracle on the same bits.
"""
import numpy as np
import pytest

import gpu_common as gpu
import oracle_lib as orc
import synth_np

pytestmark = pytest.mark.gpu

PAIRS, SECONDS, CH = 4096, 10.0, 2
N = int(SECONDS * 48000)


def _run(advanced, seed0, pairs, plant_identical=None):
    import torch
    import gstpeaq_amd
    ctx = gpu.ctx()
    ref, test = gstpeaq_amd.synth_fill(ctx, seed0, pairs, CH, N)
    if plant_identical is not None:
        test[plant_identical].copy_(ref[plant_identical])
    res = gstpeaq_amd.batch_run(ctx, advanced, ref, test, sync=False)
    torch.cuda.synchronize()
    out = res.cpu().numpy()
    del ref, test
    torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize("advanced", [0, 1], ids=["basic", "advanced"])
def test_full_batch_properties(advanced, fir_mode):
    nm = 5 if advanced else 11
    full = _run(advanced, 1, PAIRS, plant_identical=777)
    # framing: 10 s -> 467 whole frames + the flush frame; 2500 filter-bank blocks (gstpeaq.c:596-611,716-745)
    assert np.all(full[:, 14] == 468)
    assert np.all(full[:, 15] == (2500 if advanced else 0))
    ok = np.arange(PAIRS) != 777
    assert not np.isnan(full[ok][:, :nm]).any() and not np.isnan(full[ok][:, 11:13]).any()
    assert np.all(full[ok][:, 12] <= 0.3) and np.all(full[ok][:, 12] >= -3.98)       # ODG range of the MLP

    # position independence: seeds 2049..4096 computed again as pairs 0..2047 of a 2048-pair batch
    # (other batch size -> other chunking of the frames, other neighbours, other workgroup ids)
    shifted = _run(advanced, 2049, 2048)
    a, b = full[2048:], shifted
    if advanced and gpu.mode() != "default":
        # the reduced-precision engine adds its waves' partial sums with LDS atomics: the order, hence the last
        # bits, may differ from run to run
        np.testing.assert_allclose(a[:, :nm], b[:, :nm], rtol=gpu.tol("chunks"), atol=1e-12)
        np.testing.assert_allclose(a[:, 11:14], b[:, 11:14], rtol=gpu.tol("chunks"), atol=1e-9)
    else:
        # The engine's default is reproducible bit for bit in both versions: every sum has one owner and one order
        # (the reference's sums are sequential, fbearmodel.c:399-435, fftearmodel.c:657-667), and both batches cut the
        # streams into the same launches -- so "the same pair somewhere else" means the same bits.
        assert np.array_equal(a[:, :nm].view(np.uint64), b[:, :nm].view(np.uint64))
        assert np.array_equal(a[:, 11:14].view(np.uint64), b[:, 11:14].view(np.uint64))

    # the planted identical pair, against the oracle on the same bits (and unaffected neighbours above)
    r, _ = synth_np.pair(1 + 777, CH, N)
    e = orc.run_pair(advanced, r, r.copy())
    got = full[777]
    assert np.array_equal(np.isnan(got[:nm]), np.isnan(e["movs"][:nm]))
    fin = ~np.isnan(e["movs"][:nm])
    np.testing.assert_allclose(got[:nm][fin], e["movs"][:nm][fin], rtol=gpu.tol("movs", advanced), atol=1e-9)
    assert np.isnan(e["odg"]) == np.isnan(got[12]) and (np.isnan(e["odg"]) or abs(got[12] - e["odg"]) < 1e-6)

    # sparse exact check against the oracle
    worst = 0.0
    for p in range(0, PAIRS, 512):
        r, t = synth_np.pair(1 + p, CH, N)
        e = orc.run_pair(advanced, r, t)
        np.testing.assert_allclose(full[p][:nm], e["movs"][:nm], rtol=gpu.tol("movs", advanced), atol=1e-9, err_msg=f"pair {p}")
        worst = max(worst, abs(full[p][12] - e["odg"]))
        assert abs(full[p][13] - e["totalsnr"]) < 1e-9
    assert worst < 1e-6
    print(f"full size, {'advanced' if advanced else 'basic'}: max |dODG| vs oracle on 8 pairs {worst:.2e}")


@pytest.mark.parametrize("advanced", [0, 1], ids=["basic", "advanced"])
def test_one_minute_stream(advanced, fir_mode):
    """a 60 s stereo pair (2812 frames / 15 000 filter-bank blocks; BS.1387 items run 10-30 s): the
    recurrent state is carried through many chunks of the batch driver without drifting from the oracle"""
    import torch
    import gstpeaq_amd
    n = 60 * 48000
    ref, test = synth_np.pair(9001, CH, n)
    got = gstpeaq_amd.batch_run(gpu.ctx(), advanced, torch.from_numpy(ref[None]).cuda(), torch.from_numpy(test[None]).cuda())[0]
    e = orc.run_pair(advanced, ref, test)
    assert got["frames"] == e["frames"] == 2812
    nm = 5 if advanced else 11
    np.testing.assert_allclose(got["movs"][:nm], e["movs"][:nm], rtol=gpu.tol("movs", advanced), atol=1e-9)
    assert abs(got["odg"] - e["odg"]) < 1e-6 and abs(got["totalsnr"] - e["totalsnr"]) < 1e-9


def test_many_short_pairs_advanced_chunking(fir_mode):
    """8192 stereo pairs of 2 s: so many signals that the filter-bank path has to shrink its chunk
    (rows of high-passed samples are budgeted, peaq_batch.hip fb_blocks_per_chunk); results must not
    depend on it -- the same seeds in a small batch (one chunk) give the same numbers"""
    import torch
    import gstpeaq_amd
    ctx = gpu.ctx()
    n = 2 * 48000
    ref, test = gstpeaq_amd.synth_fill(ctx, 5000, 8192, CH, n)
    big = gstpeaq_amd.batch_run(ctx, 1, ref, test, sync=False)
    torch.cuda.synchronize()
    big = big.cpu().numpy()
    del ref, test
    torch.cuda.empty_cache()
    ref, test = gstpeaq_amd.synth_fill(ctx, 5000, 64, CH, n)
    small = gstpeaq_amd.batch_run(ctx, 1, ref, test, sync=False)
    torch.cuda.synchronize()
    small = small.cpu().numpy()
    assert np.all(big[:, 14] == 93) and np.all(big[:, 15] == 500)     # 92 whole frames + flush; 500 blocks
    np.testing.assert_allclose(big[:64, :5], small[:, :5], rtol=gpu.tol("chunks"), atol=1e-12)
    np.testing.assert_allclose(big[:64, 11:14], small[:, 11:14], rtol=gpu.tol("chunks"), atol=1e-9)
    e = orc.run_pair(1, *synth_np.pair(5000 + 8191, CH, n))
    np.testing.assert_allclose(big[8191, :5], e["movs"][:5], rtol=gpu.tol("movs"), atol=1e-9)


@pytest.mark.parametrize("advanced", [0, 1], ids=["basic", "advanced"])
def test_two_runs_agree_bit_for_bit(advanced):
    """The same batch twice through the default engine: the result records are equal bit for bit (round 4's FP64
    filter bank met its partial sums in LDS atomics and differed in the last bits from run to run)."""
    import torch
    import gstpeaq_amd
    ctx = gpu.ctx("default")
    ref, test = gstpeaq_amd.synth_fill(ctx, 31, 768, CH, N)
    runs = []
    for _ in range(3):
        out = gstpeaq_amd.batch_run(ctx, advanced, ref, test, sync=False)
        torch.cuda.synchronize()
        runs.append(out.cpu().numpy().copy())
    assert not np.isnan(runs[0][:, 12]).any()
    for r in runs[1:]:
        assert np.array_equal(runs[0].view(np.uint64), r.view(np.uint64))


@pytest.mark.parametrize("advanced", [0, 1], ids=["basic", "advanced"])
def test_256_seeded_pairs_against_the_real_reference_element(advanced):
    """SURVEY.md 8(d)'s parity subset as a test: the first 256 pairs of the bench workload (10 s stereo, seeds 1 .. 256)
    through the HIP path and through the reference element itself (`oracle/_ref/ref_harness`, built from the
    reference's sources in the build container and carried along; one process per host core) -- not the oracle.
    tools/parity_soak.py does the same for all 4096 (profiles/r06_parity_soak.json)."""
    import sys
    from pathlib import Path
    import torch
    import gstpeaq_amd
    root = Path(__file__).resolve().parent.parent
    if not (root / "oracle" / "_ref" / "ref_harness").exists():
        pytest.skip("oracle/_ref/ref_harness was not built (needs /root/reference at build time)")
    sys.path.insert(0, str(root))
    import bench
    n, ns = 256, 480000
    ctx = gstpeaq_amd.Context(0)
    ref, test = gstpeaq_amd.synth_fill(ctx, 1, n, 2, ns)
    res = torch.empty((n, 16), dtype=torch.float64, device=ref.device)
    gstpeaq_amd.batch_run(ctx, advanced, ref, test, results=res, sync=True)
    cpu = bench.cpu_records(ns, 2, 1, bool(advanced), n)
    assert cpu["kind"] == "reference"
    d = bench.result_deltas(res.cpu().numpy(), cpu, bool(advanced))
    assert d["delta_pairs"] == n and d["delta_nan_mismatches"] == 0, d
    assert d["odg_max_abs_delta"] < 1e-7 and d["di_max_abs_delta"] < 1e-7, d
    assert all(v == 0 for v in d["gated_movs_pairs_differing"].values()), d
    worst = max(v for v in d["mov_max_rel_delta"].values() if v is not None)
    assert worst < 1e-6, d                            # (EHS, the loosest: its own cancellation, ~ 1e-8 .. 1e-7)
    print(f"advanced={advanced}: max |dODG| {d['odg_max_abs_delta']:.2e}, worst MOV {worst:.2e}")
