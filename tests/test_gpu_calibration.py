"""peaq_calibrate / peaq_batch_last_clock (include/peaq_amd.h, "device calibration"): what bench.py prints as
`device_clock` -- the shader clock under a fixed FP64 load and while a batch ran.  Needs an MI355X (`-m gpu`)."""
import pytest

pytestmark = pytest.mark.gpu


def test_calibration_and_step_clock_are_plausible():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    import gstpeaq_amd
    ctx = gstpeaq_amd.Context(0)
    assert ctx.last_clock_mhz() == 0.0                                   # no batch yet
    cal = ctx.calibrate()
    assert cal["compute_units"] >= 64 and 500. < cal["shader_clock_mhz"] <= cal["max_clock_mhz"] * 1.05, cal
    # two waves per SIMD share the FP64 pipe: 8 cycles per multiply-add and wave when it is full (a little less: the loop)
    assert 6.5 < cal["cycles_per_fma"] < 12., cal
    peak = cal["compute_units"] * 4 * 16 * 2 * cal["shader_clock_mhz"] * 1e-6   # TFLOP/s at the clock it measured
    assert 0.8 * peak < cal["fp64_tflops"] < 1.02 * peak, (cal, peak)
    ref, test = gstpeaq_amd.synth_fill(ctx, 3, 64, 2, 96000)
    for advanced in (0, 1):
        gstpeaq_amd.batch_run(ctx, advanced, ref, test)
        mhz = ctx.last_clock_mhz()
        assert 500. < mhz <= cal["max_clock_mhz"] * 1.05, (advanced, mhz)
