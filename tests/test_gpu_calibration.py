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
    # steady state (the kernel's second half, tens of milliseconds in): a wave alone on its SIMD with sixteen independent
    # chains fills the FP64 pipe -- 4 cycles per multiply-add (a little more: the loop) -- and the device gives the rate
    # that goes with it at the clock measured
    assert 3.95 < cal["cycles_per_fma"] < 4.2, cal
    peak = cal["compute_units"] * 4 * 16 * 2 * cal["shader_clock_mhz"] * 1e-6   # TFLOP/s at the clock it measured
    assert 0.95 * peak < cal["fp64_tflops"] < 1.01 * peak, (cal, peak)
    # the ramp (first half: from the idle clock, waves dispatched one after the other) is reported, not folded in
    assert cal["ramp_cycles_per_fma"] > 3.9 and 300. < cal["ramp_clock_mhz"] <= cal["max_clock_mhz"] * 1.05, cal
    assert cal["elapsed_ms"] > 40. and cal["event_fp64_tflops"] <= cal["fp64_tflops"] * 1.02, cal
    if cal["max_waves_on_a_simd"] == 1:
        assert cal["event_fp64_tflops"] > 0.85 * cal["fp64_tflops"], cal   # only launch and ramp between the two
    # where the waves ran: at most two on any SIMD (a process's first launch is not always dealt out evenly: the kernel
    # then lasts twice a wave's lifetime, which the event-based rate shows and the steady-state figures do not)
    assert 512 <= cal["simds_used"] <= cal["compute_units"] * 4 and cal["max_waves_on_a_simd"] in (1, 2), cal
    cal2 = ctx.calibrate()                                               # two calls in a row agree in the steady state
    assert cal2["simds_used"] >= 0.85 * cal2["compute_units"] * 4 and cal2["max_waves_on_a_simd"] in (1, 2), cal2
    assert abs(cal2["shader_clock_mhz"] / cal["shader_clock_mhz"] - 1.) < 0.08, (cal, cal2)   # (a board that throttles between the two)
    ref, test = gstpeaq_amd.synth_fill(ctx, 3, 64, 2, 96000)
    for advanced in (0, 1):
        gstpeaq_amd.batch_run(ctx, advanced, ref, test)
        mhz = ctx.last_clock_mhz()
        assert 500. < mhz <= cal["max_clock_mhz"] * 1.05, (advanced, mhz)
