"""The engine's run-time version of the reference's settings.h switches (include/peaq_amd.h
peaq_settings; settings.h:47-97) against the REAL reference compiled with each switch flipped
(tests/golden/ref_e2e_settings.json, see tests/test_oracle_settings.py): batch path, sessions and the
broker.  Tolerances as for the default build's goldens (tests/gpu_common.py TOL, per arithmetic of the
filter bank; the basic version 1e-7).  Needs an MI355X (`-m gpu`)."""
import json

import numpy as np
import pytest

import cases as case_defs
import gpu_common as gpu

pytestmark = pytest.mark.gpu


def _records(golden_dir, advanced):
    return [r for r in json.loads((golden_dir / "ref_e2e_settings.json").read_text())
            if r["case"]["advanced"] == advanced]


def _check(got, rec):
    exp = np.array([float(v) for v in rec["movs"]])
    adv = rec["case"]["advanced"]
    odg_tol = gpu.tol("odg", advanced=adv)
    assert got["frames"] == rec["frames"]
    np.testing.assert_allclose(got["movs"][: len(exp)], exp, rtol=gpu.tol("movs", adv), atol=1e-9,
                               err_msg=f"{rec['variant']} {rec['case']['name']}")
    assert abs(got["odg"] - rec["odg"]) <= odg_tol and abs(got["di"] - rec["di"]) <= odg_tol, (rec["variant"], got, rec)


@pytest.mark.parametrize("advanced", [0, 1])
def test_batch_matches_the_reference_built_with_other_settings(golden_dir, advanced, fir_mode):
    ctx = gpu.ctx()
    assert ctx.settings() == dict(swap_mod_patts_for_noise_loudness_movs=1, center_ehs_correlation_window=0,
                                  ehs_subtract_dc_before_window=1, use_floor_for_steps_above_threshold=0,
                                  clamp_movs=0, swap_slope_filter_coefficients=0)
    try:
        for rec in _records(golden_dir, advanced):
            case = rec["case"]
            ctx.set_settings(**rec["settings"])
            got = gpu.run_batch([case_defs.make_inputs(case)], advanced, case["channels"])[0]
            _check(got, rec)
            # ... and the switch is what made the difference: the default build's result is another one
            if rec["odg"] != rec["odg_default"]:
                ctx.set_settings()
                dflt = gpu.run_batch([case_defs.make_inputs(case)], advanced, case["channels"])[0]
                assert abs(dflt["odg"] - rec["odg_default"]) <= gpu.tol("odg", advanced=advanced)
    finally:
        ctx.set_settings()


def test_sessions_and_broker_take_the_settings_at_creation(golden_dir):
    import gstpeaq_amd
    ctx = gpu.ctx()
    recs = [r for r in _records(golden_dir, 0) if r["case"]["name"] == "synth_s12_mono"]
    rec = next(r for r in recs if r["variant"] == "dcafter")
    case = rec["case"]
    ref, test = case_defs.make_inputs(case)
    try:
        ctx.set_settings(**rec["settings"])
        s = gstpeaq_amd.Session(ctx, 0, case["channels"])
        br = gstpeaq_amd.Broker(ctx, case["channels"], 4)
        sid = br.open()
        ctx.set_settings()                               # later changes do not reach them
        for lo in range(0, len(ref), 5000):
            s.push(0, ref[lo:lo + 5000])
            s.push(1, test[lo:lo + 5000])
            br.push(sid, 0, ref[lo:lo + 5000])
            br.push(sid, 1, test[lo:lo + 5000])
            br.tick()
        s.flush()
        br.flush(sid)
        _check(s.results(), rec)
        _check(br.results(sid), rec)
        s.close()
        br.close_session(sid)
        br.close()
    finally:
        ctx.set_settings()


def test_cli_reads_the_switches_from_the_environment(tmp_path):
    """PEAQ_AMD_SETTINGS (settings.h macro names) for the programs that, like the reference's, have no
    property for them: the CLI on 16-bit WAV files, CLAMP_MOVS + CENTER_EHS_CORRELATION_WINDOW"""
    import os
    import subprocess
    import gst_env
    import oracle_lib as orc
    from test_gpu_element import write_wav16
    ref, test = case_defs.make_inputs(dict(kind="synth", seed=12, channels=1, n=72000))
    rq = write_wav16(tmp_path / "ref.wav", ref)
    tq = write_wav16(tmp_path / "test.wav", test)
    try:
        orc.set_settings(clamp_movs=1, center_ehs_correlation_window=1)
        exp = orc.run_pair(0, rq, tq)
    finally:
        orc.set_settings()
    dflt = orc.run_pair(0, rq, tq)
    assert "%.3f" % exp["odg"] != "%.3f" % dflt["odg"]
    env = dict(os.environ, PEAQ_AMD_SETTINGS="CLAMP_MOVS=1,center_ehs_correlation_window=1")
    out = subprocess.run([str(gst_env.CLI), "--basic", str(tmp_path / "ref.wav"), str(tmp_path / "test.wav")],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().splitlines()[-2] == "Objective Difference Grade: %.3f" % exp["odg"]
    bad = subprocess.run([str(gst_env.CLI), "--basic", str(tmp_path / "ref.wav"), str(tmp_path / "test.wav")],
                         capture_output=True, text=True, env=dict(os.environ, PEAQ_AMD_SETTINGS="CLAMP=1"))
    assert bad.returncode != 0 and "unknown switch" in (bad.stdout + bad.stderr)
