"""Parity case definitions shared by tools/make_golden.py (which runs the REAL
reference on them in the build container) and by the tests (which run the
oracle / the HIP path on the same inputs).  A case is a small dict; inputs are
rebuilt from it deterministically."""
import numpy as np

import synth_np


def make_inputs(case):
    """-> (ref, test): float32 [n, channels] arrays (lengths may differ)."""
    kind = case["kind"]
    ch = case.get("channels", 1)
    if kind == "raw":                                  # the caller's own samples on both pads
        return case["_x"], case["_x"].copy()
    if kind == "synth":
        ref, test = synth_np.pair(case["seed"], ch, case["n"])
        if case.get("identical"):
            test = ref.copy()
        if case.get("swap"):
            ref, test = test, ref
        sh = case.get("atten_shift", 0)
        if sh:
            ref = ref * np.float32(2.0 ** -sh)
            test = test * np.float32(2.0 ** -sh)
        ref = ref[: case["n"] - case.get("ref_trim", 0)]
        test = test[: case["n"] - case.get("test_trim", 0)]
        return np.ascontiguousarray(ref), np.ascontiguousarray(test)
    if kind == "ats":
        ref = synth_np.audiotestsrc(case["wave_ref"], case["n"])
        test = synth_np.audiotestsrc(case["wave_test"], case["n"])
        if ch == 2:
            ref = np.repeat(ref, 2, axis=1)
            test = np.repeat(test, 2, axis=1)
        return ref, test
    if kind == "silence":
        z = np.zeros((case["n"], ch), dtype=np.float32)
        return z, z.copy()
    raise ValueError(kind)


def e2e_cases():
    """End-to-end cases run through the reference element."""
    cases = []
    for adv in (0, 1):
        # the reference's own regression pipelines (runtest-1.0.sh); basic ODGs 0.171 / -2.007
        cases.append(dict(name="ats_sine_identical", kind="ats", wave_ref="sine", wave_test="sine", n=131072, channels=1))
        cases.append(dict(name="ats_saw_triangle", kind="ats", wave_ref="saw", wave_test="triangle", n=131072, channels=1))
        cases.append(dict(name="ats_saw_triangle_stereo", kind="ats", wave_ref="saw", wave_test="triangle", n=65536, channels=2))
        for seed in range(8):
            cases.append(dict(name=f"synth_s{seed}_stereo", kind="synth", seed=seed, channels=2, n=120000))
        for seed in (10, 11, 12, 13):
            cases.append(dict(name=f"synth_s{seed}_mono", kind="synth", seed=seed, channels=1, n=72000))
        cases.append(dict(name="synth_10s_stereo", kind="synth", seed=42, channels=2, n=480000))
        cases.append(dict(name="synth_ragged_test_short", kind="synth", seed=21, channels=2, n=100000, test_trim=1500))
        cases.append(dict(name="synth_ragged_ref_short", kind="synth", seed=22, channels=1, n=100000, ref_trim=700))
        cases.append(dict(name="synth_sub_frame", kind="synth", seed=23, channels=2, n=1500))
        cases.append(dict(name="synth_exact_frame", kind="synth", seed=24, channels=1, n=2048))
        cases.append(dict(name="synth_frame_and_hop", kind="synth", seed=25, channels=2, n=3072))
        cases.append(dict(name="synth_identical", kind="synth", seed=26, channels=2, n=96000, identical=1))
        cases.append(dict(name="synth_swapped", kind="synth", seed=27, channels=2, n=96000, swap=1))
        cases.append(dict(name="synth_quiet_36dB", kind="synth", seed=28, channels=2, n=96000, atten_shift=6))
        cases.append(dict(name="synth_quiet_60dB", kind="synth", seed=29, channels=1, n=96000, atten_shift=10))
        cases.append(dict(name="synth_quiet_84dB", kind="synth", seed=30, channels=2, n=96000, atten_shift=14))
        cases.append(dict(name="silence", kind="silence", channels=2, n=48000))
        for c in cases:
            c.setdefault("advanced", adv)
    return cases


def level_cases():
    """playback_level other than the default (the level enters both ear models' input scaling)"""
    cases = []
    for adv in (0, 1):
        for level, seed in ((60.0, 3), (75.5, 4), (105.0, 5), (130.0, 6)):
            cases.append(dict(name=f"synth_s{seed}_L{level:g}", kind="synth", seed=seed, channels=2, n=96000,
                              level=level, advanced=adv))
    return cases


def resampled_cases():
    """pairs at other sampling rates than 48 kHz, which the reference's CLI takes through audioresample
    (peaq.c:154-209): the samples of a seeded pair, declared to run at `rate`"""
    cases = []
    for adv in (0, 1):
        for rate, channels, seed, seconds in ((44100, 2, 31, 4), (44100, 1, 32, 4), (32000, 2, 33, 4), (96000, 1, 34, 3)):
            cases.append(dict(name=f"synth_s{seed}_{rate}Hz", kind="synth", seed=seed, channels=channels,
                              n=rate * seconds, rate=rate, advanced=adv))
    return cases


def settings_cases():
    """cases for the settings.h variants of the reference (tests/golden/ref_e2e_settings.json)"""
    cases = []
    for adv in (0, 1):
        cases.append(dict(name="ats_saw_triangle", kind="ats", wave_ref="saw", wave_test="triangle", n=131072,
                          channels=1, advanced=adv))
        cases.append(dict(name="synth_s1_stereo", kind="synth", seed=1, channels=2, n=120000, advanced=adv))
        cases.append(dict(name="synth_s12_mono", kind="synth", seed=12, channels=1, n=72000, advanced=adv))
        cases.append(dict(name="synth_quiet_36dB", kind="synth", seed=28, channels=2, n=96000, atten_shift=6,
                          advanced=adv))
    return cases


def stage_inputs():
    """Mono signals for the stage-level dumps (ear models)."""
    ref, test = synth_np.pair(5, 1, 8192)
    return {"synth5_ref": ref[:, 0].copy(), "synth5_test": test[:, 0].copy()}
