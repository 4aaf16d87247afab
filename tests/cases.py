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
        ref, test = np.ascontiguousarray(ref), np.ascontiguousarray(test)
        # input classes beyond "a clean pair" (all float32 arithmetic, in this order)
        if case.get("chan_gain"):                      # channels at different levels (binaural maxima, movs.c:1224-1276)
            g = np.asarray(case["chan_gain"], dtype=np.float32)[None, :]
            ref, test = ref * g, test * g
        if case.get("gain"):                           # drive into full scale ...
            ref, test = ref * np.float32(case["gain"]), test * np.float32(case["gain"])
        if case.get("clip"):                           # ... and clip hard: "both" or "test"
            if case["clip"] == "both":
                ref = np.clip(ref, np.float32(-1), np.float32(1))
            test = np.clip(test, np.float32(-1), np.float32(1))
        if case.get("dc_ref"):
            ref = ref + np.float32(case["dc_ref"])
        if case.get("dc_test"):
            test = test + np.float32(case["dc_test"])
        if case.get("invert_test"):
            test = -test
        for start, length in case.get("gaps", ()):     # digital silence in mid-stream (NORMAL -> TENTATIVE -> NORMAL in the
            who = case.get("gap_who", "both")          # accumulators, movaccum.c:317-352; the detector looks at ref only)
            if who in ("both", "ref"):
                ref[start:start + length] = 0
            if who in ("both", "test"):
                test[start:start + length] = 0
        return np.ascontiguousarray(ref, dtype=np.float32), np.ascontiguousarray(test, dtype=np.float32)
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
        cases += input_class_cases()
        for c in cases:
            c.setdefault("advanced", adv)
    return cases


def input_class_cases():
    """Input classes a clean seeded pair does not reach (round 6): digital silence in MID-stream -- one gap, three
    gaps, whole frames and parts of frames, in both signals / the reference only / the test only --, hard clipping at
    full scale, DC offsets, a polarity-inverted test signal, channels 40 dB apart."""
    one, three = [(40000, 20000)], [(20000, 6000), (50000, 12000), (90000, 3000)]
    return [
        dict(name="gap1_stereo", kind="synth", seed=40, channels=2, n=120000, gaps=one),
        dict(name="gap3_stereo", kind="synth", seed=41, channels=2, n=120000, gaps=three),
        dict(name="gap1_mono", kind="synth", seed=42, channels=1, n=120000, gaps=one),
        dict(name="gap3_mono", kind="synth", seed=43, channels=1, n=120000, gaps=three),
        dict(name="gap3_ref_only_stereo", kind="synth", seed=44, channels=2, n=120000, gaps=three, gap_who="ref"),
        dict(name="gap1_test_only_mono", kind="synth", seed=45, channels=1, n=120000, gaps=one, gap_who="test"),
        dict(name="clip_both_stereo", kind="synth", seed=46, channels=2, n=96000, gain=6.0, clip="both"),
        dict(name="clip_test_mono", kind="synth", seed=47, channels=1, n=96000, gain=3.0, clip="test"),
        dict(name="dc_offsets_stereo", kind="synth", seed=48, channels=2, n=96000, dc_ref=0.05, dc_test=-0.125),
        dict(name="inverted_test_stereo", kind="synth", seed=49, channels=2, n=96000, invert_test=1),
        dict(name="channels_40dB_apart", kind="synth", seed=50, channels=2, n=96000, chan_gain=[1.0, 0.01]),
    ]


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
