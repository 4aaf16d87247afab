"""The reference's OWN integration tests (src/runtest-1.0.sh) run against our
`peaq` element, and the `peaq` CLI against WAV files -- on the GPU."""
import subprocess
import wave

import numpy as np
import pytest

import cases as case_defs
import gst_env
import oracle_lib as orc

pytestmark = pytest.mark.gpu


def launch(*args):
    cmd = ["gst-launch-1.0", "-q", f"--gst-plugin-load={gst_env.PLUGIN}", *args]
    out = subprocess.run(cmd, capture_output=True, text=True, env=gst_env.env(), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def odg_of(stdout):
    line = [l for l in stdout.splitlines() if l.startswith("Objective Difference Grade:")][-1]
    return line.split()[3]


@pytest.fixture(scope="module", autouse=True)
def _need_gst():
    if not gst_env.have_gst():
        pytest.fail("GStreamer tools / built plugin missing on the GPU box")


def test_runtest_identical_sine_through_tee():
    # runtest-1.0.sh:8-20
    out = launch("audiotestsrc", "name=src0", "num-buffers=128", "freq=440", "tee", "name=tee0",
                 "queue", "name=queue0", "queue", "name=queue1", "peaq", "name=peaq0",
                 "src0.src!tee0.sink", "tee0.src_2!queue0.sink", "tee0.src_1!queue1.sink",
                 "queue0.src!peaq0.ref", "queue1.src!peaq0.test")
    assert odg_of(out) == "0.171"


@pytest.mark.parametrize("ref_caps,test_caps", [(None, None), ("audio/x-raw,channels=2", None),
                                                (None, "audio/x-raw,channels=2")],
                         ids=["mono-mono", "stereo-mono", "mono-stereo"])
def test_runtest_saw_vs_triangle(ref_caps, test_caps):
    # runtest-1.0.sh:21-50; the last two exercise the caps negotiation between the pads
    args = ["audiotestsrc", "name=src0", "num-buffers=128", "wave=saw", "freq=440",
            "audiotestsrc", "name=src1", "num-buffers=128", "wave=triangle", "freq=440", "peaq", "name=peaq"]
    args.append("src0.src!" + (ref_caps + "!" if ref_caps else "") + "peaq.ref")
    args.append("src1.src!" + (test_caps + "!" if test_caps else "") + "peaq.test")
    out = launch(*args)
    assert odg_of(out) == "-2.007"
    if ref_caps is None and test_caps is None:
        # the MOV table printed by the reference for this pipeline
        for line in ("   BandwidthRefB: 921.000000", "  BandwidthTestB: 733.000000", "      Total NMRB: 1.713453",
                     "    WinModDiff1B: 11.064398", "            ADBB: 3.397249", "            EHSB: 0.225160",
                     "    AvgModDiff1B: 11.793056", "    AvgModDiff2B: 11.093628", "   RmsNoiseLoudB: 1.179670",
                     "           MFPDB: 0.999999", "  RelDistFramesB: 1.000000"):
            assert line in out, line


def test_element_advanced_mode():
    out = launch("audiotestsrc", "name=src0", "num-buffers=128", "wave=saw", "freq=440",
                 "audiotestsrc", "name=src1", "num-buffers=128", "wave=triangle", "freq=440",
                 "peaq", "name=peaq", "advanced=true", "src0.src!peaq.ref", "src1.src!peaq.test")
    assert "RmsModDiffA = " in out and "AvgLinDistA = " in out
    assert odg_of(out) == "-3.612"                    # tests/golden/ref_e2e.json: ats_saw_triangle, advanced


def write_wav16(path, x):
    q = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(x.shape[1])
        w.setsampwidth(2)
        w.setframerate(48000)
        w.writeframes(q.tobytes())
    return q.astype(np.float32) / np.float32(32768.0)


@pytest.mark.parametrize("advanced", [0, 1])
def test_cli_on_wav_files(tmp_path, advanced):
    ref, test = case_defs.make_inputs(dict(kind="synth", seed=31, channels=2, n=100000))
    rq = write_wav16(tmp_path / "ref.wav", ref)
    tq = write_wav16(tmp_path / "test.wav", test)
    exp = orc.run_pair(advanced, rq, tq)
    import os
    # whole files in one call (peaq_run_pair, the default) and buffer by buffer through a session
    for env in (dict(os.environ), dict(os.environ, PEAQ_AMD_CLI_STREAM="1")):
        out = subprocess.run([str(gst_env.CLI), "--advanced" if advanced else "--basic",
                              str(tmp_path / "ref.wav"), str(tmp_path / "test.wav")], capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.strip().splitlines()
        assert lines[-2] == "Objective Difference Grade: %.3f" % exp["odg"]      # peaq.c:217-220
        assert lines[-1] == "Distortion Index: %.3f" % exp["di"]


def test_many_elements_share_the_broker():
    """with PEAQ_AMD_BROKER set, every `peaq` element of the process becomes a
    session of one broker and their frames run as batched launches; each element still prints
    exactly what the reference prints for its pipeline (runtest-1.0.sh:8-50)."""
    args = []
    n = 8
    for i in range(n):
        if i % 2 == 0:
            args += ["audiotestsrc", f"name=s{i}", "num-buffers=128", "freq=440", "tee", f"name=t{i}",
                     "queue", f"name=qa{i}", "queue", f"name=qb{i}", "peaq", f"name=p{i}",
                     f"s{i}.src!t{i}.sink", f"t{i}.src_0!qa{i}.sink", f"t{i}.src_1!qb{i}.sink",
                     f"qa{i}.src!p{i}.ref", f"qb{i}.src!p{i}.test"]
        else:
            args += ["audiotestsrc", f"name=s{i}", "num-buffers=128", "wave=saw", "freq=440",
                     "audiotestsrc", f"name=r{i}", "num-buffers=128", "wave=triangle", "freq=440",
                     "peaq", f"name=p{i}", f"s{i}.src!p{i}.ref", f"r{i}.src!p{i}.test"]
    cmd = ["gst-launch-1.0", "-q", f"--gst-plugin-load={gst_env.PLUGIN}", *args]
    env = gst_env.env()
    env["PEAQ_AMD_BROKER"] = "16"
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    odgs = sorted(l.split()[3] for l in out.stdout.splitlines() if l.startswith("Objective Difference Grade:"))
    assert odgs == sorted(["0.171"] * (n // 2) + ["-2.007"] * (n // 2)), out.stdout[-1500:]
    # the advanced version goes through its own broker (saw vs triangle: -3.612, as test_element_advanced_mode)
    args = []
    for i in range(4):
        args += ["audiotestsrc", f"name=s{i}", "num-buffers=128", "wave=saw", "freq=440",
                 "audiotestsrc", f"name=r{i}", "num-buffers=128", "wave=triangle", "freq=440",
                 "peaq", f"name=p{i}", "advanced=true", f"s{i}.src!p{i}.ref", f"r{i}.src!p{i}.test"]
    out = subprocess.run(["gst-launch-1.0", "-q", f"--gst-plugin-load={gst_env.PLUGIN}", *args],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    odgs = [l.split()[3] for l in out.stdout.splitlines() if l.startswith("Objective Difference Grade:")]
    assert odgs == ["-3.612"] * 4, out.stdout[-1500:]
    # PEAQ_AMD_DEVICES: the shared broker spans several GPUs (here two device brokers on the one GPU of the box)
    env["PEAQ_AMD_DEVICES"] = "0,0"
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    odgs = sorted(l.split()[3] for l in out.stdout.splitlines() if l.startswith("Objective Difference Grade:"))
    assert odgs == sorted(["0.171"] * (n // 2) + ["-2.007"] * (n // 2)), out.stdout[-1500:]


def test_element_playback_level_property():
    """playback_level (gstpeaq.c:273-281) reaches the engine: saw vs triangle at 75.5 dB SPL; the
    printed ODG equals the oracle's for the same audiotestsrc samples"""
    import synth_np
    out = launch("audiotestsrc", "name=src0", "num-buffers=64", "wave=saw", "freq=440",
                 "audiotestsrc", "name=src1", "num-buffers=64", "wave=triangle", "freq=440",
                 "peaq", "name=peaq", "playback_level=75.5", "src0.src!peaq.ref", "src1.src!peaq.test")
    n = 64 * 1024
    e = orc.run_pair(0, synth_np.audiotestsrc("saw", n), synth_np.audiotestsrc("triangle", n), level=75.5)
    assert odg_of(out) == "%.3f" % e["odg"]
    assert odg_of(out) != "-2.007"                     # not the default level's answer
