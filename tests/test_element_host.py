"""The `peaq` element and CLI keep the reference's plugin surface;
CPU-only checks: factory, pads, caps, properties, CLI usage/exit codes."""
import subprocess

import pytest

import gst_env

pytestmark = pytest.mark.skipif(not gst_env.have_gst(), reason="GStreamer tools or the built plugin are missing")


def inspect():
    out = subprocess.run(["gst-inspect-1.0", f"--gst-plugin-load={gst_env.PLUGIN}", "peaq"],
                         capture_output=True, text=True, env=gst_env.env())
    assert out.returncode == 0, out.stderr
    return out.stdout


def test_factory_pads_and_caps():
    txt = inspect()
    assert "Sink/Audio" in txt                              # gstpeaq.c:321
    for pad in ("SINK template: 'ref'", "SINK template: 'test'"):
        assert pad in txt                                   # gstpeaq.c:154-165
    assert txt.count("Availability: Always") == 2
    assert "format: F32LE" in txt and "rate: 48000" in txt and "layout: interleaved" in txt   # :146-152


def test_properties():
    txt = inspect()
    # gstpeaq.c:273-317 (GObject shows playback_level in its canonical form)
    for prop in ("playback-level", "advanced", "console-output", "di ", "odg ", "totalsnr"):
        assert prop in txt, prop
    import re
    assert re.search(r"Range:\s+0 -\s+130 Default:\s+92", txt)       # playback_level: 0..130, default 92
    assert 'Default: "peaq0"' in txt                                 # instances are named like the reference's


def test_cli_usage_and_exit_codes(tmp_path):
    cli = str(gst_env.CLI)
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 1 and "REFFILE TESTFILE" in r.stdout          # peaq.c:127-133
    r = subprocess.run([cli, "--version"], capture_output=True, text=True)
    assert r.returncode == 0 and "gfx950" in r.stdout
    r = subprocess.run([cli, "--bogus", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([cli, str(tmp_path / "nope.wav"), str(tmp_path / "nope2.wav")], capture_output=True, text=True)
    assert r.returncode == 2
