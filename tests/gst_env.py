"""GStreamer lives under /opt/conda in this image (no system registry)."""
import os
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PLUGIN = ROOT / "gstpeaq_amd" / "gst" / "libgstpeaq.so"
CLI = ROOT / "gstpeaq_amd" / "cli" / "peaq"


def env():
    e = dict(os.environ)
    e["PATH"] = "/opt/conda/bin:" + e.get("PATH", "")
    e.setdefault("GST_PLUGIN_SYSTEM_PATH", "/opt/conda/lib/gstreamer-1.0")
    e.setdefault("GST_PLUGIN_SCANNER", "/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner")
    e.setdefault("GST_REGISTRY", "/tmp/peaq_amd_test_registry.bin")
    return e


def have_gst():
    return shutil.which("gst-launch-1.0", path=env()["PATH"]) is not None and PLUGIN.exists()
