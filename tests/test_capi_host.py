"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every
symbol include/peaq_amd.h declares, fails loudly instead of falling back, and
its framing arithmetic matches the reference element's frame counts."""
import ctypes as C
import json
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import gstpeaq_amd
    if not gstpeaq_amd.library_path().exists():
        gstpeaq_amd.build_library()
    return gstpeaq_amd.load_library()


def declared_functions():
    hdr = (ROOT / "include" / "peaq_amd.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(peaq_[a-z_0-9]+)\s*\(", hdr)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("peaq_ctx_create", "peaq_session_create", "peaq_session_push", "peaq_session_flush",
                 "peaq_session_results", "peaq_batch_run", "peaq_synth_fill", "peaq_debug_frontend"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/peaq_amd.h but not exported"


def test_no_silent_cpu_fallback(lib):
    """without a GPU the context cannot be created and says why"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.peaq_ctx_create(0, C.byref(h))
    assert rc == -2 and not h                      # PEAQ_ERR_DEVICE
    assert b"HIP" in lib.peaq_last_error() or b"device" in lib.peaq_last_error()
    import gstpeaq_amd
    with pytest.raises(gstpeaq_amd.PeaqError):
        gstpeaq_amd.Context(0)


def test_measurement_entry_points_check_their_arguments(lib):
    """peaq_calibrate / peaq_batch_last_clock / peaq_debug_backend_advanced: NULL handles are refused with a message"""
    for fn, args in ((lib.peaq_calibrate, (None, 0, None)), (lib.peaq_batch_last_clock, (None, None)),
                     (lib.peaq_debug_backend_advanced, (None, 1, 1, None, 1, None, None, None, None))):
        fn.restype = C.c_int
        assert fn(*args) == -1                                           # PEAQ_ERR_ARG
        lib.peaq_last_error.restype = C.c_char_p
        assert b"NULL" in lib.peaq_last_error()


def test_argument_checks(lib):
    assert lib.peaq_ctx_create(0, None) == -1      # PEAQ_ERR_ARG
    assert lib.peaq_session_push(None, 0, None, 0) == -1
    assert lib.peaq_session_flush(None) == -1


def test_frame_counts_match_the_reference_element(lib):
    """frames / fb_frames recorded from the real element for every golden case"""
    recs = json.loads((ROOT / "tests" / "golden" / "ref_e2e.json").read_text())
    for r in recs:
        c = r["case"]
        n_ref = c["n"] - c.get("ref_trim", 0)
        n_test = c["n"] - c.get("test_trim", 0)
        assert lib.peaq_frame_count(n_ref, n_test, 0) == r["frames"], c["name"]
        if c["advanced"]:
            assert lib.peaq_frame_count(n_ref, n_test, 1) == r["fb_frames"], c["name"]
    # a 10 s pair is 467 full frames + 1 flush frame, 2500 filter-bank blocks
    assert lib.peaq_frame_count(480000, 480000, 0) == 468
    assert lib.peaq_frame_count(480000, 480000, 1) == 2500
    assert lib.peaq_frame_count(0, 0, 0) == 0
    assert lib.peaq_frame_count(1, 0, 0) == 1      # leftover on one side only still flushes
    assert lib.peaq_frame_count(5000, 2048, 0) == 2


def test_python_layer_refuses_to_run_without_the_library(monkeypatch, tmp_path):
    import gstpeaq_amd.capi as capi
    monkeypatch.setattr(capi, "_LIB", None)
    monkeypatch.setattr(capi, "library_path", lambda: tmp_path / "libpeaq_amd.so")
    with pytest.raises(capi.PeaqError):
        capi.load_library()


def test_settings_default_are_the_reference_s_shipped_values():
    """peaq_settings_default needs no device: settings.h:47-97 as the reference ships them"""
    import ctypes as C
    import gstpeaq_amd
    from gstpeaq_amd.capi import Settings
    L = gstpeaq_amd.load_library()
    st = Settings()
    L.peaq_settings_default(C.byref(st))
    assert [getattr(st, k) for k, _ in Settings._fields_] == [1, 0, 1, 0, 0, 0]
    # without a GPU every entry point that needs one fails loudly (no CPU fallback)
    import torch
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert L.peaq_ctx_create(0, C.byref(h)) != 0 and b"HIP device" in L.peaq_last_error()


def test_fp64_filter_bank_tables_reproduce_the_plain_sums(lib):
    """The FP64 engine evaluates its 24 long filters as running block sums (three rectangular windows per Hann
    window, own coefficients on the two edge blocks) and the 16 short ones as one folded tile: both forms,
    evaluated on the host from the very tables the kernel reads, against the sums of fbearmodel.c:399-435."""
    lib.peaq_debug_fb_tables_selfcheck.restype = C.c_double
    lib.peaq_debug_fb_tables_selfcheck.argtypes = []
    worst = lib.peaq_debug_fb_tables_selfcheck()
    assert 0. < worst < 5e-14, worst


def test_multi_device_broker_argument_checks(lib):
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.peaq_broker_create_multi(None, 2, 0, 2, C.c_double(92.), 8, None, -1, C.byref(h)) == -1
    assert lib.peaq_broker_create_multi(devs, 0, 0, 2, C.c_double(92.), 8, None, -1, C.byref(h)) == -1
    assert lib.peaq_broker_create_multi(devs, 2, 0, 2, C.c_double(92.), 1, None, -1, C.byref(h)) == -1
    import torch
    if not torch.cuda.is_available():                  # no device: creation fails with the device's message, nothing leaks
        rc = lib.peaq_broker_create_multi(devs, 2, 0, 2, C.c_double(92.), 8, None, -1, C.byref(h))
        assert rc != 0 and not h and b"device 0" in lib.peaq_last_error()
    lib.peaq_broker_stats_size.restype = C.c_size_t
    from gstpeaq_amd.capi import _BrokerStats
    assert lib.peaq_broker_stats_size() == C.sizeof(_BrokerStats)
