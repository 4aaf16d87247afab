"""ctypes binding of libpeaq_amd.so (include/peaq_amd.h)."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent

# gstpeaq.c:95-108 / :86-93
MOV_NAMES_BASIC = ["BandwidthRefB", "BandwidthTestB", "TotalNMRB", "WinModDiff1B", "ADBB", "EHSB",
                   "AvgModDiff1B", "AvgModDiff2B", "RmsNoiseLoudB", "MFPDB", "RelDistFramesB"]
MOV_NAMES_ADVANCED = ["RmsModDiffA", "RmsNoiseLoudAsymA", "SegmentalNMRB", "EHSB", "AvgLinDistA"]

RESULT_DOUBLES = 16
RECORD_DOUBLES = 576


class Settings(C.Structure):
    """mirrors peaq_settings (include/peaq_amd.h): the reference's settings.h switches"""
    _fields_ = [("swap_mod_patts_for_noise_loudness_movs", C.c_int), ("center_ehs_correlation_window", C.c_int),
                ("ehs_subtract_dc_before_window", C.c_int), ("use_floor_for_steps_above_threshold", C.c_int),
                ("clamp_movs", C.c_int), ("swap_slope_filter_coefficients", C.c_int)]


class PeaqError(RuntimeError):
    pass


class _Calibration(C.Structure):
    _fields_ = [("elapsed_ms", C.c_double), ("shader_clock_mhz", C.c_double), ("fp64_tflops", C.c_double),
                ("cycles_per_fma", C.c_double), ("max_clock_mhz", C.c_double), ("compute_units", C.c_int),
                ("ramp_clock_mhz", C.c_double), ("ramp_cycles_per_fma", C.c_double), ("event_fp64_tflops", C.c_double),
                ("simds_used", C.c_int), ("max_waves_on_a_simd", C.c_int), ("dispatch_spread_ms", C.c_double)]


class _Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("frontend_ms", C.c_float), ("frontend_launches", C.c_int),
                ("backend_ms", C.c_float), ("backend_launches", C.c_int),
                ("fb_ms", C.c_float), ("fb_launches", C.c_int)]


_LIB = None


class _BrokerStats(C.Structure):
    _fields_ = [("ticks", C.c_uint64), ("launches", C.c_uint64), ("frames", C.c_uint64),
                ("max_active", C.c_uint32), ("worker_failed", C.c_uint32),
                ("tick_host_us_max", C.c_double), ("tick_host_us_p99", C.c_double), ("tick_host_us_mean", C.c_double),
                ("tick_device_us_max", C.c_double), ("tick_device_us_p99", C.c_double), ("tick_device_us_mean", C.c_double),
                ("latency_us_max", C.c_double), ("latency_us_p99", C.c_double), ("latency_us_mean", C.c_double),
                ("latency_samples", C.c_uint64)]


def library_path():
    # PEAQ_AMD_LIB: development knob for A/B runs of kernel variants (tools/variants.sh); the
    # product is the in-tree libpeaq_amd.so
    return Path(os.environ["PEAQ_AMD_LIB"]) if os.environ.get("PEAQ_AMD_LIB") else PKG / "libpeaq_amd.so"


def build_library(verbose=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run(["make", "-C", str(PKG / "csrc"), "-j", "8"], env=env, capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise PeaqError("building libpeaq_amd.so failed")
    return library_path()


def load_library():
    """Load libpeaq_amd.so; never falls back to anything else."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = library_path()
    if not so.exists():
        raise PeaqError(f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(the HIP extension is required, there is no CPU path)")
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so); whichever copy is mapped first
    # serves the whole process, and torch finds no GPU when it is not its own.  This binding uses
    # torch for device memory and streams anyway, so let it load its runtime first.
    import torch  # noqa: F401
    L = C.CDLL(str(so))
    vp, dp, fp, u32p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.peaq_last_error.restype = C.c_char_p
    L.peaq_version.restype = C.c_char_p
    L.peaq_frame_count.restype = C.c_uint32
    L.peaq_frame_count.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    L.peaq_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.peaq_ctx_destroy.argtypes = [vp]
    L.peaq_ctx_device.argtypes = [vp]
    L.peaq_settings_default.restype = None
    L.peaq_settings_default.argtypes = [C.POINTER(Settings)]
    L.peaq_ctx_set_settings.argtypes = [vp, C.POINTER(Settings)]
    L.peaq_ctx_get_settings.argtypes = [vp, C.POINTER(Settings)]
    L.peaq_ctx_set_fir_fp64.argtypes = [vp, C.c_int]
    L.peaq_ctx_get_fir_fp64.argtypes = [vp]
    L.peaq_ctx_set_fir_mode.argtypes = [vp, C.c_int]
    L.peaq_ctx_get_fir_mode.argtypes = [vp]
    L.peaq_session_create.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.POINTER(vp)]
    L.peaq_session_destroy.argtypes = [vp]
    L.peaq_session_push.argtypes = [vp, C.c_int, fp, C.c_size_t]
    L.peaq_session_flush.argtypes = [vp]
    L.peaq_session_results.argtypes = [vp, dp]
    L.peaq_session_reset.argtypes = [vp]
    L.peaq_session_set_level.argtypes = [vp, C.c_double]
    L.peaq_debug_backend.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp]
    if hasattr(L, "peaq_debug_backend_advanced"):
        L.peaq_debug_backend_advanced.argtypes = [vp, C.c_int, C.c_int, dp, C.c_int, dp, dp, dp, dp]
    L.peaq_batch_run.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, C.c_size_t,
                                 u32p, u32p, C.c_uint32, vp, vp]
    L.peaq_run_pair.argtypes = [vp, C.c_int, C.c_int, C.c_double, fp, C.c_size_t, fp, C.c_size_t, dp]
    L.peaq_batch_workspace_bytes.restype = C.c_size_t
    L.peaq_batch_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32]
    L.peaq_batch_last_timing.argtypes = [vp, C.POINTER(_Timing)]
    if hasattr(L, "peaq_calibrate"):                 # (A/B runs load older variant libraries through PEAQ_AMD_LIB)
        L.peaq_calibrate.argtypes = [vp, C.c_int, C.POINTER(_Calibration)]
        L.peaq_batch_last_clock.argtypes = [vp, dp]
    L.peaq_synth_fill.argtypes = [vp, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_size_t, vp, vp, vp]
    L.peaq_debug_frontend.argtypes = [vp, C.c_int, C.c_int, C.c_double, vp, vp, C.c_uint32, C.c_uint32,
                                      C.c_int, dp]
    L.peaq_debug_filterbank.argtypes = [vp, C.c_int, C.c_double, vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, dp]
    ip = C.POINTER(C.c_int)
    L.peaq_broker_create.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(vp)]
    L.peaq_broker_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, C.c_int,
                                           C.POINTER(vp)]
    L.peaq_broker_devices.argtypes = [vp]
    if hasattr(L, "peaq_debug_broker_fail_shard"):
        L.peaq_debug_broker_fail_shard.argtypes = [vp, C.c_int, C.c_char_p]
    L.peaq_broker_stats_size.argtypes = []
    L.peaq_broker_stats_size.restype = C.c_size_t
    L.peaq_broker_destroy.argtypes = [vp]
    L.peaq_broker_destroy.restype = None
    L.peaq_broker_open.argtypes = [vp, ip]
    L.peaq_broker_close.argtypes = [vp, C.c_int]
    L.peaq_broker_push.argtypes = [vp, C.c_int, C.c_int, fp, C.c_size_t]
    L.peaq_broker_flush.argtypes = [vp, C.c_int]
    L.peaq_broker_tick.argtypes = [vp, C.POINTER(C.c_uint)]
    L.peaq_broker_results.argtypes = [vp, C.c_int, dp]
    L.peaq_broker_start.argtypes = [vp, C.c_uint]
    L.peaq_broker_stop.argtypes = [vp]
    L.peaq_broker_stats.argtypes = [vp, C.POINTER(_BrokerStats)]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise PeaqError(f"libpeaq_amd error {rc}: {load_library().peaq_last_error().decode()}")


def _result_dict(row, advanced):
    n = 5 if advanced else 11
    return dict(movs=np.array(row[:n]), di=float(row[11]), odg=float(row[12]), totalsnr=float(row[13]),
                frames=int(row[14]), fb_blocks=int(row[15]))


class Context:
    """One per process and GPU: owns the constant tables in HBM."""

    def __init__(self, device=0):
        self.L = load_library()
        self.h = C.c_void_p()
        _check(self.L.peaq_ctx_create(int(device), C.byref(self.h)))
        self.device = device

    def close(self):
        if self.h:
            self.L.peaq_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_settings(self, **changed):
        """the reference's settings.h switches by name (include/peaq_amd.h peaq_settings); unnamed ones get
        the shipped values, no argument restores all of them.  Applies to batch calls made and to sessions /
        brokers created afterwards."""
        st = Settings()
        self.L.peaq_settings_default(C.byref(st))
        for k, v in changed.items():
            if k not in dict(Settings._fields_):
                raise PeaqError(f"unknown setting {k}")
            setattr(st, k, int(v))
        _check(self.L.peaq_ctx_set_settings(self.h, C.byref(st)))

    def settings(self):
        st = Settings()
        _check(self.L.peaq_ctx_get_settings(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in Settings._fields_}

    def set_fir_fp64(self, enable):
        """advanced version: True = every stage FP64, FIR bank on the FP64 matrix instruction (the engine's default);
        False = the opt-in split-FP16 FIR (see include/peaq_amd.h PEAQ_FIR_*)"""
        _check(self.L.peaq_ctx_set_fir_fp64(self.h, int(bool(enable))))

    def set_fir_mode(self, mode):
        """'f32' | 'f64' | 'f16x3' (include/peaq_amd.h PEAQ_FIR_*)"""
        _check(self.L.peaq_ctx_set_fir_mode(self.h, {"f32": 0, "f64": 1, "f16x3": 2}[mode]))

    def fir_mode(self):
        return ("f32", "f64", "f16x3")[self.L.peaq_ctx_get_fir_mode(self.h)]

    def fir_fp64(self):
        return bool(self.L.peaq_ctx_get_fir_fp64(self.h))

    def last_timing(self):
        t = _Timing()
        _check(self.L.peaq_batch_last_timing(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _Timing._fields_}

    def last_clock_mhz(self):
        """shader clock while the last batch ran (peaq_batch_last_clock)"""
        if not hasattr(self.L, "peaq_batch_last_clock"):
            return 0.0
        v = C.c_double(0.)
        _check(self.L.peaq_batch_last_clock(self.h, C.byref(v)))
        return v.value

    def calibrate(self, iterations=0):
        """peaq_calibrate: shader clock and FP64 rate of this device under a fixed FP64 load (include/peaq_amd.h)"""
        if not hasattr(self.L, "peaq_calibrate"):
            return None
        t = _Calibration()
        _check(self.L.peaq_calibrate(self.h, int(iterations), C.byref(t)))
        return {k: getattr(t, k) for k, _ in _Calibration._fields_}


class Session:
    """Streaming session = one `peaq` element instance (host buffers in, results out)."""

    def __init__(self, ctx, advanced, channels, playback_level=92.0):
        self.ctx, self.advanced, self.channels = ctx, bool(advanced), channels
        self.L = ctx.L
        self.h = C.c_void_p()
        _check(self.L.peaq_session_create(ctx.h, int(advanced), int(channels), float(playback_level),
                                          C.byref(self.h)))

    def push(self, pad, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        _check(self.L.peaq_session_push(self.h, int(pad), x.ctypes.data_as(C.POINTER(C.c_float)),
                                        x.size // self.channels))

    def push_ref(self, x):
        self.push(0, x)

    def push_test(self, x):
        self.push(1, x)

    def flush(self):
        _check(self.L.peaq_session_flush(self.h))

    def set_level(self, playback_level):
        _check(self.L.peaq_session_set_level(self.h, float(playback_level)))

    def results(self):
        out = np.zeros(RESULT_DOUBLES)
        _check(self.L.peaq_session_results(self.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return _result_dict(out, self.advanced)

    def close(self):
        if self.h:
            self.L.peaq_session_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Broker:
    """Many live sessions (one per hosted `peaq` element), one batched launch per tick."""

    def __init__(self, ctx, channels, max_sessions, playback_level=92.0, advanced=False, devices=None, fir_mode=None):
        """devices = None: one broker on ctx's GPU.  devices = [ordinals]: peaq_broker_create_multi -- one device
        broker (with a context of its own) per entry, sessions dealt out to the least loaded; ctx only lends the
        loaded library then."""
        self.ctx, self.channels, self.advanced = ctx, channels, bool(advanced)
        self.L = ctx.L
        self.h = C.c_void_p()
        if devices is None:
            _check(self.L.peaq_broker_create(ctx.h, int(bool(advanced)), int(channels), float(playback_level),
                                             int(max_sessions), C.byref(self.h)))
        else:
            assert self.L.peaq_broker_stats_size() == C.sizeof(_BrokerStats)
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            mode = -1 if fir_mode is None else FIR_MODES[fir_mode]
            _check(self.L.peaq_broker_create_multi(arr, len(devices), int(bool(advanced)), int(channels),
                                                   float(playback_level), int(max_sessions), None, mode, C.byref(self.h)))

    def devices(self):
        return int(self.L.peaq_broker_devices(self.h))

    def fail_shard(self, shard, message):
        """test hook (peaq_debug_broker_fail_shard): one device's share stops as after a device error"""
        _check(self.L.peaq_debug_broker_fail_shard(self.h, int(shard), message.encode()))

    def open(self):
        sid = C.c_int(-1)
        _check(self.L.peaq_broker_open(self.h, C.byref(sid)))
        return sid.value

    def close_session(self, sid):
        _check(self.L.peaq_broker_close(self.h, int(sid)))

    def push(self, sid, pad, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        _check(self.L.peaq_broker_push(self.h, int(sid), int(pad), x.ctypes.data_as(C.POINTER(C.c_float)),
                                       x.size // self.channels))

    def flush(self, sid):
        _check(self.L.peaq_broker_flush(self.h, int(sid)))

    def tick(self):
        n = C.c_uint(0)
        _check(self.L.peaq_broker_tick(self.h, C.byref(n)))
        return n.value

    def results(self, sid):
        out = np.zeros(RESULT_DOUBLES)
        _check(self.L.peaq_broker_results(self.h, int(sid), out.ctypes.data_as(C.POINTER(C.c_double))))
        return _result_dict(out, self.advanced)

    def start(self, period_us=2000):
        _check(self.L.peaq_broker_start(self.h, int(period_us)))

    def stop(self):
        _check(self.L.peaq_broker_stop(self.h))

    def stats(self):
        st = _BrokerStats()
        _check(self.L.peaq_broker_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _BrokerStats._fields_}

    def close(self):
        if self.h:
            self.L.peaq_broker_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stream_ptr(stream):
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(int(getattr(stream, "cuda_stream", stream)))   # torch.cuda.Stream or a raw hipStream_t


def batch_run(ctx, advanced, ref, test, n_ref=None, n_test=None, playback_level=92.0, results=None,
              stream=None, sync=True):
    """ref/test: CUDA float32 tensors [n_pairs, n_samples, channels] (contiguous).
    n_ref/n_test: optional per-pair lengths (samples per channel).
    Returns a list of result dicts (sync=True) or the device result tensor."""
    import torch
    assert ref.is_cuda and test.is_cuda and ref.dtype == torch.float32 and test.dtype == torch.float32
    assert ref.is_contiguous() and test.is_contiguous() and ref.shape == test.shape and ref.dim() == 3
    n_pairs, stride, channels = ref.shape
    if results is None:
        results = torch.empty((n_pairs, RESULT_DOUBLES), dtype=torch.float64, device=ref.device)
    a_ref = a_test = None
    if n_ref is not None:
        a_ref = np.ascontiguousarray(n_ref, dtype=np.uint32)
        a_test = np.ascontiguousarray(n_test, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    _check(ctx.L.peaq_batch_run(ctx.h, int(bool(advanced)), channels, float(playback_level), n_pairs,
                                C.c_void_p(ref.data_ptr()), C.c_void_p(test.data_ptr()), stride,
                                a_ref.ctypes.data_as(u32p) if a_ref is not None else None,
                                a_test.ctypes.data_as(u32p) if a_test is not None else None,
                                stride, C.c_void_p(results.data_ptr()), _stream_ptr(stream)))
    if not sync:
        return results
    torch.cuda.synchronize(ref.device)
    rows = results.cpu().numpy()
    return [_result_dict(r, advanced) for r in rows]


def run_pair(ctx, advanced, ref, test, playback_level=92.0):
    """one whole pair from host memory (peaq_run_pair): ref/test numpy float32 [n, channels]"""
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    test = np.ascontiguousarray(test, dtype=np.float32)
    ch = ref.shape[1]
    assert test.shape[1] == ch
    out = np.zeros(RESULT_DOUBLES)
    _check(ctx.L.peaq_run_pair(ctx.h, int(bool(advanced)), ch, float(playback_level),
                               ref.ctypes.data_as(C.POINTER(C.c_float)), len(ref),
                               test.ctypes.data_as(C.POINTER(C.c_float)), len(test),
                               out.ctypes.data_as(C.POINTER(C.c_double))))
    return _result_dict(out, bool(advanced))


def synth_fill(ctx, seed0, n_pairs, channels, n_samples, device="cuda:0", stream=None, out=None):
    """-> (ref, test) CUDA tensors [n_pairs, n_samples, channels] of include/peaq_synth.h pairs
    seed0 .. seed0 + n_pairs - 1.  out=(ref, test): refill existing buffers (their first n_pairs
    rows) instead of allocating -- how a GPU's share is consumed in waves."""
    import torch
    if out is not None:
        ref, test = out
        assert ref.is_cuda and ref.is_contiguous() and test.is_contiguous() and ref.shape == test.shape
        assert ref.shape[0] >= n_pairs and ref.shape[1] == n_samples and ref.shape[2] == channels
    else:
        ref = torch.empty((n_pairs, n_samples, channels), dtype=torch.float32, device=device)
        test = torch.empty_like(ref)
    _check(ctx.L.peaq_synth_fill(ctx.h, int(seed0) & 0xFFFFFFFF, n_pairs, channels, n_samples, n_samples,
                                 C.c_void_p(ref.data_ptr()), C.c_void_p(test.data_ptr()), _stream_ptr(stream)))
    return ref, test


def debug_frontend(ctx, bands, ref, test, n_frames, playback_level=92.0):
    """Stage-level access: per-frame front-end records of ONE pair.
    ref/test: CUDA float32 [n, channels] (may differ in length).  -> np [frames, channels, 576]"""
    import torch
    assert ref.is_cuda and test.is_cuda and ref.is_contiguous() and test.is_contiguous()
    channels = ref.shape[1]
    out = np.zeros((n_frames, channels, RECORD_DOUBLES))
    torch.cuda.synchronize()
    _check(ctx.L.peaq_debug_frontend(ctx.h, bands, channels, float(playback_level), C.c_void_p(ref.data_ptr()),
                                     C.c_void_p(test.data_ptr()), ref.shape[0], test.shape[0], n_frames,
                                     out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def debug_filterbank(ctx, ref, test, n_blocks, blocks_per_launch=320, playback_level=92.0):
    """Stage-level access to the filter-bank ear model of ONE pair.
    ref/test: CUDA float32 [n, channels].  -> np [blocks, channels, 168]"""
    import torch
    assert ref.is_cuda and test.is_cuda and ref.is_contiguous() and test.is_contiguous()
    channels = ref.shape[1]
    out = np.zeros((n_blocks, channels, 168))
    torch.cuda.synchronize()
    _check(ctx.L.peaq_debug_filterbank(ctx.h, channels, float(playback_level), C.c_void_p(ref.data_ptr()),
                                       C.c_void_p(test.data_ptr()), ref.shape[0], test.shape[0], n_blocks,
                                       blocks_per_launch, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


BACKEND_DEBUG_DOUBLES = 912
BACKEND_DEBUG_VECTORS = ["exc_ref", "exc_test", "adapted_ref", "adapted_test", "mod_ref", "mod_test",
                         "avgloud_ref", "avgloud_test"]
BACKEND_DEBUG_MOVS = ["moddiff1", "moddiff2", "tempwt", "noiseloud", "nmr_mean", "nmr_max", "p_detect", "steps"]


def debug_backend(ctx, records):
    """Stage-level access to the stateful back end (basic version, fresh state).
    records: np [frames, channels, 576] front-end records (from debug_frontend, or hand-built).
    -> (dict name -> np [frames, channels, 109] for BACKEND_DEBUG_VECTORS plus 'loudness' [frames, channels, 2] and
        'mov' (dict name -> np [frames, channels] for BACKEND_DEBUG_MOVS; the last two: channel 0 only),
        result dict after the last frame)"""
    rec = np.ascontiguousarray(records, dtype=np.float64)
    n_frames, channels, width = rec.shape
    assert width == RECORD_DOUBLES
    out = np.zeros((n_frames, channels, BACKEND_DEBUG_DOUBLES))
    res = np.zeros(RESULT_DOUBLES)
    dp = C.POINTER(C.c_double)
    _check(ctx.L.peaq_debug_backend(ctx.h, channels, n_frames, rec.ctypes.data_as(dp), out.ctypes.data_as(dp),
                                    res.ctypes.data_as(dp)))
    d = {name: out[:, :, 112 * i: 112 * i + 109] for i, name in enumerate(BACKEND_DEBUG_VECTORS)}
    d["loudness"] = out[:, :, 896:898]
    # the MOV layer's per-frame values before accumulation (include/peaq_amd.h, PEAQ_DEBUG_BACKEND_DOUBLES)
    d["mov"] = dict(zip(BACKEND_DEBUG_MOVS, np.moveaxis(out[:, :, 904:912], 2, 0)))
    return d, _result_dict(res, False)


ADVANCED_DEBUG_BLOCK = ["rmsmoddiff", "tempwt", "noiseloud", "missing", "lindist", "loudness_ref", "loudness_test"]
ADVANCED_DEBUG_FRAME = ["segnmr_db", "nmr_mean"]


def debug_backend_advanced(ctx, fb_records, fft_records):
    """Stage-level access to the advanced version's MOV layer (fresh state).
    fb_records: np [blocks, channels, 168] (debug_filterbank); fft_records: np [frames, channels, 576] (debug_frontend
    with 55 bands).  -> (dict name -> np [blocks, channels] for ADVANCED_DEBUG_BLOCK, dict name -> np [frames, channels]
    for ADVANCED_DEBUG_FRAME, result dict after the last block and frame): the values of EVERY block / frame before
    accumulation (include/peaq_amd.h, peaq_debug_backend_advanced)."""
    fb = np.ascontiguousarray(fb_records, dtype=np.float64)
    ff = np.ascontiguousarray(fft_records, dtype=np.float64)
    n_blocks, channels, w = fb.shape
    n_frames, ch2, w2 = ff.shape
    assert w == 168 and w2 == RECORD_DOUBLES and ch2 == channels
    ob = np.zeros((n_blocks, channels, 8))
    of = np.zeros((n_frames, channels, 2))
    res = np.zeros(RESULT_DOUBLES)
    dp = C.POINTER(C.c_double)
    _check(ctx.L.peaq_debug_backend_advanced(ctx.h, channels, n_blocks, fb.ctypes.data_as(dp), n_frames, ff.ctypes.data_as(dp),
                                             ob.ctypes.data_as(dp), of.ctypes.data_as(dp), res.ctypes.data_as(dp)))
    return (dict(zip(ADVANCED_DEBUG_BLOCK, np.moveaxis(ob[:, :, :7], 2, 0))),
            dict(zip(ADVANCED_DEBUG_FRAME, np.moveaxis(of, 2, 0))), _result_dict(res, True))
