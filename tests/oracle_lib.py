"""ctypes access to oracle/liboracle.so -- the CPU oracle (test infrastructure).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_LIB = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def lib():
    global _LIB
    if _LIB is None:
        so = ROOT / "oracle" / "liboracle.so"
        src = ROOT / "oracle" / "peaq_oracle.c"
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            subprocess.run(["make", "-C", str(ROOT / "oracle"), "liboracle.so"], check=True,
                           capture_output=True)
        L = C.CDLL(str(so))
        L.orc_run_pair.argtypes = [C.c_int, C.c_int, C.c_double, _fp, C.c_size_t, _fp, C.c_size_t, _dp, _dp, _dp]
        L.orc_session_new.restype = C.c_void_p
        L.orc_session_new.argtypes = [C.c_int, C.c_int, C.c_double]
        L.orc_session_free.argtypes = [C.c_void_p]
        L.orc_session_push_ref.argtypes = [C.c_void_p, _fp, C.c_size_t]
        L.orc_session_push_test.argtypes = [C.c_void_p, _fp, C.c_size_t]
        L.orc_session_flush.argtypes = [C.c_void_p]
        L.orc_session_results.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_session_totalsnr.restype = C.c_double
        L.orc_session_totalsnr.argtypes = [C.c_void_p]
        L.orc_session_frames.restype = C.c_uint
        L.orc_session_frames.argtypes = [C.c_void_p]
        L.orc_flat_fftear.argtypes = [C.c_int, C.c_double, _fp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _ip, _dp]
        L.orc_flat_fbear.argtypes = [C.c_double, _fp, C.c_int, _dp, _dp, _dp]
        L.orc_flat_leveladapt.argtypes = [C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_flat_modproc.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp]
        L.orc_flat_tables.argtypes = [C.c_int, _dp]
        L.orc_flat_mov_trace.argtypes = [C.c_int, C.c_double, _fp, C.c_size_t, _fp, C.c_size_t, C.c_int, _dp]
        L.orc_flat_mov_trace_advanced.argtypes = [C.c_int, C.c_double, _fp, C.c_size_t, _fp, C.c_size_t, C.c_int, C.c_int,
                                                  _dp, _dp]
        L.orc_di_basic.restype = C.c_double
        L.orc_di_basic.argtypes = [_dp]
        L.orc_di_advanced.restype = C.c_double
        L.orc_di_advanced.argtypes = [_dp]
        L.orc_odg.restype = C.c_double
        L.orc_odg.argtypes = [C.c_double]
        _LIB = L
    return _LIB


class Session:
    """Streaming session: mirrors one `peaq` element instance."""

    def __init__(self, advanced, channels, level=92.0):
        self.L = lib()
        self.h = self.L.orc_session_new(int(advanced), int(channels), float(level))
        self.channels = channels
        self.n_movs = 5 if advanced else 11

    def push_ref(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        self.L.orc_session_push_ref(self.h, _ptr(x, _fp), x.size // self.channels)

    def push_test(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        self.L.orc_session_push_test(self.h, _ptr(x, _fp), x.size // self.channels)

    def flush(self):
        self.L.orc_session_flush(self.h)

    def results(self):
        movs = np.zeros(11)
        di, odg = C.c_double(), C.c_double()
        self.L.orc_session_results(self.h, _ptr(movs, _dp), C.byref(di), C.byref(odg))
        return dict(movs=movs[: self.n_movs].copy(), di=di.value, odg=odg.value,
                    totalsnr=self.L.orc_session_totalsnr(self.h), frames=self.L.orc_session_frames(self.h))

    def close(self):
        if self.h:
            self.L.orc_session_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


SETTINGS_FIELDS = ("swap_mod_patts_for_noise_loudness_movs", "center_ehs_correlation_window",
                   "ehs_subtract_dc_before_window", "use_floor_for_steps_above_threshold", "clamp_movs",
                   "swap_slope_filter_coefficients")
SETTINGS_DEFAULT = dict(zip(SETTINGS_FIELDS, (1, 0, 1, 0, 0, 0)))


def set_settings(**changed):
    """the reference's settings.h switches (process-wide in the oracle); no arguments = the shipped values"""
    vals = dict(SETTINGS_DEFAULT, **changed)
    assert set(vals) == set(SETTINGS_FIELDS), sorted(set(vals) - set(SETTINGS_FIELDS))
    arr = (C.c_int * 6)(*[int(vals[k]) for k in SETTINGS_FIELDS])
    lib().orc_set_settings(arr)


def run_pair(advanced, ref, test, level=92.0):
    """ref/test: float32 [n, channels]; -> dict(movs, di, odg, totalsnr, frames)"""
    ch = ref.shape[1]
    s = Session(advanced, ch, level)
    s.push_ref(ref)
    s.push_test(test)
    s.flush()
    r = s.results()
    s.close()
    return r


MOV_TRACE = ["moddiff1", "moddiff2", "tempwt", "noiseloud", "nmr_mean", "nmr_max", "p_detect", "steps"]


def mov_trace(ref, test, n_frames, level=92.0):
    """basic version, one pair: the MOV layer's values of every frame before accumulation
    -> dict name -> np [frames, channels] (MOV_TRACE; the last two: channel 0 only)"""
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    test = np.ascontiguousarray(test, dtype=np.float32)
    ch = ref.shape[1]
    out = np.zeros((n_frames, ch, 8))
    lib().orc_flat_mov_trace(ch, level, _ptr(ref, _fp), ref.shape[0], _ptr(test, _fp), test.shape[0], n_frames,
                             _ptr(out, _dp))
    return dict(zip(MOV_TRACE, np.moveaxis(out, 2, 0)))


MOV_TRACE_ADV_BLOCK = ["rmsmoddiff", "tempwt", "noiseloud", "missing", "lindist", "loudness_ref", "loudness_test"]
MOV_TRACE_ADV_FRAME = ["segnmr_db", "nmr_mean"]


def mov_trace_advanced(ref, test, n_blocks, n_frames, level=92.0):
    """advanced version, one pair: the MOV layer's values of every filter-bank block and FFT frame before accumulation
    -> (dict name -> np [blocks, channels] (MOV_TRACE_ADV_BLOCK), dict name -> np [frames, channels] (MOV_TRACE_ADV_FRAME))"""
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    test = np.ascontiguousarray(test, dtype=np.float32)
    ch = ref.shape[1]
    ob = np.zeros((n_blocks, ch, 8))
    of = np.zeros((n_frames, ch, 2))
    lib().orc_flat_mov_trace_advanced(ch, level, _ptr(ref, _fp), ref.shape[0], _ptr(test, _fp), test.shape[0],
                                      n_blocks, n_frames, _ptr(ob, _dp), _ptr(of, _dp))
    return (dict(zip(MOV_TRACE_ADV_BLOCK, np.moveaxis(ob[:, :, :7], 2, 0))),
            dict(zip(MOV_TRACE_ADV_FRAME, np.moveaxis(of, 2, 0))))


def fftear(bands, x, n_frames, hop, level=92.0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = dict(power=np.zeros((n_frames, 1025)), weighted=np.zeros((n_frames, 1025)),
               unsmeared=np.zeros((n_frames, bands)), excitation=np.zeros((n_frames, bands)),
               energy=np.zeros(n_frames, dtype=np.int32), loudness=np.zeros(n_frames))
    lib().orc_flat_fftear(bands, level, _ptr(x, _fp), n_frames, hop, _ptr(out["power"], _dp),
                          _ptr(out["weighted"], _dp), _ptr(out["unsmeared"], _dp),
                          _ptr(out["excitation"], _dp), _ptr(out["energy"], _ip), _ptr(out["loudness"], _dp))
    return out


def fbear(x, n_blocks, level=92.0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = dict(unsmeared=np.zeros((n_blocks, 40)), excitation=np.zeros((n_blocks, 40)),
               loudness=np.zeros(n_blocks))
    lib().orc_flat_fbear(level, _ptr(x, _fp), n_blocks, _ptr(out["unsmeared"], _dp),
                         _ptr(out["excitation"], _dp), _ptr(out["loudness"], _dp))
    return out


def leveladapt(bands, ref, test):
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    test = np.ascontiguousarray(test, dtype=np.float64)
    o_r, o_t = np.zeros_like(ref), np.zeros_like(test)
    lib().orc_flat_leveladapt(bands, _ptr(ref, _dp), _ptr(test, _dp), ref.shape[0], _ptr(o_r, _dp), _ptr(o_t, _dp))
    return o_r, o_t


def modproc(bands, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    o_m, o_l = np.zeros_like(x), np.zeros_like(x)
    lib().orc_flat_modproc(bands, _ptr(x, _dp), x.shape[0], _ptr(o_m, _dp), _ptr(o_l, _dp))
    return o_m, o_l


TABLE_ROWS = ["fc", "internal_noise", "ear_tc", "exc_threshold", "threshold", "loud_factor", "adapt_tc", "mask_diff"]


def tables(bands):
    out = np.zeros((8, bands))
    lib().orc_flat_tables(bands, _ptr(out, _dp))
    return dict(zip(TABLE_ROWS, out))
