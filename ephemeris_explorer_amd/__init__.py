"""ephemeris_explorer_amd -- MI355X-native ephemeris propagator (one hot path of Canleskis/ephemeris-explorer).

Python view of the C ABI in include/ephemeris_amd.h (libephemeris_amd.so, hand-written HIP for gfx950). The class
and method names mirror the reference's operator surface (paths relative to the reference repository):

    NBodyIntegration   = M::new(FixedMethodParams::new(h)).integrate(NBodyProblem{..})   integration/src/lib.rs
    NBodyPropagator    = ephemeris::NBodyPropagator<D, DVec3, M, SplineInterpolators<..>>   ephemeris/src/propagators/nbody.rs
    Solution           = Vec<UniformSpline<DVec3>>                                          ephemeris/src/trajectory.rs

There is NO CPU fallback: if the shared library is missing or no HIP device is visible, every compute call
raises. (The CPU oracle lives in oracle/ and is test infrastructure only; this package never imports it.)
"""
import ctypes as C
from pathlib import Path

import numpy as np

from . import systems  # noqa: F401  (state.json / ephemeris.json / ships readers)

_HERE = Path(__file__).resolve().parent
import os as _os

# (the evaluation order of the point-mass term is a run-time choice now: set_pair_variant(k) / EPH_PAIR_VARIANT=k)
LIB_PATH = _HERE / "libephemeris_amd.so"
if _os.environ.get("EPH_AMD_LIBRARY"):            # tuning builds (scripts/): another build of the same sources
    LIB_PATH = Path(_os.environ["EPH_AMD_LIBRARY"])

FORWARD, BACKWARD = 1, -1
PATH_FAST = 4
PATH_FAST_RSQ = 5
PATH_F32_PAIRS = 6   # OPT-IN mixed precision: f32 pair arithmetic, f64 accumulation and integrator (include/ephemeris_amd.h)

OK = 0
STEP_SIZE_UNDERFLOW, MAX_ITERATIONS_REACHED, BOUND_REACHED, EVAL_FAILED, SOLOUT_EXIT = 1, 2, 3, 4, 5
ERR_BAD_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_OUT_OF_MEMORY = -1, -2, -3, -4, -5

# every symbol include/ephemeris_amd.h declares (tests check the .so exports exactly these)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p)

ABI_SYMBOLS = [
    "eph_abi_version", "eph_pair_variant", "eph_set_pair_variant", "eph_release_cached_memory", "eph_status_string", "eph_last_error", "eph_device_count", "eph_set_device",
    "eph_device_name", "eph_srkn_coeffs", "eph_elm2_coeffs", "eph_accel_eval",
    "eph_nbody_create", "eph_nbody_advance", "eph_nbody_get_state", "eph_nbody_get_acc", "eph_nbody_set_bound",
    "eph_nbody_clone", "eph_nbody_destroy", "eph_nbody_eval_count", "eph_nbody_set_path", "eph_nbody_kernel_time",
    "eph_nbody_enable_timing", "eph_nbody_sync", "eph_rccl_unique_id", "eph_nbody_shard", "eph_nbody_shard_info",
    "eph_prop_shard", "eph_peer_create", "eph_peer_create_ex", "eph_peer_memory_form", "eph_peer_handle", "eph_peer_connect", "eph_peer_destroy", "eph_nbody_shard_peer",
    "eph_prop_shard_peer", "eph_nbody_advance_many", "eph_prop_step_n_many",
    "eph_prop_create", "eph_prop_step", "eph_prop_step_n", "eph_prop_step_to", "eph_prop_time",
    "eph_prop_has_reached", "eph_prop_integrator_time", "eph_prop_get_state", "eph_prop_take_solution",
    "eph_prop_propagate", "eph_prop_clone", "eph_prop_destroy", "eph_prop_integrator",
    "eph_solution_bodies", "eph_solution_info", "eph_solution_coeffs", "eph_solution_eval", "eph_solution_append", "eph_solution_create", "eph_solution_clear", "eph_solution_between",
    "eph_solution_destroy", "eph_least_squares_fit",
    "eph_ephemeris_create", "eph_ephemeris_destroy", "eph_ephemeris_append", "eph_ephemeris_merge", "eph_ephemeris_clear", "eph_ephemeris_info",
    "eph_ephemeris_is_valid_at", "eph_ephemeris_export", "eph_ephemeris_import", "eph_craft_batch_retry_failed",
    "eph_ephemeris_interpolation_errors", "eph_craft_batch_create", "eph_craft_batch_set_body_order", "eph_craft_batch_propagate", "eph_craft_batch_step_n",
    "eph_craft_batch_status", "eph_craft_batch_state", "eph_craft_batch_summary", "eph_craft_batch_knots", "eph_craft_batch_kernel_time",
    "eph_craft_batch_clone", "eph_craft_batch_knot_slabs", "eph_craft_batch_reset_knots", "eph_craft_batch_reset_events", "eph_timeline_divergence_time", "eph_craft_batch_enable_events", "eph_craft_batch_event_counts", "eph_craft_batch_events",
    "eph_craft_batch_destroy", "eph_hermite_eval", "eph_hermite_join", "eph_transitions_join", "eph_apsides_join", "eph_plot_points",
]


class EphemerisError(RuntimeError):
    """A library / device failure (negative status)."""

    def __init__(self, status, where=""):
        self.status = status
        msg = _lib().eph_status_string(status).decode()
        detail = _lib().eph_last_error().decode()
        super().__init__(f"{where}: {msg}" + (f" [{detail}]" if detail else ""))


class StepError(Exception):
    """integration::StepError / NBodyPropagatorError (positive status): errors the reference returns as values."""

    NAMES = {1: "StepSizeUnderflow", 2: "MaxIterationsReached", 3: "BoundReached", 4: "EvalFailed", 5: "Solout"}

    def __init__(self, status):
        self.status = status
        super().__init__(self.NAMES.get(status, str(status)))


_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_L = None
KNOTS_FULL = 6


class PlotView(C.Structure):
    """eph_plot_view: camera position, the floating-origin grid's affine map and the simulation time."""
    _fields_ = [("camera_position", C.c_double * 3), ("grid_matrix3", C.c_double * 9), ("grid_translation", C.c_double * 3),
                ("cell_offset", C.c_double * 3), ("current", C.c_double)]


class PlotRequest(C.Structure):
    """eph_plot_request = PlotConfig + PlotSource (ephemeris_explorer/src/ui/world/plot.rs:15-83)."""
    _fields_ = [("source_body", C.c_int32), ("reference_body", C.c_int32), ("knot_first", C.c_int64),
                ("knot_count", C.c_int64), ("start", C.c_double), ("end", C.c_double), ("bound", C.c_int32),
                ("enabled", C.c_int32), ("tan2_angular_resolution", C.c_double), ("max_points", C.c_int64)]


class AdaptiveParams(C.Structure):
    """eph_adaptive_params = integration::AdaptiveMethodParams; defaults = the app's INITIAL_ADAPTIVE_PARAMS
    (ephemeris_explorer/src/load/mod.rs:472-486)."""
    _fields_ = [("h_init", C.c_double), ("h_max", C.c_double), ("tol_position", C.c_double),
                ("tol_velocity", C.c_double), ("fac_min", C.c_double), ("fac_max", C.c_double), ("fac", C.c_double),
                ("n_max", C.c_uint32)]

    @classmethod
    def default(cls, tolerance=1e-3):
        return cls(60.0, 1.7976931348623157e308, tolerance, tolerance, 1.0 / 5.0, 5.0 / 1.0, 9.0 / 10.0, 1_000_000)


def _lib():
    """Loads libephemeris_amd.so; raises if it has not been built (no fallback of any kind)."""
    global _L
    if _L is not None:
        return _L
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950). ephemeris_explorer_amd has no CPU fallback.")
    L = C.CDLL(str(LIB_PATH))
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.eph_abi_version.restype = i32
    L.eph_pair_variant.restype = i32
    L.eph_set_pair_variant.argtypes = [i32]
    L.eph_status_string.restype = C.c_char_p
    L.eph_status_string.argtypes = [i32]
    L.eph_last_error.restype = C.c_char_p
    L.eph_device_count.argtypes = [_i32p]
    L.eph_set_device.argtypes = [i32]
    L.eph_device_name.argtypes = [C.c_char_p, i32]
    L.eph_srkn_coeffs.argtypes = [C.c_char_p, _i32p, _i32p, _dp, _dp]
    L.eph_elm2_coeffs.argtypes = [C.c_char_p, _i32p, _dp, _dp, _dp, _dp, _dp]
    L.eph_accel_eval.argtypes = [i32, _dp, _dp, _dp]
    L.eph_nbody_create.argtypes = [i32, _dp, _dp, _dp, f64, f64, C.c_char_p, C.POINTER(vp)]
    L.eph_nbody_advance.argtypes = [vp, i64]
    L.eph_nbody_get_state.argtypes = [vp, _dp, _dp, _dp, _u32p]
    L.eph_nbody_get_acc.argtypes = [vp, _dp]
    L.eph_nbody_set_bound.argtypes = [vp, f64]
    L.eph_nbody_clone.argtypes = [vp, C.POINTER(vp)]
    L.eph_nbody_destroy.argtypes = [vp]
    L.eph_nbody_destroy.restype = None
    L.eph_nbody_eval_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.eph_nbody_set_path.argtypes = [vp, i32]
    L.eph_rccl_unique_id.argtypes = [vp]
    L.eph_nbody_shard.argtypes = [vp, i32, i32, vp, EXCHANGE_FN, vp]
    L.eph_nbody_shard_info.argtypes = [vp, _i32p, _i32p, C.POINTER(C.c_uint64)]
    L.eph_prop_shard.argtypes = [vp, i32, i32, vp, EXCHANGE_FN, vp]
    L.eph_peer_create.argtypes = [i32, i32, C.c_uint64, C.POINTER(vp)]
    L.eph_peer_handle.argtypes = [vp, vp]
    L.eph_peer_connect.argtypes = [vp, vp]
    L.eph_peer_destroy.argtypes = [vp]
    L.eph_nbody_shard_peer.argtypes = [vp, vp]
    L.eph_prop_shard_peer.argtypes = [vp, vp]
    L.eph_nbody_advance_many.argtypes = [C.POINTER(vp), i32, i64]
    L.eph_prop_step_n_many.argtypes = [C.POINTER(vp), i32, i64]
    L.eph_nbody_kernel_time.argtypes = [vp, _dp, C.POINTER(C.c_uint64)]
    L.eph_nbody_enable_timing.argtypes = [vp, i32]
    L.eph_nbody_sync.argtypes = [vp]
    L.eph_prop_create.argtypes = [i32, _dp, _dp, _dp, f64, f64, i32, C.c_char_p, _u32p, _u32p, C.POINTER(vp)]
    L.eph_prop_step.argtypes = [vp]
    L.eph_prop_step_n.argtypes = [vp, i64]
    L.eph_prop_step_to.argtypes = [vp, f64]
    L.eph_prop_time.argtypes = [vp, _dp]
    L.eph_prop_has_reached.argtypes = [vp, f64, _i32p]
    L.eph_prop_integrator_time.argtypes = [vp, _dp]
    L.eph_prop_get_state.argtypes = [vp, _dp, _dp, _dp, _u32p]
    L.eph_prop_take_solution.argtypes = [vp, C.POINTER(vp)]
    L.eph_prop_propagate.argtypes = [vp, f64, C.POINTER(vp)]
    L.eph_prop_clone.argtypes = [vp, C.POINTER(vp)]
    L.eph_prop_destroy.argtypes = [vp]
    L.eph_prop_destroy.restype = None
    L.eph_prop_integrator.argtypes = [vp]
    L.eph_prop_integrator.restype = vp
    L.eph_solution_bodies.argtypes = [vp, _i32p]
    L.eph_solution_info.argtypes = [vp, i32, _dp, _dp, C.POINTER(i64)]
    L.eph_solution_coeffs.argtypes = [vp, i32, _dp, _i32p]
    L.eph_solution_eval.argtypes = [vp, i32, i64, _dp, _dp, _dp, _u8p]
    L.eph_solution_append.argtypes = [vp, vp, i32]
    L.eph_solution_destroy.argtypes = [vp]
    L.eph_solution_destroy.restype = None
    L.eph_least_squares_fit.argtypes = [i32, i32, i64, _dp, _dp, _i32p]
    L.eph_ephemeris_create.argtypes = [vp, _dp, C.POINTER(vp)]
    L.eph_ephemeris_interpolation_errors.argtypes = [vp, vp, i64, _dp, C.POINTER(i64)]
    L.eph_ephemeris_destroy.argtypes = [vp]
    L.eph_ephemeris_destroy.restype = None
    L.eph_ephemeris_append.argtypes = [vp, vp, i32]
    L.eph_ephemeris_merge.argtypes = [vp, vp, i32]
    L.eph_ephemeris_clear.argtypes = [vp, i32, f64, i32]
    L.eph_ephemeris_info.argtypes = [vp, i32, _dp, _dp, _i64p, C.POINTER(C.c_uint64)]
    L.eph_ephemeris_is_valid_at.argtypes = [vp, f64, _i32p]
    L.eph_ephemeris_export.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.eph_ephemeris_import.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.eph_craft_batch_retry_failed.argtypes = [vp]
    L.eph_craft_batch_create.argtypes = [vp, i64, _dp, _dp, _dp, C.c_char_p, C.POINTER(AdaptiveParams), _i64p, _dp, _dp,
                                         _dp, _i32p, i32, C.POINTER(vp)]
    L.eph_craft_batch_propagate.argtypes = [vp, f64]
    L.eph_craft_batch_status.argtypes = [vp, _i32p, _i32p, _u32p, _u32p]
    L.eph_craft_batch_state.argtypes = [vp, _dp, _dp, _dp, _dp]
    L.eph_craft_batch_knots.argtypes = [vp, i64, _dp, _dp, _dp]
    L.eph_craft_batch_kernel_time.argtypes = [vp, _dp]
    L.eph_craft_batch_summary.argtypes = [vp, vp]
    L.eph_craft_batch_clone.argtypes = [vp, C.POINTER(vp)]
    L.eph_solution_create.argtypes = [i32, _dp, _dp, _i64p, _dp, _i32p, C.POINTER(vp)]
    L.eph_solution_clear.argtypes = [vp, i32, f64, i32]
    L.eph_solution_between.argtypes = [vp, f64, f64, C.POINTER(vp)]
    L.eph_craft_batch_step_n.argtypes = [vp, C.c_uint32]
    L.eph_craft_batch_knot_slabs.argtypes = [vp, i32, i32, _dp, _dp]
    L.eph_craft_batch_reset_knots.argtypes = [vp]
    L.eph_craft_batch_reset_events.argtypes = [vp]
    L.eph_timeline_divergence_time.argtypes = [i64, _dp, _dp, _dp, _i32p, i64, _dp, _dp, _dp, _i32p, f64, _dp]
    L.eph_craft_batch_enable_events.argtypes = [vp, _dp, i32, i32]
    L.eph_craft_batch_event_counts.argtypes = [vp, _i32p, _i32p, _i32p]
    L.eph_craft_batch_events.argtypes = [vp, i64, _dp, _i32p, _dp, _dp, _i32p, _i32p]
    L.eph_craft_batch_destroy.argtypes = [vp]
    L.eph_craft_batch_destroy.restype = None
    L.eph_hermite_eval.argtypes = [i64, _dp, _dp, _dp, i64, _dp, _dp, _dp, _u8p]
    L.eph_plot_points.argtypes = [vp, C.POINTER(PlotView), i64, C.POINTER(PlotRequest), i64, _dp, _dp, _dp, i64, _dp,
                                  C.POINTER(C.c_float), _i64p, _i32p, _dp]
    L.eph_hermite_join.argtypes = [i64, _dp, _dp, _dp, i64, _dp, _dp, _dp, i64, _dp, _dp, _dp, _i64p]
    L.eph_transitions_join.argtypes = [i64, _dp, _i32p, i64, _dp, _i32p, f64, i64, _dp, _i32p, _i64p]
    L.eph_apsides_join.argtypes = [i64, _dp, _dp, _i32p, _i32p, i64, _dp, _dp, _i32p, _i32p, f64, i64, _dp, _dp, _i32p,
                                   _i32p, _i64p]
    if L.eph_abi_version() != 3:
        raise ImportError("libephemeris_amd.so ABI version mismatch")
    _L = L
    return L


def _check(st, where):
    if st < 0:
        raise EphemerisError(st, where)
    return st


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


def device_count():
    n = C.c_int32()
    _check(_lib().eph_device_count(C.byref(n)), "eph_device_count")
    return n.value


def set_device(i):
    _check(_lib().eph_set_device(int(i)), "eph_set_device")


def release_cached_memory():
    """Returns the library's cache of large device blocks to the driver (eph_release_cached_memory); bytes released."""
    b = C.c_uint64()
    _lib().eph_release_cached_memory.argtypes = [C.POINTER(C.c_uint64)]
    _check(_lib().eph_release_cached_memory(C.byref(b)), "eph_release_cached_memory")
    return int(b.value)


def device_name():
    buf = C.create_string_buffer(256)
    _check(_lib().eph_device_name(buf, 256), "eph_device_name")
    return buf.value.decode()


def srkn_coeffs(name):
    A, B = np.zeros(32), np.zeros(32)
    s, f = C.c_int32(), C.c_int32()
    _check(_lib().eph_srkn_coeffs(name.encode(), C.byref(s), C.byref(f), _p(A), _p(B)), "eph_srkn_coeffs")
    return A[: s.value].copy(), B[: s.value].copy(), bool(f.value)


def elm2_coeffs(name):
    wa, wb, cw = np.zeros(16), np.zeros(16), np.zeros(16)
    o = C.c_int32()
    ib, ic = C.c_double(), C.c_double()
    _check(_lib().eph_elm2_coeffs(name.encode(), C.byref(o), _p(wa), _p(wb), C.byref(ib), _p(cw), C.byref(ic)),
           "eph_elm2_coeffs")
    k = o.value
    return dict(order=k, w_alpha=wa[:k].copy(), w_beta=wb[:k].copy(), inv_beta_d=ib.value, cowell=cw[:k].copy(),
                inv_cowell_d=ic.value)


def accel_eval(pos, mu, acc=None):
    """SecondOrderODE::eval for NewtonianGravity: returns acc (+= the accelerations, reference summation order)."""
    pos, mu = _f64(pos), _f64(mu)
    acc = np.zeros_like(pos) if acc is None else _f64(acc).copy()
    _check(_lib().eph_accel_eval(len(mu), _p(pos), _p(mu), _p(acc)), "eph_accel_eval")
    return acc


def least_squares_fit(degree, samples, backward=False):
    """LeastSquaresFit::interpolate on windows of 9 samples: samples [nwin, 9, 3] -> (coeffs [nwin, 8, 3], ncoef)."""
    samples = _f64(samples).reshape(-1, 9, 3)
    nwin = samples.shape[0]
    co = np.zeros((nwin, 8, 3))
    nc = np.zeros(nwin, dtype=np.int32)
    _check(_lib().eph_least_squares_fit(int(degree), int(bool(backward)), nwin, _p(samples), _p(co), _p(nc, _i32p)),
           "eph_least_squares_fit")
    return co, nc


def pair_variant():
    """the evaluation order of the point-mass term new handles take (eph_pair_variant)"""
    return int(_lib().eph_pair_variant())


def set_pair_variant(k):
    """eph_set_pair_variant: the order (0..6, csrc/pair_term.h) for every handle created afterwards"""
    _check(_lib().eph_set_pair_variant(int(k)), "eph_set_pair_variant")


def _shard_call(fn, name, handle, rank, world, unique_id, exchange):
    cb = None
    if exchange is not None:
        def _tramp(ctx, buf, nbytes, r, w, stream):
            try:
                return int(exchange(buf, nbytes, r, w, stream) or 0)
            except Exception:                          # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        cb = EXCHANGE_FN(_tramp)
    uid = None
    if unique_id is not None:
        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        uid = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
    _check(fn(handle, int(rank), int(world), uid, cb if cb else EXCHANGE_FN(0), None), name)
    return cb                                          # the caller keeps the trampoline alive with the handle


def advance_many(integrations, n_steps):
    """eph_nbody_advance_many: advance(n_steps) on every NBodyIntegration of the list, small systems in one launch."""
    arr = (C.c_void_p * len(integrations))(*[g._h for g in integrations])
    st = _check(_lib().eph_nbody_advance_many(arr, len(integrations), int(n_steps)), "eph_nbody_advance_many")
    if st:
        raise StepError(st)


def step_n_many(propagators, n_steps):
    """eph_prop_step_n_many: step_n(n_steps) on every NBodyPropagator of the list, small systems in shared launches."""
    arr = (C.c_void_p * len(propagators))(*[p._h for p in propagators])
    st = _check(_lib().eph_prop_step_n_many(arr, len(propagators), int(n_steps)), "eph_prop_step_n_many")
    if st:
        raise StepError(st)


class PeerTransport:
    """eph_peer: the direct-write exchange (csrc/peer.hip). Create on every rank, pass `handle` (64 bytes) to every other
    rank, `connect(handles)` with all of them in rank order, then hand it to `NBodyIntegration.shard_peer` /
    `NBodyPropagator.shard_peer` (`parallel.peer_transport(dist)` does the hand-shake over torch.distributed)."""

    MEMORY = {"auto": 0, "fine": 1, "coarse": 2}

    def __init__(self, rank, world, slot_bytes=1 << 22, memory="auto"):
        self._L = _lib()
        h = C.c_void_p()
        self._L.eph_peer_create_ex.argtypes = [C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]
        _check(self._L.eph_peer_create_ex(int(rank), int(world), int(slot_bytes), self.MEMORY[memory], C.byref(h)), "eph_peer_create_ex")
        self._h = h
        self.rank, self.world = int(rank), int(world)
        f = C.c_int32()
        self._L.eph_peer_memory_form.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        _check(self._L.eph_peer_memory_form(self._h, C.byref(f)), "eph_peer_memory_form")
        self.memory = {1: "fine", 2: "coarse"}[f.value]          # what is live (auto may have fallen back)
        buf = (C.c_char * 64)()
        _check(self._L.eph_peer_handle(self._h, buf), "eph_peer_handle")
        self.handle = bytes(buf)

    def connect(self, handles):
        table = b"".join(bytes(h) for h in handles)
        if len(table) != 64 * self.world:
            raise ValueError("need one 64-byte handle per rank")
        _check(self._L.eph_peer_connect(self._h, (C.c_char * len(table)).from_buffer_copy(table)), "eph_peer_connect")
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.eph_peer_destroy(self._h)
            self._h = None


def hip_runtime():
    """The HIP runtime libephemeris_amd.so is bound to IN THIS PROCESS, as a ctypes object with hipMemcpy and
    hipStreamSynchronize: symbols looked up through the library's own handle (dlsym searches its dependencies), not through
    whatever "libamdhip64.so" resolves to -- a process that also imported PyTorch may carry a second, bundled runtime, and
    device pointers / streams of one mean nothing to the other. For host programs (and tests) that touch the library's device
    buffers themselves, e.g. inside an eph_exchange_fn."""
    lib = _lib()
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.hipMemcpy.restype = C.c_int
    lib.hipStreamSynchronize.argtypes = [C.c_void_p]
    lib.hipStreamSynchronize.restype = C.c_int
    return lib


def rccl_unique_id():
    """ncclGetUniqueId through the library (rank 0 calls it and distributes the 128 bytes)."""
    out = (C.c_char * 128)()
    _check(_lib().eph_rccl_unique_id(out), "eph_rccl_unique_id")
    return bytes(out.raw)


class NBodyIntegration:
    """Integration<NBodyProblem<DVec3>, M> (no solout). method: "QuinlanTremaine12", "Stormer13" or an SRKN name."""

    def __init__(self, pos, vel, mu, t0, h, method="QuinlanTremaine12", _handle=None, _owned=True):
        self._L = _lib()
        self._owned = _owned
        if _handle is not None:
            self._h, self.n = _handle
            return
        pos, vel, mu = _f64(pos), _f64(vel), _f64(mu)
        self.n = len(mu)
        h_ = C.c_void_p()
        st = self._L.eph_nbody_create(self.n, _p(pos), _p(vel), _p(mu), float(t0), float(h), method.encode(),
                                      C.byref(h_))
        _check(st, "eph_nbody_create")
        self._h = h_

    def advance(self, n_steps=1):
        """n_steps x Integrator::advance; raises StepError like the reference returns Err."""
        st = _check(self._L.eph_nbody_advance(self._h, int(n_steps)), "eph_nbody_advance")
        if st:
            raise StepError(st)

    def state(self):
        pos, vel = np.zeros((self.n, 3)), np.zeros((self.n, 3))
        t, sc = C.c_double(), C.c_uint32()
        _check(self._L.eph_nbody_get_state(self._h, _p(pos), _p(vel), C.byref(t), C.byref(sc)), "eph_nbody_get_state")
        return pos, vel, t.value, sc.value

    def acc(self):
        a = np.zeros((self.n, 3))
        _check(self._L.eph_nbody_get_acc(self._h, _p(a)), "eph_nbody_get_acc")
        return a

    def set_bound(self, b):
        _check(self._L.eph_nbody_set_bound(self._h, float(b)), "eph_nbody_set_bound")

    def set_path(self, path):
        """0 auto | 1 wave kernel | 2 single workgroup | 3 workgroup kernel (all bit-identical to the reference order) |
        PATH_FAST = 4: opt-in slice-parallel sums, NOT the reference's summation order (include/ephemeris_amd.h)."""
        _check(self._L.eph_nbody_set_path(self._h, int(path)), "eph_nbody_set_path")

    def enable_timing(self, on=True):
        _check(self._L.eph_nbody_enable_timing(self._h, int(on)), "eph_nbody_enable_timing")

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        _check(self._L.eph_nbody_kernel_time(self._h, C.byref(ms), C.byref(n)), "eph_nbody_kernel_time")
        return ms.value, n.value

    def sync(self):
        _check(self._L.eph_nbody_sync(self._h), "eph_nbody_sync")

    def eval_count(self):
        n = C.c_uint64()
        _check(self._L.eph_nbody_eval_count(self._h, C.byref(n)), "eph_nbody_eval_count")
        return n.value

    def clone(self):
        h_ = C.c_void_p()
        _check(self._L.eph_nbody_clone(self._h, C.byref(h_)), "eph_nbody_clone")
        return NBodyIntegration(None, None, None, 0, 0, _handle=(h_, self.n))

    def shard(self, rank, world, unique_id=None, exchange=None):
        """Partition the system by target body over `world` ranks (eph_nbody_shard). unique_id: the 128 bytes of
        rccl_unique_id() from rank 0 (RCCL transport), or exchange: callable(device_ptr, slice_bytes, rank, world,
        hip_stream) -> 0 performing the in-place all-gather (see parallel.host_staged_exchange)."""
        self._exchange_cb = _shard_call(self._L.eph_nbody_shard, "eph_nbody_shard", self._h, rank, world, unique_id,
                                        exchange)
        return self

    def shard_peer(self, peer):
        """eph_nbody_shard_peer: the same partition with the direct-write transport (a connected PeerTransport)."""
        _check(self._L.eph_nbody_shard_peer(self._h, peer._h), "eph_nbody_shard_peer")
        self._peer = peer
        return self

    def shard_info(self):
        lo, hi, g = C.c_int32(), C.c_int32(), C.c_uint64()
        _check(self._L.eph_nbody_shard_info(self._h, C.byref(lo), C.byref(hi), C.byref(g)), "eph_nbody_shard_info")
        return lo.value, hi.value, g.value

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_h", None):
            self._L.eph_nbody_destroy(self._h)
            self._h = None


class Solution:
    """Vec<UniformSpline<DVec3>>"""

    def __init__(self, handle):
        self._L = _lib()
        self._h = handle
        n = C.c_int32()
        _check(self._L.eph_solution_bodies(handle, C.byref(n)), "eph_solution_bodies")
        self.n = n.value

    def info(self, body):
        s, i, n = C.c_double(), C.c_double(), C.c_int64()
        _check(self._L.eph_solution_info(self._h, body, C.byref(s), C.byref(i), C.byref(n)), "eph_solution_info")
        return s.value, i.value, n.value

    def coeffs(self, body):
        n = self.info(body)[2]
        co = np.zeros((max(n, 1), 8, 3))
        nc = np.zeros(max(n, 1), dtype=np.int32)
        _check(self._L.eph_solution_coeffs(self._h, body, _p(co), _p(nc, _i32p)), "eph_solution_coeffs")
        return co[:n], nc[:n]

    def eval(self, body, at, with_velocity=True):
        """EvaluateTrajectory::state_vector / position at many epochs -> (pos, vel|None, inside)."""
        at = _f64(np.atleast_1d(at))
        m = len(at)
        pos = np.zeros((m, 3))
        vel = np.zeros((m, 3)) if with_velocity else None
        inside = np.zeros(m, dtype=np.uint8)
        _check(self._L.eph_solution_eval(self._h, body, m, _p(at), _p(pos), _p(vel) if with_velocity else None,
                                         _p(inside, _u8p)), "eph_solution_eval")
        return pos, vel, inside.astype(bool)

    @classmethod
    def from_parts(cls, start, interval, polys):
        """Vec<UniformSpline> from per-body (start, interval) and a list per body of (coeffs[k][3]) polynomials (host
        only: no device needed)."""
        n = len(start)
        npoly = np.array([len(p) for p in polys], dtype=np.int64)
        tot = max(int(npoly.sum()), 1)
        co, nc, q = np.zeros((tot, 8, 3)), np.zeros(tot, dtype=np.int32), 0
        for body in polys:
            for poly in body:
                poly = np.asarray(poly, dtype=np.float64).reshape(-1, 3)
                co[q, :len(poly)] = poly
                nc[q] = len(poly)
                q += 1
        h_ = C.c_void_p()
        _check(_lib().eph_solution_create(n, _p(_f64(start)), _p(_f64(interval)), _p(npoly, _i64p), _p(co), _p(nc, _i32p),
                                          C.byref(h_)), "eph_solution_create")
        return cls(h_)

    def clear_before(self, at, body=-1):
        _check(self._L.eph_solution_clear(self._h, int(body), float(at), 0), "eph_solution_clear")

    def clear_after(self, at, body=-1):
        _check(self._L.eph_solution_clear(self._h, int(body), float(at), 1), "eph_solution_clear")

    def between(self, start, end):
        """UniformSpline::between for every body -> Solution, or None where the reference returns None."""
        h_ = C.c_void_p()
        _check(self._L.eph_solution_between(self._h, float(start), float(end), C.byref(h_)), "eph_solution_between")
        return Solution(h_) if h_.value else None

    def append(self, tail, direction=FORWARD):
        st = self._L.eph_solution_append(self._h, tail._h, int(direction))
        if st == ERR_BAD_ARGUMENT:
            raise ValueError("splines are not contiguous (UniformSpline::append/prepend assert)")
        _check(st, "eph_solution_append")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.eph_solution_destroy(self._h)
            self._h = None


class NBodyPropagator:
    """NBodyPropagator<D, DVec3, M, SplineInterpolators<D, DVec3, LeastSquaresFit>> on the device."""

    def __init__(self, pos, vel, mu, t0, dt, direction, count, degree, method="QuinlanTremaine12", _handle=None):
        self._L = _lib()
        if _handle is not None:
            self._h, self.n = _handle
            return
        pos, vel, mu = _f64(pos), _f64(vel), _f64(mu)
        count = np.ascontiguousarray(count, dtype=np.uint32)
        degree = np.ascontiguousarray(degree, dtype=np.uint32)
        self.n = len(mu)
        h_ = C.c_void_p()
        st = self._L.eph_prop_create(self.n, _p(pos), _p(vel), _p(mu), float(t0), float(dt), int(direction),
                                     method.encode(), _p(count, _u32p), _p(degree, _u32p), C.byref(h_))
        _check(st, "eph_prop_create")
        self._h = h_

    @classmethod
    def from_system(cls, system, direction=FORWARD, method="QuinlanTremaine12"):
        """CelestialTrajectory::<D>::new_propagator (ephemeris_explorer/src/dynamics/celestial.rs:156-186)."""
        return cls(system.pos, system.vel, system.mu, system.epoch, system.dt, direction, system.count, system.degree,
                   method)

    def shard(self, rank, world, unique_id=None, exchange=None):
        """eph_prop_shard: partition the propagator's system by target body over the ranks (right after creation, on
        every rank); arguments as NBodyIntegration.shard."""
        self._exchange_cb = _shard_call(self._L.eph_prop_shard, "eph_prop_shard", self._h, rank, world, unique_id,
                                        exchange)
        return self

    def shard_peer(self, peer):
        """eph_prop_shard_peer: eph_prop_shard with the direct-write transport (a connected PeerTransport)."""
        _check(self._L.eph_prop_shard_peer(self._h, peer._h), "eph_prop_shard_peer")
        self._peer = peer
        return self

    def _step_status(self, st, where):
        st = _check(st, where)
        if st:
            raise StepError(st)

    def step(self):
        self._step_status(self._L.eph_prop_step(self._h), "eph_prop_step")

    def step_n(self, n):
        self._step_status(self._L.eph_prop_step_n(self._h, int(n)), "eph_prop_step_n")

    def step_to(self, t):
        self._step_status(self._L.eph_prop_step_to(self._h, float(t)), "eph_prop_step_to")

    def time(self):
        t = C.c_double()
        _check(self._L.eph_prop_time(self._h, C.byref(t)), "eph_prop_time")
        return t.value

    def has_reached(self, t):
        f = C.c_int32()
        _check(self._L.eph_prop_has_reached(self._h, float(t), C.byref(f)), "eph_prop_has_reached")
        return bool(f.value)

    def integrator_time(self):
        t = C.c_double()
        _check(self._L.eph_prop_integrator_time(self._h, C.byref(t)), "eph_prop_integrator_time")
        return t.value

    def state(self):
        pos, vel = np.zeros((self.n, 3)), np.zeros((self.n, 3))
        t, sc = C.c_double(), C.c_uint32()
        _check(self._L.eph_prop_get_state(self._h, _p(pos), _p(vel), C.byref(t), C.byref(sc)), "eph_prop_get_state")
        return pos, vel, t.value, sc.value

    def take_solution(self):
        h_ = C.c_void_p()
        _check(self._L.eph_prop_take_solution(self._h, C.byref(h_)), "eph_prop_take_solution")
        return Solution(h_)

    def propagate(self, to):
        h_ = C.c_void_p()
        self._step_status(self._L.eph_prop_propagate(self._h, float(to), C.byref(h_)), "eph_prop_propagate")
        return Solution(h_)

    def integration(self):
        """the NBodyIntegration inside (borrowed)"""
        return NBodyIntegration(None, None, None, 0, 0, _handle=(C.c_void_p(self._L.eph_prop_integrator(self._h)), self.n),
                                _owned=False)

    def clone(self):
        h_ = C.c_void_p()
        _check(self._L.eph_prop_clone(self._h, C.byref(h_)), "eph_prop_clone")
        return NBodyPropagator(None, None, None, 0, 0, 0, None, None, _handle=(h_, self.n))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.eph_prop_destroy(self._h)
            self._h = None


class Ephemeris:
    """Device-resident table of the massive bodies' UniformSplines (what `Bodies` holds in the app)."""

    def __init__(self, solution, mu):
        self._L = _lib()
        mu = _f64(mu)
        h_ = C.c_void_p()
        _check(self._L.eph_ephemeris_create(solution._h, _p(mu), C.byref(h_)), "eph_ephemeris_create")
        self._h = h_
        self.n_bodies = len(mu)

    # ---- the table is LIVE, like the reference's Arc<RwLock<PredictionTrajectory>> (dynamics/mod.rs:84-85): every batch bound to
    # it sees the new extent at its next call
    def append(self, tail, direction=FORWARD):
        """UniformSpline::append (FORWARD) / prepend (BACKWARD) for every body (trajectory.rs:515-534); raises ValueError where the
        reference's assert_eq! would panic, the table untouched."""
        st = self._L.eph_ephemeris_append(self._h, tail._h, int(direction))
        if st == ERR_BAD_ARGUMENT:
            raise ValueError("eph_ephemeris_append: not contiguous (trajectory.rs:517-518,530-531)")
        _check(st, "eph_ephemeris_append")
        return self

    def merge(self, propagated, direction=FORWARD):
        """PredictionTarget::merge for the bodies (dynamics/celestial.rs:198-204 Forward, :220-226 Backward)."""
        st = self._L.eph_ephemeris_merge(self._h, propagated._h, int(direction))
        if st == ERR_BAD_ARGUMENT:
            raise ValueError("eph_ephemeris_merge: not contiguous (trajectory.rs:517-518,530-531)")
        _check(st, "eph_ephemeris_merge")
        return self

    def clear_before(self, at, body=-1):
        _check(self._L.eph_ephemeris_clear(self._h, int(body), float(at), 0), "eph_ephemeris_clear")
        return self

    def clear_after(self, at, body=-1):
        _check(self._L.eph_ephemeris_clear(self._h, int(body), float(at), 1), "eph_ephemeris_clear")
        return self

    def info(self, body):
        """(start, interval, npoly) of one body's spline as it is now"""
        s_, iv, npoly = C.c_double(), C.c_double(), C.c_int64()
        _check(self._L.eph_ephemeris_info(self._h, int(body), C.byref(s_), C.byref(iv), C.byref(npoly), None), "eph_ephemeris_info")
        return s_.value, iv.value, npoly.value

    @property
    def revision(self):
        r = C.c_uint64()
        _check(self._L.eph_ephemeris_info(self._h, -1, None, None, None, C.byref(r)), "eph_ephemeris_info")
        return r.value

    def is_valid_at(self, t):
        """Bodies::is_valid_at (dynamics/spacecraft.rs:206-208)"""
        f = C.c_int32()
        _check(self._L.eph_ephemeris_is_valid_at(self._h, float(t), C.byref(f)), "eph_ephemeris_is_valid_at")
        return bool(f.value)

    def export_image(self):
        """One contiguous image of the table (numpy uint8): what rank 0 broadcasts (parallel.broadcast_ephemeris)."""
        need = C.c_uint64()
        self._L.eph_ephemeris_export(self._h, None, 0, C.byref(need))
        buf = np.empty(need.value, dtype=np.uint8)
        _check(self._L.eph_ephemeris_export(self._h, buf.ctypes.data_as(C.c_void_p), need.value, C.byref(need)), "eph_ephemeris_export")
        return buf

    @classmethod
    def from_image(cls, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        L = _lib()
        h_ = C.c_void_p()
        _check(L.eph_ephemeris_import(image.ctypes.data_as(C.c_void_p), image.size, C.byref(h_)), "eph_ephemeris_import")
        e = object.__new__(cls)
        e._L, e._h = L, h_
        e.n_bodies = int(np.frombuffer(image[8:16].tobytes(), dtype=np.uint64)[0])
        return e

    def interpolation_errors(self, integration, n_steps):
        """debug.rs:182-238: advance `integration` (NBodyIntegration over the same bodies) up to n_steps steps, or to its
        bound, and return (max |position - spline position| per body in metres, steps taken)."""
        err = np.zeros(self.n_bodies)
        done = C.c_int64()
        _check(self._L.eph_ephemeris_interpolation_errors(self._h, integration._h, int(n_steps), _p(err),
                                                          C.byref(done)), "eph_ephemeris_interpolation_errors")
        return err, done.value

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.eph_ephemeris_destroy(self._h)
            self._h = None


class SpacecraftBatch:
    """n independent SpacecraftPropagator<[StateVector;1], ReferenceFrame, Bodies, <adaptive ERK pair>,
    CubicHermiteSplineSolout>, one device thread each. burns[i] = list of (start, end, acc[3], ref_body or -1)."""

    def __init__(self, ephemeris, t0, pos, vel, method="Verner87", params=None, burns=None, max_knots=4096):
        self._L = _lib()
        self.ephemeris = ephemeris
        pos, vel = _f64(pos).reshape(-1, 3), _f64(vel).reshape(-1, 3)
        self.n = len(pos)
        t0 = _f64(np.broadcast_to(np.asarray(t0, dtype=np.float64), (self.n,)))
        self.params = params or AdaptiveParams.default()
        burns = burns if burns is not None else [[] for _ in range(self.n)]
        off = np.zeros(self.n + 1, dtype=np.int64)
        flat = []
        for i, bl in enumerate(burns):
            flat.extend(bl)
            off[i + 1] = len(flat)
        bs = _f64([b[0] for b in flat] or [0.0])
        be = _f64([b[1] for b in flat] or [0.0])
        ba = _f64([b[2] for b in flat] or [[0.0, 0.0, 0.0]])
        br = np.ascontiguousarray([b[3] for b in flat] or [0], dtype=np.int32)
        h_ = C.c_void_p()
        st = self._L.eph_craft_batch_create(ephemeris._h, self.n, _p(t0), _p(pos), _p(vel), method.encode(),
                                            C.byref(self.params), _p(off, _i64p), _p(bs), _p(be), _p(ba), _p(br, _i32p),
                                            int(max_knots), C.byref(h_))
        _check(st, "eph_craft_batch_create")
        self._h = h_

    def step_n(self, n_steps=1):
        """IncrementalPropagator::step n_steps times for every craft (one knot per step)."""
        _check(self._L.eph_craft_batch_step_n(self._h, int(n_steps)), "eph_craft_batch_step_n")

    def clone(self):
        """SpacecraftPropagator: Clone -- a deep copy (state, knots, events) that can be resumed independently."""
        h_ = C.c_void_p()
        _check(self._L.eph_craft_batch_clone(self._h, C.byref(h_)), "eph_craft_batch_clone")
        c = object.__new__(SpacecraftBatch)
        c._L, c.ephemeris, c.n, c.params, c._h = self._L, self.ephemeris, self.n, self.params, h_
        return c

    def retry_failed(self):
        """Re-arm: the NEXT propagate / step_n steps the craft whose last step returned a StepError too -- the reference's next
        step() on a propagator that returned Err (how a stored ship propagator resumes once the ephemeris has grown)."""
        _check(self._L.eph_craft_batch_retry_failed(self._h), "eph_craft_batch_retry_failed")
        return self

    def propagate(self, t_end):
        """step_to(t_end) for every craft; per-craft outcomes in status()"""
        _check(self._L.eph_craft_batch_propagate(self._h, float(t_end)), "eph_craft_batch_propagate")

    def status(self):
        st, nk = np.zeros(self.n, np.int32), np.zeros(self.n, np.int32)
        at, sp = np.zeros(self.n, np.uint32), np.zeros(self.n, np.uint32)
        _check(self._L.eph_craft_batch_status(self._h, _p(st, _i32p), _p(nk, _i32p), _p(at, _u32p), _p(sp, _u32p)),
               "eph_craft_batch_status")
        return dict(status=st, nknots=nk, attempts=at, steps=sp)

    RECORD = np.dtype([("t", "f8"), ("pos", "f8", 3), ("vel", "f8", 3), ("next_h", "f8"), ("status", "i4"), ("nknots", "i4"),
                       ("attempts", "u4"), ("steps", "u4")])        # eph_craft_record

    def set_body_order(self, order):
        """The order in which the massive bodies' terms are added in the acceleration (eph_craft_batch_set_body_order):
        a permutation of range(n_bodies), or None for the table's order."""
        if order is None:
            _check(self._L.eph_craft_batch_set_body_order(self._h, None), "eph_craft_batch_set_body_order")
        else:
            o = np.ascontiguousarray(order, dtype=np.int32)
            _check(self._L.eph_craft_batch_set_body_order(self._h, o.ctypes.data_as(C.POINTER(C.c_int32))), "eph_craft_batch_set_body_order")
        return self

    def summary(self, out=None):
        """status() and state() in one device-packed record array (eph_craft_batch_summary): fields t, pos, vel, next_h,
        status, nknots, attempts, steps. `out`: a record array of n entries to fill (one the caller keeps between sweeps
        has its pages mapped already: the copy into a fresh 21 MB allocation pays a page fault per 4 KB)."""
        rec = np.empty(self.n, dtype=self.RECORD) if out is None else out
        assert rec.itemsize == 80 and rec.shape == (self.n,) and rec.flags.c_contiguous
        _check(self._L.eph_craft_batch_summary(self._h, rec.ctypes.data_as(C.c_void_p)), "eph_craft_batch_summary")
        return rec

    def state(self):
        t, h = np.zeros(self.n), np.zeros(self.n)
        p, v = np.zeros((self.n, 3)), np.zeros((self.n, 3))
        _check(self._L.eph_craft_batch_state(self._h, _p(t), _p(p), _p(v), _p(h)), "eph_craft_batch_state")
        return dict(t=t, pos=p, vel=v, next_h=h)

    def knots(self, craft, nknots=None):
        nk = int(self.status()["nknots"][craft]) if nknots is None else int(nknots)
        t, p, v = np.zeros(nk), np.zeros((nk, 3)), np.zeros((nk, 3))
        _check(self._L.eph_craft_batch_knots(self._h, int(craft), _p(t), _p(p), _p(v)), "eph_craft_batch_knots")
        return t, p, v

    def knot_slabs(self, first_knot=0, n_knots=None):
        """All craft at once: (t[k][craft], y[k][6][craft]) for knots first_knot .. first_knot + n_knots - 1; entries
        at or beyond a craft's nknots are unspecified (use status()["nknots"])."""
        if n_knots is None:
            n_knots = int(self.status()["nknots"].max()) - first_knot
        t = np.zeros((n_knots, self.n))
        y = np.zeros((n_knots, 6, self.n))
        _check(self._L.eph_craft_batch_knot_slabs(self._h, int(first_knot), int(n_knots), _p(t), _p(y)),
               "eph_craft_batch_knot_slabs")
        return t, y

    def reset_knots(self):
        """Keep only the newest knot of every craft (as knot 0) and clear KNOTS_FULL: the drain point of a long run."""
        _check(self._L.eph_craft_batch_reset_knots(self._h), "eph_craft_batch_reset_knots")

    def reset_events(self):
        """Keep only the newest SOI transition of every craft, drop the apsides, clear EVENTS_FULL (after reading)."""
        _check(self._L.eph_craft_batch_reset_events(self._h), "eph_craft_batch_reset_events")

    def enable_events(self, soi_radius, max_transitions=64, max_apsides=1024):
        """Switch to the app's SpacecraftSolout: SOI transitions + apsides per accepted step (call before propagate)."""
        r = _f64(soi_radius)
        _check(self._L.eph_craft_batch_enable_events(self._h, _p(r), int(max_transitions), int(max_apsides)),
               "eph_craft_batch_enable_events")
        return self

    def event_counts(self):
        ntr, nap, st = (np.zeros(self.n, dtype=np.int32) for _ in range(3))
        _check(self._L.eph_craft_batch_event_counts(self._h, _p(ntr, _i32p), _p(nap, _i32p), _p(st, _i32p)),
               "eph_craft_batch_event_counts")
        return ntr, nap, st

    def events(self, craft, counts=None):
        """(transitions: time[], body[]), (apsides: time[], distance[], body[], kind[]) of one craft."""
        ntr, nap, _ = counts if counts is not None else self.event_counts()
        a, b = max(int(ntr[craft]), 1), max(int(nap[craft]), 1)
        tt, tb = np.zeros(a), np.zeros(a, dtype=np.int32)
        at, ad, ab, ak = np.zeros(b), np.zeros(b), np.zeros(b, dtype=np.int32), np.zeros(b, dtype=np.int32)
        _check(self._L.eph_craft_batch_events(self._h, int(craft), _p(tt), _p(tb, _i32p), _p(at), _p(ad),
                                              _p(ab, _i32p), _p(ak, _i32p)), "eph_craft_batch_events")
        k, m = int(ntr[craft]), int(nap[craft])
        return (tt[:k], tb[:k]), (at[:m], ad[:m], ab[:m], ak[:m])

    def kernel_ms(self):
        ms = C.c_double()
        _check(self._L.eph_craft_batch_kernel_time(self._h, C.byref(ms)), "eph_craft_batch_kernel_time")
        return ms.value

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.eph_craft_batch_destroy(self._h)
            self._h = None


def _burn_arrays(burns):
    n = len(burns)
    bs = _f64([b[0] for b in burns] or [0.0])
    be = _f64([b[1] for b in burns] or [0.0])
    ba = _f64([b[2] for b in burns] or [[0.0, 0.0, 0.0]])
    br = np.ascontiguousarray([b[3] for b in burns] or [0], dtype=np.int32)
    return n, bs, be, ba, br


def timeline_divergence_time(old_burns, new_burns, before):
    """Timeline::divergence_time_before (spacecraft.rs:179-213) of new_burns against old_burns; burns are
    (start, end, acc[3], ref_body or -1). The epoch a flight plan restarts from (flight_plan.rs:263-303)."""
    no, os_, oe, oa, or_ = _burn_arrays(list(old_burns))
    nn, ns, ne, na, nr = _burn_arrays(list(new_burns))
    out = C.c_double()
    _check(_lib().eph_timeline_divergence_time(no, _p(os_), _p(oe), _p(oa), _p(or_, _i32p), nn, _p(ns), _p(ne), _p(na),
                                               _p(nr, _i32p), float(before), C.byref(out)),
           "eph_timeline_divergence_time")
    return out.value


def hermite_eval(t, pos, vel, at, with_velocity=True):
    """CubicHermiteSpline::state_vector at many epochs (device)."""
    t, pos, vel, at = _f64(t), _f64(pos), _f64(vel), _f64(np.atleast_1d(at))
    m = len(at)
    op, ov = np.zeros((m, 3)), np.zeros((m, 3))
    inside = np.zeros(m, dtype=np.uint8)
    _check(_lib().eph_hermite_eval(len(t), _p(t), _p(pos), _p(vel), m, _p(at), _p(op), _p(ov) if with_velocity else None,
                                   _p(inside, _u8p)), "eph_hermite_eval")
    return op, (ov if with_velocity else None), inside.astype(bool)


def plot_points(ephemeris, view, requests, knots=None):
    """compute_plot_points_parallel + PlotPoints::new (ephemeris_explorer/src/ui/world/plot.rs:93-149,272-374) for a batch
    of plots. view: dict(camera_position, grid_matrix3 (3x3, columns = axes), grid_translation, cell_offset, current);
    requests: list of dict(source_body | knots=(first, count), reference_body, start, end, bound, enabled,
    tan2_angular_resolution, max_points); knots: (t, pos, vel) arrays the hermite sources index into.
    -> list of (status, failed_at, t[k], xyz[k, 3] float32)."""
    v = PlotView()
    v.camera_position[:] = [float(x) for x in view["camera_position"]]
    m = np.asarray(view.get("grid_matrix3", np.eye(3)), dtype=np.float64)
    v.grid_matrix3[:] = [float(m[r, c]) for c in range(3) for r in range(3)]          # column major
    v.grid_translation[:] = [float(x) for x in view.get("grid_translation", (0.0, 0.0, 0.0))]
    v.cell_offset[:] = [float(x) for x in view.get("cell_offset", (0.0, 0.0, 0.0))]
    v.current = float(view["current"])
    n = len(requests)
    arr = (PlotRequest * max(n, 1))()
    cap = 1
    for i, r in enumerate(requests):
        first, count = r.get("knots", (0, 0))
        arr[i] = PlotRequest(int(r.get("source_body", -1)), int(r.get("reference_body", -1)), int(first), int(count),
                             float(r["start"]), float(r["end"]), int(r.get("bound", 0)), int(r.get("enabled", 1)),
                             float(r["tan2_angular_resolution"]), int(r["max_points"]))
        cap = max(cap, int(r["max_points"]))
    kt, kp, kv = (np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3))) if knots is None else (_f64(knots[0]).ravel(),
                                                                                        _f64(knots[1]).reshape(-1, 3),
                                                                                        _f64(knots[2]).reshape(-1, 3))
    ot, ox = np.zeros((max(n, 1), cap)), np.zeros((max(n, 1), cap, 3), dtype=np.float32)
    cnt, st, fail = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1))
    _check(_lib().eph_plot_points(ephemeris._h, C.byref(v), n, arr, len(kt), _p(kt), _p(kp), _p(kv), cap, _p(ot),
                                  ox.ctypes.data_as(C.POINTER(C.c_float)), _p(cnt, _i64p), _p(st, _i32p), _p(fail)),
           "eph_plot_points")
    return [(int(st[i]), float(fail[i]), ot[i, :cnt[i]].copy(), ox[i, :cnt[i]].copy()) for i in range(n)]


def hermite_join(lhs, rhs):
    """SpacecraftPropagator::join(lhs, rhs) (ephemeris/src/propagators/spacecraft.rs:558-561) on (t, pos, vel) knot
    arrays -> the joined (t, pos, vel). Host only."""
    lt, lp, lv = _f64(lhs[0]).ravel(), _f64(lhs[1]).reshape(-1, 3), _f64(lhs[2]).reshape(-1, 3)
    rt, rp, rv = _f64(rhs[0]).ravel(), _f64(rhs[1]).reshape(-1, 3), _f64(rhs[2]).reshape(-1, 3)
    cap = len(lt) + len(rt)
    t, p, v = np.zeros(max(cap, 1)), np.zeros((max(cap, 1), 3)), np.zeros((max(cap, 1), 3))
    n = C.c_int64()
    _check(_lib().eph_hermite_join(len(lt), _p(lt), _p(lp), _p(lv), len(rt), _p(rt), _p(rp), _p(rv), cap, _p(t), _p(p),
                                   _p(v), C.byref(n)), "eph_hermite_join")
    return t[:n.value].copy(), p[:n.value].copy(), v[:n.value].copy()


def transitions_join(lhs, rhs, at):
    """item.transitions.clear_after(at); item.transitions.extend(rhs) -- the SoiTransitions half of
    PredictionTarget::merge (ephemeris_explorer/src/dynamics/spacecraft.rs:838-839). lhs / rhs are (time, body) arrays;
    `at` is the merged solution's trajectory.start(). Host only."""
    lt, lb = _f64(lhs[0]).ravel(), np.ascontiguousarray(lhs[1], dtype=np.int32).ravel()
    rt, rb = _f64(rhs[0]).ravel(), np.ascontiguousarray(rhs[1], dtype=np.int32).ravel()
    cap = len(lt) + len(rt)
    t, b = np.zeros(max(cap, 1)), np.zeros(max(cap, 1), dtype=np.int32)
    n = C.c_int64()
    _check(_lib().eph_transitions_join(len(lt), _p(lt), _p(lb, _i32p), len(rt), _p(rt), _p(rb, _i32p), float(at), cap, _p(t),
                                       _p(b, _i32p), C.byref(n)), "eph_transitions_join")
    return t[:n.value].copy(), b[:n.value].copy()


def apsides_join(lhs, rhs, at):
    """item.apsides.clear_after(at); item.apsides.extend(rhs) (ephemeris_explorer/src/dynamics/spacecraft.rs:836-837).
    lhs / rhs are (time, distance, kind, body) arrays. Host only."""
    def parts(x):
        return (_f64(x[0]).ravel(), _f64(x[1]).ravel(), np.ascontiguousarray(x[2], dtype=np.int32).ravel(),
                np.ascontiguousarray(x[3], dtype=np.int32).ravel())
    lt, ld, lk, lb = parts(lhs)
    rt, rd, rk, rb = parts(rhs)
    cap = len(lt) + len(rt)
    m = max(cap, 1)
    t, d, k, b = np.zeros(m), np.zeros(m), np.zeros(m, dtype=np.int32), np.zeros(m, dtype=np.int32)
    n = C.c_int64()
    _check(_lib().eph_apsides_join(len(lt), _p(lt), _p(ld), _p(lk, _i32p), _p(lb, _i32p), len(rt), _p(rt), _p(rd), _p(rk, _i32p),
                                   _p(rb, _i32p), float(at), cap, _p(t), _p(d), _p(k, _i32p), _p(b, _i32p), C.byref(n)),
           "eph_apsides_join")
    return t[:n.value].copy(), d[:n.value].copy(), k[:n.value].copy(), b[:n.value].copy()
