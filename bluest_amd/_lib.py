"""
ctypes binding of libbluest_hip.so (include/bluest_hip.h).  There is no CPU fallback: if the library is
missing, or no GPU is visible, every compute call raises BluestHipError.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbluest_hip.so")

_lib = None

c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_vp = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f64 = ctypes.c_double

# every symbol include/bluest_hip.h declares: name -> argtypes (restype is int unless noted)
SIGNATURES = {
    "bluest_abi_version": [],
    "bluest_last_error": [],
    "bluest_device_count": [ctypes.POINTER(c_int)],
    "bluest_device_name": [ctypes.c_char_p, c_int],
    "bluest_assemble_psi": [c_vp, c_int, c_int, c_i64, c_vp, c_vp],
    "bluest_objectiveK_f64": [c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_vp],
    "bluest_objectiveK_i64": [c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_vp],
    "bluest_gradK": [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_int],
    "bluest_cleanupK": [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_int],
    "bluest_hessKQ": [c_vp, c_int, c_int, c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
    "bluest_group_pinv": [c_vp, c_int, c_int, c_i64, c_vp, c_vp],
    "bluest_plan_create": [ctypes.POINTER(c_vp), c_int, c_i64],
    "bluest_plan_destroy": [c_vp],
    "bluest_capture_guard": [c_int],
    "bluest_deferred_plans": [ctypes.POINTER(c_int)],
    "bluest_plan_add_output": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp],
    "bluest_plan_add_output_cov": [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp],
    "bluest_plan_get_invcovs": [c_vp, c_int, c_vp],
    "bluest_plan_gather_invcovs": [c_vp, c_int, c_vp, c_i64, c_vp],
    "bluest_plan_restrict": [c_vp, c_vp, c_i64, c_int, c_vp],
    "bluest_plan_output_layout": [c_vp, c_int, c_vp, c_vp, c_vp],
    "bluest_plan_finalize": [c_vp, c_int],
    "bluest_plan_n_outputs": [c_vp, ctypes.POINTER(c_int)],
    "bluest_plan_grad_layout": [c_vp, c_i64p, c_i64p],
    "bluest_plan_traffic": [c_vp, c_i64p, c_i64p],
    "bluest_plan_matfree": [c_vp, c_vp, c_i64p],
    "bluest_plan_phi_len": [c_vp, c_i64p],
    "bluest_plan_phi_chunks": [c_vp, c_vp, c_int, c_i64, c_vp],
    "bluest_plan_phi": [c_vp, c_vp, c_int, c_i64, c_vp, c_vp],
    "bluest_plan_solve": [c_vp, c_vp, c_int, c_f64, c_vp, c_vp, c_vp, c_vp],
    "bluest_plan_solve_pinv": [c_vp, c_vp, c_int, c_f64, c_vp, c_vp, c_vp, c_vp],
    "bluest_plan_grad": [c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp],
    "bluest_plan_eval": [c_vp, c_vp, c_int, c_i64, c_f64, c_vp, c_vp, c_i64, c_vp, c_vp],
    "bluest_plan_combine_grad": [c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_i64, c_vp],
    "bluest_plan_set_gate": [c_vp, c_vp, c_int],
    "bluest_plan_eval_decide": [c_vp, c_vp, c_f64, c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "bluest_plan_eval_grad_decide": [c_vp, c_vp, c_f64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "bluest_plan_solve_grad": [c_vp, c_vp, c_f64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp],
    "bluest_plan_v_workspace": [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp)],
    "bluest_spg_direction": [c_vp, c_vp, c_vp, c_f64, c_f64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "bluest_spg_converged": [c_vp, c_vp, c_vp, c_f64, c_f64, c_i64, c_vp, c_vp],
    "bluest_spg_trial": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "bluest_spg_decide": [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp],
    "bluest_spg_update_fused": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f64, c_vp, c_vp],
    "bluest_spg_finish": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f64, c_vp, c_vp],
    "bluest_spg_update": [c_vp, c_vp, c_vp, c_vp, c_vp, c_f64, c_i64, c_vp, c_vp],
    "bluest_spg_window": [c_vp] + [c_vp] * 14 + [c_f64, c_int, c_int, c_int, c_vp],
    "bluest_intproj_eval": [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp],
    "bluest_xchg_create": [ctypes.POINTER(c_vp), c_int, c_int, c_i64, c_vp],
    "bluest_xchg_connect": [c_vp, c_vp],
    "bluest_xchg_allreduce_sum": [c_vp, c_vp, c_i64, c_vp],
    "bluest_xchg_status": [c_vp, c_i64p, ctypes.POINTER(c_int)],
    "bluest_xchg_destroy": [c_vp],
    "bluest_master_max_support": [c_vp, ctypes.POINTER(c_int)],
    "bluest_master_newton": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_f64, c_vp, c_vp, c_f64, c_int, c_vp, c_vp],
    "bluest_master_newton_capped": [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_f64, c_vp, c_vp, c_f64, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp],
    "bluest_ma_update": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f64, c_vp, c_vp, c_vp],
    "bluest_plan_is_identity": [c_vp, ctypes.POINTER(ctypes.c_int)],
    "bluest_plan_eval_ma": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "bluest_support_point": [c_i64, c_int, c_vp, c_vp, c_vp, c_f64, c_vp, c_vp],
    "bluest_price": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "bluest_price_capped": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "bluest_simplex_workspace_doubles": [c_i64, c_i64p],
    "bluest_simplex_project": [c_vp, c_vp, c_f64, c_f64, c_f64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp],
}


class BluestHipError(RuntimeError):
    pass


def lib():
    """load libbluest_hip.so (build it with `python -m bluest_amd.build`); raises if it is not there"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BluestHipError(
                "%s not found: build the HIP extension first (python -m bluest_amd.build). "
                "bluest_amd has no CPU fallback." % LIB_PATH)
        # torch bundles its own libamdhip64.so.7; load it FIRST so that this library binds to the same HIP runtime
        # (two runtimes in one process cannot both own the device)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_char_p if name == "bluest_last_error" else c_int
        if L.bluest_abi_version() != 1:
            raise BluestHipError("libbluest_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().bluest_last_error()
        raise BluestHipError("libbluest_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def device_count():
    n = c_int(0)
    check(lib().bluest_device_count(ctypes.byref(n)))
    return n.value


def device_name():
    buf = ctypes.create_string_buffer(256)
    check(lib().bluest_device_name(buf, 256))
    return buf.value.decode()


def ptr(a):
    """raw address of a numpy array (host) or torch tensor (device/host); None -> NULL"""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


class capture_guard(object):
    """with capture_guard(): ... -- plans (and anything registered with `park`) dropped while a hipGraph is being captured
    are released only when the capture is over (include/bluest_hip.h: bluest_capture_guard)"""
    depth = 0
    parked = []

    def __enter__(self):
        lib().bluest_capture_guard(1)
        capture_guard.depth += 1
        return self

    def __exit__(self, *exc):
        capture_guard.depth -= 1
        lib().bluest_capture_guard(0)
        if capture_guard.depth == 0:
            del capture_guard.parked[:]          # destroys the parked hipGraphs now that no capture is running
        return False

    @staticmethod
    def park(objects):
        """keep `objects` (captured hipGraphs of a solver that is going away) alive until no capture is in progress"""
        if capture_guard.depth > 0:
            capture_guard.parked.extend(objects)
            return True
        return False
