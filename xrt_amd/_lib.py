"""ctypes binding of libxrt_hip.so (C ABI in include/xrt_hip.h).

There is NO CPU fallback: if the HIP library cannot be loaded, or a call fails,
an exception is raised. (The numpy restatements under ``oracle/`` are test
infrastructure and are never imported from this package.)
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (XRT_HIP_LIBRARY: another build of the same library, for A/B measurements)
LIB_PATH = os.environ.get('XRT_HIP_LIBRARY') or os.path.join(_HERE, 'libxrt_hip.so')

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_size_p = ctypes.POINTER(ctypes.c_size_t)
vp = ctypes.c_void_p
i64 = ctypes.c_int64

# name -> (restype, argtypes); mirrors include/xrt_hip.h one to one
SIGNATURES = {
    'xrt_hip_version': (ctypes.c_int, []),
    'xrt_hip_device_count': (ctypes.c_int, []),
    'xrt_hip_last_error': (ctypes.c_char_p, []),
    'xrt_hip_kirchhoff_plan': (ctypes.c_int, [
        i64, i64, ctypes.c_int, ctypes.c_int, c_size_p, c_int_p, c_int_p]),
    'xrt_hip_kirchhoff_f64_dev': (ctypes.c_int, [
        i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
        ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_int,
        ctypes.c_int, vp, c_float_p]),
    'xrt_hip_kirchhoff_f64': (ctypes.c_int, [
        ctypes.c_int, c_int_p, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp,
        ctypes.c_int, vp, vp, vp, vp, vp, c_float_p]),
    'xrt_hip_reflect_workspace_bytes': (ctypes.c_size_t, [i64]),
    'xrt_hip_sizeof': (ctypes.c_int, [ctypes.c_int]),
    'xrt_hip_reflect_pass_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, c_double_p,
        c_float_p]),
    'xrt_hip_reflect_screen_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp, c_int_p,
        c_float_p]),
    'xrt_hip_shine_reflect_screen_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp, c_int_p]),
    'xrt_hip_plot_tail_workspace_bytes': (ctypes.c_int, [i64, vp, c_size_p]),
    'xrt_hip_reflect_screen_plot_fusable': (ctypes.c_int, [vp, vp, vp, vp, i64]),
    'xrt_hip_reflect_screen_plot_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp,
        ctypes.c_size_t, vp, c_int_p]),
    'xrt_hip_shine_reflect_screen_plot_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp,
        ctypes.c_size_t, vp, c_int_p]),
    'xrt_hip_reflect_tail_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp, c_int_p]),
    'xrt_hip_double_reflect_tail_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_size_t, vp,
        c_int_p]),
    'xrt_hip_bounce_workspace_bytes': (ctypes.c_size_t, [i64]),
    'xrt_hip_reflect_bounce_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, ctypes.POINTER(ctypes.c_int64),
        c_double_p]),
    'xrt_hip_multiple_reflect_out_f64_dev': (ctypes.c_int, [vp, vp, vp, vp, vp, vp]),
    'xrt_hip_double_reflect_fusable': (ctypes.c_int, [vp, vp, vp, vp]),
    'xrt_hip_double_reflect_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, c_float_p]),
    'xrt_hip_surface_eval_f64_dev': (ctypes.c_int, [
        vp, ctypes.c_int, i64, vp, vp, vp, vp, vp]),
    'xrt_hip_local_to_global_f64_dev': (ctypes.c_int, [vp, vp, vp]),
    'xrt_hip_diffract_pre_f64_dev': (ctypes.c_int, [
        vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp,
        c_double_p]),
    'xrt_hip_wave_fields_f64_dev': (ctypes.c_int, [
        i64, vp, vp, vp, ctypes.c_double, ctypes.c_int, vp, vp]),
    'xrt_hip_basis_to_global_f64_dev': (ctypes.c_int, [vp, vp, ctypes.c_int, vp]),
    'xrt_hip_wave_receive_f64_dev': (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp]),
    'xrt_hip_material_amplitude_f64_dev': (ctypes.c_int, [
        vp, i64, vp, vp, vp, vp, vp, vp, vp]),
    'xrt_hip_multilayer_amplitude_f64_dev': (ctypes.c_int, [vp, i64, vp, vp, vp, vp, vp]),
    'xrt_hip_crystal_amplitude_f64_dev': (ctypes.c_int, [
        vp, i64, vp, vp, vp, vp, vp, vp, vp]),
    'xrt_hip_screen_expose_f64_dev': (ctypes.c_int, [vp, vp, vp, vp]),
    'xrt_hip_aperture_propagate_f64_dev': (ctypes.c_int, [vp, vp, vp, vp, vp]),
    'xrt_hip_screen_expose_mark_f64_dev': (ctypes.c_int, [vp, vp, vp, vp, vp]),
    'xrt_hip_user_unit_abi': (ctypes.c_int, []),
    'xrt_hip_user_surface_load': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    'xrt_hip_user_surface_unload': (ctypes.c_int, [vp]),
    'xrt_hip_geosource_shine_f64_dev': (ctypes.c_int, [vp, vp, vp]),
    'xrt_hip_geosource_probe_f64_dev': (ctypes.c_int, [vp, i64, vp, vp]),
    'xrt_hip_hist2d_f64_dev': (ctypes.c_int, [
        vp, vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
        ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double,
        ctypes.c_int, ctypes.c_double, ctypes.c_double, vp, vp, vp]),
    'xrt_hip_plot_hist_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'xrt_hip_plot_hist_ws_f64_dev': (ctypes.c_int, [
        vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    'xrt_hip_plot_hist_workspace_bytes': (ctypes.c_int, [
        i64, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    'xrt_hip_undulator_workspace_bytes': (ctypes.c_size_t, [i64]),
    'xrt_hip_undulator_f64_dev': (ctypes.c_int, [
        vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp,
        c_float_p]),
    'xrt_hip_undulator_imap_f64_dev': (ctypes.c_int, [
        vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    'xrt_hip_undulator_f64': (ctypes.c_int, [
        ctypes.c_int, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, c_float_p]),
    'xrt_hip_custom_field_f64_dev': (ctypes.c_int, [
        vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, c_float_p]),
    'xrt_hip_custom_field_f64': (ctypes.c_int, [
        ctypes.c_int, vp, i64, vp, vp, vp, vp, vp, vp, vp, c_float_p]),
    'xrt_hip_trajectory_f64_dev': (ctypes.c_int, [
        ctypes.c_int, i64, vp, vp, vp, vp, ctypes.c_double, ctypes.c_double,
        vp, vp, vp, vp, vp, vp, vp]),
    'xrt_hip_gaussian_beam_f64_dev': (ctypes.c_int, [
        vp, i64, vp, vp, vp, vp, vp, ctypes.c_double, vp, vp, vp, vp, vp]),
    'xrt_hip_bend_imap_f64_dev': (ctypes.c_int, [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    'xrt_hip_debug_bessel_k_f64_dev': (ctypes.c_int, [i64, vp, vp, vp, vp]),
    'xrt_hip_kirchhoff_report': (ctypes.c_int, [
        vp, vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint),
        ctypes.POINTER(ctypes.c_int64)]),
    'xrt_hip_debug_sqrt_f64_dev': (ctypes.c_int, [i64, vp, vp, vp, vp]),
    'xrt_hip_debug_sqrt_seeded_f64_dev': (ctypes.c_int, [i64, vp, vp, vp, vp, vp]),
    'xrt_hip_debug_divconst_f64_dev': (ctypes.c_int, [i64, vp, ctypes.c_double, vp, vp]),
    'xrt_hip_debug_sincos_f64_dev': (ctypes.c_int, [i64, vp, vp, vp, vp]),
    'xrt_hip_debug_sincos_tab_f64_dev': (ctypes.c_int, [i64, vp, vp, vp, vp]),
    'xrt_hip_debug_sincos_tab4k_f64_dev': (ctypes.c_int, [i64, vp, vp, vp, vp]),
    'xrt_hip_event_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    'xrt_hip_event_destroy': (ctypes.c_int, [vp]),
    'xrt_hip_event_elapsed_ms': (ctypes.c_int, [vp, vp, ctypes.POINTER(ctypes.c_float)]),
    'xrt_hip_reflect_time_next_pass': (ctypes.c_int, [vp, vp, vp, vp]),
}


class XrtHipError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


def load(build_if_missing=True):
    """Returns the loaded CDLL with typed signatures; raises XrtHipError if the
    library is missing and cannot be built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            if not build_if_missing:
                raise XrtHipError('%s not built' % LIB_PATH)
            try:
                from .csrc.build import build
                build()
            except Exception as e:  # noqa: BLE001
                raise XrtHipError(
                    'libxrt_hip.so is missing and could not be built with '
                    'hipcc (%s); there is no CPU fallback' % (e,))
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise XrtHipError('cannot load %s: %s' % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        from . import _structs
        for which, st in enumerate(_structs.STRUCTS):
            if lib.xrt_hip_sizeof(which) != ctypes.sizeof(st):
                raise XrtHipError(
                    'struct layout mismatch for %s: C %d B, ctypes %d B'
                    % (st.__name__, lib.xrt_hip_sizeof(which), ctypes.sizeof(st)))
        _lib = lib
        return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().xrt_hip_last_error().decode('utf-8', 'replace')
        raise XrtHipError('%s failed (%d): %s' % (what or 'xrt_hip call', rc, msg))


def device_count():
    n = load().xrt_hip_device_count()
    return max(n, 0)


def require_gpu():
    lib = load()
    n = lib.xrt_hip_device_count()
    if n <= 0:
        raise XrtHipError('no MI355X/ROCm device visible: %s (there is no CPU '
                          'fallback)' % lib.xrt_hip_last_error().decode())
    return n
