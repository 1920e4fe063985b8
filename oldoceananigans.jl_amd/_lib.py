"""ctypes binding of libocn_mi355x.so (C ABI: include/ocn_mi355x.h). The product path FAILS LOUDLY when the HIP
extension is missing or a call returns a non-zero status -- there is no CPU fallback."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("OCN_LIB") or os.path.join(_HERE, "csrc", "libocn_mi355x.so")   # OCN_LIB: experiments with alternative builds
_lib = None


class OcnError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))
            if f.endswith((".hip", ".h", ".sh"))]
    srcs.append(os.path.join(_HERE, "..", "include", "ocn_mi355x.h"))
    srcs.append(os.path.join(_HERE, "..", "include", "ocn_weno_coeffs.h"))
    if force or not os.path.exists(SO_PATH) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs):
        subprocess.check_call(["bash", os.path.join(_HERE, "csrc", "build.sh")])
    return SO_PATH


# every symbol include/ocn_mi355x.h declares: name -> (restype, argtypes)
_vp, _dp, _ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
_pp = C.POINTER(C.c_void_p)


class BC(C.Structure):
    """ocn_bc_t"""
    _fields_ = [("kind", C.c_int), ("value", C.c_double), ("array", C.c_void_p)]

class Transport(C.Structure):
    """ocn_transport_t: caller-supplied collectives (device addresses as integers)"""
    EXCHANGE_START = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    EXCHANGE_WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
    ALL_TO_ALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    ALLREDUCE_MAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))
    EXCHANGE_PEERS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    ALL_TO_ALL_GROUP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


Transport._fields_ = [("exchange_start", Transport.EXCHANGE_START), ("exchange_wait", Transport.EXCHANGE_WAIT),
                      ("all_to_all", Transport.ALL_TO_ALL), ("all_gather", Transport.ALL_GATHER),
                      ("allreduce_max", Transport.ALLREDUCE_MAX), ("user", C.c_void_p),
                      ("exchange_peers", Transport.EXCHANGE_PEERS), ("all_to_all_group", Transport.ALL_TO_ALL_GROUP)]

SYMBOLS = {
    "ocn_dist_unique_id": (C.c_int, [_vp]),
    "ocn_dist_create": (C.c_int, [_pp, _vp, C.c_int, C.c_int]),
    "ocn_dist_create_transport": (C.c_int, [_pp, C.POINTER(Transport), C.c_int, C.c_int]),
    "ocn_dist_destroy": (C.c_int, [_vp]),
    "ocn_dist_info": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "ocn_dist_comm_info": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "ocn_dist_set_self_loop": (C.c_int, [_vp, C.c_int]),
    "ocn_dist_exchange_start": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_size_t]),
    "ocn_dist_exchange_wait": (C.c_int, [_vp]),
    "ocn_dist_all_to_all": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "ocn_dist_all_gather": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "ocn_dist_allreduce_max": (C.c_int, [_vp, _dp]),
    "ocn_dist_barrier": (C.c_int, [_vp]),
    "ocn_dist_model_create": (C.c_int, [_pp, _vp, C.c_int, _vp, C.c_double]),
    "ocn_dist_model_create_sizes": (C.c_int, [_pp, _vp, C.c_int, _vp, C.c_double, _ip]),
    "ocn_dist_model_create_partition": (C.c_int, [_pp, _vp, C.c_int, _vp, C.c_double, _ip, C.c_int]),
    "ocn_dist_set_layout": (C.c_int, [_vp, C.c_int, C.c_int]),
    "ocn_dist_model_create_pencil": (C.c_int, [_pp, _vp, C.c_int, _vp, C.c_double, C.c_double, C.c_int, C.c_int, _ip, _ip, C.c_int, C.c_int]),
    "ocn_dist_model_max_abs_divergence": (C.c_int, [_vp, _dp]),
    "ocn_transposable_create": (C.c_int, [_pp, _vp, C.c_int, C.c_int, C.c_int]),
    "ocn_transposable_destroy": (C.c_int, [_vp]),
    "ocn_transposable_fields": (C.c_int, [_vp, _pp, _pp, _pp, _ip, _ip, _ip]),
    "ocn_transpose_z_to_y": (C.c_int, [_vp]),
    "ocn_transpose_y_to_x": (C.c_int, [_vp]),
    "ocn_transpose_x_to_y": (C.c_int, [_vp]),
    "ocn_transpose_y_to_z": (C.c_int, [_vp]),
    "ocn_init": (C.c_int, [C.c_int]),
    "ocn_device_count": (C.c_int, [_ip]),
    "ocn_sync": (C.c_int, []),
    "ocn_last_error": (C.c_char_p, []),
    "ocn_version": (C.c_char_p, []),
    "ocn_malloc": (C.c_int, [_pp, C.c_size_t]),
    "ocn_free": (C.c_int, [_vp]),
    "ocn_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ocn_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ocn_memcpy_d2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ocn_memset_zero": (C.c_int, [_vp, C.c_size_t]),
    "ocn_stream": (_vp, []),
    "ocn_set_stream": (C.c_int, [_vp]),
    "ocn_own_stream": (C.c_int, []),
    "ocn_pack_x_halos": (C.c_int, [_vp, _pp, _vp, C.c_int, _vp, _vp]),
    "ocn_unpack_x_halos": (C.c_int, [_vp, _pp, _vp, C.c_int, _vp, _vp]),
    "ocn_compute_linear_flux_bc": (C.c_int, [_vp, _vp, _ip, C.c_int, C.c_double, C.c_double, _vp]),
    "ocn_model_set_linear_flux_bc": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_char_p]),
    "ocn_make_pressure_correction_range": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _ip]),
    "ocn_make_pressure_correction_divide": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _ip]),
    "ocn_model_reset": (C.c_int, [_vp]),
    "ocn_dist_poisson_layout": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "ocn_hasnan": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_int)]),
    "ocn_pack_x_halos_depth": (C.c_int, [_vp, _pp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "ocn_unpack_x_halos_depth": (C.c_int, [_vp, _pp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "ocn_fill_halo_regions_bcs": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int]),
    "ocn_compute_flux_bcs": (C.c_int, [_vp, _vp, _ip, _vp]),
    "ocn_compute_tendencies_and_substep": (C.c_int, [_vp, _vp, C.c_int, _vp, _ip, _vp, _vp, C.c_double, C.c_double, C.c_double,
                                                     C.c_int]),
    "ocn_debug_rcp64_check": (C.c_int, [C.c_ulonglong, C.c_int, C.c_int, C.c_ulonglong, C.POINTER(C.c_ulonglong)]),
    "ocn_compute_closure_tendencies": (C.c_int, [_vp, _vp, _vp, _vp, _pp, C.c_int, C.c_double, _dp, _vp, _vp, _vp, _pp, _ip]),
    "ocn_model_set_closure": (C.c_int, [_vp, C.c_double, _dp]),
    "ocn_model_set_amd": (C.c_int, [_vp, C.c_double, _dp]),
    "ocn_compute_closure_tendencies_field": (C.c_int, [_vp, _vp, _vp, _vp, _pp, C.c_int, _vp, _pp, _vp, _vp, _vp, _pp, _ip]),
    "ocn_compute_amd_diffusivities": (C.c_int, [_vp, C.c_double, _dp, _vp, _vp, _vp, _pp, C.c_int, _vp, _pp, _ip]),
    "ocn_ab2_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_double]),
    "ocn_model_time_step_ab2": (C.c_int, [_vp, C.c_double, C.c_double, C.c_int]),
    "ocn_update_hydrostatic_pressure": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "ocn_add_hydrostatic_pressure_gradient": (C.c_int, [_vp, _vp, _vp, _vp, _ip]),
    "ocn_model_set_buoyancy": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]),
    "ocn_add_fplane_coriolis": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, _vp, _ip]),
    "ocn_model_set_coriolis": (C.c_int, [_vp, C.c_int, C.c_double]),
    "ocn_cell_advection_timescale": (C.c_int, [_vp, _vp, _vp, _vp, _dp]),
    "ocn_model_cell_advection_timescale": (C.c_int, [_vp, _dp]),
    "ocn_model_get_option": (C.c_int, [_vp, C.c_char_p, _ip]),
    "ocn_model_set_boundary_condition": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, C.c_double]),
    "ocn_model_set_boundary_condition_array": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, _vp]),
    "ocn_dist_poisson_create": (C.c_int, [_pp, _vp, C.c_int, C.c_int, C.c_double]),
    "ocn_dist_poisson_destroy": (C.c_int, [_vp]),
    "ocn_dist_poisson_set_buffers": (C.c_int, [_vp, _vp, _vp]),
    "ocn_dist_poisson_buffer_size": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "ocn_dist_poisson_source_term": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ocn_dist_poisson_forward_yz": (C.c_int, [_vp]),
    "ocn_dist_poisson_payload_size": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "ocn_dist_poisson_set_gather_buffers": (C.c_int, [_vp, _vp, _vp]),
    "ocn_dist_poisson_forward_local": (C.c_int, [_vp]),
    "ocn_dist_poisson_backward_local": (C.c_int, [_vp, _vp]),
    "ocn_dist_poisson_solve_x": (C.c_int, [_vp]),
    "ocn_dist_poisson_backward_yz": (C.c_int, [_vp, _vp]),
    "ocn_grid_create": (C.c_int, [_pp, _ip, _ip, _ip, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp]),
    "ocn_grid_destroy": (C.c_int, [_vp]),
    "ocn_grid_parent_size": (C.c_int, [_vp, _ip, _ip]),
    "ocn_fill_halo_regions": (C.c_int, [_vp, _pp, _vp, C.c_int, C.c_int]),
    "ocn_compute_Gu": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _ip]),
    "ocn_compute_Gv": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _ip]),
    "ocn_compute_Gw": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _ip]),
    "ocn_compute_Gc": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _ip]),
    "ocn_compute_tendencies": (C.c_int, [_vp, _vp, _vp, _vp, _pp, C.c_int, _vp, _vp, _vp, _pp, _ip]),
    "ocn_rk3_substep": (C.c_int, [_vp, _pp, _pp, _pp, _vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]),
    "ocn_cache_tendencies": (C.c_int, [_vp, _pp, _pp, _vp, C.c_int]),
    "ocn_compute_source_term": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int]),
    "ocn_make_pressure_correction": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ocn_divide_interior": (C.c_int, [_vp, _vp, C.c_double]),
    "ocn_poisson_create": (C.c_int, [_pp, _vp, C.c_int]),
    "ocn_poisson_destroy": (C.c_int, [_vp]),
    "ocn_poisson_kind": (C.c_int, [_vp]),
    "ocn_poisson_rhs": (C.c_int, [_vp, _pp]),
    "ocn_poisson_solve": (C.c_int, [_vp, _vp]),
    "ocn_solve_for_pressure": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ocn_batched_tridiagonal_solve_z": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ocn_model_create": (C.c_int, [_pp, _vp, C.c_int]),
    "ocn_model_destroy": (C.c_int, [_vp]),
    "ocn_model_field": (C.c_int, [_vp, C.c_char_p, _pp, _ip]),
    "ocn_model_update_state": (C.c_int, [_vp, C.c_int]),
    "ocn_model_set_finalize": (C.c_int, [_vp, C.c_int]),
    "ocn_model_time_step": (C.c_int, [_vp, C.c_double]),
    "ocn_model_clock": (C.c_int, [_vp, _dp, C.POINTER(C.c_int64), _ip, _dp, _dp]),
    "ocn_model_set_clock": (C.c_int, [_vp, C.c_double, C.c_int64, C.c_int, C.c_double, C.c_double]),
    "ocn_model_max_abs_divergence": (C.c_int, [_vp, _dp]),
    "ocn_max_abs_divergence": (C.c_int, [_vp, _vp, _vp, _vp, _dp]),
    "ocn_model_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "ocn_model_profile_read": (C.c_int, [_vp, _dp, _ip]),
    "ocn_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "ocn_debug_rcp_check": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]),
    "ocn_debug_permute_indices": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "ocn_debug_fft_fallbacks": (C.c_int, []),
}


LOADED_BEFORE_TORCH = False


def lib():
    global _lib, LOADED_BEFORE_TORCH
    if _lib is None:
        import sys
        LOADED_BEFORE_TORCH = "torch" not in sys.modules
        if not os.path.exists(SO_PATH):
            raise OcnError(f"HIP extension {SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback in the product path)")
        _lib = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(_lib, name)        # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
    return _lib


def check(status):
    if status != 0:
        raise OcnError(f"libocn_mi355x status {status}: {lib().ocn_last_error().decode()}")


def i3(t):
    return (C.c_int * 3)(*[int(x) for x in t])
