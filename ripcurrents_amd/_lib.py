"""ctypes binding of librcflow.so (the C ABI declared in include/rcflow.h).

The library is the product: there is no CPU fallback.  Loading fails loudly when the
built extension is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCFLOW_LIB selects another build of the same library (e.g. the diagnostic stamps build)
LIB_PATH = os.environ.get("RCFLOW_LIB") or os.path.join(_HERE, "librcflow.so")

RC_OK = 0
RC_FARNEBACK_GAUSSIAN = 256
HIST_BINS, HIST_DIRECTIONS, HIST_RESOLUTION = 50, 36, 20
HIST_WORDS = HIST_BINS + HIST_DIRECTIONS * HIST_BINS + 1 + HIST_DIRECTIONS
COMM_ID_BYTES = 128     # RC_COMM_ID_BYTES = sizeof(ncclUniqueId)

ERRORS = {-1: "RC_EINVAL", -2: "RC_ENOMEM", -3: "RC_EHIP", -4: "RC_ENODEV", -5: "RC_ESIZE",
          -6: "RC_ESTATE", -7: "RC_ECOMM"}


class RcflowError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "RC_E?"), code, text))
        self.code = code


class FarnebackParams(C.Structure):
    _fields_ = [("pyr_scale", C.c_double), ("levels", C.c_int), ("winsize", C.c_int),
                ("iterations", C.c_int), ("poly_n", C.c_int), ("poly_sigma", C.c_double),
                ("flags", C.c_int)]


class FrameLoop(C.Structure):
    """rc_frame_loop (include/rcflow.h): the per-frame analysis chain of rcflow_frame_loop_step."""
    _fields_ = [("dt", C.c_float), ("iterations", C.c_int), ("d_seeds", C.c_void_p), ("nseeds", C.c_int),
                ("seed_variant", C.c_int), ("seed_dt", C.c_float), ("seed_iterations", C.c_int), ("seed_upper", C.c_float),
                ("MID", C.c_float), ("LOWER", C.c_float), ("d_outmask", C.c_void_p), ("mask_step", C.c_size_t),
                ("d_edges", C.c_void_p), ("edges_step", C.c_size_t), ("use_graph", C.c_int)]


_vp, _sz, _i, _f, _d = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double
_pp = C.POINTER(FarnebackParams)

# name -> argtypes; every symbol include/rcflow.h declares (tests check the export list)
SIGNATURES = {
    "rcflow_create": [C.POINTER(_vp), _i, _i, _i, _i],
    "rcflow_destroy": [_vp],
    "rcflow_abi_version": [],
    "rcflow_last_error": [],
    "rcflow_sync": [_vp, _i],
    "rcflow_set_hip_stream": [_vp, _i, _vp],
    "rcflow_use_own_stream": [_vp, _i],
    "rcflow_set_option": [_vp, C.c_char_p, _i],
    "rcflow_farneback_u8": [_vp, _i, _vp, _sz, _vp, _sz, _i, _i, _vp, _sz, _d, _i, _i, _i, _i, _d, _i],
    "rcflow_farneback_dev": [_vp, _i, _vp, _sz, _vp, _sz, _i, _i, _vp, _sz, _pp],
    "rcflow_push_frame_dev": [_vp, _i, _vp, _sz, _i, _i, _vp, _sz, _pp],
    "rcflow_stream_reset": [_vp, _i],
    "rcflow_push_frame_u8": [_vp, _i, _vp, _sz, _i, _i, _pp],
    "rcflow_frame_buffer_acquire": [_vp, _i, _i, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
    "rcflow_push_frame_acquired": [_vp, _i, _pp],
    "rcflow_stream_flow_ptr": [_vp, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)],
    "rcflow_stream_flow_read": [_vp, _i, _vp, _sz],
    "rcflow_frame_loop_step": [_vp, _i, _pp, C.POINTER(FrameLoop)],
    "rcflow_farneback_clip_dev": [_vp, _i, _vp, _sz, _sz, _i, _i, _i, _vp, _sz, _sz, _pp],
    "rcflow_push_clip_dev": [_vp, _i, _vp, _sz, _sz, _i, _i, _i, _vp, _sz, _sz, _pp],
    "rcflow_push_batch_dev": [_vp, _i, _vp, _sz, _sz, _i, _i, _i, _vp, _sz, _sz, _pp, _i],
    "rcflow_batch_reset": [_vp, _i],
    "rcflow_level_geometry": [_i, _i, _d, _i, _i, C.POINTER(_i), C.POINTER(_i)],
    "rcflow_stage_pyr_level_dev": [_vp, _i, _vp, _sz, _i, _i, _d, _i, _vp],
    "rcflow_stage_polyexp_dev": [_vp, _i, _vp, _i, _i, _i, _d, _vp],
    "rcflow_stage_flow_iter_dev": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "rcflow_analysis_reset": [_vp, _i, _i, _i],
    "rcflow_histogram_dev": [_vp, _i, _vp, _sz, _i, _i],
    "rcflow_histogram_clip_dev": [_vp, _i, _vp, _sz, _sz, _i, _i, _i],
    "rcflow_thresholds_dev": [_vp, _i],
    "rcflow_thresholds_words_dev": [_vp, _i, _vp],
    "rcflow_histogram_read": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rcflow_histogram_write": [_vp, _i, _vp],
    "rcflow_histogram_reset_dev": [_vp, _i],
    "rcflow_histogram_device_ptr": [_vp, _i, C.POINTER(_vp)],
    "rcflow_classify_accumulate_dev": [_vp, _i, _vp, _sz, _i, _i, _i, _f, _f, _vp, _sz, _vp, _sz,
                                       _vp, _sz, _vp, _sz],
    "rcflow_accumulator_read": [_vp, _i, _vp],
    "rcflow_advect_field_dev": [_vp, _i, _vp, _sz, _i, _i, _f, _i, _f],
    "rcflow_advect_field_read": [_vp, _i, _vp, _vp],
    "rcflow_advect_points_dev": [_vp, _i, _vp, _i, _vp, _sz, _i, _i, _f, _i, _f, _i, _vp],
    "rcflow_get_delta_field_dev": [_vp, _i, _vp, _sz, _vp, _sz, _i, _i, _f, _f],
    "rcflow_subtract_average_dev": [_vp, _i, _vp, _sz, _i, _i],
    "rcflow_subtract_mean_magnitude_dev": [_vp, _i, _vp, _sz, _i, _i],
    "rcflow_stabilizer_dev": [_vp, _i, _vp, _sz, _i, _i],
    "rcflow_window_mean_dev": [_vp, _i, _vp, _vp, _vp, _sz, _i],
    "rcflow_vector_to_color_dev": [_vp, _i, _vp, _sz, _i, _i, _vp, _sz, C.POINTER(_f)],
    "rcflow_shear_rate_to_color_dev": [_vp, _i, _vp, _sz, _i, _i, _vp, _sz, C.POINTER(_f)],
    "rcflow_create_edges_dev": [_vp, _i, _vp, _sz, _i, _i, _vp, _sz],
    "rcflow_create_output_dev": [_vp, _i, _vp, _sz, _vp, _sz, _i, _i],
    "rcflow_resize_bgr_to_gray_dev": [_vp, _i, _vp, _sz, _i, _i, _vp, _sz, _i, _i],
    "rcflow_resize_area_bgr_to_gray_dev": [_vp, _i, _vp, _sz, _i, _i, _vp, _sz, _i, _i],
    "rcflow_streamline_display_dev": [_vp, _i, _i, _vp, _sz, C.POINTER(_f)],
    "rcflow_streamline_positions_dev": [_vp, _i, _vp, _sz],
    "rcflow_hsv_to_bgr_dev": [_vp, _i, _vp, _sz, _i, _i, _vp, _sz],
    "rcflow_jet_lut": [_vp],
    "rcflow_analysis_size": [_vp, _i, C.POINTER(_i), C.POINTER(_i)],
    "rcflow_pyrlk_dev": [_vp, _i, _vp, _sz, _vp, _sz, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _d],
    "rcflow_pyrlk_u8": [_vp, _i, _vp, _sz, _vp, _sz, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _d],
    "rcflow_pyrlk_levels": [_i, _i, _i, _i, _i],
    "rcflow_comm_unique_id": [_vp],
    "rcflow_comm_init": [_vp, _vp, _i, _i],
    "rcflow_comm_destroy": [_vp],
    "rcflow_comm_rank": [_vp, C.POINTER(_i), C.POINTER(_i)],
    "rcflow_allreduce_hist": [_vp, _i, _vp],
    "rcflow_allreduce_hist_join": [_vp, _i],
    "rcflow_allreduce_hist_status": [_vp, C.POINTER(C.c_longlong)],
    "rcflow_allreduce_hist_result": [_vp, C.POINTER(_vp)],
    "rcflow_profile_enable": [_vp, _i],
    "rcflow_profile_reset": [_vp],
    "rcflow_profile_read_buckets": [_vp, C.POINTER(C.c_char_p), C.POINTER(_d)],
    "rcflow_profile_read": [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i), C.POINTER(_d), C.POINTER(_d), C.POINTER(_d)],
}

_LIB = None


def load():
    """Loads librcflow.so; raises if the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "librcflow.so is missing at %s: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "rcflow_last_error" else C.c_int
    lib.rcflow_destroy.restype = None
    _LIB = lib
    return lib


def check(rc):
    if rc < 0:
        raise RcflowError(rc, (load().rcflow_last_error() or b"").decode())
    return rc
