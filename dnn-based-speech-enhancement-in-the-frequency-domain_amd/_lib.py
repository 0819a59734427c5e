"""ctypes binding of include/sefd.h.  There is NO CPU fallback: a missing library is a hard error."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SEFD_LIB_PATH: another build of the same library (tuning A/B on one box: tools/ab_build.sh); the product loads the in-tree file
LIB_PATH = os.environ.get("SEFD_LIB_PATH") or os.path.join(HERE, "libsefd_hip.so")


class ModelConfig(C.Structure):
    """Mirror of `sefd_model_config` (include/sefd.h)."""
    _fields_ = [("model", C.c_int32), ("B", C.c_int32), ("L", C.c_int32),
                ("win_len", C.c_int32), ("hop", C.c_int32), ("fft_len", C.c_int32),
                ("n_layers", C.c_int32), ("kernel_num", C.c_int32 * 12),
                ("rnn_layers", C.c_int32), ("rnn_units", C.c_int32), ("mask_mode", C.c_int32),
                ("lstm_complex", C.c_int32), ("skip", C.c_int32), ("act_dtype", C.c_int32),
                ("kernel_size", C.c_int32), ("training", C.c_int32), ("bn_world", C.c_int32), ("grad_buckets", C.c_int32), ("use_cbn", C.c_int32), ("window", C.c_int32),
                ("pad_", C.c_int32), ("window_values", C.POINTER(C.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). This package has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, cp = C.c_void_p, C.c_int32, C.c_int64, C.c_char_p
    sig = {
        "sefd_plan_create": (vp, [C.POINTER(ModelConfig)]),
        "sefd_plan_destroy": (None, [vp]),
        "sefd_plan_error": (cp, [vp]),
        "sefd_plan_arena_bytes": (i64, [vp, i32]),
        "sefd_plan_frames": (i32, [vp]),
        "sefd_plan_num_syncs": (i32, [vp]),
        "sefd_plan_sync": (i32, [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]),
        "sefd_plan_num_params": (i32, [vp, i32]),
        "sefd_plan_param_name": (cp, [vp, i32, i32]),
        "sefd_plan_param_offset": (i64, [vp, i32, i32]),
        "sefd_plan_param_numel": (i64, [vp, i32, i32]),
        "sefd_plan_param_shape": (i32, [vp, i32, i32, C.POINTER(i64)]),
        "sefd_plan_buffer": (i32, [vp, cp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]),
        "sefd_plan_num_buffers": (i32, [vp]),
        "sefd_plan_buffer_name": (cp, [vp, i32]),
        "sefd_plan_const_data": (vp, [vp]),
        "sefd_plan_num_ops": (i32, [vp, i32]),
        "sefd_plan_ops": (vp, [vp, i32]),
        "sefd_op_size": (i32, []),
        "sefd_plan_op_info": (i32, [vp, i32, i32, C.POINTER(i64)]),
        "sefd_plan_run": (i32, [vp, i32, i32, i32, C.POINTER(vp), vp]),
        "sefd_plan_grad_bucket": (i32, [vp, C.POINTER(i32), C.POINTER(i64)]),
        "sefd_plan_grad_bucket_range": (i32, [vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]),
        "sefd_plan_run_cb": (i32, [vp, i32, C.POINTER(vp), vp, i32, vp, vp]),
        "sefd_plan_run_timed": (i32, [vp, i32, C.POINTER(vp), vp, C.POINTER(C.c_float), i32]),
        "sefd_plan_run_flags": (i32, [vp, i32, C.POINTER(vp), vp, i32, i32, vp, vp]),
        "sefd_loss_ws_floats": (i64, [i32]),
        "sefd_loss_forward": (i32, [i32, vp, vp, i32, i32, vp, vp, vp]),
        "sefd_loss_backward": (i32, [i32, vp, vp, i32, i32, vp, vp, vp, vp]),
        "sefd_loss_dp_offset": (i64, [i64, i32]),
        "sefd_loss_dp_finish": (i32, [i32, i64, vp, i32, vp, vp]),
        "sefd_loss_rows_ws_floats": (i64, [i64]),
        "sefd_loss_rows_forward": (i32, [i32, vp, vp, i64, i32, vp, vp, vp]),
        "sefd_loss_rows_backward": (i32, [i32, vp, vp, i64, i32, vp, vp, vp, vp, vp]),
        "sefd_lms_forward": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32), i32, i32, vp, vp, vp]),
        "sefd_lms_backward": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32), i32, i32, vp, vp, vp, vp]),
        "sefd_fsn_targets": (i32, [vp, vp, i64, vp, vp, vp, vp]),
        "sefd_mix_snr": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
        "sefd_pmsqe_table_floats": (i64, []),
        "sefd_pmsqe_ws_floats": (i64, [i32, i32]),
        "sefd_pmsqe_forward": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
        "sefd_pmsqe_backward": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "sefd_adam_step": (i32, [vp, vp, vp, vp, i64, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
        "sefd_adam_step_guarded": (i32, [vp, vp, vp, vp, i64, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]),
        "sefd_adam_step_guarded_dp": (i32, [vp, vp, vp, vp, i64, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp]),
        "sefd_plan_status_poison": (i32, [vp, vp, vp]),
        "sefd_plan_status_word": (vp, [vp]),
        "sefd_plan_status": (i32, [vp, i32]),
        "sefd_plan_status_set": (i32, [vp]),
        "sefd_tuning_set": (None, [C.c_char_p, C.c_char_p]),
        "sefd_tuning_get": (C.c_char_p, [C.c_char_p]),
        "sefd_tuning_clear": (None, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED = ["sefd_plan_create", "sefd_plan_destroy", "sefd_plan_error", "sefd_plan_arena_bytes", "sefd_plan_frames",
            "sefd_plan_num_syncs", "sefd_plan_sync",
            "sefd_plan_num_params", "sefd_plan_param_name", "sefd_plan_param_offset", "sefd_plan_param_numel",
            "sefd_plan_param_shape", "sefd_plan_buffer", "sefd_plan_num_buffers", "sefd_plan_buffer_name",
            "sefd_plan_const_data", "sefd_plan_num_ops", "sefd_plan_ops", "sefd_op_size", "sefd_plan_op_info", "sefd_plan_run",
            "sefd_plan_grad_bucket", "sefd_plan_grad_bucket_range", "sefd_plan_run_cb", "sefd_plan_run_flags", "sefd_plan_run_timed",
            "sefd_loss_ws_floats", "sefd_loss_forward", "sefd_loss_backward", "sefd_loss_rows_ws_floats", "sefd_loss_rows_forward", "sefd_loss_rows_backward", "sefd_loss_dp_offset", "sefd_loss_dp_finish", "sefd_lms_forward", "sefd_lms_backward", "sefd_fsn_targets", "sefd_mix_snr", "sefd_pmsqe_table_floats", "sefd_pmsqe_ws_floats", "sefd_pmsqe_forward", "sefd_pmsqe_backward",
            "sefd_adam_step", "sefd_adam_step_guarded", "sefd_adam_step_guarded_dp", "sefd_plan_status_poison", "sefd_plan_status_word", "sefd_plan_status", "sefd_plan_status_set",
            "sefd_tuning_set", "sefd_tuning_get", "sefd_tuning_clear"]
