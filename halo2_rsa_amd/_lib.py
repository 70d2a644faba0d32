"""ctypes binding of libh2r.so (the C ABI declared in include/h2r.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises.
"""
import ctypes
import os

from . import _build

H2R_PL_COUNT = 29
PLANES = ["Q", "R", "Q_SUB", "R_SUB", "AB_LO", "AB_HI", "QN_LO", "QN_HI", "EQB_LO", "EQB_HI", "AMB_LO", "AMB_HI",
          "SUM_LO", "SUM_HI", "CARRY", "CMOD", "NQ1_LO", "NQ1_HI", "AMNQ1", "ACCX_LO", "ACCX_HI", "QACC", "MODACC",
          "NQ2_LO", "NQ2_HI", "AMNQ2", "FLAGS", "CARRY_DUP", "CARRY_SUB"]
assert len(PLANES) == H2R_PL_COUNT

H2R_OK, H2R_E_SHAPE, H2R_E_ZERO_MODULUS, H2R_E_NOT_REDUCED, H2R_E_FIELD_TOO_SMALL, H2R_E_HIP, H2R_E_UNSUPPORTED, \
    H2R_E_NULL, H2R_E_NOT_IN_FIELD, H2R_E_ASSERTION = range(10)
FIELDS = {"bn254_fr": 0, "bn254_fq": 1, "pasta_fp": 2, "pasta_fq": 3}
H2R_F_SHARED_MODULUS = 1


class H2RParams(ctypes.Structure):
    _fields_ = [("limb_width", ctypes.c_uint32), ("bits_len", ctypes.c_uint32), ("field", ctypes.c_uint32),
                ("device", ctypes.c_int32)]


class H2RAdviceRepr(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("col_stride", ctypes.c_uint64)]


H2R_ADVICE_COLUMNS, H2R_ADVICE_MONTGOMERY = 0x400, 0x800
H2R_VERSION = 4


class H2RPipelineInfo(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("depth", ctypes.c_uint32), ("side_streams", ctypes.c_uint32), ("record_form", ctypes.c_uint32),
                ("three_queues", ctypes.c_uint32), ("probe_ms", ctypes.c_float), ("probe_span_ms", ctypes.c_float)]


H2R_PIPE_ONE_LAUNCH_STEP, H2R_PIPE_TWO_QUEUE, H2R_PIPE_SIDE_STREAM = 0, 1, 2


class H2RLayout(ctypes.Structure):
    _fields_ = [("limb_width", ctypes.c_uint32), ("num_limbs", ctypes.c_uint32), ("num_cols", ctypes.c_uint32),
                ("limb_bytes", ctypes.c_uint32), ("wide_bytes", ctypes.c_uint32), ("carry_bytes", ctypes.c_uint32),
                ("limb_sub_bits", ctypes.c_uint32), ("limb_nsub", ctypes.c_uint32),
                ("carry_bits", ctypes.c_uint32), ("carry_sub_bits", ctypes.c_uint32), ("carry_nsub", ctypes.c_uint32),
                ("carry_sub_stride", ctypes.c_uint32), ("word_max_bits", ctypes.c_uint32), ("reserved0", ctypes.c_uint32),
                ("record_stride", ctypes.c_uint64), ("stream_bytes", ctypes.c_uint64),
                ("plane_off", ctypes.c_uint64 * H2R_PL_COUNT), ("plane_elem", ctypes.c_uint32 * H2R_PL_COUNT),
                ("plane_count", ctypes.c_uint32 * H2R_PL_COUNT),
                ("acc_steps_per_group", ctypes.c_uint32), ("acc_lo_row_bytes", ctypes.c_uint32),
                ("acc_lo_group_bytes", ctypes.c_uint64), ("acc_hi_group_bytes", ctypes.c_uint64)]


class H2RPowLayout(ctypes.Structure):
    _fields_ = [("num_mul_mods", ctypes.c_uint32), ("num_exp_bits", ctypes.c_uint32),
                ("elem_stride", ctypes.c_uint64), ("off_records", ctypes.c_uint64), ("off_result", ctypes.c_uint64),
                ("off_e_bits", ctypes.c_uint64), ("off_selected", ctypes.c_uint64), ("selected_stride", ctypes.c_uint64),
                ("stream_bytes", ctypes.c_uint64), ("exp_limb_bits", ctypes.c_uint32), ("e_num_limbs", ctypes.c_uint32)]


class H2RCopy(ctypes.Structure):
    _fields_ = [("row", ctypes.c_uint32), ("col", ctypes.c_uint32), ("src_row", ctypes.c_uint32), ("src_col", ctypes.c_uint32)]


class H2RAdviceLayout(ctypes.Structure):
    _fields_ = [("version", ctypes.c_uint32), ("column_of", (ctypes.c_uint8 * 5) * 256)]


H2R_COPY_SRC_A, H2R_COPY_SRC_B, H2R_COPY_SRC_N = 0xFFFFFF01, 0xFFFFFF02, 0xFFFFFF03
H2R_SRC_X, H2R_SRC_ONE = -1, -2


class H2RVerifyLayout(ctypes.Structure):
    _fields_ = [("pow", H2RPowLayout), ("off_in_field", ctypes.c_uint64), ("in_field_stream_bytes", ctypes.c_uint64),
                ("off_em", ctypes.c_uint64), ("em_stream_bytes", ctypes.c_uint64), ("elem_stride", ctypes.c_uint64),
                ("stream_bytes", ctypes.c_uint64)]


class H2RLookupConfig(ctypes.Structure):
    _fields_ = [("n_lens", ctypes.c_uint32), ("bit_len", ctypes.c_uint32 * 8), ("tag", ctypes.c_uint32 * 8),
                ("row_off", ctypes.c_uint32 * 8), ("n_rows", ctypes.c_uint32)]


class H2RFixedRow(ctypes.Structure):
    NAMES = ("sa", "sb", "sc", "sd", "se", "s_mul_ab", "s_mul_cd", "se_next", "s_const")
    _fields_ = [(nm, ctypes.c_uint64 * 4) for nm in NAMES] + [("tag_composition", ctypes.c_uint32), ("tag_overflow", ctypes.c_uint32)]

    def as_dict(self):
        d = {nm: sum(int(getattr(self, nm)[k]) << (64 * k) for k in range(4)) for nm in self.NAMES}
        d["tag_composition"], d["tag_overflow"] = int(self.tag_composition), int(self.tag_overflow)
        return d


class H2RError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().h2r_status_str(code).decode()
        if code == H2R_E_HIP:
            msg += " (" + lib().h2r_last_hip_error().decode() + ")"
        super().__init__("%s: h2r status %d: %s" % (where, code, msg))


_lib = None
EXPORTS = ["h2r_ctx_create", "h2r_ctx_create_ex", "h2r_ctx_advice_repr", "h2r_abi_version", "h2r_build_id", "h2r_ctx_destroy", "h2r_compute_range_lens", "h2r_rsa_compute_range_lens",
           "h2r_trace_layout", "h2r_pow_fixed_layout", "h2r_pow_var_layout", "h2r_workspace_bytes",
           "h2r_mul_mod_batch", "h2r_square_mod_batch", "h2r_pow_mod_fixed_exp_batch", "h2r_pow_mod_batch",
           "h2r_modpow_public_key_batch", "h2r_modpow_public_key_var_batch", "h2r_pipeline_create", "h2r_pipeline_create_ex", "h2r_pipeline_destroy",
           "h2r_pipeline_modpow_public_key", "h2r_pipeline_modpow_public_key_advice", "h2r_pipeline_modpow_public_key_var", "h2r_pipeline_verify_pkcs1v15", "h2r_pipeline_join", "h2r_pipeline_info", "h2r_pipeline_call_plan", "h2r_exp_segment_plan", "h2r_arena_create", "h2r_arena_create_ex", "h2r_image_arena_create", "h2r_arena_region", "h2r_arena_region_bytes", "h2r_arena_region_ms", "h2r_arena_measurements", "h2r_arena_destroy", "h2r_verify_layout_fixed", "h2r_verify_pkcs1v15_batch",
           "h2r_verify_trace_flatten", "h2r_fresh_op_layout", "h2r_fresh_op_batch", "h2r_fresh_op_flatten",
           "h2r_mul_stream_bytes", "h2r_is_equal_muled_stream_bytes", "h2r_refresh_stream_bytes", "h2r_mul_batch",
           "h2r_mul_trace_flatten", "h2r_is_equal_muled_batch", "h2r_is_equal_muled_flatten", "h2r_refresh_batch",
           "h2r_mul_stream_bytes_ex", "h2r_mul_batch_ex", "h2r_mul_trace_flatten_ex", "h2r_refresh_layout", "h2r_refresh_batch_ex",
           "h2r_is_equal_muled_stream_bytes_ex", "h2r_is_equal_muled_batch_ex",
           "h2r_range_decompose_batch", "h2r_hist_len", "h2r_trace_lookup_hist", "h2r_lookups_per_record",
           "h2r_trace_lookup_permutation", "h2r_trace_lookup_permutation_hist",
           "h2r_trace_flatten", "h2r_pow_trace_flatten", "h2r_stream_bytes", "h2r_pow_stream_bytes", "h2r_trace_flatten_ex",
           "h2r_pow_trace_flatten_ex", "h2r_trace_emit_stream", "h2r_pow_trace_emit_stream", "h2r_mul_mod_trace_check", "h2r_pow_trace_check",
           "h2r_modpow_public_key_advice_rows", "h2r_modpow_public_key_emit_advice",
           "h2r_advice_copy_map", "h2r_pow_operand_sources", "h2r_advice_layout_default", "h2r_advice_layout_custom", "h2r_advice_fixed_row_ex",
           "h2r_advice_apply_layout", "h2r_advice_check", "h2r_pow_copy_map", "h2r_lookup_hist_advice",
           "h2r_advice_rows", "h2r_mul_mod_emit_advice", "h2r_pow_trace_emit_advice", "h2r_pow_advice_rows", "h2r_pow_row_kinds", "h2r_advice_row_kinds",
           "h2r_advice_fixed_row", "h2r_fresh_op_advice_rows", "h2r_fresh_op_row_kinds", "h2r_fresh_op_emit_advice",
           "h2r_verify_advice_rows", "h2r_verify_row_kinds", "h2r_verify_emit_advice", "h2r_verify_layout_compact", "h2r_pipeline_verify_pkcs1v15_advice", "h2r_pow_layout_compact", "h2r_pipeline_modpow_public_key_var_advice", "h2r_pipeline_verify_pkcs1v15_var_advice", "h2r_verify_layout_var", "h2r_verify_pkcs1v15_var_batch", "h2r_pipeline_verify_pkcs1v15_var",
           "h2r_sha256_hashed_msg_batch", "h2r_signature_verifier_batch", "h2r_pipeline_signature_verifier", "h2r_hashed_msg_advice_rows", "h2r_hashed_msg_row_kinds",
           "h2r_hashed_msg_emit_advice",
           "h2r_lookup_config_default", "h2r_lookup_config_custom", "h2r_lookup_table_image", "h2r_lookup_hist_records",
           "h2r_lookup_hist_values", "h2r_lookup_hist_values_strided", "h2r_lookup_hist_verify", "h2r_lookup_hist_fresh_op", "h2r_lookup_workspace_bytes", "h2r_lookup_permuted_columns", "h2r_field_eval",
           "h2r_dist_unique_id", "h2r_dist_init", "h2r_dist_destroy", "h2r_dist_rank", "h2r_dist_world", "h2r_dist_version", "h2r_dist_shard_range",
           "h2r_dist_bcast", "h2r_dist_gather_results", "h2r_dist_allreduce_max_f64",
           "h2r_profile_enable", "h2r_profile_read", "h2r_status_str",
           "h2r_last_hip_error"]
H2R_ADVICE_ASSERT_ONE = 0x100
H2R_ADVICE_DIRECT = 0x200
KERNEL_CHAIN, KERNEL_TRACE, KERNEL_HIST, KERNEL_AUX, KERNEL_EMIT, KERNEL_STEP, KERNEL_LOOKUP, KERNEL_SHA256, KERNEL_CELLS = 0, 1, 2, 3, 4, 5, 6, 7, 8
H2R_HASHED_MSG_STREAM_BYTES = 288
H2R_STREAM_FIELD_AB = 1
FRESH_OPS = ["add", "sub", "add_mod", "sub_mod", "is_zero", "is_equal_fresh", "is_less_than", "is_less_than_or_equal",
             "is_greater_than", "is_greater_than_or_equal", "is_in_field"]


def lib_path():
    # H2R_LIB: developer override used by tools/ for same-box A/B runs of build variants
    return os.environ.get("H2R_LIB") or _build.LIB


def lib():
    """Load libh2r.so.  `import torch` must happen first so that the HIP runtime torch ships
    (same SONAME libamdhip64.so.7) is the one both share."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("libh2r.so is not built (%s). Run `python -m halo2_rsa_amd._build`; there is no CPU fallback." % path)
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except Exception:  # pragma: no cover - torch-less use links /opt/rocm's runtime
        pass
    L = ctypes.CDLL(path)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32
    L.h2r_ctx_create.argtypes = [ctypes.POINTER(H2RParams), ctypes.POINTER(vp)]
    L.h2r_ctx_create_ex.argtypes = [ctypes.POINTER(H2RParams), ctypes.POINTER(H2RAdviceRepr), ctypes.POINTER(vp)]
    L.h2r_ctx_advice_repr.argtypes = [vp, ctypes.POINTER(H2RAdviceRepr)]
    L.h2r_abi_version.argtypes = []
    L.h2r_abi_version.restype = u32
    L.h2r_build_id.argtypes = []
    L.h2r_build_id.restype = ctypes.c_char_p
    L.h2r_ctx_destroy.argtypes = [vp]
    L.h2r_ctx_destroy.restype = None
    L.h2r_compute_range_lens.argtypes = [u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.h2r_rsa_compute_range_lens.argtypes = [u32, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.h2r_trace_layout.argtypes = [vp, ctypes.POINTER(H2RLayout)]
    L.h2r_pow_fixed_layout.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(H2RPowLayout)]
    L.h2r_pow_var_layout.argtypes = [vp, u32, u32, ctypes.POINTER(H2RPowLayout)]
    L.h2r_workspace_bytes.argtypes = [vp, u64, u32]
    L.h2r_workspace_bytes.restype = u64
    L.h2r_mul_mod_batch.argtypes = [vp, vp, vp, vp, u64, u32, vp, vp, vp, vp, vp]
    L.h2r_square_mod_batch.argtypes = [vp, vp, vp, u64, u32, vp, vp, vp, vp, vp]
    L.h2r_pow_mod_fixed_exp_batch.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_size_t, u64, u32, vp, vp, vp, vp, vp]
    L.h2r_modpow_public_key_batch.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_size_t, u64, u32, vp, vp, vp, vp, vp, vp]
    L.h2r_pow_mod_batch.argtypes = [vp, vp, vp, u32, u32, vp, u64, u32, vp, vp, vp, vp, vp]
    L.h2r_modpow_public_key_var_batch.argtypes = [vp, vp, vp, u32, u32, vp, u64, u32, vp, vp, vp, vp, vp, vp]
    L.h2r_pipeline_create.argtypes = [vp, ctypes.POINTER(vp)]
    L.h2r_pipeline_create_ex.argtypes = [vp, u32, u32, ctypes.POINTER(vp)]
    L.h2r_pipeline_destroy.argtypes = [vp]
    L.h2r_pipeline_destroy.restype = None
    L.h2r_pipeline_modpow_public_key.argtypes = L.h2r_modpow_public_key_batch.argtypes
    L.h2r_pipeline_modpow_public_key_advice.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_size_t, u64, u32, vp, vp, vp, vp, vp, u64, vp]
    L.h2r_pipeline_modpow_public_key_var.argtypes = [vp, vp, vp, u32, u32, vp, u64, u32, vp, vp, vp, vp, vp, vp]
    L.h2r_pow_layout_compact.argtypes = [vp, ctypes.POINTER(H2RPowLayout), ctypes.POINTER(H2RPowLayout)]
    L.h2r_pipeline_modpow_public_key_var_advice.argtypes = [vp, vp, vp, u32, u32, vp, u64, u32, vp, vp, vp, vp, vp, vp, u64, vp]
    L.h2r_pipeline_verify_pkcs1v15_var_advice.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, u32, vp, vp, vp, vp, vp, vp, u64, vp]
    L.h2r_lookup_hist_advice.argtypes = [vp, vp, vp, vp, u64, vp, u64, u64, vp, vp, vp]
    L.h2r_verify_layout_compact.argtypes = [vp, ctypes.POINTER(H2RVerifyLayout), ctypes.POINTER(H2RVerifyLayout)]
    L.h2r_pipeline_verify_pkcs1v15_advice.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_size_t, vp, u64, u32, vp, vp, vp, vp, vp, vp, u64, vp]
    L.h2r_pipeline_join.argtypes = [vp, vp]
    L.h2r_pipeline_info.argtypes = [vp, vp, u64, ctypes.POINTER(H2RPipelineInfo)]
    L.h2r_arena_create.argtypes = [vp, u64, u64, u32, u64, u32, u32, vp, ctypes.POINTER(vp)]
    L.h2r_arena_create_ex.argtypes = [vp, u64, u64, u32, u64, u32, u32, u64, vp, ctypes.POINTER(vp)]
    L.h2r_image_arena_create.argtypes = [vp, u64, u32, u32, u64, vp, ctypes.POINTER(vp)]
    L.h2r_dist_version.argtypes = []
    L.h2r_dist_version.restype = ctypes.c_int32
    L.h2r_arena_region.argtypes = [vp, u32]
    L.h2r_arena_region.restype = vp
    L.h2r_arena_region_bytes.argtypes = [vp]
    L.h2r_arena_region_bytes.restype = u64
    L.h2r_arena_region_ms.argtypes = [vp, u32]
    L.h2r_arena_region_ms.restype = ctypes.c_double
    L.h2r_arena_measurements.argtypes = [vp, ctypes.POINTER(ctypes.c_double), u32]
    L.h2r_arena_measurements.restype = u32
    L.h2r_arena_destroy.argtypes = [vp]
    L.h2r_arena_destroy.restype = None
    L.h2r_exp_segment_plan.argtypes = [vp, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    L.h2r_pipeline_call_plan.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32,
                                         ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.h2r_verify_layout_fixed.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(H2RVerifyLayout)]
    L.h2r_verify_pkcs1v15_batch.argtypes = [vp, vp, vp, ctypes.c_char_p, ctypes.c_size_t, vp, u64, u32, vp, vp, vp, vp, vp, vp]
    L.h2r_verify_trace_flatten.argtypes = [vp, ctypes.POINTER(H2RVerifyLayout), vp, vp]
    L.h2r_verify_layout_var.argtypes = [vp, u32, u32, ctypes.POINTER(H2RVerifyLayout)]
    L.h2r_verify_pkcs1v15_var_batch.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, u32, vp, vp, vp, vp, vp, vp]
    L.h2r_pipeline_verify_pkcs1v15_var.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, u32, vp, vp, vp, vp, vp, vp]
    L.h2r_pipeline_verify_pkcs1v15.argtypes = L.h2r_verify_pkcs1v15_batch.argtypes
    L.h2r_fresh_op_layout.argtypes = [vp, u32, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u32)]
    L.h2r_fresh_op_batch.argtypes = [vp, u32, vp, vp, vp, u64, u32, vp, vp, vp, vp, vp]
    L.h2r_fresh_op_flatten.argtypes = [vp, u32, vp, vp]
    for nm in ("h2r_mul_stream_bytes", "h2r_is_equal_muled_stream_bytes", "h2r_refresh_stream_bytes"):
        getattr(L, nm).argtypes = [vp]
        getattr(L, nm).restype = u64
    L.h2r_mul_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp]
    L.h2r_mul_trace_flatten.argtypes = [vp, vp, vp]
    L.h2r_is_equal_muled_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp]
    L.h2r_is_equal_muled_flatten.argtypes = [vp, vp, vp]
    L.h2r_refresh_batch.argtypes = [vp, vp, u64, vp, vp, vp, vp]
    L.h2r_mul_stream_bytes_ex.argtypes = [vp, u32, u32]
    L.h2r_mul_stream_bytes_ex.restype = u64
    L.h2r_mul_batch_ex.argtypes = [vp, vp, u32, vp, u32, u64, vp, vp, vp]
    L.h2r_mul_trace_flatten_ex.argtypes = [vp, vp, u32, u32, vp]
    L.h2r_refresh_layout.argtypes = [vp, u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.h2r_refresh_batch_ex.argtypes = [vp, vp, u64, u32, u32, u64, vp, vp, vp, vp]
    L.h2r_is_equal_muled_stream_bytes_ex.argtypes = [vp, u32, u32, u32]
    L.h2r_is_equal_muled_stream_bytes_ex.restype = u64
    L.h2r_is_equal_muled_batch_ex.argtypes = [vp, vp, vp, u64, u32, u32, u64, u32, vp, u64, vp, vp]
    L.h2r_range_decompose_batch.argtypes = [vp, vp, u32, u64, u32, u32, vp, u32, vp, vp]
    L.h2r_hist_len.argtypes = [vp]
    L.h2r_hist_len.restype = u32
    L.h2r_trace_lookup_hist.argtypes = [vp, vp, u64, u64, u64, u32, vp, vp]
    L.h2r_lookups_per_record.argtypes = [vp]
    L.h2r_lookups_per_record.restype = u32
    L.h2r_trace_lookup_permutation.argtypes = [vp, vp, u64, u64, u64, u32, vp, vp, vp]
    L.h2r_trace_lookup_permutation_hist.argtypes = [vp, vp, u64, u64, u64, u32, vp, vp, vp, vp]
    L.h2r_trace_flatten.argtypes = [vp, vp, vp]
    L.h2r_pow_trace_flatten.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp, vp]
    L.h2r_stream_bytes.argtypes = [vp, u32]
    L.h2r_stream_bytes.restype = u64
    L.h2r_pow_stream_bytes.argtypes = [vp, ctypes.POINTER(H2RPowLayout), u32]
    L.h2r_pow_stream_bytes.restype = u64
    L.h2r_trace_flatten_ex.argtypes = [vp, vp, u32, vp]
    L.h2r_pow_trace_flatten_ex.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp, u32, vp]
    L.h2r_trace_emit_stream.argtypes = [vp, vp, u64, u32, vp, u64, u64, vp]
    L.h2r_pow_trace_emit_stream.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp, u64, u64, u32, vp, u64, u64, vp]
    L.h2r_advice_rows.argtypes = [vp]
    L.h2r_advice_rows.restype = u32
    L.h2r_pow_advice_rows.argtypes = [vp, ctypes.POINTER(H2RPowLayout)]
    L.h2r_pow_advice_rows.restype = u64
    L.h2r_advice_row_kinds.argtypes = [vp, vp]
    L.h2r_advice_fixed_row.argtypes = [vp, ctypes.POINTER(H2RLookupConfig), u32, ctypes.POINTER(H2RFixedRow)]
    L.h2r_fresh_op_advice_rows.argtypes = [vp, u32, u32]
    L.h2r_fresh_op_advice_rows.restype = u32
    L.h2r_fresh_op_row_kinds.argtypes = [vp, u32, u32, vp]
    L.h2r_fresh_op_emit_advice.argtypes = [vp, u32, u32, vp, vp, vp, vp, u64, u64, u64, vp, vp, u64, vp]
    L.h2r_verify_advice_rows.argtypes = [vp, ctypes.POINTER(H2RVerifyLayout), vp]
    L.h2r_verify_advice_rows.restype = u64
    L.h2r_verify_row_kinds.argtypes = [vp, ctypes.POINTER(H2RVerifyLayout), vp]
    L.h2r_verify_emit_advice.argtypes = [vp, ctypes.POINTER(H2RVerifyLayout), vp, vp, vp, vp, u32, vp, vp, u64, vp, vp, u64, vp]
    L.h2r_sha256_hashed_msg_batch.argtypes = [vp, vp, vp, u64, u64, vp, vp, vp, u64, vp]
    L.h2r_signature_verifier_batch.argtypes = [vp, vp, vp, u64, vp, vp, ctypes.c_char_p, ctypes.c_size_t, u64, u32, vp, vp, u64, vp, vp, vp,
                                               vp, vp, vp, vp]
    L.h2r_pipeline_signature_verifier.argtypes = [vp, vp, vp, u64, vp, vp, ctypes.c_char_p, ctypes.c_size_t, u64, u32, vp, vp, u64, vp, vp, vp,
                                                  vp, vp, vp, vp]
    L.h2r_hashed_msg_advice_rows.argtypes = [vp]
    L.h2r_hashed_msg_advice_rows.restype = u32
    L.h2r_hashed_msg_row_kinds.argtypes = [vp, vp]
    L.h2r_hashed_msg_emit_advice.argtypes = [vp, vp, u64, u64, vp, vp, u64, vp]
    L.h2r_mul_mod_emit_advice.argtypes = [vp, vp, vp, vp, u32, vp, u64, vp, vp, u64, vp]
    L.h2r_pow_row_kinds.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp]
    L.h2r_advice_copy_map.argtypes = [vp, vp, u32]
    L.h2r_advice_copy_map.restype = u32
    L.h2r_pow_operand_sources.argtypes = [vp, ctypes.POINTER(H2RPowLayout), ctypes.c_char_p, ctypes.c_size_t, vp, vp]
    L.h2r_advice_layout_default.argtypes = [vp]
    L.h2r_advice_layout_custom.argtypes = [vp, vp, vp, u32, vp]
    L.h2r_advice_fixed_row_ex.argtypes = [vp, vp, vp, u32, vp]
    L.h2r_advice_apply_layout.argtypes = [vp, vp, vp, u64, vp, u64, u64, vp, vp]
    L.h2r_advice_check.argtypes = [vp, vp, vp, vp, u64, vp, u64, u64, vp, vp, u64, vp, vp, vp, u32, vp, vp, vp]
    L.h2r_pow_copy_map.argtypes = [vp, ctypes.POINTER(H2RPowLayout), ctypes.c_char_p, ctypes.c_size_t, u64, vp, u64]
    L.h2r_pow_copy_map.restype = u64
    L.h2r_modpow_public_key_advice_rows.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp]
    L.h2r_modpow_public_key_advice_rows.restype = u64
    L.h2r_modpow_public_key_emit_advice.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp, vp, u32, vp, vp, vp, u64, vp, vp, u64, vp]
    L.h2r_pow_trace_emit_advice.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp, u32, vp, u64, vp, u64, vp, vp, u64, vp]
    L.h2r_mul_mod_trace_check.argtypes = [vp, vp, vp, vp, u32, vp, u64, vp, vp, vp, vp]
    L.h2r_pow_trace_check.argtypes = [vp, ctypes.POINTER(H2RPowLayout), vp, vp, ctypes.c_char_p, ctypes.c_size_t, u32, vp, u64, vp, u64, vp,
                                      vp, vp, vp]
    pcfg, pu32, pu64 = ctypes.POINTER(H2RLookupConfig), ctypes.POINTER(u32), ctypes.POINTER(u64)
    L.h2r_lookup_config_default.argtypes = [vp, u32, pcfg]
    L.h2r_lookup_config_custom.argtypes = [pu32, pu32, u32, pcfg]
    L.h2r_lookup_table_image.argtypes = [vp, pcfg, vp, vp]
    L.h2r_lookup_hist_records.argtypes = [vp, pcfg, vp, u64, u64, u64, u32, vp, vp, vp]
    L.h2r_lookup_hist_values.argtypes = [vp, pcfg, vp, u32, u64, u64, u32, u32, vp, vp]
    L.h2r_lookup_hist_values_strided.argtypes = [vp, pcfg, vp, u32, u64, u64, u64, u64, u32, u32, vp, vp, vp]
    L.h2r_lookup_hist_verify.argtypes = [vp, pcfg, ctypes.POINTER(H2RVerifyLayout), vp, u64, vp, vp, vp]
    L.h2r_lookup_hist_fresh_op.argtypes = [vp, pcfg, u32, vp, u64, u64, u64, vp, vp]
    L.h2r_lookup_workspace_bytes.argtypes = [pcfg, u64]
    L.h2r_lookup_workspace_bytes.restype = u64
    L.h2r_lookup_permuted_columns.argtypes = [vp, pcfg, vp, vp, u64, u32, u32, vp, vp, u64, vp, vp, vp]
    L.h2r_field_eval.argtypes = [vp, u32, pu64, pu64, pu64]
    L.h2r_dist_unique_id.argtypes = [vp]
    L.h2r_dist_init.argtypes = [vp, vp, u32, u32, ctypes.POINTER(vp)]
    L.h2r_dist_destroy.argtypes = [vp]
    L.h2r_dist_destroy.restype = None
    L.h2r_dist_rank.argtypes = [vp]
    L.h2r_dist_rank.restype = u32
    L.h2r_dist_world.argtypes = [vp]
    L.h2r_dist_world.restype = u32
    L.h2r_dist_shard_range.argtypes = [u64, u32, u32, pu64, pu64]
    L.h2r_dist_bcast.argtypes = [vp, vp, u64, u32, vp]
    L.h2r_dist_gather_results.argtypes = [vp, vp, vp, u64, vp, vp, vp]
    L.h2r_dist_allreduce_max_f64.argtypes = [vp, vp, u64, vp]
    L.h2r_profile_enable.argtypes = [u32]
    L.h2r_profile_read.argtypes = [u32, ctypes.POINTER(ctypes.c_float), u32, ctypes.POINTER(u32)]
    L.h2r_status_str.argtypes = [i32]
    L.h2r_status_str.restype = ctypes.c_char_p
    L.h2r_last_hip_error.restype = ctypes.c_char_p
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int:
            fn.restype = i32
    _lib = L
    return _lib


def check(code, where):
    if code != H2R_OK:
        raise H2RError(code, where)


def profile_enable(capacity):
    check(lib().h2r_profile_enable(capacity), "h2r_profile_enable")


def profile_read(kernel, max_count=1 << 16):
    """Durations (ms) of the recorded launches of one kernel class, in launch order."""
    n = ctypes.c_uint32(0)
    check(lib().h2r_profile_read(kernel, None, 0, ctypes.byref(n)), "h2r_profile_read")
    k = min(n.value, max_count)
    buf = (ctypes.c_float * max(k, 1))()
    check(lib().h2r_profile_read(kernel, buf, k, ctypes.byref(n)), "h2r_profile_read")
    return [float(buf[i]) for i in range(k)]
