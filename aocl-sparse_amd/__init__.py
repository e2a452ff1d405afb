"""ctypes view of libaoclsparse_mi355.so -- the C-ABI drop-in for the aoclsparse_* hot path.

The directory name carries a hyphen (repo contract), so import it through
``__graft_entry__.load_package()`` (importlib by path, module name ``aocl_sparse_amd``).

This module holds NO arithmetic: it declares the C signatures of include/aoclsparse.h and
include/aoclsparse_mi355.h and a few numpy/torch conveniences for tests and bench.py.  If the shared
library has not been built it raises -- there is no Python or CPU fallback for the product path.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_bool, c_char_p, c_double, c_float, c_int, c_int32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# AOCLSPARSE_MI355_LIB: load another build of the same library (e.g. a sanitizer build), never a different product
LIB_PATH = os.environ.get("AOCLSPARSE_MI355_LIB") or os.path.join(_HERE, "lib", "libaoclsparse_mi355.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

# enum values of include/aoclsparse.h
OP_NONE, OP_TRANSPOSE, OP_CONJ_TRANSPOSE = 111, 112, 113
BASE_ZERO, BASE_ONE = 0, 1
TYPE_GENERAL, TYPE_SYMMETRIC, TYPE_HERMITIAN, TYPE_TRIANGULAR = 0, 1, 2, 3
DIAG_NON_UNIT, DIAG_UNIT, DIAG_ZERO = 0, 1, 2
FILL_LOWER, FILL_UPPER = 0, 1
ORDER_ROW, ORDER_COLUMN = 0, 1
STAGE_NNZ_COUNT, STAGE_FINALIZE, STAGE_FULL = 0, 1, 2
MEM_MINIMAL, MEM_UNRESTRICTED = 0, 1
PTR_AUTO, PTR_HOST, PTR_DEVICE = 0, 1, 2
OPTION_SPMV_KERNEL, OPTION_SELL, OPTION_SPMV_STRICT, OPTION_ALTERNATE_SWEEPS, OPTION_TRSV_CHUNKS = 0, 1, 2, 3, 4  # aoclsparse_mi355_set_option

STATUS = {
    0: "success", 1: "not_implemented", 2: "invalid_pointer", 3: "invalid_size", 4: "internal_error",
    5: "invalid_value", 6: "invalid_index_value", 7: "maxit", 8: "user_stop", 9: "wrong_type",
    10: "memory_error", 11: "numerical_error", 12: "invalid_operation", 13: "unsorted_input",
    14: "invalid_kid",
}


class CFloat(ctypes.Structure):
    _fields_ = [("real", c_float), ("imag", c_float)]


class CDouble(ctypes.Structure):
    _fields_ = [("real", c_double), ("imag", c_double)]


class SpmvInfo(ctypes.Structure):
    _fields_ = [("kernel", c_int32), ("order", c_int32), ("row_blocks", c_int32), ("tile", c_int32),
                ("long_rows", c_int32), ("max_row_nnz", c_int32), ("device_resident", c_int32),
                ("sell_slices", c_int32), ("stored_cells", ctypes.c_longlong), ("mm_groups", c_int32),
                ("mm_window_rows", c_int32), ("mm_bell_width", c_int32), ("mm_bell_fill_permille", c_int32),
                ("tree_min", c_int32), ("mm_bell_xcd_chunk", c_int32), ("mm_bell_model_fetches_permille", c_int32),
                ("mm_bell_model_fetches_launch_order_permille", c_int32), ("mm_bell_lattice_line", c_int32),
                ("mm_bell_lattice_lines", c_int32), ("mm_bell_region_a", c_int32), ("mm_bell_region_b", c_int32)]


class TrsvInfo(ctypes.Structure):
    _fields_ = [(k, c_int32) for k in ("levels", "blocks", "block_levels", "chunks", "steps", "lds_slots", "model_chunk_us",
                                      "model_block_us", "schedule", "slices", "slice_fan_in_permille")]


MM_STATE_BUFFERS = 14


class MmState(ctypes.Structure):
    """aoclsparse_mi355_mm_state: sizes + scalars of a handle's analysed csrmm state (a POD that travels as bytes)"""
    _fields_ = [("scalars", ctypes.c_longlong * 40), ("bytes", ctypes.c_longlong * MM_STATE_BUFFERS)]


class CommId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]



# every exported symbol of include/*.h: name -> (restype, argtypes)
_I = c_int32
_P = c_void_p
SIGNATURES = {
    # auxiliary
    "aoclsparse_get_version": (c_char_p, []),
    "aoclsparse_enable_instructions": (c_int, [c_char_p]),
    "aoclsparse_debug_get": (c_int, [c_char_p, POINTER(_I), c_char_p, POINTER(c_bool), c_char_p]),
    "aoclsparse_is_avx512_build": (_I, []),
    "aoclsparse_create_mat_descr": (c_int, [POINTER(_P)]),
    "aoclsparse_copy_mat_descr": (c_int, [_P, _P]),
    "aoclsparse_destroy_mat_descr": (c_int, [_P]),
    "aoclsparse_set_mat_index_base": (c_int, [_P, c_int]),
    "aoclsparse_get_mat_index_base": (c_int, [_P]),
    "aoclsparse_set_mat_type": (c_int, [_P, c_int]),
    "aoclsparse_get_mat_type": (c_int, [_P]),
    "aoclsparse_set_mat_fill_mode": (c_int, [_P, c_int]),
    "aoclsparse_get_mat_fill_mode": (c_int, [_P]),
    "aoclsparse_set_mat_diag_type": (c_int, [_P, c_int]),
    "aoclsparse_get_mat_diag_type": (c_int, [_P]),
    "aoclsparse_create_scsr": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_create_dcsr": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_export_scsr": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I),
                                       POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_export_dcsr": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I),
                                       POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_destroy": (c_int, [POINTER(_P)]),
    "aoclsparse_sset_value": (c_int, [_P, _I, _I, c_float]),
    "aoclsparse_dset_value": (c_int, [_P, _I, _I, c_double]),
    "aoclsparse_supdate_values": (c_int, [_P, _I, _P]),
    "aoclsparse_dupdate_values": (c_int, [_P, _I, _P]),
    "aoclsparse_copy": (c_int, [_P, _P, POINTER(_P)]),
    # analysis
    "aoclsparse_optimize": (c_int, [_P]),
    "aoclsparse_set_mv_hint": (c_int, [_P, c_int, _P, _I]),
    "aoclsparse_set_mv_hint_kid": (c_int, [_P, c_int, _P, _I, _I]),
    "aoclsparse_set_sv_hint": (c_int, [_P, c_int, _P, _I]),
    "aoclsparse_set_mm_hint": (c_int, [_P, c_int, _P, _I]),
    "aoclsparse_set_2m_hint": (c_int, [_P, c_int, _P, _I]),
    "aoclsparse_set_memory_hint": (c_int, [_P, c_int]),
    # level 2
    "aoclsparse_scsrmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_dcsrmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_smv": (c_int, [c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_dmv": (c_int, [c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_sdotmv": (c_int, [c_int, c_float, _P, _P, _P, c_float, _P, _P]),
    "aoclsparse_ddotmv": (c_int, [c_int, c_double, _P, _P, _P, c_double, _P, _P]),
    "aoclsparse_strsv": (c_int, [c_int, c_float, _P, _P, _P, _P]),
    "aoclsparse_dtrsv": (c_int, [c_int, c_double, _P, _P, _P, _P]),
    "aoclsparse_strsv_kid": (c_int, [c_int, c_float, _P, _P, _P, _P, _I]),
    "aoclsparse_dtrsv_kid": (c_int, [c_int, c_double, _P, _P, _P, _P, _I]),
    "aoclsparse_strsv_strided": (c_int, [c_int, c_float, _P, _P, _P, _I, _P, _I]),
    "aoclsparse_dtrsv_strided": (c_int, [c_int, c_double, _P, _P, _P, _I, _P, _I]),
    # level 3
    "aoclsparse_strsm": (c_int, [c_int, c_float, _P, _P, c_int, _P, _I, _I, _P, _I]),
    "aoclsparse_dtrsm": (c_int, [c_int, c_double, _P, _P, c_int, _P, _I, _I, _P, _I]),
    "aoclsparse_strsm_kid": (c_int, [c_int, c_float, _P, _P, c_int, _P, _I, _I, _P, _I, _I]),
    "aoclsparse_dtrsm_kid": (c_int, [c_int, c_double, _P, _P, c_int, _P, _I, _I, _P, _I, _I]),
    "aoclsparse_create_ccsr": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_create_zcsr": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_export_ccsr": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_export_zcsr": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_cset_value": (c_int, [_P, _I, _I, CFloat]),
    "aoclsparse_zset_value": (c_int, [_P, _I, _I, CDouble]),
    "aoclsparse_cupdate_values": (c_int, [_P, _I, _P]),
    "aoclsparse_zupdate_values": (c_int, [_P, _I, _P]),
    "aoclsparse_ccsrmm": (c_int, [c_int, CFloat, _P, _P, c_int, _P, _I, _I, CFloat, _P, _I]),
    "aoclsparse_zcsrmm": (c_int, [c_int, CDouble, _P, _P, c_int, _P, _I, _I, CDouble, _P, _I]),
    "aoclsparse_ccsrmm_kid": (c_int, [c_int, CFloat, _P, _P, c_int, _P, _I, _I, CFloat, _P, _I, _I]),
    "aoclsparse_zcsrmm_kid": (c_int, [c_int, CDouble, _P, _P, c_int, _P, _I, _I, CDouble, _P, _I, _I]),
    "aoclsparse_ctrsv": (c_int, [c_int, CFloat, _P, _P, _P, _P]),
    "aoclsparse_ctrsv_kid": (c_int, [c_int, CFloat, _P, _P, _P, _P, _I]),
    "aoclsparse_ctrsv_strided": (c_int, [c_int, CFloat, _P, _P, _P, _I, _P, _I]),
    "aoclsparse_ctrsm": (c_int, [c_int, CFloat, _P, _P, c_int, _P, _I, _I, _P, _I]),
    "aoclsparse_ctrsm_kid": (c_int, [c_int, CFloat, _P, _P, c_int, _P, _I, _I, _P, _I, _I]),
    "aoclsparse_ztrsv": (c_int, [c_int, CDouble, _P, _P, _P, _P]),
    "aoclsparse_ztrsv_kid": (c_int, [c_int, CDouble, _P, _P, _P, _P, _I]),
    "aoclsparse_ztrsv_strided": (c_int, [c_int, CDouble, _P, _P, _P, _I, _P, _I]),
    "aoclsparse_ztrsm": (c_int, [c_int, CDouble, _P, _P, c_int, _P, _I, _I, _P, _I]),
    "aoclsparse_ztrsm_kid": (c_int, [c_int, CDouble, _P, _P, c_int, _P, _I, _I, _P, _I, _I]),
    "aoclsparse_scsrsv": (c_int, [c_int, _P, _I, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_dcsrsv": (c_int, [c_int, _P, _I, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_cdotmv": (c_int, [c_int, CFloat, _P, _P, _P, CFloat, _P, _P]),
    "aoclsparse_zdotmv": (c_int, [c_int, CDouble, _P, _P, _P, CDouble, _P, _P]),
    "aoclsparse_csymgs": (c_int, [c_int, _P, _P, CFloat, _P, _P]),
    "aoclsparse_csymgs_kid": (c_int, [c_int, _P, _P, CFloat, _P, _P, _I]),
    "aoclsparse_csymgs_mv": (c_int, [c_int, _P, _P, CFloat, _P, _P, _P]),
    "aoclsparse_csymgs_mv_kid": (c_int, [c_int, _P, _P, CFloat, _P, _P, _P, _I]),
    "aoclsparse_zsymgs": (c_int, [c_int, _P, _P, CDouble, _P, _P]),
    "aoclsparse_zsymgs_kid": (c_int, [c_int, _P, _P, CDouble, _P, _P, _I]),
    "aoclsparse_zsymgs_mv": (c_int, [c_int, _P, _P, CDouble, _P, _P, _P]),
    "aoclsparse_zsymgs_mv_kid": (c_int, [c_int, _P, _P, CDouble, _P, _P, _P, _I]),
    "aoclsparse_mi355_strsv_full": (c_int, [c_int, c_float, _P, _P, _P, _I, _P, _I, _I]),
    "aoclsparse_mi355_dtrsv_full": (c_int, [c_int, c_double, _P, _P, _P, _I, _P, _I, _I]),
    "aoclsparse_mi355_ctrsv_full": (c_int, [c_int, CFloat, _P, _P, _P, _I, _P, _I, _I]),
    "aoclsparse_mi355_ztrsv_full": (c_int, [c_int, CDouble, _P, _P, _P, _I, _P, _I, _I]),
    "aoclsparse_cmv": (c_int, [c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_zmv": (c_int, [c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_create_scsc": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_create_dcsc": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_create_scoo": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_create_dcoo": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_export_scsc": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_export_dcsc": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_export_scoo": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_export_dcoo": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_create_ccsc": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_create_ccoo": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_export_ccsc": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_export_ccoo": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_ccsr2csc": (c_int, [_I, _I, _I, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_create_zcsc": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_create_zcoo": (c_int, [POINTER(_P), c_int, _I, _I, _I, _P, _P, _P]),
    "aoclsparse_export_zcsc": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_export_zcoo": (c_int, [_P, POINTER(c_int), POINTER(_I), POINTER(_I), POINTER(_I), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "aoclsparse_zcsr2csc": (c_int, [_I, _I, _I, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_convert_csr": (c_int, [_P, c_int, POINTER(_P)]),
    "aoclsparse_order_mat": (c_int, [_P]),
    "aoclsparse_scsr2csc": (c_int, [_I, _I, _I, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_dcsr2csc": (c_int, [_I, _I, _I, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_sellmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "aoclsparse_dellmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "aoclsparse_selltmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "aoclsparse_delltmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "aoclsparse_sellthybmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_dellthybmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_dblkcsrmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I]),
    "aoclsparse_opt_blksize": (_I, [_I, _I, c_int, _P, _P, _P]),
    "aoclsparse_csr2blkcsr": (c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, c_int]),
    "aoclsparse_csr2ell_width": (c_int, [_I, _I, _P, _P]),
    "aoclsparse_csr2ellthyb_width": (c_int, [_I, _I, _P, _P, _P]),
    "aoclsparse_scsr2ell": (c_int, [_I, _P, _P, _P, _P, _P, _P, _I]),
    "aoclsparse_dcsr2ell": (c_int, [_I, _P, _P, _P, _P, _P, _P, _I]),
    "aoclsparse_scsr2ellt": (c_int, [_I, _P, _P, _P, _P, _P, _P, _I]),
    "aoclsparse_dcsr2ellt": (c_int, [_I, _P, _P, _P, _P, _P, _P, _I]),
    "aoclsparse_scsr2ellthyb": (c_int, [_I, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _I]),
    "aoclsparse_dcsr2ellthyb": (c_int, [_I, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _I]),
    "aoclsparse_itsol_handle_prn_options": (None, [_P]),
    "aoclsparse_itsol_option_set": (c_int, [_P, c_char_p, c_char_p]),
    "aoclsparse_itsol_c_init": (c_int, [POINTER(_P)]),
    "aoclsparse_itsol_c_rci_input": (c_int, [_P, _I, _P]),
    "aoclsparse_itsol_c_rci_solve": (c_int, [_P, POINTER(c_int), POINTER(_P), POINTER(_P), _P, _P]),
    "aoclsparse_itsol_c_solve": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_itsol_z_init": (c_int, [POINTER(_P)]),
    "aoclsparse_itsol_z_rci_input": (c_int, [_P, _I, _P]),
    "aoclsparse_itsol_z_rci_solve": (c_int, [_P, POINTER(c_int), POINTER(_P), POINTER(_P), _P, _P]),
    "aoclsparse_itsol_z_solve": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_itsol_d_init": (c_int, [POINTER(_P)]),
    "aoclsparse_itsol_s_init": (c_int, [POINTER(_P)]),
    "aoclsparse_itsol_destroy": (None, [POINTER(_P)]),
    "aoclsparse_itsol_d_rci_input": (c_int, [_P, _I, _P]),
    "aoclsparse_itsol_s_rci_input": (c_int, [_P, _I, _P]),
    "aoclsparse_itsol_d_rci_solve": (c_int, [_P, POINTER(c_int), POINTER(_P), POINTER(_P), _P, _P]),
    "aoclsparse_itsol_s_rci_solve": (c_int, [_P, POINTER(c_int), POINTER(_P), POINTER(_P), _P, _P]),
    "aoclsparse_itsol_d_solve": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_itsol_s_solve": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_ssymgs": (c_int, [c_int, _P, _P, c_float, _P, _P]),
    "aoclsparse_dsymgs": (c_int, [c_int, _P, _P, c_double, _P, _P]),
    "aoclsparse_ssymgs_kid": (c_int, [c_int, _P, _P, c_float, _P, _P, _I]),
    "aoclsparse_dsymgs_kid": (c_int, [c_int, _P, _P, c_double, _P, _P, _I]),
    "aoclsparse_ssymgs_mv": (c_int, [c_int, _P, _P, c_float, _P, _P, _P]),
    "aoclsparse_dsymgs_mv": (c_int, [c_int, _P, _P, c_double, _P, _P, _P]),
    "aoclsparse_ssymgs_mv_kid": (c_int, [c_int, _P, _P, c_float, _P, _P, _P, _I]),
    "aoclsparse_dsymgs_mv_kid": (c_int, [c_int, _P, _P, c_double, _P, _P, _P, _I]),
    "aoclsparse_silu_smoother": (c_int, [c_int, _P, _P, POINTER(_P), _P, _P, _P]),
    "aoclsparse_dilu_smoother": (c_int, [c_int, _P, _P, POINTER(_P), _P, _P, _P]),
    "aoclsparse_cilu_smoother": (c_int, [c_int, _P, _P, POINTER(_P), _P, _P, _P]),
    "aoclsparse_zilu_smoother": (c_int, [c_int, _P, _P, POINTER(_P), _P, _P, _P]),
    "aoclsparse_set_dotmv_hint": (c_int, [_P, c_int, _P, _I]),
    "aoclsparse_set_lu_smoother_hint": (c_int, [_P, c_int, _P, _I]),
    "aoclsparse_set_sm_hint": (c_int, [_P, c_int, _P, c_int, _I]),
    "aoclsparse_set_symgs_hint": (c_int, [_P, c_int, _P, _I]),
    "aoclsparse_set_sorv_hint": (c_int, [_P, _P, c_int, _I]),
    "aoclsparse_scsrmm": (c_int, [c_int, c_float, _P, _P, c_int, _P, _I, _I, c_float, _P, _I]),
    "aoclsparse_dcsrmm": (c_int, [c_int, c_double, _P, _P, c_int, _P, _I, _I, c_double, _P, _I]),
    "aoclsparse_scsrmm_kid": (c_int, [c_int, c_float, _P, _P, c_int, _P, _I, _I, c_float, _P, _I, _I]),
    "aoclsparse_dcsrmm_kid": (c_int, [c_int, c_double, _P, _P, c_int, _P, _I, _I, c_double, _P, _I, _I]),
    "aoclsparse_sp2m": (c_int, [c_int, _P, _P, c_int, _P, _P, c_int, POINTER(_P)]),
    "aoclsparse_spmm": (c_int, [c_int, _P, _P, POINTER(_P)]),
    "aoclsparse_dcsr2m": (c_int, [c_int, _P, _P, c_int, _P, _P, c_int, POINTER(_P)]),
    "aoclsparse_scsr2m": (c_int, [c_int, _P, _P, c_int, _P, _P, c_int, POINTER(_P)]),
    "aoclsparse_ssp2md": (c_int, [c_int, _P, _P, c_int, _P, _P, c_float, c_float, _P, c_int, _I]),
    "aoclsparse_sspmmd": (c_int, [c_int, _P, _P, c_int, _P, _I]),
    "aoclsparse_scsr2dense": (c_int, [_I, _I, _P, _P, _P, _P, _P, _I, c_int]),
    "aoclsparse_sadd": (c_int, [c_int, _P, c_float, _P, POINTER(_P)]),
    "aoclsparse_dsp2md": (c_int, [c_int, _P, _P, c_int, _P, _P, c_double, c_double, _P, c_int, _I]),
    "aoclsparse_dspmmd": (c_int, [c_int, _P, _P, c_int, _P, _I]),
    "aoclsparse_dcsr2dense": (c_int, [_I, _I, _P, _P, _P, _P, _P, _I, c_int]),
    "aoclsparse_dadd": (c_int, [c_int, _P, c_double, _P, POINTER(_P)]),
    "aoclsparse_csp2md": (c_int, [c_int, _P, _P, c_int, _P, _P, CFloat, CFloat, _P, c_int, _I]),
    "aoclsparse_cspmmd": (c_int, [c_int, _P, _P, c_int, _P, _I]),
    "aoclsparse_ccsr2dense": (c_int, [_I, _I, _P, _P, _P, _P, _P, _I, c_int]),
    "aoclsparse_cadd": (c_int, [c_int, _P, CFloat, _P, POINTER(_P)]),
    "aoclsparse_zsp2md": (c_int, [c_int, _P, _P, c_int, _P, _P, CDouble, CDouble, _P, c_int, _I]),
    "aoclsparse_zspmmd": (c_int, [c_int, _P, _P, c_int, _P, _I]),
    "aoclsparse_zcsr2dense": (c_int, [_I, _I, _P, _P, _P, _P, _P, _I, c_int]),
    "aoclsparse_zadd": (c_int, [c_int, _P, CDouble, _P, POINTER(_P)]),
    "aoclsparse_saxpyi": (c_int, [_I, c_float, _P, _P, _P]),
    "aoclsparse_sdoti": (c_float, [_I, _P, _P, _P]),
    "aoclsparse_sroti": (c_int, [_I, _P, _P, _P, c_float, c_float]),
    "aoclsparse_daxpyi": (c_int, [_I, c_double, _P, _P, _P]),
    "aoclsparse_ddoti": (c_double, [_I, _P, _P, _P]),
    "aoclsparse_droti": (c_int, [_I, _P, _P, _P, c_double, c_double]),
    "aoclsparse_caxpyi": (c_int, [_I, _P, _P, _P, _P]),
    "aoclsparse_cdotci": (c_int, [_I, _P, _P, _P, _P]),
    "aoclsparse_cdotui": (c_int, [_I, _P, _P, _P, _P]),
    "aoclsparse_zaxpyi": (c_int, [_I, _P, _P, _P, _P]),
    "aoclsparse_zdotci": (c_int, [_I, _P, _P, _P, _P]),
    "aoclsparse_zdotui": (c_int, [_I, _P, _P, _P, _P]),
    "aoclsparse_sgthr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_sgthrz": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_sgthrs": (c_int, [_I, _P, _P, _I]),
    "aoclsparse_ssctr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_ssctrs": (c_int, [_I, _P, _I, _P]),
    "aoclsparse_dgthr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_dgthrz": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_dgthrs": (c_int, [_I, _P, _P, _I]),
    "aoclsparse_dsctr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_dsctrs": (c_int, [_I, _P, _I, _P]),
    "aoclsparse_cgthr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_cgthrz": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_cgthrs": (c_int, [_I, _P, _P, _I]),
    "aoclsparse_csctr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_csctrs": (c_int, [_I, _P, _I, _P]),
    "aoclsparse_zgthr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_zgthrz": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_zgthrs": (c_int, [_I, _P, _P, _I]),
    "aoclsparse_zsctr": (c_int, [_I, _P, _P, _P]),
    "aoclsparse_zsctrs": (c_int, [_I, _P, _I, _P]),
    "aoclsparse_saxpyi_kid": (c_int, [_I, c_float, _P, _P, _P, _I]),
    "aoclsparse_sdoti_kid": (c_float, [_I, _P, _P, _P, _I]),
    "aoclsparse_sroti_kid": (c_int, [_I, _P, _P, _P, c_float, c_float, _I]),
    "aoclsparse_daxpyi_kid": (c_int, [_I, c_double, _P, _P, _P, _I]),
    "aoclsparse_ddoti_kid": (c_double, [_I, _P, _P, _P, _I]),
    "aoclsparse_droti_kid": (c_int, [_I, _P, _P, _P, c_double, c_double, _I]),
    "aoclsparse_caxpyi_kid": (c_int, [_I, _P, _P, _P, _P, _I]),
    "aoclsparse_cdotci_kid": (c_int, [_I, _P, _P, _P, _P, _I]),
    "aoclsparse_cdotui_kid": (c_int, [_I, _P, _P, _P, _P, _I]),
    "aoclsparse_zaxpyi_kid": (c_int, [_I, _P, _P, _P, _P, _I]),
    "aoclsparse_zdotci_kid": (c_int, [_I, _P, _P, _P, _P, _I]),
    "aoclsparse_zdotui_kid": (c_int, [_I, _P, _P, _P, _P, _I]),
    "aoclsparse_sgthr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_sgthrz_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_sgthrs_kid": (c_int, [_I, _P, _P, _I, _I]),
    "aoclsparse_ssctr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_ssctrs_kid": (c_int, [_I, _P, _I, _P, _I]),
    "aoclsparse_dgthr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_dgthrz_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_dgthrs_kid": (c_int, [_I, _P, _P, _I, _I]),
    "aoclsparse_dsctr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_dsctrs_kid": (c_int, [_I, _P, _I, _P, _I]),
    "aoclsparse_cgthr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_cgthrz_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_cgthrs_kid": (c_int, [_I, _P, _P, _I, _I]),
    "aoclsparse_csctr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_csctrs_kid": (c_int, [_I, _P, _I, _P, _I]),
    "aoclsparse_zgthr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_zgthrz_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_zgthrs_kid": (c_int, [_I, _P, _P, _I, _I]),
    "aoclsparse_zsctr_kid": (c_int, [_I, _P, _P, _P, _I]),
    "aoclsparse_zsctrs_kid": (c_int, [_I, _P, _I, _P, _I]),
    "aoclsparse_csr2dia_ndiag": (c_int, [_I, _I, _P, _I, _P, _P, _P]),
    "aoclsparse_csr2bsr_nnz": (c_int, [_I, _I, _P, _P, _P, _I, _P, _P]),
    "aoclsparse_scsr2dia": (c_int, [_I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "aoclsparse_sdiamv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "aoclsparse_sdiamv_kid": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I]),
    "aoclsparse_sbsrmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_dcsr2dia": (c_int, [_I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "aoclsparse_ddiamv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "aoclsparse_ddiamv_kid": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I]),
    "aoclsparse_dbsrmv": (c_int, [c_int, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "aoclsparse_scsr2bsr": (c_int, [_I, _I, _P, c_int, _P, _P, _P, _I, _P, _P, _P]),
    "aoclsparse_dcsr2bsr": (c_int, [_I, _I, _P, c_int, _P, _P, _P, _I, _P, _P, _P]),
    "aoclsparse_ccsr2bsr": (c_int, [_I, _I, _P, c_int, _P, _P, _P, _I, _P, _P, _P]),
    "aoclsparse_zcsr2bsr": (c_int, [_I, _I, _P, c_int, _P, _P, _P, _I, _P, _P, _P]),
    "aoclsparse_ssorv": (c_int, [c_int, _P, _P, c_float, c_float, _P, _P]),
    "aoclsparse_dsorv": (c_int, [c_int, _P, _P, c_double, c_double, _P, _P]),
    "aoclsparse_csorv": (c_int, [c_int, _P, _P, CFloat, CFloat, _P, _P]),
    "aoclsparse_zsorv": (c_int, [c_int, _P, _P, CDouble, CDouble, _P, _P]),
    # include/aoclsparse_mi355.h
    "aoclsparse_mi355_set_pointer_mode": (c_int, [c_int]),
    "aoclsparse_mi355_set_stream": (c_int, [_P]),
    "aoclsparse_mi355_get_stream": (_P, []),
    "aoclsparse_mi355_hip_runtime_path": (c_int, [c_char_p, ctypes.c_size_t]),
    "aoclsparse_mi355_synchronize": (c_int, []),
    "aoclsparse_mi355_device_info": (c_int, [POINTER(_I), POINTER(_I), c_char_p]),
    "aoclsparse_mi355_timer_start": (c_int, []),
    "aoclsparse_mi355_timer_stop": (c_int, [POINTER(c_float)]),
    "aoclsparse_mi355_timer_mark": (c_int, []),
    "aoclsparse_mi355_timer_laps": (c_int, [POINTER(c_float), _I, POINTER(_I)]),
    "aoclsparse_mi355_column_shard": (c_int, [_I, _I, _I, POINTER(_I), POINTER(_I)]),
    "aoclsparse_mi355_plan_block_row_order": (c_int, [_I, _I, _I, c_void_p, _I, c_void_p, _I, POINTER(_I), POINTER(_I)]),
    "aoclsparse_mi355_dcsrmm_shard": (c_int, [c_int, c_double, _P, _P, c_int, _P, _I, _I, c_double, _P, _I, _I, _I]),
    "aoclsparse_mi355_dcsrmm_multi": (c_int, [c_int, c_double, _P, _P, c_int, _P, _I, _I, c_double, _P, _I, _I, _P]),
    "aoclsparse_mi355_set_option": (c_int, [c_int, _I]),
    "aoclsparse_mi355_mm_state_export": (c_int, [_P, POINTER(MmState), _P]),
    "aoclsparse_mi355_mm_state_adopt": (c_int, [_P, POINTER(MmState), _P]),
    "aoclsparse_mi355_comm_unique_id": (c_int, [POINTER(CommId)]),
    "aoclsparse_mi355_comm_init": (c_int, [_I, _I, POINTER(CommId)]),
    "aoclsparse_mi355_comm_destroy": (c_int, []),
    "aoclsparse_mi355_comm_info": (c_int, [POINTER(_I), POINTER(_I), POINTER(_I)]),
    "aoclsparse_mi355_comm_broadcast": (c_int, [_P, ctypes.c_size_t, _I]),
    "aoclsparse_mi355_comm_allgather": (c_int, [_P, _P, ctypes.c_size_t]),
    "aoclsparse_mi355_comm_broadcast_matrix": (c_int, [_P, _I]),
    "aoclsparse_mi355_replica_count": (c_int, [_P]),
    "aoclsparse_mi355_multi_last_ms": (c_int, [_P, c_int]),
    "aoclsparse_mi355_replicas_cloned": (c_int, [_P]),
    "aoclsparse_mi355_scsrmm_multi": (c_int, [c_int, c_float, _P, _P, c_int, _P, _I, _I, c_float, _P, _I, _I, _P]),
    "aoclsparse_mi355_dcsrmm_multi_slabs": (c_int, [c_int, c_double, _P, _P, c_int, _P, _I, _I, c_double, _P, _I, _I, _P]),
    "aoclsparse_mi355_scsrmm_shard": (c_int, [c_int, c_float, _P, _P, c_int, _P, _I, _I, c_float, _P, _I, _I, _I]),
    "aoclsparse_mi355_export_diag": (c_int, [_P, POINTER(_P), POINTER(_P), POINTER(_I)]),
    "aoclsparse_mi355_get_spmv_info": (c_int, [_P, c_int, POINTER(SpmvInfo)]),
    "aoclsparse_mi355_get_trsv_levels": (c_int, [_P, c_int, c_int, POINTER(_I)]),
    "aoclsparse_mi355_get_trsv_info": (c_int, [_P, c_int, c_int, POINTER(TrsvInfo)]),
    "aoclsparse_mi355_trsv_status": (c_int, [_P]),
    "aoclsparse_mi355_set_trsv_schedule": (c_int, [_I]),
    "aoclsparse_mi355_set_csrmm_beta0_overwrite": (c_int, [c_int]),
    "aoclsparse_mi355_invalidate": (c_int, [_P]),
    "aoclsparse_mi355_release_staging": (c_int, [POINTER(ctypes.c_size_t)]),
    "mi355_csrmv_plan_bound": (_I, [_I, _I]),
    "mi355_csrmv_plan_host": (_I, [_I, _I, _I, _P, _P]),
    "mi355_dcsrmv": (c_int, [_P, _I, _I, _I, _I, c_double, _I, _P, _P, _P, _P, _I, _P, c_double, _P]),
    "mi355_scsrmv": (c_int, [_P, _I, _I, _I, _I, c_float, _I, _P, _P, _P, _P, _I, _P, c_float, _P]),
    "mi355_dcsrmm": (c_int, [_P, _I, _I, c_double, _I, _I, _P, _P, _P, _P, _I, _I, c_double, _P, _I]),
}

_lib = None


def lib():
    """Load the shared library (once) and attach the signatures.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libaoclsparse_mi355.so is not built (%s): run `python -c 'import __graft_entry__ as g; "
                "g.build()'` -- the HIP library is the product, there is no fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = symbol declared in include/ but not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    """numpy array / torch tensor / int / None -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, int):
        return c_void_p(a)
    if hasattr(a, "data_ptr"):  # torch tensor (host or device)
        return c_void_p(a.data_ptr())
    return c_void_p(a.ctypes.data)


class Descr:
    """aoclsparse_mat_descr wrapper."""

    def __init__(self, base=0, mtype=TYPE_GENERAL, fill=FILL_LOWER, diag=DIAG_NON_UNIT):
        self.h = c_void_p()
        st = lib().aoclsparse_create_mat_descr(byref(self.h))
        assert st == 0, STATUS[st]
        L = lib()
        assert L.aoclsparse_set_mat_index_base(self.h, base) == 0
        assert L.aoclsparse_set_mat_type(self.h, mtype) == 0
        assert L.aoclsparse_set_mat_fill_mode(self.h, fill) == 0
        assert L.aoclsparse_set_mat_diag_type(self.h, diag) == 0

    def __del__(self):
        try:
            if self.h:
                lib().aoclsparse_destroy_mat_descr(self.h)
                self.h = c_void_p()
        except Exception:
            pass


class Matrix:
    """aoclsparse_matrix wrapper; keeps the aliased numpy arrays alive (the library does not copy)."""

    def __init__(self, base, m, n, row_ptr, col_ind, val):
        import numpy as np

        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.col_ind = np.ascontiguousarray(col_ind, dtype=np.int32)
        self.val = np.ascontiguousarray(val)
        assert self.val.dtype in (np.float64, np.float32)
        self.double = self.val.dtype == np.float64
        self.m, self.n, self.nnz, self.base = m, n, int(self.row_ptr[m]) - base if len(self.row_ptr) > m else 0, base
        self.h = c_void_p()
        fn = lib().aoclsparse_create_dcsr if self.double else lib().aoclsparse_create_scsr
        self.status = fn(byref(self.h), base, m, n, self.nnz, _ptr(self.row_ptr), _ptr(self.col_ind),
                         _ptr(self.val))

    @classmethod
    def from_handle(cls, h, double=True):
        """wrap a handle the LIBRARY created and whose arrays it owns (aoclsparse_mi355_mm_state_adopt,
        aoclsparse_mi355_comm_broadcast_matrix); m / n / nnz / base are read back through aoclsparse_export_?csr"""
        self = cls.__new__(cls)
        self.h, self.double, self.status = h, double, 0
        self.row_ptr = self.col_ind = self.val = None
        e = self.export()
        assert e["status"] == 0, STATUS.get(e["status"], e["status"])
        self.m, self.n, self.nnz, self.base = e["m"], e["n"], e["nnz"], e["base"]
        return self

    def mm_state_export(self):
        """-> (status, MmState, [MM_STATE_BUFFERS device pointers]) : aoclsparse_mi355_mm_state_export"""
        st = MmState()
        ptrs = (c_void_p * MM_STATE_BUFFERS)()
        rc = lib().aoclsparse_mi355_mm_state_export(self.h, byref(st), ctypes.cast(ptrs, c_void_p))
        return rc, st, [ptrs[i] for i in range(MM_STATE_BUFFERS)]

    @classmethod
    def mm_state_adopt(cls, state, device_ptrs, double=True):
        """device_ptrs: MM_STATE_BUFFERS device addresses (int / None) in THIS process's device memory -> (status, Matrix or None)"""
        ptrs = (c_void_p * MM_STATE_BUFFERS)(*[c_void_p(p) if p else None for p in device_ptrs])
        h = c_void_p()
        rc = lib().aoclsparse_mi355_mm_state_adopt(byref(h), byref(state), ctypes.cast(ptrs, c_void_p))
        return rc, (cls.from_handle(h, double) if rc == 0 else None)

    def destroy(self):
        if self.h:
            lib().aoclsparse_destroy(byref(self.h))

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def export(self):
        """-> dict(base, m, n, nnz, row_ptr, col_ind, val) copies of aoclsparse_export_?csr."""
        import numpy as np

        base, m, n, nnz = c_int(), c_int32(), c_int32(), c_int32()
        rp, ci, v = c_void_p(), c_void_p(), c_void_p()
        fn = lib().aoclsparse_export_dcsr if self.double else lib().aoclsparse_export_scsr
        st = fn(self.h, byref(base), byref(m), byref(n), byref(nnz), byref(rp), byref(ci), byref(v))
        if st != 0:
            return dict(status=st)
        ct = c_double if self.double else c_float
        row_ptr = np.ctypeslib.as_array(ctypes.cast(rp, POINTER(c_int32)), (m.value + 1,)).copy()
        col = np.ctypeslib.as_array(ctypes.cast(ci, POINTER(c_int32)), (max(nnz.value, 1),))[: nnz.value].copy()
        val = np.ctypeslib.as_array(ctypes.cast(v, POINTER(ct)), (max(nnz.value, 1),))[: nnz.value].copy()
        return dict(status=0, base=base.value, m=m.value, n=n.value, nnz=nnz.value, row_ptr=row_ptr,
                    col_ind=col, val=val, aliased=(self.row_ptr is not None and rp.value == self.row_ptr.ctypes.data))

    def export_diag(self):
        import numpy as np

        d, u, internal = c_void_p(), c_void_p(), c_int32()
        st = lib().aoclsparse_mi355_export_diag(self.h, byref(d), byref(u), byref(internal))
        if st != 0:
            return dict(status=st)
        k = max(self.m, 1)
        idiag = np.ctypeslib.as_array(ctypes.cast(d, POINTER(c_int32)), (k,))[: self.m].copy()
        iurow = np.ctypeslib.as_array(ctypes.cast(u, POINTER(c_int32)), (k,))[: self.m].copy()
        return dict(status=0, idiag=idiag, iurow=iurow, is_internal=bool(internal.value))

    def spmv_info(self, op=OP_NONE):
        info = SpmvInfo()
        st = lib().aoclsparse_mi355_get_spmv_info(self.h, op, byref(info))
        assert st == 0
        return info

    def trsv_info(self, fill, op=OP_NONE):
        info = TrsvInfo()
        assert lib().aoclsparse_mi355_get_trsv_info(self.h, fill, op, byref(info)) == 0
        return info

    def trsv_levels(self, fill, op=OP_NONE):
        lv = c_int32(-2)
        assert lib().aoclsparse_mi355_get_trsv_levels(self.h, fill, op, byref(lv)) == 0
        return lv.value


def scalar(v, double=True):
    return (c_double if double else c_float)(v)


def dmv(op, alpha, A, descr, x, beta, y):
    a, b = c_double(alpha), c_double(beta)
    return lib().aoclsparse_dmv(op, byref(a), A.h, descr.h, _ptr(x), byref(b), _ptr(y))


def smv(op, alpha, A, descr, x, beta, y):
    a, b = c_float(alpha), c_float(beta)
    return lib().aoclsparse_smv(op, byref(a), A.h, descr.h, _ptr(x), byref(b), _ptr(y))


def dcsrmv(op, alpha, m, n, nnz, val, col, row, descr, x, beta, y):
    a, b = c_double(alpha), c_double(beta)
    return lib().aoclsparse_dcsrmv(op, byref(a), m, n, nnz, _ptr(val), _ptr(col), _ptr(row), descr.h,
                                   _ptr(x), byref(b), _ptr(y))


def scsrmv(op, alpha, m, n, nnz, val, col, row, descr, x, beta, y):
    a, b = c_float(alpha), c_float(beta)
    return lib().aoclsparse_scsrmv(op, byref(a), m, n, nnz, _ptr(val), _ptr(col), _ptr(row), descr.h,
                                   _ptr(x), byref(b), _ptr(y))


def dtrsv(op, alpha, A, descr, b, x, kid=None, incb=None, incx=None):
    L = lib()
    if incb is not None or incx is not None:
        return L.aoclsparse_dtrsv_strided(op, alpha, A.h, descr.h, _ptr(b), incb or 1, _ptr(x), incx or 1)
    if kid is None:
        return L.aoclsparse_dtrsv(op, alpha, A.h, descr.h, _ptr(b), _ptr(x))
    return L.aoclsparse_dtrsv_kid(op, alpha, A.h, descr.h, _ptr(b), _ptr(x), kid)


def strsv(op, alpha, A, descr, b, x, kid=None):
    L = lib()
    if kid is None:
        return L.aoclsparse_strsv(op, alpha, A.h, descr.h, _ptr(b), _ptr(x))
    return L.aoclsparse_strsv_kid(op, alpha, A.h, descr.h, _ptr(b), _ptr(x), kid)


def dcsrmm(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, kid=None):
    L = lib()
    if kid is None:
        return L.aoclsparse_dcsrmm(op, alpha, A.h, descr.h, order, _ptr(B), n, ldb, beta, _ptr(C), ldc)
    return L.aoclsparse_dcsrmm_kid(op, alpha, A.h, descr.h, order, _ptr(B), n, ldb, beta, _ptr(C), ldc, kid)


def scsrmm(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, kid=None):
    if kid is None:
        return lib().aoclsparse_scsrmm(op, alpha, A.h, descr.h, order, _ptr(B), n, ldb, beta, _ptr(C), ldc)
    return lib().aoclsparse_scsrmm_kid(op, alpha, A.h, descr.h, order, _ptr(B), n, ldb, beta, _ptr(C), ldc, kid)


def device_info():
    dev, cus = c_int32(-1), c_int32(0)
    name = ctypes.create_string_buffer(256)
    st = lib().aoclsparse_mi355_device_info(byref(dev), byref(cus), name)
    return st, dev.value, cus.value, name.value.decode()


def hip_runtime_path():
    """path of the libamdhip64 the library is bound to (streams given to aoclsparse_mi355_set_stream must come from it)"""
    buf = ctypes.create_string_buffer(4096)
    st = lib().aoclsparse_mi355_hip_runtime_path(buf, 4096)
    return st, buf.value.decode()


def timer_start():
    return lib().aoclsparse_mi355_timer_start()


def timer_stop():
    ms = c_float(0)
    st = lib().aoclsparse_mi355_timer_stop(byref(ms))
    assert st == 0, STATUS[st]
    return ms.value


def timer_mark():
    st = lib().aoclsparse_mi355_timer_mark()
    assert st == 0, STATUS[st]


def timer_laps(capacity=65536):
    """-> list of elapsed milliseconds between consecutive timer_mark() calls (waits for the last one)."""
    buf = (c_float * capacity)()
    cnt = c_int32(0)
    st = lib().aoclsparse_mi355_timer_laps(buf, capacity, byref(cnt))
    assert st == 0, STATUS[st]
    return [buf[i] for i in range(min(cnt.value, capacity))]


def column_shard(ncols, world, rank):
    """[j0, j1) of rank `rank`: the reference's per-thread column split (csrmm_kt.cpp:68-82) with ranks for threads."""
    j0, j1 = c_int32(0), c_int32(0)
    st = lib().aoclsparse_mi355_column_shard(ncols, world, rank, byref(j0), byref(j1))
    if st != 0:
        raise ValueError("column_shard(%d, %d, %d): %s" % (ncols, world, rank, STATUS[st]))
    return j0.value, j1.value


def dcsrmm_shard(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, world, rank):
    return lib().aoclsparse_mi355_dcsrmm_shard(op, alpha, A.h, descr.h, order, _ptr(B), n, ldb, beta, _ptr(C), ldc,
                                               world, rank)


def dcsrmm_multi(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, devices):
    """aoclsparse_mi355_dcsrmm_multi: full host operands, one worker per entry of `devices` (list of HIP ordinals)."""
    dv = (c_int32 * len(devices))(*devices)
    return lib().aoclsparse_mi355_dcsrmm_multi(op, alpha, A.h, descr.h, order, _ptr(B), n, ldb, beta, _ptr(C), ldc,
                                               len(devices), ctypes.cast(dv, c_void_p))


def dcsrmm_multi_slabs(op, alpha, A, descr, order, B_slabs, n, ldb, beta, C_slabs, ldc, devices):
    """aoclsparse_mi355_dcsrmm_multi_slabs: B_slabs[i] / C_slabs[i] are tensors resident on devices[i]."""
    dv = (c_int32 * len(devices))(*devices)
    bp = (c_void_p * len(devices))(*[_ptr(b) for b in B_slabs])
    cp = (c_void_p * len(devices))(*[_ptr(c) for c in C_slabs])
    return lib().aoclsparse_mi355_dcsrmm_multi_slabs(op, alpha, A.h, descr.h, order, ctypes.cast(bp, c_void_p), n, ldb, beta,
                                                     ctypes.cast(cp, c_void_p), ldc, len(devices), ctypes.cast(dv, c_void_p))
