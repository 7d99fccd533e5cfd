"""ctypes binding of ``libhs_solver.so`` (C ABI: ``include/hs_solver.h``, ``include/hs_kernels.h``).

There is no CPU fallback: if the shared library is missing, or it finds no HIP
device, the calls raise.  ``build()`` compiles the library in-tree with hipcc
(cross-compiles for gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhs_solver.so")
CSRC = os.path.join(_HERE, "csrc")

i64 = C.c_int64
p_i64 = C.POINTER(C.c_int64)
p_f64 = C.POINTER(C.c_double)


class hs_options(C.Structure):
    _fields_ = [
        ("swlevel", i64), ("swsize", i64), ("atol", C.c_double), ("rtol", C.c_double), ("c_tol", C.c_double),
        ("leafsize", i64), ("kest", i64), ("stepsize", i64), ("verbose", C.c_uint8),
        ("keep_schur", C.c_uint8), ("profile", C.c_uint8), ("split", C.c_uint8), ("hss_d", C.c_uint8), ("hss_dexp", C.c_uint8), ("mf", C.c_uint8), ("dist_top", C.c_uint8), ("seed", i64),
    ]


class hs_tree(C.Structure):
    _fields_ = [
        ("nnodes", i64), ("left", p_i64), ("right", p_i64),
        ("int_ptr", p_i64), ("int_idx", p_i64), ("bnd_ptr", p_i64), ("bnd_idx", p_i64),
        ("iloc_ptr", p_i64), ("iloc_idx", p_i64), ("bloc_ptr", p_i64), ("bloc_idx", p_i64),
    ]


class hs_stats(C.Structure):
    _fields_ = [
        ("n", i64), ("nnodes", i64), ("nlevels", i64), ("max_ni", i64), ("max_nb", i64),
        ("flops_factor", C.c_double), ("bytes_factors", C.c_double), ("bytes_solve", C.c_double),
        ("t_symbolic", C.c_double), ("t_upload", C.c_double), ("t_assemble", C.c_double), ("t_panel", C.c_double),
        ("t_trsm", C.c_double), ("t_gemm", C.c_double), ("t_total", C.c_double), ("t_solve", C.c_double),
        ("gemm_flops", C.c_double), ("gemm_launches", i64),
        ("t_mfma_kernel", C.c_double), ("mfma_kernel_launches", i64), ("gemm_bytes", C.c_double),
    ]


class hs_sparse_dev(C.Structure):
    _fields_ = [("n", i64), ("colptr", C.c_void_p), ("rowval", C.c_void_p), ("nzval", C.c_void_p),
                ("rowptr", C.c_void_p), ("colind", C.c_void_p), ("nzval_r", C.c_void_p)]


class hs_hss_blockop(C.Structure):
    _fields_ = [("n1", i64), ("n2", i64), ("H1", C.c_void_p), ("H2", C.c_void_p), ("gid", p_i64), ("A", C.POINTER(hs_sparse_dev)), ("lpos", C.c_void_p)]


class hs_hss_options(C.Structure):
    _fields_ = [("leafsize", i64), ("first_split", i64), ("atol", C.c_double), ("rtol", C.c_double), ("kest", i64), ("pad", i64), ("seed", i64),
                ("level_scale", C.c_double)]


# hs_transfer_fn (include/hs_solver.h): one host message per peer and direction
HS_TRANSFER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, i64, p_i64, C.POINTER(C.c_void_p), p_i64, i64, p_i64, C.POINTER(C.c_void_p), p_i64)

HS_OK = 0
HS_ERR_ARGUMENT, HS_ERR_DIMENSION, HS_ERR_TREE, HS_ERR_SINGULAR = -1, -2, -3, -4
HS_ERR_HSS_LEAF, HS_ERR_DEVICE, HS_ERR_NOMEM, HS_ERR_UNSUPPORTED = -5, -6, -7, -8
HS_BLK_LU, HS_BLK_LBI, HS_BLK_UIB, HS_BLK_S = 0, 1, 2, 3

# every symbol include/*.h declares (tests check the library exports all of them)
EXPORTS = [
    "hs_options_default", "hs_factor_d", "hs_factor_z", "hs_ldiv_d", "hs_ldiv_z", "hs_ldiv_dev_d", "hs_ldiv_dev_z",
    "hs_maxrank", "hs_is_complex", "hs_size", "hs_free", "hs_last_error", "hs_last_error_info", "hs_get_stats",
    "hs_node_info", "hs_node_ranks", "hs_node_export", "hs_node_export_piv", "hs_device_info",
    "hs_analyze", "hs_plan", "hs_numeric_begin", "hs_numeric_levels", "hs_numeric_end", "hs_solve_fwd_levels", "hs_solve_bwd_levels",
    "hs_nlevels", "hs_cut_level", "hs_node_owner", "hs_num_exchanges", "hs_exchange_info", "hs_set_schur_buffer",
    "hs_pack_bnd", "hs_unpack_bnd", "hs_extract_owned", "hs_gmres_d", "hs_gmres_z",
    "hs_exchange_kind", "hs_schur_pack_size", "hs_schur_pack", "hs_schur_unpack", "hs_flow_info", "hs_hss_pack_size", "hs_hss_pack", "hs_hss_unpack", "hs_hss_qr_order",
    "hs_comm_unique_id", "hs_comm_create_rccl", "hs_comm_create_host", "hs_comm_free", "hs_comm_kind", "hs_comm_selftest", "hs_comm_bandwidth", "hs_set_comm",
    "hs_symbolic_from_elimtree", "hs_symbolic_from_graph", "hs_symbolic_size", "hs_symbolic_perm", "hs_symbolic_tree", "hs_symbolic_free",
    "hs_hss_options_default", "hs_hss_compress_d", "hs_hss_compress_z", "hs_hss_compress_ex_d", "hs_hss_compress_ex_z", "hs_hss_compress_lru_d", "hs_hss_compress_lru_z", "hs_hss_compress_lru_multi_d", "hs_hss_compress_lru_multi_z", "hs_hss_set_stream", "hs_hss_rank", "hs_hss_size", "hs_hss_samples", "hs_hss_num_nodes",
    "hs_hss_node_info", "hs_hss_node_data", "hs_hss_getindex", "hs_hss_basis", "hs_hss_expand", "hs_hss_mul", "hs_hss_mul_t", "hs_hss_child", "hs_hss_factor", "hs_hss_ldiv", "hs_hss_time", "hs_hss_trim", "hs_hss_free", "hs_node_schur_hss",
    "hs_hss_offdiag", "hs_hss_bytes", "hs_hss_prune_leaves", "hs_hss_compatible", "hs_hss_depth", "hs_hss_compress_blockop_d", "hs_hss_compress_blockop_z", "hs_hss_blockop_apply",
    "hsk_gemm_d", "hsk_gemm_z", "hsk_lowrank_d", "hsk_lowrank_z", "hsk_front_factor_d", "hsk_front_factor_z", "hsk_mfma_f64_peak", "hsk_mfma_f64_peak_random", "hsk_flow_pingpong_us", "hsk_bisect_perm",
    "hs_probs_stats_mode", "hs_probs_stats", "hs_trim", "hs_stream_order",
]

_lib = None


def build(force=False, verbose=False):
    """Compile ``libhs_solver.so`` for gfx950 in-tree (``make`` in ``csrc/``)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libhs_solver.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib():
    """Load the shared library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build() or `make -C {CSRC}`); "
            "there is no CPU fallback"
        )
    # PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).  Two HSA
    # runtimes cannot share a process, so when torch is installed let it load its runtime FIRST; this
    # library then binds to the already-loaded one.  (A Julia host has no torch and uses /opt/rocm's.)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.hs_options_default.argtypes = [C.POINTER(hs_options)]
    L.hs_options_default.restype = None
    for f in (L.hs_factor_d, L.hs_factor_z):
        f.argtypes = [i64, p_i64, p_i64, p_f64, C.POINTER(hs_tree), C.POINTER(hs_options), C.POINTER(vp)]
        f.restype = C.c_int
    for f in (L.hs_ldiv_d, L.hs_ldiv_z):
        f.argtypes = [vp, p_f64, i64, p_f64, i64, i64, i64]
        f.restype = C.c_int
    for f in (L.hs_ldiv_dev_d, L.hs_ldiv_dev_z):
        f.argtypes = [vp, vp, i64, vp, i64, i64, i64, vp]
        f.restype = C.c_int
    L.hs_analyze.argtypes = [C.c_int, i64, p_i64, p_i64, C.POINTER(hs_tree), C.POINTER(hs_options), i64, i64, C.POINTER(vp)]
    L.hs_analyze.restype = C.c_int
    L.hs_plan.argtypes = [C.c_int, i64, C.POINTER(hs_tree), C.POINTER(hs_options), i64, i64, C.POINTER(vp)]
    L.hs_plan.restype = C.c_int
    L.hs_numeric_begin.argtypes = [vp, vp, C.c_int]
    L.hs_numeric_begin.restype = C.c_int
    L.hs_numeric_levels.argtypes = [vp, i64, i64]
    L.hs_numeric_levels.restype = C.c_int
    L.hs_numeric_end.argtypes = [vp]
    L.hs_numeric_end.restype = C.c_int
    for f in (L.hs_solve_fwd_levels, L.hs_solve_bwd_levels):
        f.argtypes = [vp, vp, i64, i64, vp]
        f.restype = C.c_int
    for f in (L.hs_nlevels, L.hs_cut_level, L.hs_num_exchanges):
        f.argtypes = [vp]
        f.restype = i64
    L.hs_node_owner.argtypes = [vp, i64]
    L.hs_node_owner.restype = i64
    L.hs_exchange_info.argtypes = [vp, i64, p_i64]
    L.hs_exchange_info.restype = C.c_int
    L.hs_set_schur_buffer.argtypes = [vp, i64, vp]
    L.hs_set_schur_buffer.restype = C.c_int
    L.hs_exchange_kind.argtypes = [vp, i64]
    L.hs_exchange_kind.restype = i64
    L.hs_schur_pack_size.argtypes = [vp, i64, p_i64]
    L.hs_schur_pack_size.restype = C.c_int
    L.hs_schur_pack.argtypes = [vp, i64, vp, i64, vp]
    L.hs_schur_pack.restype = C.c_int
    L.hs_schur_unpack.argtypes = [vp, i64, vp, i64, vp]
    L.hs_schur_unpack.restype = C.c_int
    L.hs_flow_info.argtypes = [vp, p_i64]
    L.hs_flow_info.restype = C.c_int
    L.hs_hss_qr_order.argtypes = [C.c_int]
    L.hs_hss_qr_order.restype = C.c_int
    L.hs_hss_pack_size.argtypes = [vp, p_i64]
    L.hs_hss_pack_size.restype = C.c_int
    L.hs_hss_pack.argtypes = [vp, vp, i64, vp]
    L.hs_hss_pack.restype = C.c_int
    L.hs_hss_unpack.argtypes = [vp, i64, C.c_int, vp, C.POINTER(vp)]
    L.hs_hss_unpack.restype = C.c_int
    L.hs_pack_bnd.argtypes = [vp, i64, vp, vp, vp]
    L.hs_pack_bnd.restype = C.c_int
    L.hs_unpack_bnd.argtypes = [vp, i64, vp, vp, vp]
    L.hs_unpack_bnd.restype = C.c_int
    L.hs_extract_owned.argtypes = [vp, vp, vp, vp]
    L.hs_extract_owned.restype = C.c_int
    L.hs_comm_unique_id.argtypes = [vp]
    L.hs_comm_unique_id.restype = C.c_int
    L.hs_comm_create_rccl.argtypes = [vp, i64, i64, C.POINTER(vp)]
    L.hs_comm_create_rccl.restype = C.c_int
    L.hs_comm_create_host.argtypes = [HS_TRANSFER_FN, vp, i64, i64, C.POINTER(vp)]
    L.hs_comm_create_host.restype = C.c_int
    L.hs_comm_free.argtypes = [vp]
    L.hs_comm_free.restype = None
    L.hs_comm_kind.argtypes = [vp]
    L.hs_comm_kind.restype = C.c_char_p
    L.hs_comm_selftest.argtypes = [vp, i64]
    L.hs_comm_selftest.restype = C.c_int
    L.hs_comm_bandwidth.argtypes = [vp, i64, i64, C.POINTER(C.c_double)]
    L.hs_comm_bandwidth.restype = C.c_int
    L.hs_set_comm.argtypes = [vp, vp]
    L.hs_set_comm.restype = C.c_int
    for f in (L.hs_gmres_d, L.hs_gmres_z):
        f.argtypes = [vp, i64, p_i64, p_i64, vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, i64, i64, p_f64, p_i64, C.POINTER(C.c_int), vp]
        f.restype = C.c_int
    L.hs_maxrank.argtypes = [vp]
    L.hs_maxrank.restype = i64
    L.hs_is_complex.argtypes = [vp]
    L.hs_is_complex.restype = C.c_int
    L.hs_size.argtypes = [vp]
    L.hs_size.restype = i64
    L.hs_free.argtypes = [vp]
    L.hs_free.restype = None
    L.hs_last_error.argtypes = []
    L.hs_last_error.restype = C.c_char_p
    L.hs_last_error_info.argtypes = []
    L.hs_last_error_info.restype = i64
    L.hs_get_stats.argtypes = [vp, C.POINTER(hs_stats)]
    L.hs_get_stats.restype = C.c_int
    L.hs_node_info.argtypes = [vp, i64, p_i64, p_i64, p_i64]
    L.hs_node_info.restype = C.c_int
    L.hs_symbolic_from_elimtree.argtypes = [i64, p_i64, p_i64, p_i64, p_i64, p_i64, i64, p_i64, p_i64, i64, C.POINTER(vp)]
    L.hs_symbolic_from_elimtree.restype = C.c_int
    L.hs_symbolic_from_graph.argtypes = [i64, p_i64, p_i64, i64, C.POINTER(vp)]
    L.hs_symbolic_from_graph.restype = C.c_int
    L.hs_symbolic_size.argtypes = [vp]
    L.hs_symbolic_size.restype = i64
    L.hs_symbolic_perm.argtypes = [vp]
    L.hs_symbolic_perm.restype = p_i64
    L.hs_symbolic_tree.argtypes = [vp, C.POINTER(hs_tree)]
    L.hs_symbolic_tree.restype = C.c_int
    L.hs_symbolic_free.argtypes = [vp]
    L.hs_symbolic_free.restype = None
    L.hs_node_ranks.argtypes = [vp, i64, p_i64, p_i64]
    L.hs_node_ranks.restype = C.c_int
    L.hs_node_export.argtypes = [vp, i64, C.c_int, p_f64]
    L.hs_node_export.restype = C.c_int
    L.hs_node_export_piv.argtypes = [vp, i64, p_i64]
    L.hs_node_export_piv.restype = C.c_int
    L.hs_device_info.argtypes = [C.c_char_p, i64, p_i64, p_i64]
    L.hs_device_info.restype = C.c_int
    for f in (L.hsk_gemm_d, L.hsk_gemm_z):
        f.argtypes = [i64, i64, i64, p_f64, i64, p_f64, i64, p_f64, i64, C.c_int, C.c_int, p_f64]
        f.restype = C.c_int
    for f in (L.hsk_front_factor_d, L.hsk_front_factor_z):
        f.argtypes = [i64, i64, i64, p_f64, p_f64, p_f64, p_f64, p_i64, p_i64, p_f64]
        f.restype = C.c_int
    for f in (L.hsk_lowrank_d, L.hsk_lowrank_z):
        f.argtypes = [i64, i64, p_f64, C.c_double, C.c_double, i64, i64, p_i64, p_f64, p_f64, i64]
        f.restype = C.c_int
    L.hs_hss_options_default.argtypes = [C.POINTER(hs_hss_options)]
    L.hs_hss_options_default.restype = None
    for f in (L.hs_hss_compress_d, L.hs_hss_compress_z):
        f.argtypes = [i64, vp, i64, C.c_int, C.POINTER(hs_hss_options), C.POINTER(vp)]
        f.restype = C.c_int
    for f in (L.hs_hss_compress_ex_d, L.hs_hss_compress_ex_z):
        f.argtypes = [i64, vp, i64, C.c_int, p_i64, C.POINTER(hs_hss_options), vp, C.POINTER(vp)]
        f.restype = C.c_int
    for f in (L.hs_hss_compress_lru_d, L.hs_hss_compress_lru_z):
        f.argtypes = [i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, C.c_int, p_i64, C.POINTER(hs_hss_options), vp, C.POINTER(vp)]
        f.restype = C.c_int
    for f in (L.hs_hss_compress_lru_multi_d, L.hs_hss_compress_lru_multi_z):
        f.argtypes = [i64, p_i64, C.POINTER(vp), p_i64, C.POINTER(vp), p_i64, C.POINTER(vp), p_i64, C.POINTER(vp), p_i64, p_i64, p_i64, C.POINTER(p_i64),
                      C.POINTER(C.POINTER(hs_hss_options)), vp, C.POINTER(vp)]
        f.restype = C.c_int
    for f in (L.hs_hss_rank, L.hs_hss_size, L.hs_hss_samples, L.hs_hss_num_nodes, L.hs_hss_bytes):
        f.argtypes = [vp]
        f.restype = i64
    L.hs_node_schur_hss.argtypes = [vp, i64, C.POINTER(hs_hss_options), C.POINTER(vp)]
    L.hs_node_schur_hss.restype = C.c_int
    L.hs_hss_set_stream.argtypes = [vp, vp]
    L.hs_hss_set_stream.restype = C.c_int
    L.hs_hss_node_info.argtypes = [vp, i64, p_i64]
    L.hs_hss_node_info.restype = C.c_int
    L.hs_hss_node_data.argtypes = [vp, i64, p_i64, vp, vp, vp, vp]
    L.hs_hss_node_data.restype = C.c_int
    L.hs_hss_getindex.argtypes = [vp, p_i64, i64, p_i64, i64, vp, i64, C.c_int]
    L.hs_hss_getindex.restype = C.c_int
    L.hs_hss_trim.argtypes = []
    L.hs_hss_trim.restype = i64
    L.hs_hss_basis.argtypes = [vp, i64, vp, i64, C.c_int]
    L.hs_hss_basis.restype = C.c_int
    L.hs_hss_expand.argtypes = [vp, vp, i64, C.c_int]
    L.hs_hss_expand.restype = C.c_int
    L.hs_hss_mul.argtypes = [vp, vp, i64, vp, i64, i64, C.c_int]
    L.hs_hss_mul.restype = C.c_int
    L.hs_hss_mul_t.argtypes = [vp, vp, i64, vp, i64, i64, C.c_int]
    L.hs_hss_mul_t.restype = C.c_int
    L.hs_hss_child.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.hs_hss_child.restype = C.c_int
    L.hs_hss_prune_leaves.argtypes = [vp, C.POINTER(vp)]
    L.hs_hss_prune_leaves.restype = C.c_int
    L.hs_hss_compatible.argtypes = [vp, vp]
    L.hs_hss_compatible.restype = C.c_int
    L.hs_hss_depth.argtypes = [vp]
    L.hs_hss_depth.restype = i64
    L.hs_hss_offdiag.argtypes = [vp, C.c_int, vp, i64, vp, i64, C.c_int]
    L.hs_hss_offdiag.restype = C.c_int
    for f in (L.hs_hss_compress_blockop_d, L.hs_hss_compress_blockop_z):
        f.argtypes = [C.POINTER(hs_hss_blockop), vp, i64, vp, i64, vp, i64, i64, i64, p_i64, C.POINTER(hs_hss_options), vp, C.POINTER(vp)]
        f.restype = C.c_int
    L.hs_hss_blockop_apply.argtypes = [C.POINTER(hs_hss_blockop), C.c_int, vp, i64, vp, i64, i64, C.c_int, vp]
    L.hs_hss_blockop_apply.restype = C.c_int
    L.hs_hss_factor.argtypes = [vp]
    L.hs_hss_factor.restype = C.c_int
    L.hs_hss_ldiv.argtypes = [vp, vp, i64, i64, C.c_int]
    L.hs_hss_ldiv.restype = C.c_int
    L.hs_hss_time.argtypes = [vp, C.c_int]
    L.hs_hss_time.restype = C.c_double
    L.hs_hss_free.argtypes = [vp]
    L.hs_hss_free.restype = None
    L.hsk_bisect_perm.argtypes = [i64, p_i64, p_i64, i64, p_i64, p_i64]
    L.hsk_bisect_perm.restype = C.c_int
    L.hs_stream_order.argtypes = [vp, vp, C.c_int]
    L.hs_stream_order.restype = C.c_int
    L.hs_trim.argtypes = []
    L.hs_trim.restype = i64
    L.hs_probs_stats_mode.argtypes = [C.c_int]
    L.hs_probs_stats_mode.restype = C.c_int
    L.hs_probs_stats.argtypes = [p_f64]
    L.hs_probs_stats.restype = C.c_int
    L.hsk_flow_pingpong_us.argtypes = [C.c_int, C.c_int]
    L.hsk_flow_pingpong_us.restype = C.c_double
    L.hsk_mfma_f64_peak.argtypes = [C.c_int, C.c_int]
    L.hsk_mfma_f64_peak.restype = C.c_double
    L.hsk_mfma_f64_peak_random.argtypes = [C.c_int, C.c_int]
    L.hsk_mfma_f64_peak_random.restype = C.c_double
    _lib = L
    return L


class DimensionMismatch(ValueError):
    """Julia ``DimensionMismatch`` (blockmatrix.jl:13-16,116-117; nesteddissection.jl:107)."""


class SingularException(ArithmeticError):
    """``LinearAlgebra.SingularException`` -- what ``\\`` raises in the reference on an exactly singular block."""


class DeviceError(RuntimeError):
    """No usable gfx950 device / HIP runtime failure.  There is no CPU fallback."""


class UnsupportedError(NotImplementedError):
    pass


def check(status):
    if status == HS_OK:
        return
    msg = lib().hs_last_error().decode("utf-8", "replace")
    info = lib().hs_last_error_info()
    exc = {
        HS_ERR_ARGUMENT: ValueError,  # ArgumentError
        HS_ERR_DIMENSION: DimensionMismatch,
        HS_ERR_TREE: RuntimeError,  # ErrorException (factorization.jl:25)
        HS_ERR_SINGULAR: SingularException,
        HS_ERR_HSS_LEAF: RuntimeError,
        HS_ERR_DEVICE: DeviceError,
        HS_ERR_NOMEM: MemoryError,
        HS_ERR_UNSUPPORTED: UnsupportedError,
    }.get(status, RuntimeError)
    e = exc(msg)
    e.info = info
    e.status = status
    raise e
