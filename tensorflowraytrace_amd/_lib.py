"""
ctypes binding of libtfrt_hip.so (declared in include/tfrt_hip.h).

There is deliberately NO fallback: if the shared library is missing, or a tensor handed to a
kernel wrapper is not on a HIP device, the call raises.  Build the library with
``python -m tensorflowraytrace_amd._build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFRT_LIB_PATH") or os.path.join(HERE, "libtfrt_hip.so")

ABI_VERSION = 107          # TFRT_VERSION of include/tfrt_hip.h this module's signatures are written for
F32, F64, F16 = 0, 1, 2
OPTICAL, STOP, TARGET = 0, 1, 2
CLS_ACTIVE, CLS_FINISHED, CLS_STOPPED, CLS_DEAD = 0, 1, 2, 3
COMPILE_ACTIVE, COMPILE_FINISHED, COMPILE_STOPPED, COMPILE_DEAD = 1, 2, 4, 8
COUNTS_PER_PASS = 8

c_i32, c_i64, c_f64, c_u32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_uint32
c_vp, c_sz = ctypes.c_void_p, ctypes.c_size_t


class Scene3D(ctypes.Structure):
    """struct tfrt_scene3d"""
    _fields_ = [
        ("face_verts", c_vp), ("catagory", c_vp), ("mat_in", c_vp), ("mat_out", c_vp),
        ("n_in", c_vp), ("n_out", c_vp), ("n_faces", c_i64),
        ("n_table", c_vp), ("n_table_stride", c_i64), ("n_materials", c_i32),
        ("intersect_epsilion", c_f64), ("size_epsilion", c_f64), ("ray_start_epsilion", c_f64),
        ("face_grad_mask", c_vp), ("cluster_order", c_vp), ("n_table_uniform", c_i32),
        ("deterministic", c_i32), ("coherent_rays", c_i32), ("coherent_only", c_i32),
        ("grad_n_in", c_vp), ("grad_n_out", c_vp), ("clear_buffer", c_vp), ("clear_count", c_i64),
        ("in_place", c_i32), ("ray_slot", c_vp),
    ]


class Scene2D(ctypes.Structure):
    """struct tfrt_scene2d"""
    _fields_ = [
        ("seg", c_vp), ("seg_cat", c_vp), ("seg_mat_in", c_vp), ("seg_mat_out", c_vp),
        ("seg_n_in", c_vp), ("seg_n_out", c_vp), ("n_segments", c_i64),
        ("arc", c_vp), ("arc_cat", c_vp), ("arc_mat_in", c_vp), ("arc_mat_out", c_vp),
        ("arc_n_in", c_vp), ("arc_n_out", c_vp), ("n_arcs", c_i64),
        ("n_table", c_vp), ("n_table_stride", c_i64), ("n_materials", c_i32),
        ("intersect_epsilion", c_f64), ("size_epsilion", c_f64), ("ray_start_epsilion", c_f64),
        ("finite_tir_gradient", c_i32),
    ]


MAX_SURFACES = 8   # TFRT_MAX_SURFACES


class FaceSurface(ctypes.Structure):
    """tfrt_face_surface (include/tfrt_hip.h)."""
    _fields_ = [("zero_points", ctypes.c_void_p), ("vectors", ctypes.c_void_p),
                ("parameters", ctypes.c_void_p), ("faces", ctypes.c_void_p),
                ("n_vertices", ctypes.c_int64), ("n_faces", ctypes.c_int64),
                ("face_verts", ctypes.c_void_p), ("norm", ctypes.c_void_p),
                ("copy_from", ctypes.c_void_p)]


class FaceSurfaceGrad(ctypes.Structure):
    """tfrt_face_surface_grad (include/tfrt_hip.h)."""
    _fields_ = [("grad_face_verts", ctypes.c_void_p), ("grad_norm", ctypes.c_void_p),
                ("face_verts", ctypes.c_void_p), ("update_mask", ctypes.c_void_p),
                ("vectors", ctypes.c_void_p), ("corner_start", ctypes.c_void_p),
                ("corner_list", ctypes.c_void_p), ("n_vertices", ctypes.c_int64),
                ("grad_parameters", ctypes.c_void_p)]


class GoalPending(ctypes.Structure):
    """tfrt_goal_pending (include/tfrt_hip.h)."""
    _fields_ = [("partial", ctypes.c_void_p), ("n_partial", ctypes.c_int32),
                ("n_finished", ctypes.c_void_p), ("n_fields", ctypes.c_int32),
                ("error_out", ctypes.c_void_p), ("tests_lo_hi", ctypes.c_void_p),
                ("tests_total", ctypes.c_void_p), ("partial_counts", ctypes.c_void_p),
                ("n_faces", ctypes.c_int64), ("counts_tail", ctypes.c_void_p)]


class RayOut(ctypes.Structure):
    """struct tfrt_ray_out"""
    _fields_ = [("rays", c_vp), ("ray_id", c_vp), ("face", c_vp), ("capacity", c_i64)]


class PointsProgram(ctypes.Structure):
    """tfrt_points_program (include/tfrt_hip.h)."""
    _fields_ = [("kind", c_i32), ("stream", c_i32), ("count", c_i64), ("table", c_vp),
                ("p", c_f64 * 4), ("scale", c_f64 * 3), ("quat", c_f64 * 4), ("shift", c_f64 * 3),
                ("has_scale", c_i32), ("has_quat", c_i32), ("has_shift", c_i32),
                ("reserved0", c_i32), ("seed", ctypes.c_uint64), ("epoch", c_vp)]


class Source3DProgram(ctypes.Structure):
    """tfrt_source3d_program (include/tfrt_hip.h)."""
    _fields_ = [("kind", c_i32), ("swap", c_i32), ("a", PointsProgram), ("b", PointsProgram),
                ("center", c_f64 * 3), ("quat", c_f64 * 4), ("has_quat", c_i32),
                ("reserved0", c_i32), ("ray_length", c_f64), ("n_rays", c_i64)]


PTS_TABLE, PTS_CIRCLE, PTS_SQUARE, PTS_SPHERE_UNIFORM, PTS_SPHERE_LAMBERT = 0, 1, 2, 3, 4
SRC_APERTURE, SRC_POINT, SRC_ANGULAR = 0, 1, 2

_P = ctypes.POINTER

# name -> (restype, argtypes); must list every symbol include/tfrt_hip.h declares
SIGNATURES = {
    "tfrt_version": (c_i32, []),
    "tfrt_strerror": (ctypes.c_char_p, [c_i32]),
    "tfrt_profile_enable": (c_i32, [c_i32]),
    "tfrt_profile_read": (c_i32, [c_vp, c_i32]),
    "tfrt_profile_read_kind": (c_i32, [c_i32, c_vp, c_i32]),
    "tfrt_build_faces_forward": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "tfrt_build_faces_backward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp,
                                          c_vp, c_vp]),
    "tfrt_param_faces_forward": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "tfrt_param_faces_backward": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp,
                                          c_vp, c_vp, c_vp]),
    "tfrt_param_faces_forward_multi": (c_i32, [c_vp, c_i32, c_vp]),
    "tfrt_param_faces_backward_multi": (c_i32, [c_vp, c_i32, c_vp]),
    "tfrt_line_intersect": (c_i32, [c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_f64,
                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_line_triangle_intersect": (c_i32, [c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                             c_f64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_line_circle_intersect": (c_i32, [c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                           c_f64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_sgd_process": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_f64, c_f64, c_f64, c_vp]),
    "tfrt_sgd_process_dev": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "tfrt_sgd_process_multi": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_sgd_process_multi_finish": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_csr_matvec": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tfrt_trace3d_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32, c_i32]),
    "tfrt_trace3d_forward": (c_i32, [
        c_vp, c_i64, c_i64, _P(Scene3D), c_f64, c_f64, c_i32, c_i32, c_u32,
        _P(RayOut), _P(RayOut), _P(RayOut), _P(RayOut), c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_trace3d_compact": (c_i32, [
        c_vp, c_i64, c_i64, c_f64, c_i32, c_i32, c_u32,
        _P(RayOut), _P(RayOut), _P(RayOut), _P(RayOut), c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz,
        c_vp]),
    "tfrt_trace3d_in_place": (c_i32, [_P(Scene3D), c_i64, c_i32]),
    "tfrt_trace3d_executed": (c_i32, [c_i64, c_i64, c_i32, c_i32, c_vp, c_sz, c_vp, c_vp]),
    "tfrt_trace3d_backward": (c_i32, [
        c_vp, c_i64, c_i64, _P(Scene3D), c_f64, c_f64, c_i32, c_i32,
        c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_goal_error3d_workspace_bytes": (c_sz, [c_i64]),
    "tfrt_goal_error3d": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i64,
                                  c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_goal_error3d_deferred": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i64,
                                  c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "tfrt_goal_finish": (c_i32, [c_vp, c_vp]),
    "tfrt_trace3d_backward_goal_workspace_bytes": (c_sz, [c_i64]),
    "tfrt_trace3d_backward_goal": (c_i32, [
        c_vp, c_i64, c_i64, c_vp, c_f64, c_f64, c_i32, c_i32,          # rays, scene, lengths, P, dtype
        c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp,   # goal
        c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_intersect3d_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "tfrt_intersect3d": (c_i32, [
        c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_f64, c_f64, c_f64,
        c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_snell3d": (c_i32, [c_i64] + [c_vp] * 9 + [c_f64, c_vp, c_vp]),
    "tfrt_snell2d": (c_i32, [c_i64] + [c_vp] * 7 + [c_f64, c_vp, c_vp]),
    "tfrt_selftest_f64": (c_i32, [c_i32, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_segment_intersection": (c_i32, [
        c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_f64, c_f64, c_f64,
        c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_arc_intersection": (c_i32, [
        c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_f64, c_f64, c_f64,
        c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tfrt_trace2d_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i32, c_i32]),
    "tfrt_trace2d_forward": (c_i32, [
        c_vp, c_i64, c_i64, _P(Scene2D), c_f64, c_f64, c_i32, c_i32, c_u32,
        _P(RayOut), _P(RayOut), _P(RayOut), _P(RayOut), c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_trace2d_backward": (c_i32, [
        c_vp, c_i64, c_i64, _P(Scene2D), c_f64, c_f64, c_i32, c_i32,
        c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
        c_vp]),
    "tfrt_ray_order_workspace_bytes": (c_sz, [c_i64]),
    "tfrt_ray_order": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz,
                               c_vp]),
    "tfrt_permute_rays_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "tfrt_permute_rays": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "tfrt_gather_rows": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "tfrt_cluster_order_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "tfrt_cluster_order": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_restore_order_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "tfrt_restore_order": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp,
                                   c_vp, c_vp, c_vp, c_sz, c_vp]),
    "tfrt_source3d_order": (c_i32, [_P(Source3DProgram), c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp,
                                    c_vp, c_sz, c_vp]),
    "tfrt_source3d_order_cells": (c_i32, [_P(Source3DProgram), c_i64, c_i64, c_vp, c_i64, c_vp, c_vp,
                                          c_vp, c_vp, c_sz, c_vp]),
    "tfrt_epoch_advance": (c_i32, [c_vp, c_i32, c_vp]),
    "tfrt_points_generate": (c_i32, [_P(PointsProgram), c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp,
                                     c_vp]),
    "tfrt_source3d_generate": (c_i32, [_P(Source3DProgram), c_vp, c_i64, c_i64, c_i32, c_vp, c_i64,
                                       c_vp, c_i64, c_vp]),
}

_lib = None


class TfrtError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TfrtError(
                f"{LIB_PATH} not found: the tfrt HIP extension is not built.  Run "
                "`python -m tensorflowraytrace_amd._build`.  There is no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().tfrt_strerror(code).decode()
        raise TfrtError(f"{what} failed: {msg} (code {code})")
