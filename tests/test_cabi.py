"""
The C-ABI library loads without a GPU and exports every symbol include/tfrt_hip.h declares;
host-only entry points (version, strerror, workspace sizing, argument validation) behave.
No kernel is launched here.
"""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tfrt_hip.h")


@pytest.fixture(scope="module")
def lib():
    from tensorflowraytrace_amd import _build, _lib
    _build.build()
    return _lib.lib()


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfrt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from tensorflowraytrace_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 18
    assert sorted(_lib.SIGNATURES) == declared, "ctypes binding and header disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r"\bT (tfrt_[a-z0-9_]+)", out))
    assert set(declared) <= exported, set(declared) - exported


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "tfrt_hip.h"\nint main(void){return TFRT_COUNTS_LEN(3) == 32 ? 0 : 1;}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.dirname(HEADER), str(src),
                    "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_version_and_strerror(lib):
    from tensorflowraytrace_amd import _lib
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "tfrt_hip.h")).read()
    declared = int(re.search(r"#define TFRT_VERSION (\d+)", header).group(1))
    assert lib.tfrt_version() == declared == _lib.ABI_VERSION == 107
    assert lib.tfrt_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -99):
        assert len(lib.tfrt_strerror(code)) > 0


def test_workspace_sizes(lib):
    a = lib.tfrt_trace3d_workspace_bytes(1000, 100, 3, 0)
    b = lib.tfrt_trace3d_workspace_bytes(2000, 100, 3, 0)
    c = lib.tfrt_trace3d_workspace_bytes(1000, 100, 6, 0)
    d = lib.tfrt_trace3d_workspace_bytes(1000, 100, 3, 1)
    assert 0 < a < b and a < c and a < d
    assert lib.tfrt_trace3d_workspace_bytes(-1, 100, 3, 0) == 0
    assert lib.tfrt_trace2d_workspace_bytes(1000, 10, 10, 3, 0) > 0
    assert lib.tfrt_intersect3d_workspace_bytes(1000, 100) > 0
    # 1M rays x 10k faces x 5 passes stays far below HBM capacity
    assert lib.tfrt_trace3d_workspace_bytes(1_000_000, 10_574, 5, 0) < 2 * 1024 ** 3


def test_bad_arguments_are_rejected_before_any_launch(lib):
    from tensorflowraytrace_amd import _lib
    sc = _lib.Scene3D()
    sc.n_faces = -1
    code = lib.tfrt_trace3d_forward(None, 0, 10, ctypes.byref(sc), 1.0, 0.0, 3, 0, 3, None, None,
                                    None, None, None, None, None, None, 0, None)
    assert code == -1
    assert lib.tfrt_build_faces_forward(None, -1, None, 5, None, None, None) == -1
    assert lib.tfrt_snell3d(-5, *([None] * 9), 1.0, None, None) == -1
    assert lib.tfrt_segment_intersection(None, 0, -1, 0, None, 0, 0.0, 0.0, 0.0, None, None, None,
                                         None, None, None, None) == -1
    # round 4 entry points: the folded reverse sweep, the orders, the source programs
    ok = _lib.Scene3D()
    ok.n_faces = 0
    fin, pend = _lib.RayOut(), _lib.GoalPending()
    fields = (ctypes.c_int32 * 6)(4, 5, 0, 0, 0, 0)
    dummy = ctypes.create_string_buffer(1 << 16)
    ptr = ctypes.cast(dummy, ctypes.c_void_p)

    def backward_goal(n_fields=2, goal_ws_bytes=1 << 16, pending=ctypes.byref(pend), stride=10):
        return lib.tfrt_trace3d_backward_goal(
            None, 0, 10, ctypes.byref(ok), 1.0, 0.0, 3, 0, ctypes.byref(fin), fields, n_fields, ptr,
            stride, 1, ptr, None, ptr, goal_ws_bytes, pending, None, 0, None, 0, None, 0, ptr, None,
            ptr, ptr, 1 << 16, None)
    assert backward_goal(n_fields=0) == -1 and backward_goal(n_fields=7) == -1
    assert backward_goal(goal_ws_bytes=0) == -1 and backward_goal(pending=None) == -1
    assert backward_goal(stride=-1) == -1
    assert backward_goal() == -1                     # (no finished-ray block: rays is NULL)
    assert lib.tfrt_trace3d_backward_goal_workspace_bytes(1_000_000) >= 15_625 * 8
    assert lib.tfrt_trace3d_backward_goal_workspace_bytes(-1) == 0
    assert lib.tfrt_ray_order(None, 0, -1, 0, None, 0, None, None, None, None, 0, None) == -1
    assert lib.tfrt_ray_order_workspace_bytes(1_000_000) > 0
    assert lib.tfrt_source3d_order(None, 0, 10, None, 0, None, None, None, None, 0, None) == -1
    # a program tfrt_source3d_generate refuses is refused by tfrt_source3d_order too (same check):
    # unknown kind, a random distribution without an epoch counter, a table without storage, an
    # input that has neither one sample nor one per ray
    def program(kind=_lib.SRC_APERTURE, a_kind=_lib.PTS_CIRCLE, a_count=10, epoch=ptr, table=None):
        sp = _lib.Source3DProgram()
        sp.kind, sp.n_rays = kind, 10
        for pg, knd, cnt in ((sp.a, a_kind, a_count), (sp.b, _lib.PTS_CIRCLE, 10)):
            pg.kind, pg.count, pg.epoch, pg.table = knd, cnt, epoch, table
        return sp

    for bad in (program(kind=7), program(epoch=None), program(a_kind=_lib.PTS_TABLE),
                program(a_kind=9), program(a_count=3)):
        args = (ctypes.byref(bad), 0, 10, None, 0, None, ptr, None, ptr, 1 << 16, None)
        assert lib.tfrt_source3d_order(*args) == -1
        assert lib.tfrt_source3d_generate(ctypes.byref(bad), None, 0, 10, 0, ptr, 10, None, 0,
                                          None) == -1
    assert lib.tfrt_epoch_advance(None, 9, None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from tensorflowraytrace_amd import ops, _lib
    with pytest.raises(_lib.TfrtError):
        ops.build_faces(torch.zeros(3, 3, dtype=torch.float64), torch.zeros(1, 3, dtype=torch.int32))
    with pytest.raises(_lib.TfrtError):
        ops.intersect3d(torch.zeros(6, 4), torch.zeros(2, 9, dtype=torch.float64))
