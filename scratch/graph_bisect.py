"""Which part of the fused step breaks hipGraph capture?  Usage: graph_bisect.py STAGE"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import faulthandler; faulthandler.enable()
import numpy as np, torch
from tensorflowraytrace_amd import ops, _lib
stage = sys.argv[1]
dev = "cuda:0"
L = _lib.lib()
if stage == "torch":
    x = torch.zeros(1000, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        y = x * 2 + 1
    g.replay(); torch.cuda.synchronize(); print("torch ok", float(y.sum()))
elif stage == "selftest":
    a = torch.rand(1000, dtype=torch.float64, device=dev) + 1; b = a.clone(); out = torch.empty_like(a)
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        _lib.check(L.tfrt_selftest_f64(0, 1000, ops._p(a), ops._p(b), ops._p(out), ops._stream(a)), "x")
    g.replay(); torch.cuda.synchronize(); print("selftest ok", float(out.sum()))
else:
    from test_gpu_fused_step import _make
    opt, eng, system, lens, *_ = _make(2000, "eager")
    for _ in range(3): opt.single_step(None)
    torch.cuda.synchronize()
    fs = opt._fused_step
    st = fs._state
    src = eng._source_set(); block, scene, fv = eng._trace_inputs(src)
    g = torch.cuda.CUDAGraph()
    if stage == "update":
        with torch.cuda.graph(g):
            system.update()
    elif stage == "forward":
        fvc = fv.detach(); sc = scene.struct(fvc); o = st["outs"]
        with torch.cuda.graph(g):
            stream = ops._stream(block)
            _lib.check(L.tfrt_trace3d_forward(
                ops._p(block), block.shape[1], st["N"], ctypes.byref(sc), 1.0, 0.0, st["P"], st["dt"], st["flags"],
                ctypes.byref(o["finished"]), ctypes.byref(o["active"]), ctypes.byref(o["stopped"]), ctypes.byref(o["dead"]),
                ops._p(st["aux"]["unfinished"]), ops._p(st["aux"]["unfinished_id"]), ops._p(st["counts"]),
                ops._p(st["ws"]), st["wsb"], stream), "fwd")
    elif stage in ("goal", "add", "backward", "autograd"):
        erf = opt.error_function; goal = erf.table(src); P = st["P"]; dt = st["dt"]
        fvc = fv.detach(); sc = scene.struct(fvc)
        n_fin_ptr = ctypes.c_void_p(st["counts"].data_ptr() + 4 * (P * 8 + 1))
        with torch.cuda.graph(g):
            stream = ops._stream(block)
            if stage == "goal":
                _lib.check(L.tfrt_goal_error3d(
                    ops._p(st["full"]["finished"]), st["capN"], ops._p(st["aux"]["finished_id"]), dt, n_fin_ptr,
                    st["fields"], 2, ops._p(goal), goal.shape[1], ops._p(st["g_fin"]),
                    ops._p(st["err"]), ops._p(st["goal_ws"]), st["gws"], stream), "goal")
            elif stage == "add":
                tail = st["counts"][P * 8 + 4:P * 8 + 6]
                fs.tests_total.add_(tail.view(torch.int64))
                st["g_fv"].zero_()
            elif stage == "backward":
                _lib.check(L.tfrt_trace3d_backward(
                    ops._p(block), block.shape[1], st["N"], ctypes.byref(sc), 1.0, 0.0, P, dt,
                    ops._p(st["g_fin"]), st["capN"], None, 0, None, 0, None, 0, ops._p(st["g_fv"]),
                    None, ops._p(st["counts"]), ops._p(st["ws"]), st["wsb"], stream), "bwd")
            else:
                system.update()
                fv2 = system._merged_face_verts
                with torch.autograd.set_multithreading_enabled(False):
                    gr = torch.autograd.grad([fv2], opt.parameters, grad_outputs=[st["g_fv"]], allow_unused=True)
    elif stage == "gradient":
        with torch.cuda.graph(g):
            fs._enqueue_gradient()
    elif stage == "all":
        with torch.cuda.graph(g):
            fs._sequence([None, None], 1)
    print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed ok")
