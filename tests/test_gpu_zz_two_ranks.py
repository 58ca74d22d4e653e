"""
The sharded FUSED step with a real process group of two ranks.  A 1-GPU box cannot hold two RCCL
ranks (RCCL wants distinct devices), so both processes use cuda:0 and the collective goes through
gloo (host-staged all-reduce of the ~3 KB gradient buffer): everything except the transport is the
code an 8-GPU run executes -- contiguous ray shards (rank r of 2), the two captured HIP graphs with
the all-reduce between them, the device-side error count, identical parameters on both ranks.
Eight steps (three eager, capture, replays) must reproduce the single-process run.
(Runs last, in child processes; two ranks on the card are within the pool's process limit.)
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import bench
import tensorflowraytrace_amd as tfa
from tensorflowraytrace_amd import distributed as tdist
import tfrt.optimizer as optimizer
rank, world, _ = tdist.init_from_env(backend="gloo")
torch.cuda.set_device(0)
tfa.set_device("cuda:0")
eng, system, params = bench.build_scene(60_000, 9, 5, torch.float64)
opt = optimizer.SGD_Optimizer(eng, params, bench.make_error_function(), trace_depth=3,
                              learning_rate=1e-5, grad_clip=1e-3)
opt.suppress_warnings = True
errs = [float(opt.single_step(None, lr_scale=1.0 - 0.05 * k)) for k in range(8)]
fs = opt._fused_step
n_local = int(eng.finished_rays["x_start"].shape[0])
np.savez(os.path.join(sys.argv[2], f"w{world}_r{rank}.npz"), errs=np.array(errs),
         p0=params[0].detach().cpu().numpy(), p1=params[1].detach().cpu().numpy(), n_local=n_local,
         replays=fs.graph_replays, capture_error=str(fs.capture_error),
         terms=float(opt.last_error_terms))
if world > 1:
    torch.distributed.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, out_dir):
    root = os.path.dirname(HERE)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TFRT_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, root, out_dir], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode(errors="replace")[-3000:]


@pytest.mark.timeout(600)
def test_two_ranks_sharing_the_card_reproduce_the_single_process_run(tmp_path):
    out = str(tmp_path)
    _run(1, out)
    _run(2, out)
    one = np.load(os.path.join(out, "w1_r0.npz"))
    a = np.load(os.path.join(out, "w2_r0.npz"))
    b = np.load(os.path.join(out, "w2_r1.npz"))
    for r in (one, a, b):
        assert str(r["capture_error"]) == "None" and int(r["replays"]) >= 4
    assert int(a["n_local"]) + int(b["n_local"]) == int(one["n_local"])     # rays really sharded
    assert 0 < int(a["n_local"]) < int(one["n_local"])
    assert float(a["terms"]) == float(b["terms"]) == float(one["terms"])    # reduced error count
    np.testing.assert_array_equal(a["errs"], b["errs"])
    np.testing.assert_allclose(a["errs"], one["errs"], rtol=1e-11, atol=0)
    for k in ("p0", "p1"):
        np.testing.assert_array_equal(a[k], b[k])                # parameters stay in lock-step
        np.testing.assert_allclose(a[k], one[k], rtol=0, atol=1e-12)   # summation order only
