"""
N>1 path on CPU: two gloo ranks each trace their contiguous block of the source rays
(oracle-backed ops stand-ins, tests/cpu_backend.py) and all-reduce the parameter gradients;
the result must equal the single-process gradient, and after a full optimiser step the
parameters must be bit-identical on both ranks.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, n_rays=240, scene_kw=None, tag=""):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import cpu_backend

    class MP:  # minimal monkeypatch
        @staticmethod
        def setattr(obj, name, val):
            setattr(obj, name, val)

    cpu_backend.install(MP)
    from tensorflowraytrace_amd import distributed as tdist
    import tfrt.optimizer as optimizer
    from test_host_logic import _lens_api
    if world > 1:
        tdist.init_from_env(backend="gloo")
    eng, system, lens, target = _lens_api(n_rays, k=2, **(scene_kw or {}))

    def erf(engine):
        fin = engine.finished_rays
        out = torch.stack([fin["y_end"], fin["z_end"]], 1)
        return (out + fin["object_coords"][:, 1:]) ** 2

    opt = optimizer.SGD_Optimizer(eng, lens.parameters, erf, 3, learning_rate=1.0, grad_clip=1e9)
    grads, err_sum, n_terms = opt.raw_gradient()
    n_local = eng.finished_rays["x_start"].shape[0]
    opt.single_step(None)
    np.savez(os.path.join(out_dir, f"r{tag}{world}_{rank}.npz"),
             g0=grads[0].numpy(), g1=grads[1].numpy(), err=float(err_sum), n=n_terms,
             n_local=n_local, p0=lens.parameters[0].detach().numpy(),
             p1=lens.parameters[1].detach().numpy())
    if world > 1:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_equals_single_process(tmp_path):
    out = str(tmp_path)
    mp.spawn(_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    one = np.load(os.path.join(out, "r1_0.npz"))
    a = np.load(os.path.join(out, "r2_0.npz"))
    b = np.load(os.path.join(out, "r2_1.npz"))
    assert int(a["n_local"]) + int(b["n_local"]) == int(one["n_local"])  # rays really sharded
    assert 0 < int(a["n_local"]) < int(one["n_local"])
    assert int(a["n"]) == int(b["n"]) == int(one["n"])
    for k in ("g0", "g1"):
        np.testing.assert_array_equal(a[k], b[k])               # identical after all-reduce
        scale = np.abs(one[k]).max()
        assert np.abs(a[k] - one[k]).max() <= 1e-12 * scale     # equals the unsharded gradient
    assert abs(float(a["err"]) - float(one["err"])) <= 1e-12 * abs(float(one["err"]))
    for k in ("p0", "p1"):
        np.testing.assert_array_equal(a[k], b[k])               # parameters stay in lock-step
        np.testing.assert_allclose(a[k], one[k], rtol=0, atol=1e-11)  # summation order only


@pytest.mark.timeout(300)
def test_four_ranks_uneven_shards_and_a_rank_without_finished_rays(tmp_path):
    """243 rays over 4 ranks (61 + 61 + 61 + 60: N not divisible by the world size), on a scene
    whose outer rays miss lens and target -- the source lists its rays from the axis outwards, so
    the last rank's whole shard dies and contributes no error term and a zero gradient: the
    all-reduced gradient, error sum and term count must still equal the single-process ones, and
    every rank must apply the same update."""
    out = str(tmp_path)
    kw = dict(end_radius=1.3, target_size=3.0)
    mp.spawn(_worker, args=(1, _free_port(), out, 243, kw, "u"), nprocs=1, join=True)
    mp.spawn(_worker, args=(4, _free_port(), out, 243, kw, "u"), nprocs=4, join=True)
    one = np.load(os.path.join(out, "ru1_0.npz"))
    ranks = [np.load(os.path.join(out, f"ru4_{r}.npz")) for r in range(4)]
    n_local = [int(r["n_local"]) for r in ranks]
    assert sum(n_local) == int(one["n_local"]) > 0
    assert n_local[3] == 0 and n_local[0] > 0              # the outermost shard finishes nothing
    assert 0 < int(one["n_local"]) < 243                   # ... and some rays of the scene die
    for r in ranks:
        assert int(r["n"]) == int(one["n"])
        for k in ("g0", "g1"):
            np.testing.assert_array_equal(r[k], ranks[0][k])
            assert np.abs(r[k] - one[k]).max() <= 1e-12 * np.abs(one[k]).max()
        assert abs(float(r["err"]) - float(one["err"])) <= 1e-12 * abs(float(one["err"]))
        for k in ("p0", "p1"):
            np.testing.assert_array_equal(r[k], ranks[0][k])
            np.testing.assert_allclose(r[k], one[k], rtol=0, atol=1e-11)
