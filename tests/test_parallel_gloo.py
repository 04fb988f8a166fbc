"""world_size-2 CPU tests (gloo) of the data-parallel glue: replicas that sum gradients between
backward and ADAM stay bit-identical and reproduce single-process training on the union minibatch."""
import os
import socket
import sys

import numpy as np
import pytest

import util as U


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import importlib
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(U.ROOT, "oracle"))
    import ddpg_oracle as DO
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = importlib.import_module(U.PKG_NAME + ".parallel")
    # the native communicator (RCCL in the update's stream, csrc/shems_dp.hip) cannot exist without a GPU: every rank must come out of
    # the attempt with None -- by vote, with the same number of collectives on every rank (a mismatch would hang right here) -- and
    # the gradients then travel through torch.distributed
    msgs = []
    assert P.native_comm(dist, timeout_s=30.0, log=msgs.append) is None and msgs
    assert P.direct_comm(dist, log=msgs.append) is None  # the direct exchange needs device memory too: dropped on every rank, by vote
    os.environ["SHEMS_DP"] = "torch"
    assert P.native_comm(dist) is None                  # switched off: no attempt, no collective
    del os.environ["SHEMS_DP"]
    sync = P.GradSync(dist)
    assert sync.world == world and sync.rank == rank and sync.grad_scale == 1.0 / world and sync.native is None
    rng = np.random.default_rng(0)                      # same data on both ranks, each takes its half
    B = 120
    s = rng.random((2 * B, 9)).astype(np.float32); a = (rng.random((2 * B, 2)) * 2 - 1).astype(np.float32)
    r = rng.normal(size=2 * B).astype(np.float32); s2 = rng.random((2 * B, 9)).astype(np.float32)
    done = np.zeros(2 * B, bool)
    sl = slice(rank * B, (rank + 1) * B)
    # rank 1 starts from different weights: broadcast must overwrite them
    actor = torch.from_numpy(DO.init_params(1231 + rank, 9, 2, 0)); critic = torch.from_numpy(DO.init_params(1231 + rank, 11, 1, 1))
    sync.broadcast(actor, critic)
    L = DO.Learner(actor.numpy(), critic.numpy(), np.zeros(9, np.float32), np.ones(9, np.float32))

    def allreduce(g):
        t = torch.from_numpy(g.copy())
        sync.sum_(t)
        return (t.numpy().astype(np.float64) * sync.grad_scale).astype(np.float32)

    for _ in range(3):
        L.replay(s[sl], a[sl], r[sl], s2[sl], done[sl], allreduce=allreduce)
    mn = torch.from_numpy(s[sl].min(0).copy()); mx = torch.from_numpy(s[sl].max(0).copy())
    sync.minmax_(mn, mx)
    score = sync.mean_scalar(float(rank + 1), weight=10 * (rank + 1))      # (1*10 + 2*20) / 30
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), actor=L.actor, critic=L.critic, actor_t=L.actor_t, mn=mn.numpy(),
             mx=mx.numpy(), score=score, off=np.array(P.shard_envs(65536 + 3, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_stay_identical_and_match_union_batch(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    import ddpg_oracle as DO
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for k in ("actor", "critic", "actor_t", "mn", "mx"):
        assert (r0[k] == r1[k]).all(), k                       # replicas bit-identical
    assert abs(float(r0["score"]) - 50.0 / 30.0) < 1e-12 and float(r0["score"]) == float(r1["score"])
    assert tuple(r0["off"]) == (0, 32770) and tuple(r1["off"]) == (32770, 32769)
    # single process on the union minibatch (mean over 240 samples = mean of the two 120-sample means)
    rng = np.random.default_rng(0)
    B = 120
    s = rng.random((2 * B, 9)).astype(np.float32); a = (rng.random((2 * B, 2)) * 2 - 1).astype(np.float32)
    r = rng.normal(size=2 * B).astype(np.float32); s2 = rng.random((2 * B, 9)).astype(np.float32)
    L = DO.Learner(DO.init_params(1231, 9, 2, 0), DO.init_params(1231, 11, 1, 1), np.zeros(9, np.float32), np.ones(9, np.float32))
    for _ in range(3):
        L.replay(s, a, r, s2, np.zeros(2 * B, bool))
    # same update up to fp32 summation order; ADAM's first steps are sign-like, so compare loosely but meaningfully
    assert np.abs(L.critic - r0["critic"]).max() < 2.5e-3 and np.mean(np.abs(L.critic - r0["critic"]) < 1e-5) > 0.97
    assert np.mean(np.abs(L.actor - r0["actor"]) < 1e-6) > 0.97
    assert (r0["mn"] == s.min(0)).all() and (r0["mx"] == s.max(0)).all()


def test_shard_envs_partition():
    import importlib
    P = importlib.import_module(U.PKG_NAME + ".parallel")
    for total, world in ((65536, 8), (65536, 1), (10, 4), (7, 8)):
        parts = [P.shard_envs(total, r, world) for r in range(world)]
        assert sum(c for _, c in parts) == total
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1)) and parts[0][0] == 0
    assert P.shard_envs(65536, 3, 8) == (3 * 8192, 8192)
    g = P.GradSync(None)
    assert g.world == 1 and g.grad_scale == 1.0 and g.mean_scalar(3.5) == 3.5


def test_native_comm_gives_up_a_late_rendezvous_and_the_late_thread_destroys_its_own_communicator(monkeypatch):
    """the helper thread of native_comm used to write into the caller's handle after it had been abandoned -- a
    rendezvous completing late left a live communicator nobody destroyed.  With a stand-in library whose create() takes longer than the
    timeout: the call returns None (vote), and the late thread destroys what it created, exactly once."""
    import importlib
    import threading
    import time
    import types
    import torch
    import torch.distributed as dist
    P = importlib.import_module(U.PKG_NAME + ".parallel")
    capi = importlib.import_module(U.PKG_NAME + "._capi")
    destroyed, created = [], threading.Event()

    def unique_id(buf):
        return 0

    def create(idb, rank, world, href):
        time.sleep(0.6)
        href._obj.value = 0x1234
        created.set()
        return 0

    def destroy(h):
        destroyed.append(h.value)
        return 0

    def allreduce(*a):
        raise AssertionError("the self-test must not run on an attempt that was given up")

    def last_error():
        return b"stand-in"
    fake = types.SimpleNamespace(shems_dp_unique_id=unique_id, shems_dp_create=create, shems_dp_destroy=destroy,
                                 shems_dp_allreduce_sum=allreduce, shems_last_error=last_error)
    monkeypatch.setattr(capi, "lib", lambda: fake)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        msgs = []
        assert P.native_comm(dist, timeout_s=0.1, log=msgs.append) is None and msgs
        assert destroyed == []                          # nothing existed yet when the attempt was given up
        assert created.wait(5.0)
        for _ in range(50):
            if destroyed:
                break
            time.sleep(0.02)
        assert destroyed == [0x1234]
        # and an attempt that completes in time hands its handle over (self-test fails on the stand-in -> dropped and destroyed by the caller)
        destroyed.clear()
        fake.shems_dp_allreduce_sum = lambda *a: 1
        assert P.native_comm(dist, timeout_s=5.0, log=msgs.append) is None
        assert destroyed == [0x1234]
    finally:
        dist.destroy_process_group()
