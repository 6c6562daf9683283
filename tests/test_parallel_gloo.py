"""The N>1 path on CPU: world_size-2 and world_size-8 gloo processes shard the chunk rows, evaluate their share
(the oracle stands in for the GPU kernel here -- tests may use it as a checker/evaluator, the
product never does) and combine value + gradient in ONE all-reduce.  The result must equal the
unsharded evaluation, including a rank that owns nothing from the minibatch."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleKern:
    """Stands in for PSMCKernel.value_and_grad on CPU (checker role only)."""

    def __init__(self, rows, overlap):
        self.rows, self.overlap, self.device = rows, overlap, torch.device("cpu")

    def value_and_grad(self, pp, inds, reduce_chunks=True):
        from oracle import cport

        P = pp.stack().numpy()[:, None]
        ll, g = cport.batch(P, self.rows, np.asarray(inds), self.overlap, nthreads=2)
        return torch.tensor(ll.sum(1)), torch.tensor(g.sum(1))


def _worker(rank, size, port, minibatch, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from phlash_amd import parallel
        from phlash_amd.params import MCMCParams, PSMCParams

        rng = np.random.default_rng(0)
        chunks = (rng.uniform(size=(7, 260)) < 0.06).astype(np.int8)
        mine = parallel.local_rows(len(chunks), rank, size)
        kern = _OracleKern(chunks[mine] if len(mine) else chunks[:0], overlap=40)
        init = MCMCParams.from_linear("14*1+1*2", 1e-4, 15.0, np.ones(15), 1e-2, 1e-2)
        x = (init.flat[None] + 0.2 * torch.tensor(np.random.default_rng(1).normal(size=(3, 18)))).requires_grad_(True)
        pp = PSMCParams.from_dm(init.from_flat(x).to_dm())
        local = parallel.split_minibatch(minibatch, rank, size)
        ll = parallel.sharded_loglik_sum(kern, pp, local)
        (g,) = torch.autograd.grad(ll.sum(), x)
        if rank == 0:
            torch.save({"ll": ll.detach(), "g": g}, out)
    finally:
        dist.destroy_process_group()


def _worker_particles(rank, size, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from phlash_amd import parallel

        x = torch.tensor(np.random.default_rng(3).normal(size=(5, 4)))

        def logp(xl):  # any per-particle function: rows must come back in place
            return -(xl**2).sum(1) + xl[:, 0] * xl[:, 1]

        lp, g = parallel.particle_sharded_value_and_grad(logp, x)
        if rank == 0:
            torch.save({"lp": lp, "g": g, "x": x}, out)
    finally:
        dist.destroy_process_group()


def test_particle_sharding(tmp_path):
    out = str(tmp_path / "p.pt")
    mp.start_processes(_worker_particles, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out)
    x = got["x"].clone().requires_grad_(True)
    lp = -(x**2).sum(1) + x[:, 0] * x[:, 1]
    (g,) = torch.autograd.grad(lp.sum(), x)
    np.testing.assert_allclose(got["lp"], lp.detach(), rtol=1e-14)
    np.testing.assert_allclose(got["g"], g, rtol=1e-14)


@pytest.mark.parametrize("size,minibatch", [(2, [0, 1, 2, 3, 4, 5, 6]), (2, [2, 2, 4]), (2, [5]),
                                            (8, [0, 1, 2, 3, 4, 5, 6]), (8, [6, 6, 1])])
def test_sharded_equals_unsharded(tmp_path, size, minibatch):
    """World size 2, and 8 (the node the driver benchmarks): 7 chunk rows over 8 ranks leave rank 7 without a row at
    all, and a 3-row minibatch leaves most ranks with an empty share -- they contribute zeros to the one all-reduce."""
    from oracle import cport
    from phlash_amd.params import MCMCParams, PSMCParams

    out = str(tmp_path / "r0.pt")
    mp.start_processes(_worker, args=(size, _free_port(), minibatch, out), nprocs=size, join=True, start_method="spawn")
    got = torch.load(out)
    # unsharded evaluation, same inputs
    rng = np.random.default_rng(0)
    chunks = (rng.uniform(size=(7, 260)) < 0.06).astype(np.int8)
    init = MCMCParams.from_linear("14*1+1*2", 1e-4, 15.0, np.ones(15), 1e-2, 1e-2)
    x = (init.flat[None] + 0.2 * torch.tensor(np.random.default_rng(1).normal(size=(3, 18)))).requires_grad_(True)
    P = PSMCParams.from_dm(init.from_flat(x).to_dm()).stack()
    ll, g = cport.batch(P.detach().numpy()[:, None], chunks, np.asarray(minibatch), 40)
    ll_sum = ll.sum(1)
    (gx,) = torch.autograd.grad((P * torch.tensor(g.sum(1))).sum(), x)
    np.testing.assert_allclose(got["ll"], ll_sum, rtol=1e-12)
    np.testing.assert_allclose(got["g"], gx, rtol=1e-9, atol=1e-9)


def test_sharding_helpers():
    from phlash_amd import parallel

    rows = [parallel.local_rows(11, r, 4) for r in range(4)]
    assert sorted(np.concatenate(rows).tolist()) == list(range(11))
    mb = np.array([0, 5, 5, 10, 3])
    for r in range(4):
        loc = parallel.split_minibatch(mb, r, 4)
        np.testing.assert_array_equal(rows[r][loc], mb[mb % 4 == r])
    assert parallel.world() == (0, 1)
    assert parallel.shard_mode(5, 8) == "particles" and parallel.shard_mode(8, 8) == "chunks"
    assert parallel.shard_mode(500, 8, "particles") == "particles"
