"""Child process of tests/test_multirank_gpu.py: one rank of a world_size-N run of ``phlash_amd.fit`` on
the real HIP kernels.  All ranks share GPU 0 (RCCL refuses two ranks on one device, so the process
group is gloo, which all-reduces CUDA tensors through the host); every rank is a FRESH process started
by the test with ``subprocess`` -- nothing here re-executes a process that has touched the GPU.

    python tests/mr_worker.py <rank> <world> <port> <out prefix> <shard> <scenario>

scenario:
  plain       fit() with float64 kernels, held-out contig (ELPD path), nothing forced
  real_flag   float32 kernels; every particle's emission rows are replaced by (1 - 1e-11, 1e-11) and
              only the ODD chunk rows hold a run of het sites, so in chunk mode the forward kernel of
              rank 1 (which owns the odd rows) raises the underflow flag and rank 0's does not
  fake_flag   float32 kernels; rank (world - 1) reports an underflow flag at its first hand-over that
              the device never raised (particle mode has no data-driven way to flag one rank only)
  early_stop  float64 kernels, ONE long held-out row (ranks >= 1 own none of it), elpd_cutoff = 5 and a held-out
              score that is made to fall with every evaluation: the rule stops at the evaluation of iteration 10, which
              runs beside the sampler and is read at iteration 20 -- every rank must read the same value, stop there
              and roll back to the state of iteration 10 (ranks without held-out rows used to read the
              buffer before the side stream had written it)
Writes <out prefix>.rank<r>.pt: final models, the local flags this rank handed over (before the
all-reduce), and the rescale intervals it switched to.
"""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def contigs(scenario):
    from phlash_amd.data import RawContig

    rng = np.random.default_rng(5)
    out = []
    for i in range(8):  # one chunk row per contig: chunk_size 600 + overlap 50
        if scenario == "real_flag":
            het = np.zeros(650, dtype=np.int8)
            het[60::80] = 1  # isolated hets only: at most one between two rescales (the scalar-code path of the
            # one-state-per-lane kernels rescales after 64 "debt" units: 1 per hom site, 16 per het)
            if i % 2 == 1:
                het[304:312] = 1  # a run: a whole group of hets -> mass 1e-44 between two rescales
        else:
            het = (rng.uniform(size=650) < 0.06).astype(np.int8)
            het[rng.integers(0, 650, 6)] = -1
            het[0] = max(het[0], 0)
        out.append(RawContig(het_matrix=het[None], afs=np.ones(1), window_size=100))
    return out


def main():
    rank, world, port, out, shard, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    torch.cuda.set_device(0)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import phlash_amd.mcmc as mcmc
    from phlash_amd.engine import HipEngine
    from phlash_amd.params import PSMCParams

    handed = []  # what this rank's kernels handed over, before the all-reduce mixes it with the peers'
    orig_take = HipEngine.take_flags_async
    state = {"faked": False}

    def take(self, dst):
        orig_take(self, dst)
        if scenario == "fake_flag" and rank == world - 1 and not state["faked"]:
            state["faked"] = True
            dst[0] = 1.0
        handed.append(dst.clone())

    HipEngine.take_flags_async = take
    # ... the fused step (phlash_amd/step.py) hands the flags over inside phk_reduce_chunks: row B of its buffer
    orig_reduce = HipEngine.reduce_chunks

    def reduce(self, ll, g, buf):
        orig_reduce(self, ll, g, buf)
        dst = buf[ll.shape[0], :2]
        if scenario == "fake_flag" and rank == world - 1 and not state["faked"]:
            state["faked"] = True
            dst[0] = 1.0
        handed.append(dst.clone())

    HipEngine.reduce_chunks = reduce
    nrm_calls = []
    orig_nrm = HipEngine.set_rescale_interval

    def set_nrm(self, nrm=0):
        nrm_calls.append(int(nrm))
        orig_nrm(self, nrm)

    HipEngine.set_rescale_interval = set_nrm

    if scenario == "real_flag":
        orig_map = mcmc.particles_to_psmc

        def extreme(template, x):
            pp = orig_map(template, x)
            tiny = torch.full_like(pp.emis1, 1e-11)
            return PSMCParams(b=pp.b, d=pp.d, u=pp.u, v=pp.v, emis0=1.0 - tiny, emis1=tiny, pi=pp.pi)

        mcmc.particles_to_psmc = extreme
        # ... and the fused step's parameter map
        import phlash_amd.step as fstep

        orig_pp = fstep.particle_params

        def extreme_blocks(template, x, double_precision):
            params, jac, pk = orig_pp(template, x, double_precision)
            for t in {id(params): params, id(pk): pk}.values():
                t[:, :, 4, :] = 1.0 - 1e-11
                t[:, :, 5, :] = 1e-11
            return params, jac, pk

        fstep.particle_params = extreme_blocks

    held_out = contigs("plain")[3] if scenario == "plain" else None
    extra = {}
    n_evals = []
    if scenario == "early_stop":
        from phlash_amd.data import RawContig

        rng = np.random.default_rng(11)
        long_row = (rng.uniform(size=200_000) < 0.05).astype(np.int8)  # ~4 ms per evaluation: longer than an iteration here
        held_out = RawContig(het_matrix=long_row[None], afs=np.ones(1), window_size=100)
        extra = dict(elpd_cutoff=5)
        orig_lp = mcmc._log_density_population

        def falling(x, template, c, kern, local_inds, afs, afs_transform, reduce=True):
            lp = orig_lp(x, template, c, kern, local_inds, afs, afs_transform, reduce=reduce)
            if float(c[0]) == 0.0:  # the held-out score (weights [0, 1, 1]): 1e6 lower with every evaluation, on rank 0's share
                n_evals.append(1)
                if rank == 0 or reduce:
                    lp = lp - 1e6 * len(n_evals)
            return lp

        mcmc._log_density_population = falling
    try:
        res = mcmc.fit(contigs("plain" if scenario == "early_stop" else scenario), test_data=held_out, key=3,
                       niter={"plain": 12, "early_stop": 40}.get(scenario, 4), overlap=50,
                       chunk_size=600, minibatch_size=4 if shard == "chunks" else 1, num_particles=6,
                       double_precision=scenario in ("plain", "early_stop"), shard=shard, deterministic=True, progress=False,
                       learning_rate=0.05, **extra)
        torch.cuda.synchronize()
        torch.save({
            "c": torch.stack([r.eta.c for r in res]), "t": torch.stack([r.eta.t for r in res]),
            "rho": torch.tensor([r.rho for r in res], dtype=torch.float64),
            "handed": torch.stack(handed).cpu() if handed else torch.zeros(0, 2),
            "nrm_calls": nrm_calls, "n_elpd_evals": len(n_evals),
        }, f"{out}.rank{rank}.pt")
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
