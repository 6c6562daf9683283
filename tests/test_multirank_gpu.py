"""The multi-rank path on the REAL kernels: two fresh child processes (tests/mr_worker.py) share the one
GPU of the test box through the gloo backend (RCCL refuses two ranks on one device; gloo all-reduces
CUDA tensors) and run ``phlash_amd.fit`` sharded by chunks and by particles.  Every run must reproduce
the single-process run of the same scenario, on every rank -- including when only ONE rank's forward
kernel raises the underflow flag: the redo it triggers contains an all-reduce, so the decision has to
be the same on all ranks (the flags ride in the step's all-reduce; round-1 decided rank-locally and
would pair the collectives of different iterations).  Replaces the reference's thread fan-out
(src/phlash/gpu.py:386-438)."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "mr_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, out, shard, scenario, timeout=600):
    port = str(_free_port())
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), port, out, shard, scenario], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # the exact PIDs this test started
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} of {world} ({shard}, {scenario}) failed:\n{logs[r][-3000:]}"
    return [torch.load(f"{out}.rank{r}.pt") for r in range(world)]


@pytest.mark.parametrize("shard,scenario", [("chunks", "plain"), ("particles", "plain"), ("chunks", "real_flag"),
                                            ("chunks", "fake_flag"), ("particles", "fake_flag")])
def test_two_ranks_reproduce_one(tmp_path, shard, scenario):
    one = _launch(1, str(tmp_path / "w1"), shard, scenario)[0]
    two = _launch(2, str(tmp_path / "w2"), shard, scenario)
    # float64 kernels (plain): the sharding re-associates sums and a rank with fewer sequences may run another kernel
    # variant (lanes per sequence), i.e. 1e-14 differences in the gradients, which 12 AMSGrad-normalised SVGD
    # iterations amplify to at most 2.5e-9 here (measured); float32 kernels after the switch to per-site
    # rescaling are per-sequence deterministic as well (no dense hom-run steps)
    tol = dict(rtol=2e-8, atol=1e-11)
    for r in range(2):
        for key in ("c", "t", "rho"):
            np.testing.assert_allclose(two[r][key], one[key], err_msg=f"rank {r} {key}", **tol)
        assert torch.isfinite(two[r]["c"]).all()
    if scenario == "real_flag":
        # the device flag was raised on rank 1 only (it owns the odd rows, which hold the het runs) ...
        first = [two[r]["handed"][0] for r in range(2)]
        assert first[0][0] == 0 and first[1][0] == 1, first
        # ... and BOTH ranks switched to per-site rescaling, exactly once, like the single process
        assert two[0]["nrm_calls"] == two[1]["nrm_calls"] == one["nrm_calls"] == [1]
    if scenario == "fake_flag":
        assert two[0]["handed"][0][0] == 0 and two[1]["handed"][0][0] == 1
        assert two[0]["nrm_calls"] == two[1]["nrm_calls"] == [1]
    if scenario == "plain":
        assert two[0]["nrm_calls"] == two[1]["nrm_calls"] == []


@pytest.mark.parametrize("shard", ["chunks", "particles"])
def test_speculative_early_stop_is_taken_by_every_rank(tmp_path, shard):
    """One long held-out row (rank 1 owns none of it in chunk mode), a score that falls with every evaluation and
    elpd_cutoff = 5: the evaluation of iteration 10 runs beside the sampler, is read at iteration 20 and stops the run
    there, rolling back to iteration 10.  Both ranks must return the particles of the single-process run and must have
    evaluated the held-out score the same number of times (in line at 0, beside the sampler at 10 -- not at 20, 30)."""
    one = _launch(1, str(tmp_path / "w1"), shard, "early_stop")[0]
    two = _launch(2, str(tmp_path / "w2"), shard, "early_stop")
    assert one["n_elpd_evals"] == 2, one["n_elpd_evals"]
    for r in range(2):
        assert two[r]["n_elpd_evals"] == 2, (r, two[r]["n_elpd_evals"])
        for key in ("c", "t", "rho"):
            np.testing.assert_allclose(two[r][key], one[key], rtol=2e-8, atol=1e-11, err_msg=f"rank {r} {key}")
    for key in ("c", "t", "rho"):
        assert torch.equal(two[0][key], two[1][key]), key  # the ranks agree to the bit


def test_eight_ranks_reproduce_one(tmp_path):
    """World size 8 on the real kernels (gloo, all ranks on the one GPU): 8 chunk rows -> one row per rank, a
    minibatch of 4 rows -> ranks whose share of a minibatch is empty, one held-out row -> seven ranks without a
    held-out row (they contribute zeros to the ELPD all-reduce and still take the same branches)."""
    one = _launch(1, str(tmp_path / "w1"), "chunks", "plain")[0]
    eight = _launch(8, str(tmp_path / "w8"), "chunks", "plain", timeout=900)
    for r in range(8):
        for key in ("c", "t", "rho"):
            np.testing.assert_allclose(eight[r][key], one[key], rtol=2e-8, atol=1e-11, err_msg=f"rank {r} {key}")
        assert eight[r]["nrm_calls"] == []


def _plain_bench(args, timeout=900):
    """``python bench.py --gpus N ...`` exactly as the driver types it: no torchrun, no rank environment."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


def test_bench_plain_form_self_launches_two_ranks():
    """The form that exited with an error in round 2: plain ``python bench.py --gpus 2`` starts its own rank
    processes (before touching the GPU), relays rank 0's line and exits 0."""
    line = _plain_bench(["--gpus", "2", "--steps", "3", "--warmup", "2", "--backend", "gloo", "--particles", "12",
                         "--chunks", "40", "--chunk-size", "4000", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["backend"] == "gloo"
    assert line["ranks_identical_after_timed_loop"] is True
    pr = line["per_rank"]
    assert len(pr["plan"]) == 2 and len(set(pr["plan"])) == 1, pr  # rank 0's plan was installed on both
    assert len(pr["ms_per_step"]["all"]) == 2 and pr["ms_per_step"]["max"] == pytest.approx(line["ms_per_step"])
    assert 0 < line["config"]["het_rate"] < 0.2 and 0 < line["config"]["all_hom_word16_frac"] < 1


def test_bench_default_command_on_one_gpu_carries_the_other_shapes():
    """``python bench.py`` at its default sizes: after the cfg2 loop the same process times the reference's production
    shape (``secondary``) and BASELINE.json's cfg1, K = 64 and K = 32 configs (``other_configs``), each with the full inner
    step and its own checks; ``value`` stays the cfg2 figure."""
    line = _plain_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-reference-kernel"], timeout=1500)
    assert line["n_gpus"] == 1 and line["config"]["name"] == "cfg2" and "strong_cfg3" not in line
    assert line["value"] == pytest.approx(100 * 500 * 60000 / (line["ms_per_step"] * 1e-3), rel=1e-6)
    sec = line["secondary"]
    assert sec["checks_passed"] is True and sec["value"] == pytest.approx(500 * 5 * 100000 / (sec["ms_per_step"] * 1e-3), rel=1e-6)
    other = line["other_configs"]
    # (round 6: the headline shape at 5 % / 10 % het rows and cfg3 -- 5,000 rows, AFS n = 20 -- at N = 1 ride along too)
    assert set(other) == {"cfg1", "cfg4", "cfg5", "cfg2_het5", "cfg2_het10", "cfg3"}
    assert "Bernoulli(0.1)" in other["cfg2_het10"]["workload"] and "Bernoulli(0.05)" in other["cfg2_het5"]["workload"]
    for name, work in (("cfg1", 1 * 1 * 100000), ("cfg4", 100 * 500 * 60000), ("cfg5", 500 * 500 * 60000),
                       ("cfg2_het5", 100 * 500 * 60000), ("cfg2_het10", 100 * 500 * 60000), ("cfg3", 100 * 5000 * 60000)):
        ex = other[name]
        assert ex["checks_passed"] is True and ex["scaling"] == ("strong" if name == "cfg3" else "weak") and ex["n_gpus"] == 1
        assert ex["value"] == pytest.approx(work / (ex["ms_per_step"] * 1e-3), rel=1e-6)


def test_bench_default_command_at_two_ranks_carries_the_strong_scaling_extra():
    """``python bench.py --gpus 2`` with the default (cfg2, weak scaling) sizes: after the headline loop the same launch
    times the cfg3 strong-scaling problem (a fixed total of chunk rows sharded over the ranks, AFS n = 20) and prints it
    as ``strong_cfg3``, leaving ``value`` the weak cfg2 figure.  Two ranks share the one GPU here (gloo), so the extra's
    5,000 rows are cut to 64 (--extras-chunks); the headline runs at full size."""
    line = _plain_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline",
                         "--extras-chunks", "64"], timeout=1500)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["name"] == "cfg2"
    assert line["value"] == pytest.approx(2 * 100 * 500 * 60000 / (line["ms_per_step"] * 1e-3), rel=1e-6)
    ex = line["strong_cfg3"]
    assert ex["scaling"] == "strong" and ex["n_gpus"] == 2 and ex["checks_passed"] is True
    assert ex["value"] == pytest.approx(100 * 64 * 60000 / (ex["ms_per_step"] * 1e-3), rel=1e-6)
    assert "speedup_vs_expectation_n1" in ex and line["rccl_ranks"] == 0  # (gloo: no RCCL communicator here)


@pytest.mark.parametrize("fault,code,msg", [("ranks", 4, "communicator joins 1 rank(s) but WORLD_SIZE=2"),
                                            ("diverge", 3, "ranks disagree on the particles")])
def test_bench_multi_rank_run_that_is_not_one_job_fails_loudly(fault, code, msg):
    """A multi-rank run whose communicator joins another number of ranks than it was launched with, or whose replicated
    particles differ between the ranks after the timed loop, prints NO result line and exits non-zero (round 5 reported
    both as fields of a line that exited 0).  Provoked through the test hook PHK_BENCH_TEST_FAULT: a faked count, a
    perturbed replica."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PHK_BENCH_TEST_FAULT"] = fault
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--particles", "12", "--chunks", "40", "--chunk-size", "4000", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode != 0 and not any(ln.startswith("{") for ln in p.stdout.splitlines()), (p.returncode, p.stdout[-500:])
    assert "FAILED" in p.stderr and msg in p.stderr and f"rank exit codes [{code}, {code}]" in p.stderr, p.stderr[-2000:]


def test_bench_cfg3_eight_ranks_gloo_tiny():
    """cfg3's code path (fixed total of chunk rows sharded over the ranks, AFS term in the step) at world size 8,
    tiny shapes: 40 rows -> 5 per rank (625 at full size)."""
    line = _plain_bench(["--gpus", "8", "--config", "cfg3", "--steps", "2", "--warmup", "2", "--backend", "gloo",
                         "--particles", "6", "--chunks", "40", "--chunk-size", "2000", "--overlap", "100",
                         "--no-cpu-baseline"])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["chunks_per_gpu"] == 5
    assert line["config"]["chunks_total"] == 40 and line["ranks_identical_after_timed_loop"] is True
    assert len(line["per_rank"]["plan"]) == 8


def test_bench_torchrun_form_world1_rccl():
    """The torchrun form at world size 1 initialises RCCL (backend "nccl") on the one GPU of the box."""
    port = str(_free_port())
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--steps", "2", "--warmup", "1", "--particles", "12", "--chunks", "40", "--chunk-size", "4000",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["rccl_ranks"] == 1 and line["n_gpus"] == 1


def test_bench_two_ranks_gloo(tmp_path):
    """bench.py's own step (param map -> kernels -> all-reduce -> chain rule -> SVGD) with two ranks on
    the real kernels, tiny shapes; bench.py reports whether the replicated particles are identical on all
    ranks after the timed loop, and they must be."""
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
             "--particles", "12", "--chunks", "40", "--chunk-size", "4000", "--no-cpu-baseline"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, outs[r][1][-3000:]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["ranks_identical_after_timed_loop"] is True
    assert not any(ln.startswith("{") for ln in outs[1][0].splitlines())  # only rank 0 prints the JSON line
