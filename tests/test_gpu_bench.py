"""-m gpu: bench.py's contract and its N > 1 flow on a one-GPU box. The driver launches N ranks with
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`; here two ranks share the GPU
(EPH_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device) for the three workloads: per-rank replicas, the sharded
massless sweep with its result all-gather, and one system partitioned by target body (host-staged exchange)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
NEED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"]


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(args, ranks):
    env = dict(os.environ, EPH_BENCH_BACKEND="gloo")
    if ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), str(ROOT / "bench.py"), "--gpus", str(ranks)] + args
    else:
        cmd = [sys.executable, str(ROOT / "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert not [k for k in NEED if k not in d], d.keys()
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    return d


def test_default_line_single_gpu():
    d = _run(["--steps", "10", "--warmup", "3", "--cpu-steps", "3", "--horizon", "100", "--prewarm", "0.2"], 1)
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["metric"] == "body-steps/s" and d["dtype"] == "f64"
    assert d["value"] > 1e6                                        # the north star's floor
    assert d["parity"]["max_abs_dpos"] == 0.0 and set(d["parity"]["horizon_max_abs_dpos"].values()) == {0.0}
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"]
    assert "plummer_4096_f64_qt12" in d["config"]["workload"]
    # the companions of the default line: configs[1] in-process, the configs[3] sweep from a child process
    assert d["pair_variant"] == 0 and set(d["other_variants"]) == {"1", "2", "3", "4", "5", "6"}
    assert d["roofline"]["binding"] == "fp64_valu" and 0.05 < d["roofline"]["fp64"]["frac"] < 0.5
    assert d["long_region"]["steps"] == 100 and d["long_region"]["ms_per_step"] <= d["ms_per_step"] * 1.02
    oc = d["other_configs"]
    assert oc["configs1_full_solar_system"]["bodies"] == 32 and 0.1 < oc["configs1_full_solar_system"]["us_per_step"] < 5.0
    assert oc["configs3_craft_sweep"]["value"] > 1e7 and "craft" in oc["configs3_craft_sweep"]["workload"]
    assert oc["configs3_craft_sweep"]["fp64"]["frac"] > 0.05 and oc["configs3_craft_sweep"]["cpu_baseline"]["cores"] == 1
    assert oc["configs3_craft_sweep"]["wall_over_kernel"] < 1.2
    # round 5: the shipping system's line has its own CPU baseline (same steps, one thread), parity and a latency roofline
    c1 = oc["configs1_full_solar_system"]
    assert c1["cpu_baseline"]["cores"] == 1 and c1["cpu_baseline"]["kind"] == "port" and c1["cpu_baseline"]["seconds"] > c1["seconds"]
    assert c1["parity"] == {"max_abs_dpos": 0.0, "max_abs_dvel": 0.0, "steps": 1020000, "vs": "oracle (port)"}
    assert 0.2 < c1["latency_roofline"]["frac"] < 1.0
    # counter figures are printed only while their kernel's sources are the ones they were counted on
    r = d["roofline"]
    assert (r["traffic"] is None) == (r["traffic_stale"] is not None)
    if r["traffic"] is not None:
        assert r["traffic_profile_commit"] and r["fp64"]["valu_issue"]["frac"] > 0.3


def test_two_ranks_replicas():
    d = _run(["--steps", "10", "--warmup", "3", "--prewarm", "0"], 2)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 1e6


def test_gpus_flag_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` the way the driver calls `--gpus 1`: no RANK / WORLD_SIZE in the environment. bench.py
    becomes the launcher (torch.distributed.run on 127.0.0.1) and still prints ONE JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["EPH_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--prewarm", "0"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["blocks"] >= 3
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    s4 = d["sharded_4096"]                              # the strong-scaling figure of the metric's own system
    assert s4["ranks"] == 2 and s4["ms_per_step"] > 0 and s4["bit_identical_to_single_device"] is True
    # the preflight ran every transport this job can use against the single-device bits before the sharded timing
    assert d["transports"]["peer"] == "ok" and d["transports"]["host"] == "ok" and d["transports"]["peer_memory"][0] in ("fine", "coarse")


def test_craft_workload_spawns_its_own_ranks_too():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["EPH_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "craft", "--gpus", "2", "--craft", "32768", "--steps", "2",
                        "--warmup", "1", "--prewarm", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900,
                       cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["metric"] == "craft-steps/s" and d["roofline"]["binding"] == "fp64_valu" and d["roofline"]["fp64"]["frac"] > 0


def test_a_strong_scaling_leg_that_never_returns_does_not_cost_the_line():
    """The sharded_4096 leg runs after the main line is complete, under a watchdog: with the limit at zero it expires before
    the leg can finish; rank 0 still prints the one line (the leg marked as timed out) and the job ends with status 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(EPH_BENCH_BACKEND="gloo", EPH_BENCH_SHARDED_TIMEOUT="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--prewarm", "0"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 1e6 and "no result within" in d["sharded_4096"]["error"]


@pytest.mark.parametrize("population", ["transfer", "mixed"])
def test_two_ranks_craft_sweep_with_result_gather(population):
    d = _run(["--workload", "craft", "--craft", "4001", "--craft-days", "0.05", "--steps", "1", "--prewarm", "0",
              "--population", population], 2)
    assert d["n_gpus"] == 2 and d["metric"] == "craft-steps/s" and "all-gather" in d["config"]["exchange"]
    assert d["divergence"]["attempts_max_over_mean_per_wave"] >= 1.0
    assert (population == "mixed") == ("population" in d["config"]["workload"])
    # rank 0 built the table once and broadcast its image; rank 1 imported it; bit-identical to a local rebuild
    bc = d["ephemeris_broadcast_s"]
    assert bc["identical_to_the_local_rebuild"] is True and bc["rank0_parts"]["bytes"] > 100000
    assert bc["max_over_ranks"] > 0.0 and d["ephemeris_rebuild_s"]["max_over_ranks"] > 0.0


@pytest.mark.parametrize("transport", ["host", "peer"])
def test_two_ranks_one_sharded_system(transport):
    d = _run(["--workload", "nbody-sharded", "--bodies", "1024", "--transport", transport, "--steps", "5", "--warmup", "2",
              "--prewarm", "0"], 2)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["bodies_per_gpu"] == 512


def test_two_ranks_f32_pairs_on_a_partition():
    """BASELINE configs[4] as stated: `--workload nbody-sharded --path f32-pairs` (binary32 pair arithmetic on a target partition)"""
    d = _run(["--workload", "nbody-sharded", "--path", "f32-pairs", "--bodies", "4096", "--transport", "peer", "--steps", "5",
              "--warmup", "2", "--prewarm", "0.1"], 2)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["path"] == "f32-pairs"
    assert "all-gather of 65536 B" in d["config"]["parallelism"]            # 16 B per body
    assert d["roofline"]["binding"] == "fp32_valu" and d["roofline"]["fp32"]["frac"] > 0


def test_configs4_leg_of_the_default_multi_rank_line():
    """With every rank on its own device the default `--gpus N` line also measures configs[4] as stated (65 536 bodies, f32 pairs,
    partitioned) against rank 0's single-device f32 run; on this one-GPU box the leg is asked for explicitly at a small size."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(EPH_BENCH_BACKEND="gloo", EPH_BENCH_CONFIGS4_BODIES="4096")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--prewarm", "0"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c4 = d["configs4_f32_sharded"]
    assert c4["ranks"] == 2 and c4["bodies"] == 4096 and c4["exchange_bytes_per_step"] == 16 * 4096 and c4["transport"] == "peer"
    assert c4["bit_identical_to_single_device_f32"] is True and c4["ms_per_step"] > 0 and c4["single_device_ms_per_step"] > 0
    assert d["sharded_4096"]["bit_identical_to_single_device"] is True
