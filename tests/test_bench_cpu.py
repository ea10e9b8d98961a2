"""bench.py's own launcher for --gpus N > 1 (no torch.distributed.run around it): spawn, rendezvous, one JSON line,
exit-code propagation.  `--dry-run` keeps the GPU out of it so this runs on the CPU-only build box; the full N = 2
bench over gloo on one card is tests/test_gpu_sampler.py::test_bench_self_launch_two_ranks."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_ranks_prints_one_line():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert {k: d[k] for k in ("dry_run", "n_gpus", "ranks_seen")} == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2}
    # what the one-GPU measurements predict for this N rides along
    p = d["predicted"]
    assert p["n_gpus"] == 2 and p["strong_config4_8192_total"]["walkers_per_gpu"] == 4096
    assert p["weak_config2_1024_per_gpu"]["walkers_total"] == 2048 and 1.0 < p["weak_config2_1024_per_gpu"]["speedup_vs_one_gpu"] <= 2.0


def test_self_launch_propagates_a_failing_rank():
    r = _run(["--gpus", "2", "--dry-run"], env={"MAGPROP_BENCH_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 7
    assert "rank 1 exited with code 7" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_launcher_path_still_works():
    """The driver's form: python -m torch.distributed.run ... bench.py --gpus N (WORLD_SIZE comes from the launcher)."""
    e = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29741", BENCH, "--gpus", "2", "--dry-run"],
                       env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2


def test_single_process_needs_no_group():
    r = _run(["--dry-run"])
    assert r.returncode == 0 and json.loads(r.stdout.strip())["ranks_seen"] == 1


def test_self_launch_eight_ranks():
    """The driver's largest form (N = 8; BASELINE configs[3]) through bench.py's own launcher: eight rank processes
    rendezvous over 127.0.0.1, rank 0 prints the one line.  (On the GPU boxes of this pool at most six processes may share the
    card, so eight real ranks are only ever run by the driver's 8-GPU node; profiles/r04_bench_gloo6.json is the largest
    rehearsal with kernels.)"""
    r = _run(["--gpus", "8", "--dry-run"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    d = json.loads(lines[0])
    assert len(lines) == 1 and {k: d[k] for k in ("dry_run", "n_gpus", "ranks_seen")} == {"dry_run": True, "n_gpus": 8, "ranks_seen": 8}
    p = d["predicted"]
    assert p["strong_config4_8192_total"]["walkers_per_gpu"] == 1024 and p["strong_config5_4096_total"]["walkers_per_gpu"] == 512
    assert "WEAK" in p["claim"] and 4.0 < p["weak_config2_1024_per_gpu"]["speedup_vs_one_gpu"] <= 8.0


def test_committed_counters_and_kernel_times_belong_to_this_build():
    """profiles/pmc_figures.json (PMC counters behind bench.py's roofline object) and profiles/scaling_inputs.json (one-GPU
    kernel times behind `predicted`) record the hash of the kernel sources they were collected on; bench.py flags a
    mismatch as `stale` at run time, and the committed pair must match the committed sources."""
    sys.path.insert(0, ROOT)
    import bench
    h = bench.csrc_hash()
    assert json.load(open(os.path.join(ROOT, "profiles", "pmc_figures.json"))).get("build") == h
    assert json.load(open(os.path.join(ROOT, "profiles", "scaling_inputs.json"))).get("build") == h
