"""bench.py started plainly with --gpus N must run N ranks: the parent launches N fresh rank processes before it touches
the GPU, relays rank 0's single JSON line and propagates failures.  The reference has no counterpart (single GPU, device 0: cuInit.cu:688)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AZTOT_CTL_TOKEN", "AZTOT_BENCH_LAUNCHER_PID")}
    return env


@pytest.mark.parametrize("n", [2, 4, 8])
def test_plain_start_launches_n_ranks_and_the_parent_stays_off_the_gpu(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run"], capture_output=True, text=True, timeout=120, env=clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == n
    ranks = out["ranks"]
    assert [k["rank"] for k in ranks] == list(range(n)) and [k["local_rank"] for k in ranks] == list(range(n))
    assert all(k["world"] == n and k["token_set"] for k in ranks)
    assert len({k["pid"] for k in ranks}) == n             # N processes ...
    assert {k["ppid"] for k in ranks} == {out["launcher_pid"]}   # ... all children of the one launcher
    assert len({k["master"] for k in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")
    assert out["launcher_gpu_libraries"] == []             # the launcher never loaded libaztot / HIP / HSA / RCCL
    assert all(k["gpu_libraries"] == [] for k in ranks)    # (and a dry run touches no GPU anywhere)


def test_under_a_launcher_the_process_is_one_rank():
    """with RANK / WORLD_SIZE in the environment (torch.distributed.run's contract) bench.py does not launch anything itself"""
    port = 29871
    procs = []
    for rank in range(2):
        env = clean_env()
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    out = json.loads(outs[0][0].strip().splitlines()[-1])
    assert outs[1][0].strip() == "" and out["launcher_pid"] is None
    assert [k["pid"] for k in out["ranks"]] == [p.pid for p in procs]


def test_a_failing_rank_fails_the_launch():
    """no GPU here: every rank refuses to run ('no CPU fallback'), and the launcher must hand that on instead of printing a line"""
    from aztotmd_amd import api
    if api.device_count() > 0:
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=120, env=clean_env())
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_more_ranks_than_gpus_is_a_rehearsal_and_says_so_in_the_exit_code():
    from aztotmd_amd import api
    if api.device_count() >= 2:
        pytest.skip("this box has a GPU per rank: the RCCL path is covered by test_gpu_slab.py::test_slabs_over_rccl")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "C2", "--steps", "20", "--warmup", "5", "--no-profile"], capture_output=True, text=True, timeout=600,
                       env=clean_env())
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_share_gpus"] is True and "REHEARSAL" in out["config"]["transport"]
    assert r.returncode == 4


@pytest.mark.gpu
def test_one_gpu_line_is_what_it_was():
    """--gpus 1 runs in the process itself (no launcher): metric, unit and the blocks the driver reads are there"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--workload", "C2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["metric"] == "ns_per_day" and out["n_gpus"] == 1 and out["steps"] == 20 and out["value"] > 0
    assert out["roofline"]["frac"] > 0 and out["config"]["transport"] == "single GPU"
    # the driver's contract, field by field
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "library"):
        assert k in out, k
    assert out["unit"] == "ns/day" and out["higher_is_better"] is True and out["dtype"] == "f64" and out["data"] == "synthetic" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # PMC-derived numbers are either those of THIS build or null with the reason
    assert (r["traffic"] is None) == (r.get("counters_from") is None)
    assert out["steady_state"]["ms_per_step"] > 0 and out["call_overhead"]["step1_ms_per_step"] > 0


@pytest.mark.gpu
def test_cpu_baseline_leg_and_same_run_parity():
    """the serial CPU path timed beside the GPU in the same run (the reference's own code when its binary travelled, else our port), with the energies of both"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--workload", "C2", "--steps", "20", "--warmup", "5", "--no-steady", "--cpu-steps", "3"], capture_output=True, text=True,
                       timeout=600, env=clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    c = out["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0
    if c["kind"] == "reference":
        assert c["same_run_parity"]["engTot_rel_diff"] < 1e-11, c["same_run_parity"]


def test_the_launcher_ends_ranks_that_hang():
    """a rank stuck in a collective must not hold the node: the launcher's time limit ends the process groups it started and reports 124"""
    env = clean_env()
    env.update(AZTOT_BENCH_TIMEOUT="3", AZTOT_DRY_RUN_HANG="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert "time limit" in r.stderr
