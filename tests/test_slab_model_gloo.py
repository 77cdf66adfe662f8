"""CPU test (gloo, world_size 2, 3 and 4): the slab-decomposition protocol (ownership by cell layer, emigrants kept as
ghosts, one message per neighbour and step) reproduces the single-domain oracle.  The device implementation
(aztotmd_amd/csrc/slab.hip.h) follows the same rules and is tested against the single-rank engine with -m gpu."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,port", [(2, 29631), (3, 29632), (4, 29635)])
def test_slab_protocol_model_matches_single_domain(world, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "slab_model.py"), "12"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("SLAB_MODEL ")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[-1][len("SLAB_MODEL "):])
    assert out["world"] == world and out["boundary_counts_checked"] == 12      # every step: senders' and receivers' boundary counts agree on both sides
    assert out["err"][""] < 1e-11 and out["err"]["v"] < 1e-10 and out["err"]["f"] < 1e-11 * max(out["fmax"], 1.0), out


@pytest.mark.parametrize("world,K,port", [(2, 12, 29633), (3, 6, 29634), (4, 12, 29636)])
def test_slab_ranks_repair_a_skin_violation_together(world, K, port):
    """The lazy re-sort on slab ranks (cells rebuilt every K-th step, coordinates only in between) with looks every 8 steps: K = 12 lets the fastest atoms of
    this 3 000 K gas leave the slack (0.35 A) before the scheduled rebuild - the look that finds it takes every rank back to its snapshot and runs the window
    again with the cells rebuilt every step (the protocol of Engine::step / replay_from_snapshot, modelled on the CPU); K = 6 stays inside the slack and must
    never repair.  Either way the run equals the single-domain oracle."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "slab_model.py"), "24", "lazy", str(K), "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("SLAB_MODEL ")]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[-1][len("SLAB_MODEL "):])
    assert out["world"] == world
    assert (out["repairs"] > 0) == (K == 12), out
    assert out["err"][""] < 1e-11 and out["err"]["v"] < 1e-10 and out["err"]["f"] < 1e-11 * max(out["fmax"], 1.0), out
