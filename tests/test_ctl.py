"""CPU test (world_size 2 and 3) of the torch-free control plane the rank processes of bench.py / slab_worker.py use (aztotmd_amd/ctl.py):
broadcast of the RCCL id, barrier, max / min / sum reductions and the relayed ring exchange of the host-staged test transport."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, sys
sys.path.insert(0, %r)
from aztotmd_amd import ctl
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cp = ctl.Control(rank=rank, world=world, addr="127.0.0.1", port=port)
out = {}
out["bcast"] = cp.broadcast(b"id-from-rank-0" if rank == 0 else None).decode()
cp.barrier()
out["max"] = cp.all_max(10.0 + rank)
out["min"] = cp.all_min(1 if rank != 1 else 0)
out["sum"] = cp.all_sum([1.0, float(rank)])
out["gather"] = cp.all_gather(rank * rank)
left, right = (rank + world - 1) %% world, (rank + 1) %% world
# the two messages of a slab step: leftward (arrives from the right neighbour), then rightward
out["from_right"] = cp.sendrecv(left, b"L%%d" %% rank, right).decode()
out["from_left"] = cp.sendrecv(right, b"R%%d" %% rank, left).decode()
cp.barrier()
cp.close()
print("CTL " + json.dumps(out))
""" % ROOT


@pytest.mark.parametrize("world,port", [(2, 29811), (3, 29812)])
def test_control_plane_collectives(world, port, tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("CTL ")][-1][4:]))
    for r, o in enumerate(outs):
        assert o["bcast"] == "id-from-rank-0"
        assert o["max"] == 10.0 + world - 1 and o["min"] == 0
        assert o["sum"] == [float(world), float(sum(range(world)))]
        assert o["gather"] == [k * k for k in range(world)]
        assert o["from_right"] == "L%d" % ((r + 1) % world) and o["from_left"] == "R%d" % ((r + world - 1) % world)


def test_single_rank_needs_no_sockets():
    sys.path.insert(0, ROOT)
    from aztotmd_amd import ctl
    cp = ctl.Control(rank=0, world=1)
    assert cp.broadcast("x") == "x" and cp.all_max(3) == 3 and cp.all_sum([1.0, 2.0]) == [1.0, 2.0]
    cp.barrier()
    cp.close()
