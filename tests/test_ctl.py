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


STRANGER_WORKER = r"""
import json, socket, struct, sys, threading, time
sys.path.insert(0, %r)
from aztotmd_amd import ctl
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if rank == 1:
    # before the real rank 1 connects, strangers try: a pickle that would run code if it were unpickled, a frame with the wrong token,
    # a rank id out of range, and (after the real one) a duplicate of rank 1
    import pickle, os
    class Boom:
        def __reduce__(self):
            return (os.system, ("touch %%s" %% sys.argv[4],))
    def knock(payload):
        t0 = time.time()
        while time.time() - t0 < 60:
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                break
            except OSError:
                time.sleep(0.05)
        s.sendall(payload)
        try:
            s.settimeout(5.0)
            s.recv(64)
        except OSError:
            pass
        s.close()
    evil = pickle.dumps(Boom(), protocol=4)
    knock(struct.pack("<Q", len(evil)) + evil)                       # the old wire format
    knock(ctl.encode({"rank": 1, "token": "not-the-token"}))
    knock(ctl.encode({"rank": world + 5, "token": "tok"}))
    knock(ctl.encode({"rank": True, "token": "tok"}))
cp = ctl.Control(rank=rank, world=world, addr="127.0.0.1", port=port, token="tok")
if rank == 1:
    try:
        ctl.Control(rank=1, world=world, addr="127.0.0.1", port=port, token="tok", timeout=3.0)      # a duplicate once everybody is in: nobody listens any more
        dup = "accepted"
    except OSError:
        dup = "refused"
else:
    dup = None
import numpy as np
got = cp.all_gather({"r": rank, "a": np.arange(3, dtype=np.float64) * rank, "m": np.array([[rank, 1]], dtype=np.int32), "b": b"\x00\xff"})
cp.barrier()
print("CTL " + json.dumps({"rejected": cp.rejected, "dup": dup, "ranks": [g["r"] for g in got], "a": [g["a"].tolist() for g in got],
                           "mshape": [list(g["m"].shape) for g in got], "mdtype": [str(g["m"].dtype) for g in got], "b": [list(g["b"]) for g in got]}))
cp.close()
""" % ROOT


def test_strangers_are_turned_away_and_nothing_is_unpickled(tmp_path):
    """rank 0 must not unpickle network data, must check token / rank range / duplicates, and a stray connection must not break the run"""
    script = tmp_path / "w.py"
    script.write_text(STRANGER_WORKER)
    canary = tmp_path / "pwned"
    world, port = 3, 29813
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), str(canary)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (1, 0, 2)]
    outs = {}
    for r, p in zip((1, 0, 2), procs):
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se[-2000:]
        outs[r] = json.loads([ln for ln in so.splitlines() if ln.startswith("CTL ")][-1][4:])
    assert not canary.exists()
    assert outs[0]["rejected"] in (4, 5)          # 5: the duplicate of rank 1 knocked while rank 0 was still waiting for rank 2
    assert outs[1]["dup"] == "refused"
    for r in range(world):
        assert outs[r]["ranks"] == [0, 1, 2]
        assert outs[r]["a"] == [[0.0, 0.0, 0.0], [0.0, 1.0, 2.0], [0.0, 2.0, 4.0]]
        assert outs[r]["mshape"] == [[1, 2]] * 3 and outs[r]["mdtype"] == ["int32"] * 3
        assert outs[r]["b"] == [[0, 255]] * 3


def test_wire_format_refuses_what_is_not_plain_data():
    sys.path.insert(0, ROOT)
    import numpy as np
    from aztotmd_amd import ctl
    with pytest.raises(TypeError):
        ctl.encode({"f": print})
    with pytest.raises(TypeError):
        ctl.encode(np.array([object()], dtype=object))
    with pytest.raises(TypeError):
        ctl.encode({"__b": 0})
    # an object dtype smuggled into a header is refused on the way in
    with pytest.raises(ctl.ProtocolError):
        ctl._unpack({"__a": 0, "dtype": "|O", "shape": [1]}, [b"12345678"])
    with pytest.raises(ctl.ProtocolError):
        ctl._unpack({"__a": 0, "dtype": "<f8", "shape": [3]}, [b"12345678"])
    with pytest.raises(ctl.ProtocolError):
        ctl._unpack({"__b": 4}, [b""])
