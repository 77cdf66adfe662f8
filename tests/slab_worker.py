"""Worker for the slab-decomposition tests: run under torch.distributed.run with N ranks (gloo control plane).

Every rank owns one x-slab of the SAME system and drives the real HIP engine; halos/migrants travel either
through RCCL (one GPU per rank) or - when several ranks have to share one GPU - through the host-staged callback
transport over gloo.  Rank 0 gathers the per-atom state and compares it with a single-rank engine and the oracle.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from aztotmd_amd import api, inputs  # noqa: E402


def make_transport():
    state = {"n": 0}

    def sendrecv(send_peer, data, recv_peer, rcap):
        tag = state["n"] & 1
        state["n"] += 1
        ts = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        tr = torch.empty(rcap, dtype=torch.uint8)
        if send_peer == dist.get_rank():          # single-rank ring: message to self
            return bytes(data)
        req = dist.isend(ts, dst=send_peer, tag=tag)
        dist.recv(tr, src=recv_peer, tag=tag)
        req.wait()
        return tr.numpy().tobytes()

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t)
        return t.numpy()

    return sendrecv, allreduce


def main():
    name, nsteps, transport = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    extra = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ngpu = torch.cuda.device_count()
    dev = rank % max(ngpu, 1)
    if name == "thermo":
        case = inputs.lj_case((12, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, T=298.0, tstat="radi", vel_T=150.0,
                              radii=[(2.73, 4.731, 0.2)], nEq=10, freqEq=5)
    elif name == "nose":
        case = inputs.lj_case((12, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, T=120.0, vel_T=80.0)
        case.update(tstat_type=1, tau=0.05, nEq=10, freqEq=5)
    elif name == "fennel":
        case = inputs.lj_case((14, 6, 6), a=5.26, seed=7, charges=(0.2, -0.2), elec="fenn", vel_T=400.0)
    elif name == "mol":      # bonded molecules straddle the slab boundaries and migrate across them
        case = inputs.molecular_case((16, 6, 6), seed=9, charges=(-0.2, 0.1), elec="fenn", vel_T=900.0)
    elif name == "ewald":    # structure factors are summed over the ranks every step
        case = inputs.lj_case((14, 5, 5), a=5.26, seed=8, rc=6.5, cell_list=6.5, charges=(0.4, -0.4), elec="fenn", r_real=6.5, alpha=0.45, vel_T=300.0)
        case.update(elec_type=2, ewald_k=(8, 5, 5))
    elif name.startswith("dyn"):    # random system + thermostat / equilibration schedule / bonds and angles
        from util import add_random_dynamics, random_case
        case = add_random_dynamics(random_case(100 + int(name[3:]), x_cells=2 * world + 2, vel=0.3), int(name[3:]))
    elif name.startswith("rand"):   # seeded random system (tests/util.py random_case), box stretched along x to fit the ranks
        from util import random_case
        case = random_case(int(name[4:]), x_cells=2 * world + 2)
    elif name == "big":      # long enough along x for 8 slabs of >= 2 cell layers
        case = inputs.lj_case((40, 5, 5), a=5.26, seed=13, rc=6.5, cell_list=6.5, vel_T=600.0)
    elif name == "hot":
        case = inputs.lj_case((14, 5, 5), a=5.4, seed=3, rc=7.0, cell_list=7.0, vel_T=4000.0)
    else:
        case = inputs.lj_case((16, 6, 6), a=5.26, seed=5, vel_T=120.0)
    model = api.Model.from_case(case)
    if transport == "rccl":
        idb = [api.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(idb, src=0)
        slab = {"rank": rank, "nranks": world, "rccl_id": idb[0]}
    else:
        sr, ar = make_transport()
        slab = {"rank": rank, "nranks": world, "sendrecv": sr, "allreduce": ar}
    eng = api.Engine(model, device=dev, slab=slab, **extra)
    first = max(1, nsteps // 3)
    eng.step(first)                      # two calls: the second one runs on the sort interval the first one measured
    eng.step(nsteps - first)
    st = eng.stats()
    spec_cross = eng.species_crossings()
    s = eng.state()
    keys = ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius")
    owned = ~np.isnan(s["x"])
    counts = torch.tensor([int(owned.sum())])
    dist.all_reduce(counts)
    merged = {}
    for k in keys:
        t = torch.from_numpy(np.nan_to_num(s[k], nan=0.0))
        dist.all_reduce(t)
        merged[k] = t.numpy()
    cover = torch.from_numpy(owned.astype(np.int32))
    dist.all_reduce(cover)
    out = None
    if rank == 0:
        ref = api.Engine(api.Model.from_case(case), device=dev, **extra)
        ref.step(first)
        ref.step(nsteps - first)
        rs, rst = ref.state(), ref.stats()
        from util import rel_err
        errs = {k: rel_err(merged[k], rs[k]) for k in keys if np.abs(rs[k]).max() > 0}
        out = {"world": world, "transport": transport, "rccl_ranks": eng.comm_ranks(), "n_atoms": len(case["types"]), "sort_interval": st["sort_interval"], "owned_total": int(counts.item()),
               "every_atom_owned_once": bool((cover.numpy() == 1).all()), "max_rel_err_vs_single": max(errs.values()), "errs": errs,
               "energy_rel": {k: abs(st[k] - rst[k]) / (abs(rst[k]) + 1e-300) for k in ("engTot", "engVdW", "engKin", "engCoul", "engTemp", "engBond", "engAngle", "engCoulRec", "engCoulConst") if abs(rst[k]) > 0},
               "cross": [st["negCross"], st["posCross"], rst["negCross"], rst["posCross"]],
               "species_cross_equal": bool(np.array_equal(spec_cross, ref.species_crossings())),
               "mom_rel": rel_err(st["posMom"] + st["negMom"], rst["posMom"] + rst["negMom"]) if any(rst["posMom"] + rst["negMom"]) else 0.0}
        print("SLAB_RESULT " + json.dumps(out))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
