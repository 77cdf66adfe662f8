"""Worker for the slab-decomposition tests: run under torch.distributed.run with N ranks.

Every rank owns one x-slab of the SAME system and drives the real HIP engine; halos/migrants travel either
through RCCL (one GPU per rank) or - when several ranks have to share one GPU - through the host-staged callback
transport, relayed by the control plane.  Rank 0 gathers the per-atom state and compares it with a single-rank engine.
torch is NOT imported here (its wheel's HIP / HSA copies would sit in front of the system runtime RCCL needs): the control
plane is aztotmd_amd.ctl (plain TCP; rank, world size and the master address come from the launcher's environment).
transport 'callback_corrupt': rank 0 lies by one atom in the count message of the rebuild steps - every rank of that
boundary must then report AZTOT_ERR_COMM before any plain step's exchange is posted.
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from aztotmd_amd import api, ctl, inputs  # noqa: E402


def make_transport(cp, corrupt=False):
    def sendrecv(send_peer, data, recv_peer, rcap):
        data = bytes(data)
        if corrupt and cp.rank == 0 and len(data) == 8:
            # the count message of a rebuild step ({atoms I will send you per plain step, ghosts I hold on your side}): claim one atom more
            n, g = struct.unpack("<ii", data)
            data = struct.pack("<ii", n + 1, g)
        if send_peer == cp.rank and recv_peer == cp.rank:          # single-rank ring: message to self
            return data
        got = cp.sendrecv(send_peer, data, recv_peer)
        if corrupt and cp.rank == 0 and len(got) == 8:
            # ... and see the neighbour's numbers off by one too: a real disagreement is visible from both sides of the boundary (both ranks evaluate
            # the same two equalities), so both must fail
            n, g = struct.unpack("<ii", got)
            got = struct.pack("<ii", n, g + 1)
        return got

    def allreduce(a):
        return np.asarray(cp.all_sum(np.asarray(a, dtype=np.float64).tolist()))

    return sendrecv, allreduce


def main():
    name, nsteps, transport = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    extra = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    cp = ctl.Control()
    rank, world = cp.rank, cp.world
    ngpu = api.device_count()
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
    elif name == "heat":     # cold start (long sort interval), then the velocities are tripled between two calls
        case = inputs.lj_case((16, 6, 6), a=5.26, seed=6, vel_T=60.0)
    elif name == "hot":
        case = inputs.lj_case((14, 5, 5), a=5.4, seed=3, rc=7.0, cell_list=7.0, vel_T=4000.0)
    else:
        case = inputs.lj_case((16, 6, 6), a=5.26, seed=5, vel_T=120.0)
    model = api.Model.from_case(case)
    if transport == "rccl":
        idb = cp.broadcast(api.rccl_unique_id() if rank == 0 else None)
        slab = {"rank": rank, "nranks": world, "rccl_id": idb}
    else:
        sr, ar = make_transport(cp, corrupt=(transport == "callback_corrupt"))
        slab = {"rank": rank, "nranks": world, "sendrecv": sr, "allreduce": ar}
    eng = api.Engine(model, device=dev, slab=slab, **extra)
    first = max(1, nsteps // 3)
    if transport == "callback_corrupt":
        # every rank must come back with AZTOT_ERR_COMM (-5) instead of hanging or shifting coordinates onto the wrong atoms
        code, msg = 0, ""
        try:
            eng.step(first)
            eng.step(nsteps - first)
        except api.AztotError as ex:
            code, msg = ex.code, str(ex)
        # ... and the handle is dead for stepping from then on (the first error is repeated), alive for a post-mortem read of the clock
        again, again_msg, clock_ok = 0, "", False
        try:
            eng.step(1)
        except api.AztotError as ex:
            again, again_msg = ex.code, str(ex)
        try:
            clock_ok = eng.clock()["step"] >= 0
        except api.AztotError:
            clock_ok = False
        codes = cp.all_gather(code)
        agains = cp.all_gather([again, "earlier call" in again_msg, bool(clock_ok)])
        if rank == 0:
            print("SLAB_RESULT " + json.dumps({"world": world, "transport": transport, "codes": codes, "message": msg, "after_failure": agains}))
        cp.barrier()
        cp.close()
        return
    def heat(e):
        # a heating protocol between two calls: velocities x 3 through aztot_set_state.  The sort interval measured on the slow atoms must not be carried
        # over (a slab rank repairs a skin violation by running a window of steps again - exact, but slow): set_state forgets it, every step rebuilds until the next look
        sv = e.state(("vx", "vy", "vz"))
        e.set_state(**{k: np.nan_to_num(sv[k], nan=0.0) * 3.0 for k in ("vx", "vy", "vz")})

    eng.step(first)                      # two calls: the second one runs on the sort interval the first one measured
    if name == "heat":
        assert eng.stats()["sort_interval"] > 1
        heat(eng)
    eng.step(nsteps - first)
    st = eng.stats()
    spec_cross = eng.species_crossings()
    s = eng.state()
    keys = ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius")
    owned = ~np.isnan(s["x"])
    parts = cp.all_gather((owned, {k: np.nan_to_num(s[k], nan=0.0) for k in keys}))
    owned_total = int(sum(int(o.sum()) for o, _ in parts))
    merged = {k: sum(d[k] for _, d in parts) for k in keys}
    cover = sum(o.astype(np.int32) for o, _ in parts)
    out = None
    if rank == 0:
        ref = api.Engine(api.Model.from_case(case), device=dev, **extra)
        ref.step(first)
        if name == "heat":
            heat(ref)
        ref.step(nsteps - first)
        rs, rst = ref.state(), ref.stats()
        from util import rel_err
        errs = {k: rel_err(merged[k], rs[k]) for k in keys if np.abs(rs[k]).max() > 0}
        out = {"world": world, "transport": transport, "rccl_ranks": eng.comm_ranks(), "n_atoms": len(case["types"]), "sort_interval": st["sort_interval"], "sort_violations": st["sort_violations"], "pair_lists": st["pair_lists"], "cells_without_list": st["cells_without_list"], "owned_total": owned_total,
               "every_atom_owned_once": bool((cover == 1).all()), "max_rel_err_vs_single": max(errs.values()), "errs": errs,
               "energy_rel": {k: abs(st[k] - rst[k]) / (abs(rst[k]) + 1e-300) for k in ("engTot", "engVdW", "engKin", "engCoul", "engTemp", "engBond", "engAngle", "engCoulRec", "engCoulConst") if abs(rst[k]) > 0},
               "cross": [st["negCross"], st["posCross"], rst["negCross"], rst["posCross"]],
               "species_cross_equal": bool(np.array_equal(spec_cross, ref.species_crossings())),
               "mom_rel": rel_err(st["posMom"] + st["negMom"], rst["posMom"] + rst["negMom"]) if any(rst["posMom"] + rst["negMom"]) else 0.0}
        print("SLAB_RESULT " + json.dumps(out))
    eng.close()
    cp.barrier()
    cp.close()


if __name__ == "__main__":
    main()
