"""CPU model of the slab decomposition protocol of aztotmd_amd/csrc/slab.hip.h, run under torch.distributed (gloo).

Same rules as the device code:
  * global cell grid, rank r owns a contiguous run of x-layers [lo, hi); it also holds hw ghost layers on each side;
  * after the drift every rank bins its atoms by layer: atoms that left the owned layers are sent (full state) to
    the neighbour owning that layer AND are kept as ghosts by the sender; atoms in the hw boundary layers are sent
    as halo (position/type only); one message per neighbour and step, leftward message first;
  * forces are evaluated for owned atoms from owned + ghost atoms with the ordinary minimum image.
Forces come from the CPU oracle (tests may use it); the integrator is velocity Verlet as integrators.cpp:292,486.
Usage: python -m torch.distributed.run --nproc-per-node 2 tests/slab_model.py <nsteps> [lazy <K> <window>]

'lazy K window': the lazy re-sort of Engine::step on slab ranks.  The cells are rebuilt (migration + halo records) every K-th step only; in between the ranks
keep their atoms and ghosts and exchange the boundary atoms' coordinates, and every rank checks that none of its atoms has moved farther from where it was at
the last rebuild than the slack (hw x layer width - rc) / 2 - beyond that a pair inside the cut-off could be missing from a rank's ghost layers.  A slab rank
cannot widen its stencil, so every `window` steps the ranks look: the violation flags are all-reduced, a clean look leaves a snapshot of the dynamic state
(Engine::take_snapshot), and a look that finds a violation takes EVERY rank back to its snapshot and runs the window again with the cells rebuilt on every step
(Engine::replay_from_snapshot).  The result must equal the single-domain oracle whatever K is.
"""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aztotmd_amd import inputs      # noqa: E402
from oracle import oracle           # noqa: E402

M_SCALE = 0.00010364272224473431


def forces_for(case, idx_owned, idx_ghost, pos, types):
    """forces on the owned atoms from owned + ghost atoms (oracle all-pairs with minimum image in the global box)"""
    idx = np.concatenate([idx_owned, idx_ghost]).astype(int)
    sub = dict(case)
    sub.update(types=types[idx], x=pos[idx, 0], y=pos[idx, 1], z=pos[idx, 2], vx=np.zeros(len(idx)), vy=np.zeros(len(idx)),
               vz=np.zeros(len(idx)), use_clist=0)
    o = oracle.Oracle(sub)
    o.forces(0)
    s = o.state()
    n = len(idx_owned)
    return np.stack([s["fx"][:n], s["fy"][:n], s["fz"][:n]], axis=1)


def exchange(rank, world, to_left, to_right):
    left, right = (rank - 1) % world, (rank + 1) % world
    if world == 1:
        return to_right, to_left
    box = [None]
    # leftward messages first, then rightward ones (same order on every rank)
    if rank % 2 == 0:
        dist.send_object_list([to_left], dst=left); dist.recv_object_list(box, src=right); from_right = box[0]
        dist.send_object_list([to_right], dst=right); dist.recv_object_list(box, src=left); from_left = box[0]
    else:
        dist.recv_object_list(box, src=right); from_right = box[0]; dist.send_object_list([to_left], dst=left)
        dist.recv_object_list(box, src=left); from_left = box[0]; dist.send_object_list([to_right], dst=right)
    return from_left, from_right


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    lazy = len(sys.argv) > 2 and sys.argv[2] == "lazy"
    K, window = (int(sys.argv[3]), int(sys.argv[4])) if lazy else (1, 1 << 30)
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case = inputs.lj_case((12, 5, 5), a=5.4, seed=21, rc=6.5, cell_list=6.5, vel_T=3000.0)
    N = len(case["types"])
    L = np.array(case["box"])
    rc = 6.5
    ncx = int(np.floor(L[0] / rc)); csx = L[0] / ncx; hw = 1
    lo, hi = ncx * rank // world, ncx * (rank + 1) // world
    assert hi - lo >= 2 * hw and (hi - lo) + 2 * hw <= ncx
    pos = np.stack([case["x"], case["y"], case["z"]], 1)
    vel = np.stack([case["vx"], case["vy"], case["vz"]], 1)
    types = np.asarray(case["types"])
    mass = np.array([m for m, _ in case["species"]])[types] * M_SCALE
    layer = lambda x: np.floor(x / csx).astype(int) % ncx
    owned = np.where((layer(pos[:, 0]) >= lo) & (layer(pos[:, 0]) < hi))[0]
    # state this rank knows: positions of everything it holds (indexed by global id for simplicity of the model)
    P, V = pos.copy(), vel.copy()

    def halo_ids(own):
        lay = layer(P[own, 0])
        return own[lay < lo + hw], own[lay >= hi - hw]

    def ghosts_from(from_left, from_right):
        ids = []
        for msg in (from_left, from_right):
            for rec in msg["halo"]:
                P[rec[0]] = rec[1:4]; ids.append(rec[0])
        return np.array(ids, dtype=int)

    # initial forces: halo only
    hl, hr = halo_ids(owned)
    fl, fr = exchange(rank, world, {"mig": [], "halo": [(i, *P[i]) for i in hl]}, {"mig": [], "halo": [(i, *P[i]) for i in hr]})
    ghosts = ghosts_from(fl, fr)
    F = np.zeros_like(P)
    F[owned] = forces_for(case, owned, ghosts, P, types)
    dt = case["dt"]
    slack = 0.5 * (hw * csx - rc)
    st = {"owned": owned, "ghosts": ghosts, "checked": 0, "ref": P.copy(), "violated": False, "send": None}

    def rebuild_step():
        nonlocal P, V, F
        owned = st["owned"]
        V[owned] += (0.5 * dt / mass[owned])[:, None] * F[owned]
        P[owned] += V[owned] * dt
        P[owned] -= np.floor(P[owned] / L) * L
        lay = layer(P[owned, 0])
        rel = (lay - lo) % ncx                       # position inside the rank's window, periodic
        mig_left = owned[rel == ncx - 1]             # moved into the layer just left of the slab
        mig_right = owned[rel == (hi - lo)]
        stay = owned[(rel >= 0) & (rel < hi - lo)]
        assert len(mig_left) + len(mig_right) + len(stay) == len(owned), "an atom moved more than one layer"
        hl, hr = halo_ids(stay)
        fl, fr = exchange(rank, world,
                          {"mig": [(i, *P[i], *V[i]) for i in mig_left], "halo": [(i, *P[i]) for i in hl]},
                          {"mig": [(i, *P[i], *V[i]) for i in mig_right], "halo": [(i, *P[i]) for i in hr]})
        arrivals = []
        for msg in (fl, fr):
            for rec in msg["mig"]:
                P[rec[0]] = rec[1:4]; V[rec[0]] = rec[4:7]; arrivals.append(rec[0])
        ghosts = np.concatenate([ghosts_from(fl, fr), mig_left, mig_right]).astype(int)    # emigrants are kept as ghosts
        owned = np.concatenate([stay, np.array(arrivals, dtype=int)]).astype(int)
        assert len(set(owned) & set(ghosts)) == 0 and len(set(ghosts)) == len(ghosts)
        # The invariant the device engine checks before the plain steps of a lazy run post their coordinate exchange (Engine::adopt_halo_info; the counts travel
        # in fixed-size messages, Exchanger::exchange_counts): what a rank will send to a neighbour - its outermost hw owned layers as they are AFTER the
        # migration - is exactly what that neighbour holds as ghosts on that side (halo it received + its own emigrants, kept as ghosts).  Both ranks of
        # a boundary see both numbers.
        lay_now = layer(P[owned, 0])
        to_l, to_r = owned[(lay_now - lo) % ncx < hw], owned[(lay_now - lo) % ncx >= (hi - lo) - hw]
        send_left, send_right = len(to_l), len(to_r)
        ghosts_left, ghosts_right = len(fl["halo"]) + len(mig_left), len(fr["halo"]) + len(mig_right)
        cl, cr = exchange(rank, world, (send_left, ghosts_left), (send_right, ghosts_right))
        # cl = the left neighbour's (what it sends rightward, its right ghosts) ; cr = the right neighbour's (what it sends leftward, its left ghosts)
        assert cr[0] == ghosts_right and cr[1] == send_right and cl[0] == ghosts_left and cl[1] == send_left, (rank, cl, cr, send_left, ghosts_left, send_right, ghosts_right)
        st["checked"] += 1
        F[owned] = forces_for(case, owned, ghosts, P, types)
        V[owned] += (0.5 * dt / mass[owned])[:, None] * F[owned]
        st.update(owned=owned, ghosts=ghosts, ref=P.copy(), send=(to_l, to_r))

    def plain_step():
        # atoms keep their rank and their role; coordinates stay unwrapped; only the boundary atoms' coordinates travel (Exchanger::exchange_ranges)
        nonlocal P, V, F
        owned, ghosts = st["owned"], st["ghosts"]
        V[owned] += (0.5 * dt / mass[owned])[:, None] * F[owned]
        P[owned] += V[owned] * dt
        if np.any(np.sum((P[owned] - st["ref"][owned]) ** 2, axis=1) > slack * slack):
            st["violated"] = True                    # (the forces from here to the next rebuild may miss a pair: found at the next look)
        to_l, to_r = st["send"]
        fl, fr = exchange(rank, world, [(i, *P[i]) for i in to_l], [(i, *P[i]) for i in to_r])
        for msg in (fl, fr):
            for rec in msg:
                P[rec[0]] = rec[1:4]
        F[owned] = forces_for(case, owned, ghosts, P, types)
        V[owned] += (0.5 * dt / mass[owned])[:, None] * F[owned]

    def snapshot():
        return {"P": P.copy(), "V": V.copy(), "F": F.copy(), "owned": st["owned"].copy(), "ghosts": st["ghosts"].copy(), "checked": st["checked"]}

    def restore(sn):
        nonlocal P, V, F
        P, V, F = sn["P"].copy(), sn["V"].copy(), sn["F"].copy()
        st.update(owned=sn["owned"].copy(), ghosts=sn["ghosts"].copy(), checked=sn["checked"], violated=False)

    done, since_rebuild, repairs = 0, 1 << 30, 0
    snap, snap_at = snapshot(), 0
    while done < nsteps:
        n = min(window, nsteps - done)
        for _ in range(n):
            if since_rebuild >= K - 1:
                rebuild_step(); since_rebuild = 0
            else:
                plain_step(); since_rebuild += 1
        done += n
        if lazy:
            # the look: every rank arrives at the same verdict (Engine::adapt_sort_interval: one all-reduce)
            flags = [None] * world
            dist.all_gather_object(flags, bool(st["violated"]))
            if any(flags):
                restore(snap)
                for _ in range(done - snap_at):
                    rebuild_step()
                since_rebuild = 0
                repairs += 1
            snap, snap_at = snapshot(), done         # (the interval runs on across the look: the restored state of a later repair opens with a rebuild anyway)
    owned = st["owned"]
    # gather and compare with the single-domain oracle
    out = [None] * world
    dist.all_gather_object(out, (owned, P[owned], V[owned], F[owned]))
    if rank == 0:
        ids = np.concatenate([o[0] for o in out])
        assert sorted(ids.tolist()) == list(range(N)), "every atom owned exactly once"
        Pm, Vm, Fm = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3))
        for o in out:
            Pm[o[0]], Vm[o[0]], Fm[o[0]] = o[1], o[2], o[3]
        ref = oracle.Oracle(case)
        ref.forces(0)
        ref.step(nsteps)
        s = ref.state()
        Pm -= np.floor(Pm / L) * L                   # (plain steps of the lazy schedule keep coordinates unwrapped)
        dP = np.abs(Pm - np.stack([s[c] for c in "xyz"], 1))
        dP = np.minimum(dP, L - dP)                 # (an atom that sits on a wall may be reported at either end)
        err = {"": float(dP.max()), "v": float(np.abs(Vm - np.stack([s["v" + c] for c in "xyz"], 1)).max()),
               "f": float(np.abs(Fm - np.stack([s["f" + c] for c in "xyz"], 1)).max())}
        print("SLAB_MODEL " + json.dumps({"world": world, "err": err, "fmax": float(np.abs(Fm).max()), "boundary_counts_checked": st["checked"], "repairs": repairs}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
