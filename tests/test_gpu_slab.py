"""GPU tests of the slab decomposition: several ranks drive the real HIP engine on ONE GPU, halos and
migrants travel through the host-staged gloo transport; results must equal the single-rank engine's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def transport_for(nranks):
    """RCCL (ncclSend/ncclRecv over xGMI, RcclExchanger::exchange) whenever every rank can have a GPU of its own; the host-staged
    gloo callback transport when the ranks have to share the one GPU of the development box (two RCCL ranks cannot share a device).
    The device count comes from libaztot (the system ROCm runtime); torch is NOT imported into this process: its wheel bundles a second
    copy of the HSA runtime, and RCCL brought up later in the same process would then find that un-initialised copy first
    ("no ROCm-capable device is detected").  The workers are fresh torchrun children."""
    from aztotmd_amd import api
    return "rccl" if api.device_count() >= nranks else "callback"


def run_ranks(nranks, name, nsteps, extra=None, port=29611, transport=None, env_extra=None):
    transport = transport or transport_for(nranks)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "slab_worker.py"), name, str(nsteps), transport, json.dumps(extra or {})]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("SLAB_RESULT ")]
    assert r.returncode == 0 and lines, (r.stdout[-3000:], r.stderr[-3000:])
    out = json.loads(lines[-1][len("SLAB_RESULT "):])
    assert out["transport"] == transport
    return out


def test_torch_in_the_process_is_what_rccl_cannot_live_with():
    """Why the rank processes (bench.py, slab_worker.py) do not import torch: documents the observation the control plane of aztotmd_amd.ctl rests on.
    Either outcome is accepted - the test records which one this image shows - but libaztot's own RCCL bring-up without torch must work."""
    code = "import sys; sys.path.insert(0, %r); from aztotmd_amd import api; api.rccl_selftest(0); print('OK')" % os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    r2 = subprocess.run([sys.executable, "-c", "import torch; torch.cuda.is_available(); " + code], capture_output=True, text=True, timeout=300)
    print("RCCL self-test with torch imported and CUDA initialised first: rc", r2.returncode, r2.stderr[-300:])


@pytest.mark.parametrize("nranks,name,nsteps,port", [(2, "lj", 40, 29611), (3, "fennel", 25, 29612), (4, "hot", 60, 29613), (2, "thermo", 20, 29614), (2, "nose", 15, 29616),
                                                     (2, "mol", 60, 29617), (3, "mol", 30, 29618), (2, "ewald", 30, 29619)])
def test_slabs_match_single_rank(nranks, name, nsteps, port):
    out = run_ranks(nranks, name, nsteps, port=port)
    assert out["every_atom_owned_once"] and out["owned_total"] == out["n_atoms"]
    assert out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-10 for v in out["energy_rel"].values()), out["energy_rel"]
    assert out["cross"][0] == out["cross"][2] and out["cross"][1] == out["cross"][3] and out["species_cross_equal"]
    assert out["mom_rel"] < 1e-10


@pytest.mark.parametrize("seed,nranks", [(3, 2), (6, 3), (9, 2), (13, 4)])
def test_slabs_on_random_systems(seed, nranks):
    """seeded random systems (potential mixes, Ewald / Fennell / direct, several species, external field) cut into slabs"""
    out = run_ranks(nranks, "rand%d" % seed, 20, port=29630 + seed)
    assert out["every_atom_owned_once"] and out["owned_total"] == out["n_atoms"]
    assert out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-9 for v in out["energy_rel"].values()), out["energy_rel"]
    assert out["species_cross_equal"]


@pytest.mark.parametrize("seed,nranks", [(2, 2), (7, 2), (9, 3), (13, 2)])
def test_slabs_on_random_dynamics(seed, nranks):
    """random systems with the radiative / Nose-Hoover thermostat, equilibration rescaling (all-rank kinetic energy) and bonded
    terms that straddle the slab boundaries"""
    out = run_ranks(nranks, "dyn%d" % seed, 20, port=29660 + seed)
    assert out["every_atom_owned_once"] and out["owned_total"] == out["n_atoms"]
    assert out["max_rel_err_vs_single"] < 1e-8, out["errs"]
    assert all(v < 1e-8 for v in out["energy_rel"].values()), out["energy_rel"]


def test_slabs_survive_a_heating_step_between_two_calls():
    """aztot_set_state with faster velocities between two aztot_step calls (a heating protocol, a restart): the interval measured on the slow atoms is
    forgotten (every step rebuilds until the next look), so the slab ranks do not run into a skin violation they would have to repair by running a
    window of steps again."""
    out = run_ranks(2, "heat", 45, port=29623)
    assert out["every_atom_owned_once"] and out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-10 for v in out["energy_rel"].values()), out["energy_rel"]


@pytest.mark.parametrize("name,nranks,nsteps,port", [("hot", 2, 70, 29626), ("hot", 3, 45, 29627), ("thermo", 2, 40, 29628)])
def test_slab_ranks_repair_a_skin_violation_by_running_the_window_again(name, nranks, nsteps, port):
    """A slab rank holds hw ghost layers, so it cannot fall back on a wider stencil when an atom leaves its cell's slack between two sorts.  Debug bit 8192
    holds the interval at 32 steps on atoms far too fast for it: the look that finds the violation takes every rank back to the snapshot the last clean look
    left (per-atom arrays, DevStats, Counts, partial sums - device to device) and runs the steps since again with the cells rebuilt every step.  The result
    must equal the single-rank engine's (which repairs its own violations with the wider stencil): x / v / f 1e-9, energies, wall counters, and - 'thermo' -
    the radiative thermostat's per-atom state and random numbers."""
    out = run_ranks(nranks, name, nsteps, extra={"sort_every": 32, "debug": 8192}, port=port)
    assert out["every_atom_owned_once"] and out["owned_total"] == out["n_atoms"]
    assert out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-10 for v in out["energy_rel"].values()), out["energy_rel"]
    assert out["cross"][0] == out["cross"][2] and out["cross"][1] == out["cross"][3] and out["species_cross_equal"]
    if name == "hot":
        assert out["sort_violations"] > 0, out


def test_slabs_with_several_waves_per_cell():
    """options.waves_per_cell = 2: two waves share the LDS tile of a cell in the list kernel (and two waves split the stencil's columns in the staging
    kernel) - on slab ranks, where ghost cells are candidates but never served"""
    out = run_ranks(2, "fennel", 30, extra={"split": 2}, port=29624)
    assert out["every_atom_owned_once"] and out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-10 for v in out["energy_rel"].values()), out["energy_rel"]


def test_slab_ranks_grow_their_lists_together():
    """Lists that start too small (AZTOT_ITER_CAP forces 16 iterations per cell: every cell of this liquid needs more) are re-allocated at the first look;
    the rebuild that must follow is agreed between the ranks (a rebuild step carries the full exchange: one rank rebuilding alone would desynchronise the
    protocol), and the run stays equal to the single-rank engine's."""
    out = run_ranks(3, "lj", 60, port=29625, env_extra={"AZTOT_ITER_CAP": "16", "AZTOT_VERBOSE": "1"})
    assert out["every_atom_owned_once"] and out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-10 for v in out["energy_rel"].values()), out["energy_rel"]
    assert out["sort_interval"] > 1 and out["pair_lists"] == 1 and out["cells_without_list"] == 0, out


def test_slabs_with_deferred_half_kick():
    """debug bit 256: the large-system path (second half-kick applied by the next step's k_integrate1_bin, which in slab mode also
    packs the migrants and the halo from the freshly kicked velocities)."""
    out = run_ranks(2, "lj", 25, extra={"debug": 256}, port=29620)
    assert out["every_atom_owned_once"] and out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-10 for v in out["energy_rel"].values()), out["energy_rel"]


def test_slabs_with_per_atom_kernel():
    out = run_ranks(2, "lj", 10, extra={"pair_variant": 1}, port=29615)
    assert out["max_rel_err_vs_single"] < 1e-9


@pytest.mark.parametrize("name,nranks,port", [("lj", 2, 29621), ("fennel", 3, 29622)])
def test_slabs_with_the_per_atom_kernel(name, nranks, port):
    """pair_variant 1 (one thread per atom, any geometry: the fallback where no cell tile fits) on slab ranks: ghosts are read like owned atoms"""
    out = run_ranks(nranks, name, 15, extra={"pair_variant": 1}, port=port)
    assert out["every_atom_owned_once"] and out["max_rel_err_vs_single"] < 1e-9, out["errs"]


def test_neighbours_that_disagree_about_their_boundary_atoms_fail_before_any_plain_exchange():
    """RCCL first-contact hardening: behind every sort that opens an interval of plain steps the ranks exchange {atoms I will send you per plain step,
    ghosts I hold on your side} (fixed-size messages, Exchanger::exchange_counts) and compare them with their own ranges BEFORE the first plain step
    posts a send / receive whose counts each rank derives locally (RcclExchanger::exchange_ranges would hang or deliver coordinates to the wrong atoms on a
    mismatch).  Here the callback transport of rank 0 lies by one atom: both ranks of the boundary must come back with AZTOT_ERR_COMM."""
    out = run_ranks(2, "lj", 40, port=29690, transport="callback_corrupt")
    assert out["codes"] == [-5, -5], out
    assert "disagree about their boundary atoms" in out["message"], out
    # a failed handle refuses to step on (first error repeated) and still answers a post-mortem read
    for code, repeated, clock_ok in out["after_failure"]:
        assert code < 0 and repeated and clock_ok, out


def test_rccl_library_selftest():
    """The slab transport's RCCL calls (dlopen'ed librccl, ncclCommInitRank, one ncclGroup of two sends + two receives in the
    N-GPU order, ncclAllReduce on the engine stream) on a one-rank communicator: the only way to run them on a one-GPU box."""
    from aztotmd_amd import api
    api.rccl_selftest(0)


@pytest.mark.parametrize("nranks,name,nsteps", [(2, "lj", 40), (2, "mol", 40), (4, "hot", 60), (8, "big", 30)])
def test_slabs_over_rccl(nranks, name, nsteps):
    """The production transport between real devices: one process per GPU, ncclSend/ncclRecv ring exchange + ncclAllReduce.  Needs
    as many GPUs as ranks (skipped on the one-GPU development box; runs as soon as the suite meets a multi-GPU node)."""
    from aztotmd_amd import api
    if api.device_count() < nranks:
        pytest.skip("needs %d GPUs, this machine has %d" % (nranks, api.device_count()))
    out = run_ranks(nranks, name, nsteps, port=29700 + nranks, transport="rccl")
    assert out["rccl_ranks"] == nranks
    assert out["every_atom_owned_once"] and out["owned_total"] == out["n_atoms"]
    assert out["max_rel_err_vs_single"] < 1e-9, out["errs"]
    assert all(v < 1e-10 for v in out["energy_rel"].values()), out["energy_rel"]


def test_overlapped_coordinate_exchange_changes_nothing():
    """opt-in (debug bit 16384): plain steps of a slab rank run the coordinate exchange on a second stream beside the interior cells' pair
    forces and launch the boundary cells afterwards (three launches instead of one).  One rank of 2 talking to itself (loopback transport: the
    only device-side transport a one-GPU box has) with and without the overlap: per-atom state bit for bit, energies to round-off."""
    from aztotmd_amd import api, inputs
    import numpy as np
    case = inputs.lj_case((42, 5, 5), a=5.735, seed=31, rc=8.5, vel_T=8.0)            # 28 cell layers along x: 14 owned layers per rank, 10 of them interior
    res = []
    for dbg in (0, 16384):
        e = api.Engine(api.Model.from_case(case), slab={"rank": 1, "nranks": 2, "loopback": True}, debug=dbg)   # 14 layers = 21 lattice cells: the loopback seam matches the lattice
        for n in (6, 30, 30):
            e.step(n)
        st = e.stats()
        assert st["sort_interval"] > 1
        s = e.state()
        own = ~np.isnan(s["x"])
        res.append((s, st, own))
    assert np.array_equal(res[0][2], res[1][2]) and res[0][2].sum() > 1000
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert np.array_equal(res[0][0][k][res[0][2]], res[1][0][k][res[1][2]]), k
    for k in ("engVdW", "engKin", "engTot"):
        assert abs(res[0][1][k] - res[1][1][k]) <= 1e-12 * abs(res[1][1][k]), k
