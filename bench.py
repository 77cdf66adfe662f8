#!/usr/bin/env python3
"""bench.py - ns/day of the azTotMD per-step hot path on MI355X (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W
N > 1 either way: under a launcher that sets RANK / WORLD_SIZE (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
--gpus N ...) this process IS one rank; started plainly (no WORLD_SIZE in the environment) it becomes the launcher itself - before it loads
libaztot or makes any HIP call it starts N fresh rank processes of this script, relays rank 0's JSON line and exits with the worst exit
code among them (launch_ranks).

Workload (BASELINE.json config "1 000 000 Ar LJ", SURVEY 8d "C4"): 1 000 188 argon atoms, FCC 63^3 cells of
5.735 A with +-0.15 A jitter (seed 20240502), LJ eps 0.01006 eV sigma 3.3952 A, cut-off 8.5 A, dt 1 fs, NVE,
fp64.  With N GPUs the SAME box is split into N slabs along x (strong scaling), one process per GPU, ghost
atoms exchanged with the two ring neighbours over RCCL.

A "step" is one pass of the hot path (half-kick+drift+wrap+bin, scan, sort, pair forces, half-kick, energy
bookkeeping) over all atoms.  Atoms are resident in HBM when the timed region starts.

Output: ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s, MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_VECTOR_PEAK = 78.6e12   # FLOP/s, FP64 vector (256 CUs x 128 FLOP/clk x 2.4 GHz)
PAIR_BYTES_PER_ATOM = 52.0   # x,y,z,type read + fx,fy,fz written
PAIR_BYTES_PER_CELL = 8.0    # cellStart/cellCount
# algorithmic bytes per atom of the streaming kernels: the ones for which ">= 40 % of HBM peak" is the meaningful target
STREAM_BYTES_PER_ATOM = {"integrate1_bin": 132.0, "integrate1": 124.0, "place": 28.0, "rank_gather": 144.0, "integrate2": 76.0, "post_tstat": 84.0}
# (integrate1 = a plain step of the lazy re-sort: no cell id / slot written (-8); the reference position (24 B) is read only for atoms the displacement
#  bound cannot clear - none in a normal run; rank_gather writes that reference (+24))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C4", help="C4 (1 000 188 Ar LJ, default), C4T (the same at 85 K), C4X, C3 / C3T (+Fennell Coulomb), C2 / C2T (40 000 Ar LJ), M4, S4, S40, B3, E2")
    ap.add_argument("--case-study", type=int, default=0, help="1 or 2: run the reference's shipped example input verbatim (BASELINE configs 1 and 5; atoms.xyz / field.txt / control.txt / "
                                                              "cuda.txt written back from tests/golden/case_study_K.npz and parsed by aztot_init_md) instead of a synthetic workload")
    ap.add_argument("--pair-variant", type=int, default=0)
    ap.add_argument("--cell-size", type=float, default=0.0)
    ap.add_argument("--sort-every", type=int, default=0, help="cell-list rebuild schedule: 0 adaptive lazy re-sort (default), 1 every step (the reference's), n at most every n-th step")
    ap.add_argument("--skin", type=float, default=0.0, help="Verlet skin in A (0: automatic, < 0: none - cells exactly as control.cell_list gives them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="steps of the CPU baseline sample (0: sized for ~15 s)")
    ap.add_argument("--debug", type=int, default=0,
                    help="engine path switches for A/B measurements: 128 k_integrate2 every step, 256 large-system kick path, 512 generic pair kernel")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="measurement aid: time rank 0 of an N-rank slab run on one GPU (loopback halo)")
    ap.add_argument("--no-profile", action="store_true", help="do not time individual kernels (enables hipGraph replay)")
    ap.add_argument("--no-graph", action="store_true", help="A/B aid: launch every kernel eagerly instead of replaying captured cycles")
    ap.add_argument("--split", type=int, default=0, help="A/B aid: waves per cell in the staging pair kernel (0: the engine decides)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady_state / call_overhead blocks")
    ap.add_argument("--steady-steps", type=int, default=300)
    ap.add_argument("--dry-run", action="store_true", help="launcher check: every rank reports its environment through the control plane and exits without touching a GPU")
    return ap.parse_args()


EXIT_RCCL_FAILED = 3        # one GPU per rank and RCCL did not come up
EXIT_REHEARSAL = 4          # fewer GPUs than ranks: the line printed is a rehearsal (host-staged halo), never a result
GPU_LIBRARIES = ("libaztot", "libamdhip64", "libhsa-runtime", "librccl")


def mapped_gpu_libraries(pid="self"):
    """GPU runtime libraries mapped into a process (the launcher must have none: it never re-executes or forks a process that touched the GPU)"""
    try:
        maps = open("/proc/%s/maps" % pid).read()
    except OSError:
        return None
    return sorted({g for g in GPU_LIBRARIES if g in maps})


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (fresh interpreters, one per GPU), hand them RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* and a per-run control-plane token, relay rank 0's stdout (the ONE JSON line), send every other rank's stdout to stderr, and
    return the worst exit code.  Nothing here imports the package, loads libaztot or calls HIP."""
    import secrets
    import signal
    import socket
    import threading
    assert not mapped_gpu_libraries(), "the launcher must not have touched the GPU"
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:           # a free port for the control plane (the ranks use nothing else on it)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AZTOT_CTL_PORT_OFFSET="0",
               AZTOT_CTL_TOKEN=secrets.token_hex(16), AZTOT_BENCH_LAUNCHER_PID=str(os.getpid()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e, stdout=subprocess.PIPE, stderr=None, start_new_session=True))

    def relay(proc, to):
        for line in proc.stdout:
            to.write(line)
            to.flush()

    threads = [threading.Thread(target=relay, args=(p, sys.stdout.buffer if r == 0 else sys.stderr.buffer), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    worst, first_failure = 0, None
    live = set(range(n))
    # a rank that hangs (a collective its neighbour never joins) must not hold the node for ever: after AZTOT_BENCH_TIMEOUT seconds (default 900) the launcher ends
    # exactly the process groups it started and reports 124, like timeout(1)
    deadline = time.time() + float(os.environ.get("AZTOT_BENCH_TIMEOUT", "900"))
    try:
        while live:
            if time.time() > deadline:
                sys.stderr.write("bench.py launcher: ranks %s still running after the time limit - ending them\n" % sorted(live))
                for r in sorted(live):
                    try:
                        os.killpg(procs[r].pid, signal.SIGTERM)
                    except OSError:
                        pass
                time.sleep(5.0)
                for r in sorted(live):
                    if procs[r].poll() is None:
                        try:
                            os.killpg(procs[r].pid, signal.SIGKILL)
                        except OSError:
                            pass
                worst = max(worst, 124)
                break
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0:
                    worst = max(worst, rc if rc > 0 else 128 - rc)
                    if first_failure is None:
                        first_failure = time.time()
                        sys.stderr.write("bench.py launcher: rank %d exited with code %d\n" % (r, rc))
            if first_failure is not None and live and time.time() - first_failure > 30.0:
                # a rank died: the others would wait for it in a collective until the control plane's timeout - end exactly the process groups started here
                for r in sorted(live):
                    sys.stderr.write("bench.py launcher: ending rank %d (pid %d) after another rank failed\n" % (r, procs[r].pid))
                    try:
                        os.killpg(procs[r].pid, signal.SIGTERM)
                    except OSError:
                        pass
                first_failure = time.time() + 1e9
                worst = max(worst, 1)
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        worst = max(worst, 130)
    for t in threads:
        t.join(timeout=5.0)
    return worst


def cpu_baseline(case, steps):
    """The reference's own serial code (oracle/_ref, built from /root/reference) when the binary travelled here,
    else our C port of it (oracle/liboracle.so); one thread, as the reference is serial (main.cpp, no OpenMP)."""
    from oracle import oracle
    import numpy as np
    n = len(case["types"])
    dt = case["dt"]
    if oracle.ref_available():
        c = dict(case)
        c.update(nsteps=steps, dump=[], init_forces=0, use_clist=1, center_box=0)
        drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
        from oracle import casefile
        with tempfile.TemporaryDirectory() as td:
            cp, op = os.path.join(td, "case.bin"), os.path.join(td, "out.bin")
            casefile.write_case(cp, c)
            r = subprocess.run([drv, cp, op], capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                raise RuntimeError("ref_driver failed: " + r.stderr[-500:])
            summ = json.loads(r.stdout.strip().splitlines()[-1])
        wall = summ["wall_s"]
        energies = {"engVdW": summ.get("engVdW"), "engTot": summ.get("engTot")}
        kind = "reference"
    else:
        o = oracle.Oracle(case)
        o.forces(1)
        t0 = time.perf_counter()
        o.step(steps)
        wall = time.perf_counter() - t0
        energies = None          # started from the initial forces, unlike the reference sample below: no same-run comparison
        kind = "port"
    nsday = steps * dt * 1e-3 / wall * 86400.0
    return {"value": nsday, "unit": "ns/day", "cores": 1, "kind": kind,
            "sample": "%d steps of the same %d-atom workload, linked-cell serial path, 1 thread; %.2f s wall; %.3f Matom-steps/s"
                      % (steps, n, wall, n * steps / wall / 1e6),
            "host_cpu": _cpu_model(), "host_cores": os.cpu_count(), "_energies": energies}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# kernels that run only on a step that rebuilds the cell list (everything else runs on every step)
REBUILD_ONLY = ("integrate1_bin", "bin", "scan_cells", "place", "rank_gather", "build_lists", "exchange", "unpack_halo")
WORKLOADS = {"C4": "1 000 188 Ar, LJ rc 8.5 A, FCC 63^3 a=5.735 jitter 0.15, dt 1 fs, NVE, init_vel zero (BASELINE config '1 000 000 Ar LJ', SURVEY C4)",
             "C4T": "C4 thermalised: Maxwell velocities at 85 K (init_vel gaus), equilibrates to ~44 K kinetic + lattice potential",
             "C4X": "C4's lattice in a box of exactly 42 x 8.5 A (no overhang of the cells over the cut-off)",
             "C4L": "the C4 liquid on a 64^3-cell lattice (1 048 576 atoms; with --cell-size 9.176: 40 cell layers = 8 ranks x 5)", "C4LT": "C4L thermalised: Maxwell velocities at 85 K",
             "C3": "1 000 188 atoms, LJ rc 8.5 A + Fennell/DSF Coulomb q=+-0.2 (SURVEY C3)",
             "C3T": "C3 thermalised: Maxwell velocities at 85 K",
             "C2": "40 000 Ar, LJ rc 8.5 A (SURVEY C2)", "C2T": "C2 thermalised: Maxwell velocities at 85 K",
             "S4": "4 000 atoms, surk rc 6.0 + radii on 2.7 A cells, radiative thermostat 500 K (periodic analogue of case study 2)",
             "S40": "40 000 atoms, surk rc 6.0 + radii on 2.7 A cells, radiative thermostat 500 K",
             "M4": "1 029 000 atoms in 343 000 bonded triatomics (5 bond potentials, hcos angles), LJ + Fennell",
             "B3": "1 000 188 ions, Born-Mayer-Huggins + Fennell (heats up from a lattice)",
             "E2": "40 000 ions, LJ + full Ewald sum (4 231 k-vectors)"}


def main():
    a = parse()
    if a.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "AZTOT_BENCH_LAUNCHER_PID" not in os.environ:
        # started plainly with --gpus N: this process only launches the ranks (a plain `python bench.py --gpus 8` ran ONE rank)
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    # the contract is ONE JSON line on stdout: keep library chatter (RCCL, HIP runtime) away from it
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np  # noqa: F401
    # NO torch in this process: the ranks bring up RCCL inside libaztot (system ROCm runtime), and torch's wheel would put its own HIP / HSA copies in
    # front of it.  The launcher (python -m torch.distributed.run) only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; the control plane
    # (RCCL id, barriers, a few scalars) is aztotmd_amd.ctl over plain TCP.
    from aztotmd_amd import api, ctl, inputs

    cp = ctl.Control()
    rank, world = cp.rank, cp.world
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1:
        a.gpus = world
    if a.dry_run:
        # what a rank was handed, gathered over the control plane; no GPU library is loaded in any process of a dry run
        lp = os.environ.get("AZTOT_BENCH_LAUNCHER_PID")
        mine = {"rank": rank, "local_rank": local_rank, "world": world, "pid": os.getpid(), "ppid": os.getppid(), "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
                "token_set": bool(os.environ.get("AZTOT_CTL_TOKEN")), "gpu_libraries": mapped_gpu_libraries()}
        ranks = cp.all_gather(mine)
        if os.environ.get("AZTOT_DRY_RUN_HANG") and rank == 1:
            time.sleep(3600)            # (test of the launcher's time limit: a rank that never joins the next collective)
        if rank == 0:
            os.dup2(real_stdout, 1)
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": ranks, "launcher_pid": int(lp) if lp else None,
                              "launcher_gpu_libraries": mapped_gpu_libraries(lp) if lp else None}), flush=True)
        cp.barrier()
        cp.close()
        return
    ndev = api.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    oversubscribed = world > ndev          # rehearsal on a box with fewer GPUs than ranks: ranks share devices, RCCL cannot be used
    dev = local_rank % ndev

    if a.case_study in (1, 2):
        # the reference's own input surface: four text files in a directory, parsed by the library (sys_init.cpp:1036-1119)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import materialise_case_study
        td = tempfile.mkdtemp(prefix="aztot_cs%d_" % a.case_study)
        materialise_case_study(a.case_study, td)
        model = api.Model.from_dir(td)
        n_atoms = int(model.query("n_atoms")[0])
        case = {"dt": float(model.query("dt")[0]), "types": [0] * n_atoms, "tstat_type": 2}
        a.workload = "case study %d" % a.case_study
        WORKLOADS[a.workload] = "the reference's shipped 'case study %d' input, verbatim (%d atoms)" % (a.case_study, n_atoms)
        a.no_cpu_baseline = True
    else:
        case = inputs.config(a.workload)
        n_atoms = len(case["types"])
        model = api.Model.from_case(case)
    slab = None
    if world > 1:
        idb = cp.broadcast(api.rccl_unique_id() if rank == 0 else None)
        slab = {"rank": rank, "nranks": world, "rccl_id": idb}
    if a.emulate_ranks > 1 and world == 1:
        slab = {"rank": a.emulate_ranks // 2, "nranks": a.emulate_ranks, "loopback": True}
    transport = "single GPU"
    kw = dict(device=dev, initial_forces=1, pair_variant=a.pair_variant, cell_size=a.cell_size, debug=a.debug, sort_every=a.sort_every, split=a.split, skin=a.skin)
    err = ""
    try:
        eng = api.Engine(model, use_graph=0 if a.no_graph else 1, profile=0, slab=slab, **kw)
        if world > 1:
            transport = "RCCL send/recv over xGMI"
        if a.emulate_ranks > 1:
            transport = "LOOPBACK EMULATION of one rank of %d - not a result" % a.emulate_ranks
        ok = 1
    except api.AztotError as ex:
        if world == 1:
            raise
        ok, err = 0, str(ex)
    if world > 1 and cp.all_min(ok) == 0:
        if ok:
            eng.close()                  # ranks whose engine came up: release it before anything else is built
        if not oversubscribed:
            # one GPU per rank and RCCL does not come up: that is a failed run, not a slower one - a scaling record must
            # never report "ok" with the halo travelling through the host
            sys.stderr.write("bench.py rank %d: RCCL initialisation failed with one GPU per rank: %s\n" % (rank, err if not ok else "on another rank"))
            cp.barrier()
            cp.close()
            raise SystemExit(EXIT_RCCL_FAILED)
        # rehearsal on a box with fewer GPUs than ranks (RCCL cannot put two ranks on one device): host-staged transport relayed by the
        # control plane, clearly labelled, never a result
        slab = {"rank": rank, "nranks": world, "sendrecv": lambda sp, data, rp, rcap: cp.sendrecv(sp, bytes(data), rp),
                "allreduce": lambda arr: np.asarray(cp.all_sum(arr.tolist()))}
        eng = api.Engine(model, use_graph=0, profile=0, slab=slab, **kw)
        transport = "REHEARSAL: host-staged over TCP, ranks share GPUs - not a result"

    def barrier():
        eng.sync()                # (one GPU: aztot_step may leave the end of a call - deferred half-kick, statistics, look - to whoever reads next; a timed region includes it)
        cp.barrier()
        api.device_synchronize(dev)

    rccl_ranks = eng.comm_ranks()            # ncclCommCount of the communicator the halo travels on (0: no RCCL)
    if world > 1 and not oversubscribed and rccl_ranks != world:
        raise SystemExit("bench.py: %d ranks but the RCCL communicator has %d" % (world, rccl_ranks))

    # ---- timed region: EXACTLY a.steps steps, no per-kernel instrumentation (the step is replayed as a hipGraph where that pays)
    eng.step(a.warmup)
    st0 = eng.stats()
    barrier()
    t0 = time.perf_counter()
    eng.step(a.steps)             # returns after the engine's stream has drained
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        wall = cp.all_max(wall)
    st = eng.stats()
    rebuilds_in_window = st["rebuilds"] - st0["rebuilds"]
    # ---- second pass over the same number of steps with HIP events around every kernel (on the engine's own stream):
    #      source of the per-kernel durations / the roofline figure; its wall time is reported separately
    ktimes, wall_events = {}, None
    if not a.no_profile:
        eng.set_profile(1)
        eng.reset_kernel_times()
        barrier()
        t0 = time.perf_counter()
        eng.step(max(a.steps, 2 * int(st["sort_interval"]) + 2))      # (at least two sort intervals, so that every kernel of the cycle is seen)
        barrier()
        wall_events = (time.perf_counter() - t0) / max(a.steps, 2 * int(st["sort_interval"]) + 2) * a.steps
        ktimes = eng.kernel_times()
        eng.set_profile(0)
    profile = 0 if a.no_profile else 1
    if world > 1 and profile:
        # slowest rank per kernel
        names = sorted(ktimes)
        mine = [ktimes[k]["ms"] / max(ktimes[k]["calls"], 1) for k in names]
        for k, v in zip(names, [max(col) for col in zip(*cp.all_gather(mine))]):
            ktimes[k]["avg_ms_max_over_ranks"] = v

    if rank == 0:
        dt_ps = case["dt"]
        K = max(int(st["sort_interval"]), 1)
        out = {
            "metric": "ns_per_day", "value": None, "unit": "ns/day", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "library": api.lib().aztot_version().decode(),
            "config": {"workload": WORKLOADS.get(a.workload, a.workload),
                       "n_atoms": n_atoms, "n_cells": st["n_cells"], "decomposition": "single GPU" if world == 1 else "%d slabs along x" % world, "transport": transport, "rccl_ranks": rccl_ranks, "ranks_share_gpus": oversubscribed,
                       "pair_variant": a.pair_variant, "sort_interval": st.get("sort_interval"), "sort_violations": st.get("sort_violations"), "skin_A": st.get("skin"),
                       "temperature_K": st.get("temperature"),
                       "pair_lists": st.get("pair_lists"), "cells_without_list": st.get("cells_without_list"),
                       "kernel_timing": "second pass with HIP events on the engine stream" if profile else "off"},
            "energy": {"engTot": st["engTot"], "engVdW": st["engVdW"], "engKin": st["engKin"], "pairs_dropped": st["pairs_dropped"]},
        }
        kern = {}
        if ktimes:
            kern = {k: {"avg_us": 1e3 * v.get("avg_ms_max_over_ranks", v["ms"] / max(v["calls"], 1)), "calls": v["calls"]} for k, v in ktimes.items()}
            out["kernels"] = kern
        # ---- what the timed window held.  The cell list is rebuilt every K-th step only (the reference: every step, main.cu:300-326); a window of fewer
        # steps than K may hold no rebuild at all.  `value` never profits from that: when the window holds fewer rebuilds than its share steps / K, the
        # missing ones are charged at the measured cost of a rebuild step (event-timed kernels that run on rebuild steps only, minus the plain
        # integrate kernel they replace).
        rebuild_us = None
        if kern:
            rebuild_us = sum(kern[k]["avg_us"] for k in REBUILD_ONLY if k in kern and kern[k]["calls"] > 0)
            if "integrate1" in kern and kern["integrate1"]["calls"] > 0 and rebuild_us > 0:
                rebuild_us -= kern["integrate1"]["avg_us"]
        share = a.steps / K
        missing = max(0.0, share - rebuilds_in_window) if K > 1 else 0.0
        wall_charged = wall + (missing * rebuild_us * 1e-6 if rebuild_us else 0.0)
        out["ms_per_step"] = wall_charged / a.steps * 1e3
        out["value"] = a.steps * dt_ps * 1e-3 / wall_charged * 86400.0
        out["matom_steps_per_s"] = n_atoms * a.steps / wall_charged / 1e6
        out["timed_window"] = {"ms_per_step_as_measured": wall / a.steps * 1e3, "rebuilds_in_timed_region": rebuilds_in_window, "fair_share_of_rebuilds": share,
                               "rebuild_step_extra_us": rebuild_us, "amortised_rebuild_us_per_step": (rebuild_us / K) if rebuild_us else None,
                               "charged_for_missing_rebuilds_ms": (wall_charged - wall) * 1e3,
                               "pair_energies": "booked on the last step of the call only (the only one whose statistics the caller can see); "
                                                "energies_on_every_step_ms_per_step is the same run with options.energies_every_step = 1"}
        out["ms_per_step_with_events"] = (wall_events / a.steps * 1e3) if wall_events else None
        if kern:
            # the pair kernel that carries the run: largest total time (plain steps of the lazy re-sort: pair_list; steps that rebuild the cells: pair_tile)
            pair_name = max((k for k in kern if k.startswith("pair")), key=lambda k: kern[k]["avg_us"] * kern[k]["calls"], default=None)
            if pair_name:
                t_pair = kern[pair_name]["avg_us"] * 1e-6
                n_rank = n_atoms / (a.emulate_ranks if a.emulate_ranks > 1 else world)       # atoms one rank's pair kernel serves
                alg = PAIR_BYTES_PER_ATOM * n_rank + PAIR_BYTES_PER_CELL * st["n_cells"] / world
                # PMC counters cannot be collected inside this run (rocprofv3 wraps the process); they come from profiles/pmc_traffic.json, whose entries name
                # the library build they were measured on (tools/make_profiles.py records aztot_version's source digest and the kernel symbol).  An entry
                # from another build is NOT replayed: traffic / fp64 figures are null then, and `counters_stale` says what was found
                traffic = flop = None
                rec, stale = {}, None
                tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if os.path.exists(tp) and a.emulate_ranks <= 1:        # (the counters were collected on the whole system: they say nothing about one rank's share)
                    try:
                        rec = json.load(open(tp)).get("%s:%s:%d" % (a.workload, pair_name, world)) or {}
                    except Exception:
                        rec = {}
                    if rec and rec.get("library") == out["library"]:
                        traffic = rec.get("hbm_bytes_per_launch")
                        flop = rec.get("fp64_flop_per_launch")
                    elif rec:
                        stale = {"measured_on": rec.get("library", "a build that left no fingerprint (%s)" % rec.get("round")), "running": out["library"]}
                        rec = {}
                # SURVEY 8d asks for three numbers side by side: (1) compulsory-byte fraction of the HBM roofline, (2) the measured HBM
                # rate (PMC bytes / launch time; a traffic measure), (3) FP64 FLOP / time / FP64-vector peak (what actually bounds it)
                out["roofline"] = {"bound": "hbm", "kernel": pair_name, "achieved": alg / t_pair / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                   "frac": alg / t_pair / HBM_PEAK, "traffic": traffic,
                                   "algorithmic_bytes_per_launch": alg, "avg_launch_us": t_pair * 1e6,
                                   "measured_hbm_gbps": (traffic / t_pair / 1e9) if traffic else None,
                                   "fp64_flop_per_launch": flop,
                                   "fp64_tflops": (flop / t_pair / 1e12) if flop else None,
                                   "fp64_frac": (flop / t_pair / FP64_VECTOR_PEAK) if flop else None,
                                   "fp64_peak_tflops": FP64_VECTOR_PEAK / 1e12,
                                   "counters_from": rec.get("round"), "counters_kernel": rec.get("kernel"), "counters_stale": stale,
                                   "note": "frac counts the COMPULSORY bytes (52 B/atom + 8 B/cell); traffic is what the kernel really moves - pair_list streams its "
                                           "candidate and pair lists (recorded when the cells were rebuilt) once per step, which is what replaces "
                                           "staging, filtering and mask handling; fp64_flop_per_launch = (2 FMA + ADD + MUL + TRANS) x 64 from the "
                                           "SQ_INSTS_VALU_*_F64 counters in profiles/"}
            # the streaming kernels: algorithmic bytes / launch time / HBM peak
            n_local = n_atoms / world
            stream = {}
            for k, b in STREAM_BYTES_PER_ATOM.items():
                if k in kern and kern[k]["calls"] > 0:
                    bb = b + (32.0 if (k == "rank_gather" and case.get("tstat_type", 0) == 2) else 0.0) + (16.0 if (k == "rank_gather" and st.get("pair_lists")) else 0.0)
                    t_k = kern[k]["avg_us"] * 1e-6
                    stream[k] = {"bytes_per_atom": bb, "avg_us": kern[k]["avg_us"], "gbps": bb * n_local / t_k / 1e9, "hbm_frac": bb * n_local / t_k / HBM_PEAK}
            out["streaming"] = stream
        if world == 1 and a.emulate_ranks <= 1 and not a.no_steady:
            # (a) steady state: the timed window above is whatever --steps / --warmup the caller chose (the driver: 20 after 5 - inside the clock ramp of a
            # fresh process); this is the same engine a few hundred steps later, one long call.  (b) call overhead: the reference's loop is per step
            # (main.cu:281-410), so a caller that couples something to every step calls aztot_step(1) - each call ends with the deferred half-kick,
            # the statistics reduction, a stream synchronisation and the look at the sort interval
            try:
                ns = a.steady_steps
                eng.step(ns)
                eng.sync()
                r0 = eng.stats()["rebuilds"]
                t0 = time.perf_counter()
                eng.step(ns)
                eng.sync()
                w = time.perf_counter() - t0
                s1 = eng.stats()
                out["steady_state"] = {"steps": ns, "warmup": ns, "ms_per_step": w / ns * 1e3, "ns_per_day": ns * dt_ps * 1e-3 / w * 86400.0,
                                       "rebuilds_in_timed_region": s1["rebuilds"] - r0, "sort_interval": s1["sort_interval"]}
                n1 = 200
                t0 = time.perf_counter()
                for _ in range(n1):
                    eng.step(1)
                eng.sync()
                w1 = time.perf_counter() - t0
                stat = max(int(model.query("stat")[0]), 1)
                ncall = max(2, 400 // stat)
                t0 = time.perf_counter()
                for _ in range(ncall):
                    eng.step(stat)
                    eng.stats()
                eng.sync()
                w2 = time.perf_counter() - t0
                out["call_overhead"] = {"step1_calls": n1, "step1_ms_per_step": w1 / n1 * 1e3, "step1_over_long_call": (w1 / n1) / (w / ns),
                                        "stat_interval": stat, "step_stat_calls": ncall, "step_stat_ms_per_step": w2 / (ncall * stat) * 1e3,
                                        "step_stat_over_long_call": (w2 / (ncall * stat)) / (w / ns)}
            except Exception as ex:   # noqa: BLE001
                out["steady_state"] = "failed: %r" % (ex,)
        if world == 1 and not a.no_cpu_baseline and a.emulate_ranks <= 1:
            # the same timed region with pair energies booked on every step (options.energies_every_step): what a caller who asks for statistics after every
            # single step would see
            try:
                e2 = api.Engine(model, use_graph=0 if a.no_graph else 1, profile=0, energies_every_step=1, **kw)
                e2.step(a.warmup)
                e2.sync()
                t0 = time.perf_counter()
                e2.step(a.steps)
                e2.sync()
                out["timed_window"]["energies_on_every_step_ms_per_step"] = (time.perf_counter() - t0) / a.steps * 1e3
                e2.close()
            except Exception as ex:   # noqa: BLE001
                out["timed_window"]["energies_on_every_step_ms_per_step"] = "failed: %r" % (ex,)
            # the same box thermalised (Maxwell velocities at 85 K): the reference rebuilds its cell list every step, so its cost does not depend on
            # temperature - the lazy schedule's does, and a number quoted on a lattice at rest alone would hide that
            if a.workload in ("C4", "C3", "C2"):
                try:
                    hot = inputs.config(a.workload + "T")
                    e3 = api.Engine(api.Model.from_case(hot), use_graph=0 if a.no_graph else 1, profile=0, **kw)
                    e3.step(200)
                    e3.sync()
                    s0 = e3.stats()
                    t0 = time.perf_counter()
                    e3.step(200)
                    e3.sync()
                    w3 = time.perf_counter() - t0
                    s3 = e3.stats()
                    out["thermalised"] = {"workload": WORKLOADS.get(a.workload + "T"), "steps": 200, "warmup": 200, "ms_per_step": w3 / 200 * 1e3,
                                          "ns_per_day": 200 * hot["dt"] * 1e-3 / w3 * 86400.0, "sort_interval": s3["sort_interval"], "rebuilds_in_timed_region": s3["rebuilds"] - s0["rebuilds"],
                                          "sort_violations": s3["sort_violations"], "temperature_K": s3["temperature"], "cells_without_list": s3["cells_without_list"]}
                    e3.close()
                except Exception as ex:   # noqa: BLE001
                    out["thermalised"] = "failed: %r" % (ex,)
            steps_cpu = a.cpu_steps or max(2, int(round(5.0e6 / n_atoms * 1.0)))   # ~3 us per atom-step -> about 15 s
            steps_cpu = min(steps_cpu, 200)
            try:
                out["cpu_baseline"] = cpu_baseline(case, steps_cpu)
                ref_e = out["cpu_baseline"].pop("_energies", None)
                if ref_e and ref_e.get("engTot") is not None:
                    # same inputs, same number of steps, same start (F = 0 as the reference sample): the HIP path's energies next to
                    # the CPU reference's, in this very run (force-level parity is the test suite's job)
                    chk = api.Engine(model, device=dev, initial_forces=0, pair_variant=a.pair_variant, cell_size=a.cell_size, skin=a.skin)
                    chk.step(steps_cpu)
                    cs = chk.stats()
                    chk.close()
                    out["cpu_baseline"]["same_run_parity"] = {
                        "steps": steps_cpu, "engVdW_gpu": cs["engVdW"], "engVdW_cpu": ref_e["engVdW"],
                        "engVdW_rel_diff": abs(cs["engVdW"] - ref_e["engVdW"]) / abs(ref_e["engVdW"]),
                        "engTot_rel_diff": abs(cs["engTot"] - ref_e["engTot"]) / abs(ref_e["engTot"])}
            except Exception as ex:   # noqa: BLE001 - the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "ns/day", "cores": 1, "kind": "port", "sample": "failed: %r" % (ex,)}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    eng.close()
    if world > 1:
        cp.barrier()
    cp.close()
    if oversubscribed:
        # the line above says "REHEARSAL ... not a result"; the exit code says so too (a scaling record must not be built from it)
        raise SystemExit(EXIT_REHEARSAL)


if __name__ == "__main__":
    main()
