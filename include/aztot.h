/* aztot.h - C ABI of the MI355X-native azTotMD per-step hot path.
 *
 * The reference (raadyn/aztotmd) has no plugin/FFI interface: its hot path is the set of free
 * functions and kernels that main.cu calls on one opaque device struct.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference's src/).
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * All functions return AZTOT_OK (0) on success and a negative code on failure;
 * aztot_last_error() returns a human readable message for the calling thread.
 * A handle is single-threaded (as the reference: one host thread, main.cu:239).
 *
 * Units everywhere: Angstrom, ps, eV, e; masses in amu at this boundary (the reference's input units,
 * const.h:17-49).
 */
#ifndef AZTOT_H
#define AZTOT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZTOT_OK 0
#define AZTOT_ERR_IO -1          /* cannot open / parse an input file (reference: printf("ERROR[..]") + return 0) */
#define AZTOT_ERR_INPUT -2       /* semantically invalid or out-of-scope input */
#define AZTOT_ERR_DEVICE -3      /* HIP runtime error / no device */
#define AZTOT_ERR_ARG -4
#define AZTOT_ERR_COMM -5

/* pair potential ids: vdw.h:15-21 */
enum { AZTOT_VDW_LJ = 1, AZTOT_VDW_BUCK = 2, AZTOT_VDW_746 = 3, AZTOT_VDW_BHM = 4, AZTOT_VDW_ELIN = 5, AZTOT_VDW_EINV = 6, AZTOT_VDW_SURK = 7 };
/* electrostatics: elec.h:8-12 */
enum { AZTOT_ELEC_NONE = 0, AZTOT_ELEC_DIRECT = 1, AZTOT_ELEC_EWALD = 2, AZTOT_ELEC_FENNEL = 3 };
/* thermostats: temperature.h:10-12 */
enum { AZTOT_TSTAT_NONE = 0, AZTOT_TSTAT_NOSE = 1, AZTOT_TSTAT_RADI = 2 };
/* init_vel: dataStruct.h:7-10 */
enum { AZTOT_VEL_ZERO = 0, AZTOT_VEL_GAUSS = 1, AZTOT_VEL_CONST = 2, AZTOT_VEL_KENG = 3 };

typedef struct aztot_model aztot_model;   /* host model: Atoms+Field+Sim+Elec+TStat+Box of dataStruct.h */
typedef struct aztot_md aztot_md;         /* device state: cudaMD + hostManagMD of cuStruct.h:81-423 */

/* one line of the 'spec' section of field.txt (sys_init.cpp:83) + its 'radii' line (sys_init.cpp:468-480) */
typedef struct
{
    char name[8];
    double mass_amu, charge;
    int32_t frozen;              /* 'frozensp' section, sys_init.cpp:241-255 */
    double radA, radB, mxEng;
} aztot_species;

/* one line of the 'vdw' section of field.txt (vdw.cpp:244): raw user parameters */
typedef struct
{
    int32_t spec_a, spec_b, type;
    double rcut;
    double p[5];
} aztot_vdw;

/* the directives of control.txt that touch the hot path (sys_init.cpp:676-989) */
typedef struct
{
    double timestep;             /* ps */
    int32_t nstep, nequil, eqfreq;
    double temperature;          /* K */
    int32_t tstat_type;          /* AZTOT_TSTAT_* */
    double tstat_tau;            /* 'nose tau' */
    int32_t elec_type;           /* AZTOT_ELEC_* */
    double r_real, alpha;        /* elec cut-off, Ewald/Fennell alpha */
    int32_t init_vel;            /* AZTOT_VEL_* */
    double init_vel_par[3];      /* const vx vy vz | keng e */
    double elecfield[3];         /* dU/dx, dU/dy, dU/dz */
    int32_t use_cell_list;       /* 'cell_list' present */
    double cell_list;            /* desired cell edge */
    int32_t stat;                /* statistics period */
    int32_t ewald_k[3];          /* 'elec pme rReal alpha kx ky kz' (read_elec elec.cpp:33-38): k-vectors per axis, 1..16 */
} aztot_control;

/* array form of atoms.xyz + field.txt + control.txt (used by tests / bench; same state as aztot_init_md) */
typedef struct
{
    int32_t n_atoms, n_species, n_vdw;
    double box[3];
    const int32_t *types;
    const double *x, *y, *z, *vx, *vy, *vz;   /* vx..vz may be NULL (zeros) */
    const aztot_species *species;
    const aztot_vdw *vdw;
    aztot_control control;
} aztot_system;

/* bonded terms ("next" row: constant bonds + harmonic-cosine angles) */
enum { AZTOT_BOND_HARM = 1, AZTOT_BOND_MORSE = 2, AZTOT_BOND_PEDONE = 3, AZTOT_BOND_BUCK = 4, AZTOT_BOND_E612 = 5 };   /* bonds.cpp:158-252 */
enum { AZTOT_ANGLE_HCOS = 1 };                                                                                      /* angles.cpp:111 */

/* one line of the 'bonds' section of field.txt (read_bond, bonds.cpp:125-364) with the 'con con' tail:
   harm k r0 | mors D a r0 C | pdn D a r0 C E | buck A ro C | e612 A ro C D F */
typedef struct
{
    int32_t spec_a, spec_b, type;
    double p[5];
} aztot_bond_type;

/* one line of the 'angles' section of field.txt (read_angle, angles.cpp:78-128): centralSpec hcos k cos0 */
typedef struct
{
    int32_t central, type;
    double k, cos0;
} aztot_angle_type;

/* type tables + the contents of bonds.txt (read_bondlist, bonds.cpp:25-110) and angles.txt (read_anglelist,
   angles.cpp:22-60).  Atom indices are 0-based; type ids are 1-based as in the files (id k = entry k-1 of the table). */
typedef struct
{
    int32_t n_bond_types, n_angle_types, n_bonds, n_angles;
    const aztot_bond_type *bond_types;
    const aztot_angle_type *angle_types;
    const int32_t *bond_a, *bond_b, *bond_type;
    const int32_t *angle_c, *angle_l1, *angle_l2, *angle_type;
} aztot_bonded;

/* run-time switches that replace the reference's compile-time defines.h (defines.h:5-23) and the kernel-launch file cuda.txt (cuInit.cu:701-749),
   and settle the differences between the reference's two mains.  Always start from aztot_default_options(): it fills struct_size, and a
   library that is handed a struct of another size refuses it (AZTOT_ERR_ARG) instead of reading fields that are not there.
   Measurement switches (A/B comparisons of kernel paths, phase timing) are NOT part of this struct: they are read from the environment variable
   AZTOT_DEBUG when a device handle is created (a bit mask; the bits are listed in aztotmd_amd/csrc/engine.h, enum DebugBit). */
typedef struct
{
    uint32_t struct_size;        /* sizeof(aztot_options) of the caller, set by aztot_default_options */
    int32_t device;              /* HIP device ordinal (reference: device 0 hard-coded, cuInit.cu:688) */
    int32_t initial_forces;      /* 1: forces of the initial configuration are computed at init (serial path,
                                       sys_init.cpp:1181-1184); 0: start from F = 0 (GPU path, sys_init.cpp:551-553) */
    int32_t center_box;          /* 1: apply center_box at init (serial path only, sys_init.cpp:1145) */
    uint64_t seed;               /* seed of the counter-based RNG (thermostat, tables, init_vel) */
    int32_t pair_variant;        /* 0: auto (= 2 where the geometry allows, else 1); 1: per-atom gather kernel; 2: LDS-tiled wave-per-cell kernels (+ pair lists) */
    double cell_size;            /* 0: derive from control.cell_list / cut-off (+ skin); >0: force this cell edge */
    int32_t use_graph;           /* 1: replay the step as a captured hipGraph when possible */
    int32_t profile;             /* 1: time every kernel with HIP events (aztot_kernel_times) */
    int32_t sort_every;          /* cell-list rebuild schedule.  0 (default): adaptive - the cells are rebuilt only when an atom could have used up its
                                    share of the skin (exact; the reference rebuilds every step, main.cu:300-326, and results then differ from an
                                    every-step run in summation order only); 1: every step, the reference's schedule; n > 1: adaptive, at most every n-th step */
    double skin;                 /* Verlet skin in Angstrom: the pair lists hold every pair within cut-off + skin, an atom may move skin / 2 between two rebuilds.
                                    0 (default): automatic (about 4 % of the cut-off; cells are sized cut-off + skin when control.cell_list asks for cells
                                    of about the cut-off); > 0: this value; < 0: no skin - cells exactly as control.cell_list / split_cells
                                    (cuCellList.cu:9-34) give them, the slack is whatever the cell edge happens to overhang the cut-off */
    int32_t waves_per_cell;      /* waves that share one cell in the pair kernels (list kernel: 1 / 2 / 4, they split the cell's atoms over one LDS tile;
                                    staging kernel: 1 / 2 / 4 / 8, they split the stencil's columns).  0 (default): the engine decides - several where
                                    cells are few, tiles large or stencils wide */
    int32_t energies_every_step; /* 1: pair energies are booked on every step (0, default: only on the last step of an aztot_step call, the only one whose
                                    statistics the caller can see; forces and trajectories are bit-identical either way) */
    int32_t loopback_ranks;      /* measurement aid for aztot_init_device_slab without a transport: 1 = this rank exchanges its halo with itself
                                    (one rank of N on one GPU; never a result) */
    int32_t reserved[5];         /* must be zero */
} aztot_options;

/* per-step scalars: the fields tracked by stat.dat (cuStat.cu:241-261) + serial calc_chars (integrators.cpp:63-73) */
typedef struct
{
    int64_t step;                /* number of completed steps */
    double time;                 /* ps */
    double engTot, engKin, engVdW, engCoul, engElecField, engTemp, engPot;
    double temperature;          /* 2 engKin / (degFree kB) */
    double posMom[3], negMom[3]; /* wall momentum accumulators (box.cpp:230-295; cuMDfunc.cu:72-106) */
    int64_t posCross[3], negCross[3];
    double pressure;             /* from wall momentum over the last 'stat' window (main.cpp:146-152) */
    int64_t pairs_dropped;       /* pairs skipped by the |f|^2 > 1e10 rule (integrators.cpp:170) */
    int64_t n_cells;
    double nose_chit, nose_conint;  /* Nose-Hoover friction and conserved-quantity integral (temperature.h:24-25) */
    double engBond, engAngle;    /* exec_bondlist bonds.cpp:1218, exec_anglelist angles.cpp:240; both are part of engTot */
    double engCoulRec, engCoulConst; /* Ewald sum: reciprocal part (engElec2, elec.cpp:333 ; cudaMD::engCoul2) and constant part
                                        (engElec1, ewald_const elec.cpp:144 ; engCoul3); engCoul above is the real-space part */
    int64_t sort_interval;       /* steps between two rebuilds of the cell list that the next aztot_step call will use (1: every step) */
    int64_t sort_violations;     /* calls so far in which an atom left its cell's slack before the scheduled rebuild (handled exactly: by a wider stencil on one GPU, by running the steps since the last look again - from a device-side snapshot, cells rebuilt every step - on slab ranks and on small systems) */
    int64_t pair_lists;          /* 1: the steps between two rebuilds walk per-atom pair lists recorded at the rebuild (k_pair_list) ; 0: they stage every cell */
    int64_t cells_without_list;  /* cells that did not fit the lists at the last rebuild (more than 64 atoms, tile or list full): staged in full every step */
    int64_t rebuilds;            /* steps so far that rebuilt the cell list (clear_clist .. sort_atoms of main.cu:300-326; the reference: every step) */
    double skin;                 /* the Verlet skin in force, Angstrom: pair lists reach cut-off + skin, an atom may move skin / 2 between two rebuilds (0: none) */
} aztot_stats;

/* host copy of the per-atom state, fp64 SoA, in ORIGINAL atom order (id order); any pointer may be NULL */
typedef struct
{
    int32_t n_atoms;
    double *x, *y, *z, *vx, *vy, *vz, *fx, *fy, *fz, *U, *radius;
    int32_t *types;
} aztot_state;

/* ---- host model: replaces init_md / free_md (sys_init.h:11,17) ------------------------------------- */
/* reads atoms.xyz, field.txt, control.txt, cuda.txt from `dir` (reference: from the cwd) */
int aztot_init_md(const char *dir, aztot_model **out);
int aztot_model_create(const aztot_system *sys, aztot_model **out);
/* attach bonded terms to a model made by aztot_model_create (aztot_init_md reads them from field.txt + bonds.txt +
   angles.txt itself); replaces sys_init.cpp:289-314,411-427,626-673.  Must precede aztot_init_device. */
int aztot_model_set_bonded(aztot_model *m, const aztot_bonded *b);
/* string-keyed read-out of parsed/derived values as doubles; returns the number of values written
   (or needed, if cap is too small), negative on unknown key.  Keys: see aztotmd_amd/csrc/capi.cpp */
int aztot_model_query(const aztot_model *m, const char *key, double *out, int cap);
/* name of species i as written in field.txt (NUL-terminated, at most cap-1 characters) */
int aztot_model_species_name(const aztot_model *m, int i, char *buf, int cap);
void aztot_free_md(aztot_model *m);

/* ---- device: replaces init_cudaMD / md_to_host / free_device_md (cuInit.h:4,6,7) --------------------- */
/* number of HIP devices this process can use (0: none, never negative); the reference takes device 0 unasked (cuInit.cu:688) */
int aztot_device_count(void);
/* blocks until device `device` has finished all work queued by this process (hipDeviceSynchronize): what a launcher brackets a timed region with */
int aztot_device_synchronize(int device);
void aztot_default_options(aztot_options *opt);
int aztot_init_device(const aztot_model *m, const aztot_options *opt, aztot_md **out);
void aztot_free_device(aztot_md *md);

/* ---- the hot path ----------------------------------------------------------------------------------- */
/* n iterations of the loop body of main.cu:281-410 (reset_quantities, verlet_1stage, iter_fastCellList,
   verlet_2stage, apply_tstat, calc_quantities) with the serial path's fp64 arithmetic */
int aztot_step(aztot_md *md, int nsteps);
/* aztot_step on ONE GPU may return with the kernels of its steps queued and the end of the call - the last step's second half-kick where it is folded into
   the next step, the reduction of the statistics, the look at the cell-list rebuild schedule - still to come: every entry point that reads or writes state
   (aztot_get_stats, aztot_md_to_host, aztot_get_clock, aztot_set_state, aztot_forces, ...) completes it first, so callers see no difference, and a caller
   that steps one step at a time (the reference's loop is per step, main.cu:281-410, with a cudaThreadSynchronize behind every kernel) does not pay for a
   synchronisation per step.  aztot_sync completes everything explicitly and returns when the device has finished: what a timed region ends with.  An error
   detected only then is reported by the call that detects it.  After any error from aztot_step / aztot_sync the handle no longer steps (the first error is
   repeated); reads still work for a post-mortem.  Slab ranks and runs with bonded terms end every call synchronously. */
int aztot_sync(aztot_md *md);
/* cell-list build + sort + pair forces for the current positions: iter_fastCellList (cuPairs.h:8) alone */
int aztot_forces(aztot_md *md);
int aztot_get_stats(aztot_md *md, aztot_stats *out);
/* per-species wall crossings since init: out[6 * s + {0..5}] = species s through the walls Xn, Xp, Yn, Yp, Zn, Zp - the
   specAcBoxNeg / specAcBoxPos counters of put_periodic (cuMDfunc.cu:35-106) that the reference writes to msd.dat
   (cuStat.cu:278-288,345-350).  cap = number of int64 slots in out (>= 6 * n_species).  Collective on several ranks. */
int aztot_species_crossings(aztot_md *md, int64_t *out, int cap);
int aztot_md_to_host(aztot_md *md, aztot_state *out);
int aztot_set_state(aztot_md *md, const aztot_state *in);
/* Restart support (checkpoint row): the scalars of the device state that aztot_md_to_host / aztot_set_state do not carry.  A run restarted with
   aztot_set_state (x, v, f, U, radius of the checkpoint) + aztot_set_clock continues exactly: the step number drives the equilibration schedule
   (sys_init.cpp:700-712, nequil / eqfreq) and keys the thermostat's counter-based random numbers; the Nose-Hoover pair (temperature.h:24-25) and the
   kinetic energy the thermostats saw last (sim->engKin, integrators.cpp:305) are the thermostat's memory.  Wall-momentum and crossing counters restart
   from zero (they are running sums for the statistics, never fed back into the dynamics). */
typedef struct
{
    int64_t step;                /* completed steps (aztot_stats.step) */
    double nose_chit, nose_conint;
    double eng_kin;              /* kinetic energy after the last completed step, eV */
} aztot_clock;
int aztot_get_clock(aztot_md *md, aztot_clock *out);
int aztot_set_clock(aztot_md *md, const aztot_clock *in);
/* the cell list as the device holds it after the last sort: cudaMD::firstAtomInCell / cellIndexes (cuStruct.h:219-222;
   calc_firstAtomInCell cuSort.cu:130-143, sort_atoms cuSort.cu:145-197).  dims[3] = cells per axis of this rank's window;
   cell_start[c] = first slot of cell c = (ix * ny + iy) * nz + iz (n_cells + 1 entries, the last one = resident atoms);
   atom_id[s] = original index of the atom in slot s.  Either array may be NULL; caps are entry counts.  Returns the number of
   cells (or a negative error); never part of a step - a read-back for tests and restarts. */
int aztot_cell_table(aztot_md *md, int32_t dims[3], int32_t *cell_start, int cap_cells, int32_t *atom_id, int cap_atoms);

/* ---- measurement ------------------------------------------------------------------------------------ */
/* per-kernel HIP-event times accumulated since the last reset (options.profile = 1).
   names: NUL-separated list written into `names` (cap bytes); ms / calls: arrays of length >= returned count */
int aztot_kernel_times(aztot_md *md, char *names, int cap, double *ms, int64_t *calls, int max_kernels);
int aztot_reset_kernel_times(aztot_md *md);
/* switch per-kernel HIP-event timing on/off at run time (off: the step may be replayed as a hipGraph) */
int aztot_set_profile(aztot_md *md, int on);

/* ---- multi-GPU slab decomposition (one process per GPU) ---------------------------------------------- */
/* size of the opaque RCCL unique id; rank 0 creates it, the launcher broadcasts it (e.g. torch.distributed) */
int aztot_comm_id_bytes(void);
int aztot_comm_make_id(void *id_bytes);
/* diagnostic: brings up a ONE-rank RCCL communicator on `device` and runs the slab ring exchange with itself (both messages in
   the N-GPU call order) plus both all-reduce flavours; AZTOT_OK if RCCL is usable on this node and delivers what was sent */
int aztot_comm_selftest(int device);
/* number of ranks in the RCCL communicator that carries this handle's halo exchange (ncclCommCount); 0 when the handle does not
   use RCCL (single GPU, host-staged callbacks, loopback).  Lets a launcher prove which transport a run really used. */
int aztot_comm_ranks(aztot_md *md);
/* host-staged exchange callback for tests without RCCL (gloo): send `sbytes` to `peer`, receive into rbuf */
typedef int (*aztot_sendrecv_fn)(void *ctx, int send_peer, const void *sbuf, int64_t sbytes,
                                 int recv_peer, void *rbuf, int64_t rcap, int64_t *rbytes);
typedef int (*aztot_allreduce_fn)(void *ctx, double *buf, int n);
/* as aztot_init_device, but this rank owns slab `rank` of `nranks` along x; exactly one of id_bytes / callbacks is used */
int aztot_init_device_slab(const aztot_model *m, const aztot_options *opt, int rank, int nranks,
                           const void *rccl_id_bytes, aztot_sendrecv_fn sendrecv, aztot_allreduce_fn allreduce, void *ctx,
                           aztot_md **out);

const char *aztot_last_error(void);
const char *aztot_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AZTOT_H */
