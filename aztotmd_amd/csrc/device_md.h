// Device-side state of one rank: the MI355X-native counterpart of the reference's cudaMD
// (cuStruct.h:188-423).  cudaMD is kept as the *field inventory*; the byte layout is ours:
// fp64 structure-of-arrays (the serial path's precision, dataStruct.h:309-311) so that every
// per-atom stream is a unit-stride 8-byte access, double-buffered for the per-step counting sort.
#pragma once
#include <cstdint>

namespace aztot {

constexpr int kSpecCap = 16;      // >= MX_SPEC (defines.h:14)

// per-atom arrays (one "buffer"; two of them ping-pong through the sort)
struct AtomArrays
{
    double *x, *y, *z;            // cudaMD::xyz   (cuStruct.h:195)
    double *vx, *vy, *vz;         // cudaMD::vls   (cuStruct.h:196)
    double *fx, *fy, *fz;         // cudaMD::frs   (cuStruct.h:197)
    double *U, *rad;              // cudaMD::engs, radii (cuStruct.h:384-385)
    int32_t *type;                // cudaMD::types (cuStruct.h:199)
    int32_t *id;                  // persistent atom id (ours: stable key for RNG / output order / determinism)
};

// reference positions of the lazy re-sort: where every atom was when the cells were last rebuilt (sorted order); x == nullptr: not in use
struct RefPos { double *x, *y, *z; };

// species table, small enough to travel in the kernel-argument segment (scalar loads, no pointer chasing)
struct SpecTable                  // cudaSpec, cuStruct.h:10-47
{
    double mass[kSpecCap], charge[kSpecCap], rMhdt[kSpecCap];
    double radA[kSpecCap], radB[kSpecCap], mxEng[kSpecCap];
    int32_t charged[kSpecCap], frozen[kSpecCap];
};

struct DevPot                     // cudaVdW, cuStruct.h:50-63, without the device function pointers
{
    int32_t type, use_radii;
    double p0, p1, p2, p3, p4, r2cut;
};

// bonded terms ("next" row f2).  The reference keeps bond / angle lists of atom *indices* and rewrites them after every
// sort (cuSort.cu:199-236); here the lists are static per-atom CSR tables keyed by the persistent atom id, and the sort
// publishes one id -> current-index map (idxOfId) instead.
struct DevBondType { int32_t type, pad; double p0, p1, p2, p3, p4; };     // cudaBond, cuStruct.h:66-78 (constant bonds)
struct DevAngleType { double k, cos0; };                                 // cudaAngle: harmonic cosine only
struct BondEntry { int32_t partner; int32_t typeFirst; };                // partner id ; type id | (this atom is at1) << 30
struct AngleEntry { int32_t roleType, c, l1, l2; };                      // role (0 central, 1 lig1, 2 lig2) | type id << 2 ; atom ids
struct BondedTables
{
    const int32_t* bondStart;        // [nAtGlobal + 1]
    const BondEntry* bondEnt;
    const int32_t* angStart;         // [nAtGlobal + 1]
    const AngleEntry* angEnt;
    const DevBondType* btypes;       // [nBondTypes + 1], [0] unused (the reference's reserved 'none')
    const DevAngleType* atypes;
    int32_t* idxOfId;                // [nAtGlobal]: index in the sorted arrays, verified against A.id on use
};

// reciprocal-space Ewald sum ("next" row f4): k-vector table in the order ewald_rec visits it (elec.cpp:229-330)
enum { EWK_NEW_L = 1, EWK_NEW_LM = 2 };
struct EwaldK
{
    int32_t l, m, n, flags;          // flags: first k-vector with this l / with this (l, m)
    double rkx, rky, rkz, akk;       // rk[] and exprk2[] of cuInit.cu:1017-1046
};
struct EwaldW                        // work item of the structure-factor kernel: k-vectors (l, m, +n) and (l, m, -n)
{
    int32_t l, m, n;                 // n >= 0
    int32_t kPlus, kMinus;           // indices into the k-vector table, -1 if that one is not in the sum
};
struct EwaldG                        // work item of the force kernel: the run of k-vectors (l, m, nLo..nHi), contiguous in the table
{
    int32_t l, m, nLo, nHi, kStart;
};
struct EwaldTables
{
    const EwaldK* kv;
    const EwaldW* work;
    const EwaldG* groups;
    double* T;                       // [nK][2]: scale2 * akk(k) * S(k), refreshed with S
    int32_t nK, nW, nG, kx, ky, kz, nBlocksA;
    double* partial;                 // [nBlocksA][nK][2]: per-block partial structure factors
    double* S;                       // [nK][2]: (sum q cos kr, sum q sin kr) over ALL atoms
    double scale, scale2;            // elec.cpp:380-381
};

// Few cells, wide stencils (case study 2: 4 000 atoms on 2.7 A cells, 7^3 = 343 stencil cells, five tiles' worth of candidates per cell): one wave per
// cell leaves most of the chip idle and makes every wave long.  The stencil's (x, y) columns are then dealt to `n` waves per cell (a power of two); every
// wave writes its partial forces to scratch, and the last one to arrive (one atomic per wave) adds them up IN FIXED ORDER and finishes the atoms - no
// floating-point atomics, bit-reproducible.
struct SplitArgs
{
    int n = 1;                                   // waves per cell
    double *fx = nullptr, *fy = nullptr, *fz = nullptr;   // [n][capacity]: partial forces by sub-wave and atom
    int32_t* arrived = nullptr;                  // [nCell]: sub-waves that have delivered (left at zero by the last one)
    int capacity = 0;
};

// uniform parameters of the step (kernel argument, by value)
struct StepParams
{
    int32_t nOwned;               // atoms this rank integrates
    int32_t nTotal;               // owned + ghost atoms resident in the arrays (== nOwned on one GPU)
    int32_t nSpec;
    int32_t single_lj;            // fast path: one species, LJ only, no Coulomb
    double L[3], invL[3], half[3];
    double dt;
    // cell grid (global grid; a slab rank stores cell layers [cx0, cx0 + ncxLocal) incl. ghost layers)
    int32_t nc[3];                // global cell counts (split_cells, cuCellList.cu:9-34)
    int32_t ncxLocal, cx0;        // local x-layers and global index of local layer 0 (may be negative: periodic ghost)
    int32_t nCellLocal;
    int32_t hw[3];                // stencil half-widths: ceil(rMax / cell edge)
    int32_t nOff[3];              // offsets visited per axis = min(2 hw + 1, nc)
    double csz[3], icsz[3];
    double r2Max;
    // electrostatics (cuStruct.h:388-391 ; elec.cpp:399-405)
    int32_t elec_type, use_radii;
    double alpha, el_scale, el_scale2, daipi2, rReal, fcoul, sqrtpi;
    double E[3];                  // external field gradient
    // thermostat
    int32_t tstat, nEq, freqEq, pad0;
    double tKin, revDegFree, rkB;
    double rQmass, qMassTau2;     // Nose-Hoover: 1/(2 tKin tau^2), 2 tKin (sys_init.cpp:1107-1112)
    double revLight, radFrac, radThr, numPi;
    uint64_t seed;
    int32_t nAtGlobal, pad1;
    // slab decomposition
    double xlo, xhi;              // owned x-range [xlo, xhi)
    int32_t rank, nranks;
    int32_t fuseKick;             // 1: the pair kernel also applies the second half-kick and books the kinetic energy (plain NVE steps)
    int32_t vdwFamily;            // pad1 == 2: the one potential type all defined species pairs share (1 lnjs, 2 buck, 3 p746, 4 bmhs)
    int32_t cycleStep;            // lazy re-sort: which step since the last rebuild of the cells this launch belongs to (0: the rebuilding step itself)
    int32_t pad2;                 // 1: plain steps may skip the per-atom displacement check while the displacement bound allows (Counts::cycMaxRun)
    double lazySlack2;            // lazy re-sort: square of the displacement an atom may have since the last sort ((hw * cell edge - rc) / 2 per axis, minimum);
                                  // 0: the cells are rebuilt every step
    double pruneR2;               // tile kernels: atoms farther than this (squared) from the centre cell's box are not staged: (rc + 2 slack)^2
    double ljDropR2;              // single_lj: r^2 beyond which |f| <= 1e5 is certain, so the 'pair dropped' rule (integrators.cpp:170) need not be evaluated
    int32_t optimistic, pad3;     // 1: no clean-up launch follows k_pair_list (small systems on one GPU, Engine::step): it walks its lists whatever the violation flag
                                  //    says; a look that finds a violation - or a cell without a list - goes back to the snapshot and runs the window again with the launch
};

// reduction slots of the per-block partial buffer (deterministic two-stage sums)
enum PartialSlot
{
    PS_EFIELD = 0, PS_MOM_XN, PS_MOM_XP, PS_MOM_YN, PS_MOM_YP, PS_MOM_ZN, PS_MOM_ZP,
    PS_CNT_XN, PS_CNT_XP, PS_CNT_YN, PS_CNT_YP, PS_CNT_ZN, PS_CNT_ZP,
    PS_EVDW, PS_ECOUL, PS_DROPPED, PS_EKIN, PS_ETEMP, PS_EBOND, PS_EANGLE, PS_COUNT
};

// device-resident scalars (the energy / momentum block of cudaMD, cuStruct.h:230-247)
struct DevStats
{
    long long step;
    long long stepAtSort;         // number of the last step that rebuilt the cells (k_integrate1_bin<1>): lets a launch derive the number of the step it
                                  // belongs to from StepParams::cycleStep without reading a counter another thread of the same launch advances
    double engKin, engVdW, engCoul, engElecField, engTemp, engTot, engPot, temperature;
    double engBond, engAngle;     // exec_bondlist bonds.cpp:1218 / exec_anglelist angles.cpp:240
    double engCoulRec, engCoulConst;   // Ewald sum: reciprocal (engElec2) and constant (engElec1) parts; the same on every rank
    double mom[6];                // Xn, Xp, Yn, Yp, Zn, Zp accumulated over the run (box.cpp:230-295)
    long long cross[6];
    long long dropped;
    double vscale;                // velocity scale decided for the END of the current step: equilibration x Nose-Hoover (1 = none)
    double vscaleBegin;           // Nose-Hoover scale applied at the BEGINNING of the step (integrate1, integrators.cpp:305)
    double ekSim;                 // the reference's sim->engKin as the thermostats see it (value left by the previous integrate2)
    double chit, conint;          // Nose-Hoover friction and conserved-quantity integral (temperature.h:24-25)
    long long pendingKick;        // 1: the second half-kick of the previous step is still owed (applied by the next k_integrate1_bin,
                                  //    or by k_integrate2 before anybody can look at the velocities)
    double local[PS_COUNT];       // this rank's per-step sums before the cross-rank reduction
    unsigned long long specCross[kSpecCap * 6];   // per species: crossings of the walls Xn, Xp, Yn, Yp, Zn, Zp (specAcBoxNeg/Pos, cuStruct.h)
};

}  // namespace aztot
