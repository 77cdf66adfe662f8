// Host-side model of one azTotMD system (fp64): what the reference keeps in
// Atoms / Field / Sim / Elec / TStat / Box (dataStruct.h:40-416, temperature.h:15) after init_md
// (sys_init.cpp:1036-1119), restricted to the fields the per-step hot path reads.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/aztot.h"

namespace aztot {

// ---- units: const.h:11-49 -------------------------------------------------------------------------
namespace units {
constexpr double pi = 3.14159265359;                       // const.h:11 (the reference's truncated value)
constexpr double r_SI = 1.0E-10, t_SI = 1.0E-12, E_SI = 1.60217733E-19, q_SI = 1.60217657E-19;
constexpr double kB_SI = 1.3806488E-23, e0_SI = 8.854187817E-12, amu_SI = 1.6605402E-27;
constexpr double m_SI = E_SI * t_SI * t_SI / r_SI / r_SI;  // const.h:27
constexpr double F_SI = E_SI / r_SI;                       // const.h:28
constexpr double Fcoul_SI = 0.25 / pi / e0_SI * q_SI * q_SI / r_SI / r_SI;   // const.h:29
constexpr double m_scale = amu_SI / m_SI;                  // const.h:43
constexpr double Fcoul_scale = Fcoul_SI / F_SI;            // const.h:44
constexpr double kB = kB_SI / (1.0 * E_SI);                // const.h:47
constexpr double pressure_factor = 1.58e6;                 // main.cpp:147 / main.cu:135
}  // namespace units

constexpr int kMaxSpecies = 15;                            // defines.h:14 MX_SPEC
constexpr int kNumUnitVectors = 3072;                      // cuTemp.h:4 nUvect
constexpr int kEwaldKMax = 48;                             // k-vectors per axis: the Ewald kernels keep kx + ky + kz harmonics x 1 KiB in LDS (<= 160 KiB, checked in upload_ewald; reference: NKVEC_MX 100)

struct Species                                             // Spec, dataStruct.h:244-291
{
    std::string name, nucleus;
    double mass = 0;        // internal units (amu * m_scale)
    double mass_amu = 0;
    double charge = 0;
    double energy = 0;
    int charged = 0, frozen = 0;
    double rMass_hdt = 0;   // 0.5 dt / m  (sys_init.cpp:1056)
    double radA = 0, radB = 0, mxEng = 0;
    int number = 0;
};

struct PairPot                                             // VdW, dataStruct.h:293-303, after read_vdw's preparation
{
    int type = 0;           // 0 = no potential for this species pair (NULL in the reference)
    int use_radii = 0;
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
    double r2cut = 0;
    double rcut = 0;
};

struct BondType                                            // Bond, dataStruct.h:321-339 ('con con' bonds only)
{
    int type = 0;           // 1 harm, 2 mors, 3 pdn, 4 buck, 5 e612 (bonds.cpp:158-252)
    int spec1 = 0, spec2 = 0;
    double p[5] = {0, 0, 0, 0, 0};
};

struct AngleType                                           // Angle, dataStruct.h:341-346
{
    int type = 0;           // 1 hcos
    int central = 0;
    double k = 0, cos0 = 0;
};

struct KVec                                                // one k-vector of the Ewald sum: rk[] / exprk2[] of cuInit.cu:1017-1046
{
    int l, m, n;
    double rkx, rky, rkz;
    double akk;             // exp(-rk^2 / 4 alpha^2) / rk^2  (elec.cpp:310)
};

struct Model
{
    // Atoms (dataStruct.h:305-318)
    int nAt = 0;
    std::vector<int32_t> types;
    std::vector<double> x, y, z, vx, vy, vz;
    // Box (dataStruct.h:205-241), rectangular only (box type 1)
    double L[3] = {0, 0, 0};
    // Field
    std::vector<Species> species;
    std::vector<PairPot> pairpots;     // [nSpec * nSpec]
    int nVdW = 0;
    double minRvdw = 999999.9, maxRvdw = 0.0;
    int charged_spec = 0;
    int has_radii = 0;
    // bonded terms (Field, dataStruct.h:389-410).  Type ids are the reference's: 1-based, [0] is its reserved 'none'
    // (sys_init.cpp:293-295,414-420), so bondTypes[k] below is id k + 1.
    std::vector<BondType> bondTypes;
    std::vector<AngleType> angleTypes;
    std::vector<int32_t> bondA, bondB, bondT;         // at1 (carries the type's spec1 after read_bondlist's turn), at2, type id
    std::vector<int32_t> angC, angL1, angL2, angT;    // central, ligands, type id
    // Sim / control.txt
    double tSt = 0;
    int nSt = 0, nEq = 0, freqEq = 0;
    double tSim = 0;
    int init_vel = AZTOT_VEL_ZERO;
    double init_vel_par[3] = {0, 0, 0};
    double E[3] = {0, 0, 0};           // elecfield Ux Uy Uz
    int use_clist = 0;
    double desired_cell_size = 0;
    int stat = 1000;
    int max_neigh = 50;
    // Elec (dataStruct.h:349-366)
    int elec_type = AZTOT_ELEC_NONE;
    double rReal = 0, r2Real = 0, alpha = 0, eps = 1.0;
    double el_scale = 0, el_scale2 = 0, daipi2 = 0;
    int ewald_k[3] = {0, 0, 0};        // 'elec pme rReal alpha kx ky kz': exclusive upper bounds of |k| per axis (elec.cpp:36)
    double mr4a2 = 0, rkcut2 = 0;      // prepare_elec elec.cpp:382-394
    double engElec1 = 0;               // constant part of the Ewald sum (ewald_const elec.cpp:144-164)
    std::vector<KVec> kvecs;
    // TStat (temperature.h:15)
    int tstat_type = AZTOT_TSTAT_NONE;
    double Temp = 0, tau = 0, tKin = 0;
    int tstat_step = 0;
    // derived (sys_init.cpp:1053-1112)
    double rMax = 0, r2Max = 0;
    int degFree = 0;
    double revDegFree = 0;
    // cuda.txt (cuInit.cu:684-754): only the two block sizes touch the hot path in the reference
    int nthread_a = 16, nthread_b = 32, nstep_stat = 10;
    std::vector<std::string> warnings;

    int nSpec() const { return (int)species.size(); }
    const PairPot& pot(int a, int b) const { return pairpots[(size_t)a * species.size() + b]; }
};

// init_md: read field.txt, atoms.xyz, control.txt (+ cuda.txt) from `dir`; throws std::runtime_error
// with the reference's ERROR[..] code in the message.
void init_md(const std::string& dir, Model& m);
// the same state from arrays
void model_from_system(const aztot_system& sys, Model& m);
// prepare_elec + derived parameters + initial velocities (init_md tail, sys_init.cpp:1047-1119)
void finish_model(Model& m, uint64_t seed);
// bonded lists (read_bondlist bonds.cpp:25-110, read_anglelist angles.cpp:22-60): validates against the type tables
// and the atoms' species, turns bonds so that the first atom carries spec1.  Replaces any previous lists.
void set_bond_list(Model& m, int n, const int32_t* a, const int32_t* b, const int32_t* t);
void set_angle_list(Model& m, int n, const int32_t* c, const int32_t* l1, const int32_t* l2, const int32_t* t);
void add_bond_type(Model& m, int spec1, int spec2, int type, const double p[5]);
void add_angle_type(Model& m, int central, int type, double k, double cos0);
// raw user vdw line -> prepared potential (read_vdw, vdw.cpp:261-299)
PairPot prepare_vdw(int type, double rcut, const double p[5]);
void center_box(Model& m);   // box.cpp:337-384

// tables of the radiative thermostat (temperature.cpp:28-89, 165-223) with the counter-based RNG
void photon_engs(int n, double* engs, double T, uint64_t seed);
void unit_vectors(double* ux, double* uy, double* uz);
double initial_radius(uint64_t seed, uint64_t id);

}  // namespace aztot
