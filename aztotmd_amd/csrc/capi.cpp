// C ABI (include/aztot.h) over the C++ host side.  Every entry point catches exceptions and turns them
// into an error code + thread-local message (the reference prints "ERROR[..]" and returns 0).
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "engine.h"
#include "exchange.h"
#include "model.h"

using namespace aztot;

struct aztot_model { Model m; bool finished = false; };
struct aztot_md
{
    std::unique_ptr<Exchanger> xch;
    std::unique_ptr<Engine> eng;
};

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

int classify(const std::string& msg)
{
    if (msg.find("out of scope") != std::string::npos) return AZTOT_ERR_INPUT;
    if (msg.find("can't open") != std::string::npos) return AZTOT_ERR_IO;
    if (msg.find("HIP") != std::string::npos) return AZTOT_ERR_DEVICE;
    if (msg.find("RCCL") != std::string::npos || msg.find("slab") != std::string::npos) return AZTOT_ERR_COMM;
    return AZTOT_ERR_INPUT;
}

template <typename F>
int guarded(F&& f)
{
    try { f(); g_err.clear(); return AZTOT_OK; }
    catch (const std::exception& e) { return fail(classify(e.what()), e.what()); }
    catch (...) { return fail(AZTOT_ERR_ARG, "unknown exception"); }
}
}  // namespace

extern "C" {

const char* aztot_last_error(void) { return g_err.c_str(); }
#ifndef AZTOT_SRC_HASH
#define AZTOT_SRC_HASH "unknown"
#endif
// "... src <16 hex digits>": a digest of every source file the library was compiled from (csrc/Makefile)
const char* aztot_version(void) { return "aztotmd_amd 0.4 (gfx950, fp64) src " AZTOT_SRC_HASH; }

int aztot_init_md(const char* dir, aztot_model** out)
{
    if (!dir || !out) return fail(AZTOT_ERR_ARG, "null argument");
    *out = nullptr;
    return guarded([&] {
        auto h = std::make_unique<aztot_model>();
        init_md(dir, h->m);
        *out = h.release();
    });
}

int aztot_model_create(const aztot_system* sys, aztot_model** out)
{
    if (!sys || !out) return fail(AZTOT_ERR_ARG, "null argument");
    *out = nullptr;
    return guarded([&] {
        auto h = std::make_unique<aztot_model>();
        model_from_system(*sys, h->m);
        *out = h.release();
    });
}

int aztot_model_set_bonded(aztot_model* h, const aztot_bonded* b)
{
    if (!h || !b) return fail(AZTOT_ERR_ARG, "null argument");
    if (b->n_bond_types < 0 || b->n_angle_types < 0 || b->n_bonds < 0 || b->n_angles < 0) return fail(AZTOT_ERR_ARG, "negative count in aztot_bonded");
    if ((b->n_bond_types && !b->bond_types) || (b->n_angle_types && !b->angle_types) || (b->n_bonds && !(b->bond_a && b->bond_b && b->bond_type)) ||
        (b->n_angles && !(b->angle_c && b->angle_l1 && b->angle_l2 && b->angle_type)))
        return fail(AZTOT_ERR_ARG, "null array in aztot_bonded");
    return guarded([&] {
        Model m = h->m;                 // all-or-nothing: the model is replaced only if every line is accepted
        m.bondTypes.clear(); m.angleTypes.clear();
        for (int i = 0; i < b->n_bond_types; i++) add_bond_type(m, b->bond_types[i].spec_a, b->bond_types[i].spec_b, b->bond_types[i].type, b->bond_types[i].p);
        for (int i = 0; i < b->n_angle_types; i++) add_angle_type(m, b->angle_types[i].central, b->angle_types[i].type, b->angle_types[i].k, b->angle_types[i].cos0);
        set_bond_list(m, b->n_bonds, b->bond_a, b->bond_b, b->bond_type);
        set_angle_list(m, b->n_angles, b->angle_c, b->angle_l1, b->angle_l2, b->angle_type);
        h->m = std::move(m);
    });
}

void aztot_free_md(aztot_model* m) { delete m; }

int aztot_model_species_name(const aztot_model* m, int i, char* buf, int cap)
{
    if (!m || !buf || cap <= 0 || i < 0 || i >= m->m.nSpec()) return fail(AZTOT_ERR_ARG, "bad argument");
    std::snprintf(buf, (size_t)cap, "%s", m->m.species[i].name.c_str());
    return AZTOT_OK;
}

// Keys (all values returned as doubles):
//  n_atoms n_species box dt nstep nequil eqfreq temperature tstat_type elec_type r_real alpha scale scale2 daipi2
//  rmax r2max degfree tkin cell_list use_cell_list stat init_vel elecfield nthread kB m_scale fcoul
//  species (per species: mass_amu, mass, charge, charged, frozen, rMass_hdt, radA, radB, mxEng, number)
//  vdw (per ordered species pair a*nSpec+b: type, r2cut, p0..p4, use_radii)
//  ewald (kx, ky, kz, mr4a2, rkcut2, engElec1, number of k-vectors)  kvecs (l, m, n, rkx, rky, rkz, akk per k-vector)
//  n_bonded (bond types, angle types, bonds, angles)  bond_types (type, spec1, spec2, p0..p4)  angle_types (type, central, k, cos0)
//  bonds (at1, at2, type id per bond, after the turn of read_bondlist)  angles (central, lig1, lig2, type id)
//  types x y z vx vy vz   (per atom)    photons uvx uvy uvz (radiative thermostat tables; seed = first element of `out` on entry for photons)
int aztot_model_query(const aztot_model* h, const char* key, double* out, int cap)
{
    if (!h || !key) return fail(AZTOT_ERR_ARG, "null argument");
    Model m = h->m;                       // derived values are computed on a copy: querying never mutates the model
    int rc = AZTOT_ERR_ARG;
    const uint64_t seed_in = (out && cap > 0 && out[0] >= 0 && out[0] < 1.8e19) ? (uint64_t)out[0] : 12345;
    int status = guarded([&] {
        finish_model(m, 12345);
        std::vector<double> v;
        const std::string k = key;
        auto per_atom = [&](const std::vector<double>& a) { v = a; };
        if (k == "n_atoms") v = {(double)m.nAt};
        else if (k == "n_species") v = {(double)m.nSpec()};
        else if (k == "box") v = {m.L[0], m.L[1], m.L[2]};
        else if (k == "dt") v = {m.tSt};
        else if (k == "nstep") v = {(double)m.nSt};
        else if (k == "nequil") v = {(double)m.nEq};
        else if (k == "eqfreq") v = {(double)m.freqEq};
        else if (k == "temperature") v = {m.Temp};
        else if (k == "tstat_type") v = {(double)m.tstat_type};
        else if (k == "elec_type") v = {(double)m.elec_type};
        else if (k == "r_real") v = {m.rReal};
        else if (k == "alpha") v = {m.alpha};
        else if (k == "scale") v = {m.el_scale};
        else if (k == "scale2") v = {m.el_scale2};
        else if (k == "daipi2") v = {m.daipi2};
        else if (k == "rmax") v = {m.rMax};
        else if (k == "r2max") v = {m.r2Max};
        else if (k == "degfree") v = {(double)m.degFree};
        else if (k == "tkin") v = {m.tKin};
        else if (k == "cell_list") v = {m.desired_cell_size};
        else if (k == "use_cell_list") v = {(double)m.use_clist};
        else if (k == "stat") v = {(double)m.stat};
        else if (k == "init_vel") v = {(double)m.init_vel, m.init_vel_par[0], m.init_vel_par[1], m.init_vel_par[2]};
        else if (k == "elecfield") v = {m.E[0], m.E[1], m.E[2]};
        else if (k == "nthread") v = {(double)m.nthread_a, (double)m.nthread_b, (double)m.nstep_stat};
        else if (k == "kB") v = {units::kB};
        else if (k == "m_scale") v = {units::m_scale};
        else if (k == "fcoul") v = {units::Fcoul_scale};
        else if (k == "n_warnings") v = {(double)m.warnings.size()};
        else if (k == "species")
            for (const auto& s : m.species)
            { double r[] = {s.mass_amu, s.mass, s.charge, (double)s.charged, (double)s.frozen, s.rMass_hdt, s.radA, s.radB, s.mxEng, (double)s.number}; v.insert(v.end(), r, r + 10); }
        else if (k == "vdw")
            for (const auto& p : m.pairpots)
            { double r[] = {(double)p.type, p.r2cut, p.p0, p.p1, p.p2, p.p3, p.p4, (double)p.use_radii}; v.insert(v.end(), r, r + 8); }
        else if (k == "ewald") { v = {(double)m.ewald_k[0], (double)m.ewald_k[1], (double)m.ewald_k[2], m.mr4a2, m.rkcut2, m.engElec1, (double)m.kvecs.size()}; }
        else if (k == "kvecs")
            for (const auto& q : m.kvecs) { double r[] = {(double)q.l, (double)q.m, (double)q.n, q.rkx, q.rky, q.rkz, q.akk}; v.insert(v.end(), r, r + 7); }
        else if (k == "n_bonded") v = {(double)m.bondTypes.size(), (double)m.angleTypes.size(), (double)m.bondA.size(), (double)m.angC.size()};
        else if (k == "bond_types")
            for (const auto& b : m.bondTypes)
            { double r[] = {(double)b.type, (double)b.spec1, (double)b.spec2, b.p[0], b.p[1], b.p[2], b.p[3], b.p[4]}; v.insert(v.end(), r, r + 8); }
        else if (k == "angle_types")
            for (const auto& a : m.angleTypes) { double r[] = {(double)a.type, (double)a.central, a.k, a.cos0}; v.insert(v.end(), r, r + 4); }
        else if (k == "bonds")
            for (size_t i = 0; i < m.bondA.size(); i++) { v.push_back(m.bondA[i]); v.push_back(m.bondB[i]); v.push_back(m.bondT[i]); }
        else if (k == "angles")
            for (size_t i = 0; i < m.angC.size(); i++) { v.push_back(m.angC[i]); v.push_back(m.angL1[i]); v.push_back(m.angL2[i]); v.push_back(m.angT[i]); }
        else if (k == "types") { v.resize(m.nAt); for (int i = 0; i < m.nAt; i++) v[i] = m.types[i]; }
        else if (k == "x") per_atom(m.x);
        else if (k == "y") per_atom(m.y);
        else if (k == "z") per_atom(m.z);
        else if (k == "vx") per_atom(m.vx);
        else if (k == "vy") per_atom(m.vy);
        else if (k == "vz") per_atom(m.vz);
        else if (k == "photons") { v.resize(m.nAt); photon_engs(m.nAt, v.data(), m.Temp, seed_in); }
        else if (k == "uvects") { v.resize(3 * kNumUnitVectors); unit_vectors(v.data(), v.data() + kNumUnitVectors, v.data() + 2 * kNumUnitVectors); }
        else throw std::runtime_error(std::string("unknown query key: ") + key);
        rc = (int)v.size();
        if (out && cap >= rc) std::memcpy(out, v.data(), sizeof(double) * v.size());
    });
    return status == AZTOT_OK ? rc : status;
}

int aztot_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n < 0 ? 0 : n;
}

int aztot_device_synchronize(int device)
{
    return guarded([&] {
        check_hip(hipSetDevice(device), "hipSetDevice");
        check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
    });
}

void aztot_default_options(aztot_options* opt)
{
    if (!opt) return;
    std::memset(opt, 0, sizeof(*opt));
    opt->struct_size = (uint32_t)sizeof(*opt);
    opt->device = 0;
    opt->initial_forces = 1;
    opt->center_box = 0;
    opt->seed = 12345;
    opt->pair_variant = 0;
    opt->cell_size = 0.0;
    opt->use_graph = 1;
    opt->profile = 0;
    opt->sort_every = 0;
    opt->skin = 0.0;
}

static int init_device_common(const aztot_model* h, const aztot_options* opt, int rank, int nranks, const void* id_bytes,
                              aztot_sendrecv_fn sr, aztot_allreduce_fn ar, void* ctx, aztot_md** out)
{
    if (!h || !out) return fail(AZTOT_ERR_ARG, "null argument");
    *out = nullptr;
    aztot_options o;
    if (opt)
    {
        if (opt->struct_size != (uint32_t)sizeof(aztot_options))
            return fail(AZTOT_ERR_ARG, "aztot_options.struct_size does not match this library (start from aztot_default_options)");
        o = *opt;
        for (int r : o.reserved) if (r != 0) return fail(AZTOT_ERR_ARG, "aztot_options.reserved must be zero");
        if (o.pair_variant < 0 || o.pair_variant > 2) return fail(AZTOT_ERR_ARG, "aztot_options.pair_variant: 0 (automatic), 1 or 2 (variant 3 was retired: measured slower than 2)");
    }
    else aztot_default_options(&o);
    return guarded([&] {
        Model m = h->m;
        finish_model(m, o.seed);
        if (o.center_box) center_box(m);
        auto md = std::make_unique<aztot_md>();
        if (nranks > 1)
        {
            if (rank < 0 || rank >= nranks) throw std::runtime_error("slab: bad rank");
            if (id_bytes)
            {
                int ndev = 0;
                if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available: the azTotMD hot path has no CPU fallback");
                check_hip(hipSetDevice(o.device), "hipSetDevice");
                md->xch.reset(new RcclExchanger(rank, nranks, id_bytes));
            }
            else if (sr && ar) md->xch.reset(new CallbackExchanger(sr, ar, ctx));
            else if (!o.loopback_ranks) throw std::runtime_error("slab: neither an RCCL id nor exchange callbacks were given");
        }
        md->eng.reset(new Engine(m, o, rank, nranks, md->xch.get()));
        *out = md.release();
    });
}

int aztot_init_device(const aztot_model* m, const aztot_options* opt, aztot_md** out)
{
    return init_device_common(m, opt, 0, 1, nullptr, nullptr, nullptr, nullptr, out);
}

int aztot_init_device_slab(const aztot_model* m, const aztot_options* opt, int rank, int nranks, const void* rccl_id_bytes,
                           aztot_sendrecv_fn sendrecv, aztot_allreduce_fn allreduce, void* ctx, aztot_md** out)
{
    return init_device_common(m, opt, rank, nranks, rccl_id_bytes, sendrecv, allreduce, ctx, out);
}

void aztot_free_device(aztot_md* md) { delete md; }

int aztot_step(aztot_md* md, int nsteps)
{
    if (!md) return fail(AZTOT_ERR_ARG, "null handle");
    return guarded([&] { md->eng->step(nsteps); });
}

int aztot_sync(aztot_md* md)
{
    if (!md) return fail(AZTOT_ERR_ARG, "null handle");
    return guarded([&] { md->eng->sync_all(); });
}

int aztot_forces(aztot_md* md)
{
    if (!md) return fail(AZTOT_ERR_ARG, "null handle");
    return guarded([&] { md->eng->forces(); });
}

int aztot_get_stats(aztot_md* md, aztot_stats* out)
{
    if (!md || !out) return fail(AZTOT_ERR_ARG, "null argument");
    return guarded([&] { md->eng->get_stats(*out); });
}

int aztot_species_crossings(aztot_md* md, int64_t* out, int cap)
{
    if (!md || !md->eng || !out) return fail(AZTOT_ERR_ARG, "null argument");
    return guarded([&] { md->eng->species_crossings(out, cap); });
}

int aztot_md_to_host(aztot_md* md, aztot_state* out)
{
    if (!md || !out) return fail(AZTOT_ERR_ARG, "null argument");
    return guarded([&] { md->eng->md_to_host(*out); });
}

int aztot_set_state(aztot_md* md, const aztot_state* in)
{
    if (!md || !in) return fail(AZTOT_ERR_ARG, "null argument");
    return guarded([&] { md->eng->set_state(*in); });
}

int aztot_get_clock(aztot_md* md, aztot_clock* out)
{
    if (!md || !out) return fail(AZTOT_ERR_ARG, "null argument");
    return guarded([&] { md->eng->get_clock(*out); });
}

int aztot_set_clock(aztot_md* md, const aztot_clock* in)
{
    if (!md || !in) return fail(AZTOT_ERR_ARG, "null argument");
    return guarded([&] { md->eng->set_clock(*in); });
}

int aztot_cell_table(aztot_md* md, int32_t dims[3], int32_t* cell_start, int cap_cells, int32_t* atom_id, int cap_atoms)
{
    if (!md || !md->eng || !dims) return fail(AZTOT_ERR_ARG, "null argument");
    int n = 0;
    const int rc = guarded([&] { n = md->eng->cell_table(dims, cell_start, cap_cells, atom_id, cap_atoms); });
    return rc < 0 ? rc : n;
}

int aztot_kernel_times(aztot_md* md, char* names, int cap, double* ms, int64_t* calls, int max_kernels)
{
    if (!md) return fail(AZTOT_ERR_ARG, "null handle");
    int n = 0;
    int status = guarded([&] {
        std::vector<KernelTimer> t;
        n = md->eng->kernel_times(t);
        int pos = 0;
        for (int i = 0; i < n && i < max_kernels; i++)
        {
            if (ms) ms[i] = t[i].ms;
            if (calls) calls[i] = t[i].calls;
            const int len = (int)t[i].name.size() + 1;
            if (names && pos + len <= cap) { std::memcpy(names + pos, t[i].name.c_str(), len); pos += len; }
        }
        if (names && pos < cap) names[pos] = 0;
    });
    return status == AZTOT_OK ? n : status;
}

int aztot_reset_kernel_times(aztot_md* md)
{
    if (!md) return fail(AZTOT_ERR_ARG, "null handle");
    return guarded([&] { md->eng->reset_kernel_times(); });
}

int aztot_set_profile(aztot_md* md, int on)
{
    if (!md) return fail(AZTOT_ERR_ARG, "null handle");
    return guarded([&] { md->eng->set_profile(on != 0); });
}

int aztot_comm_id_bytes(void) { return RcclExchanger::id_bytes(); }
int aztot_comm_make_id(void* id_bytes)
{
    if (!id_bytes) return fail(AZTOT_ERR_ARG, "null argument");
    return guarded([&] { RcclExchanger::make_id(id_bytes); });
}

int aztot_comm_ranks(aztot_md* md)
{
    if (!md || !md->eng) return fail(AZTOT_ERR_ARG, "null handle");
    int n = 0;
    const int rc = guarded([&] { n = md->eng->comm_ranks(); });
    return rc < 0 ? rc : n;
}

int aztot_comm_selftest(int device)
{
    return guarded([&] { RcclExchanger::selftest(device); });
}

}  // extern "C"
