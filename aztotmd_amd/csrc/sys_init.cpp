// Input surface of the hot path: atoms.xyz / field.txt / control.txt / cuda.txt -> aztot::Model.
// Mirrors the reference's init_md (sys_init.cpp:1036-1119) for the directives that reach the per-step
// path; the grammar kept is the one *implemented* by the reference, i.e.
// order-free, first-match-wins keyword scanning (utils.cpp:87-195).
#include "model.h"
#include "rng.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace aztot {
namespace {

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error(msg); }

struct File
{
    FILE* f = nullptr;
    File(const std::string& path, const char* code)
    {
        f = std::fopen(path.c_str(), "r");
        if (!f) fail(std::string(code) + " can't open file '" + path + "'");
    }
    ~File() { if (f) std::fclose(f); }
    File(const File&) = delete;
};

// Keyword scanner with the reference's semantics (utils.cpp:87-195): rewind, then try the scanf template
// at the current position; on mismatch swallow one whitespace-delimited token and retry until EOF.
// On success the stream is left right after the converted value, so the caller continues with plain fscanf.
template <typename T>
bool seek_value(FILE* f, const char* templ, T* value)
{
    char token[128];
    std::rewind(f);
    while (!std::feof(f))
    {
        if (std::fscanf(f, templ, value) > 0) return true;
        if (std::fscanf(f, "%127s", token) == EOF) break;
    }
    return false;
}

int seek_int_or(FILE* f, const char* templ, int fallback)
{
    int v = 0;
    return seek_value(f, templ, &v) ? v : fallback;
}

int vdw_type_by_name(const char* s)
{   // vdw.cpp:193
    static const char* names[] = {"lnjs", "buck", "p746", "bmhs", "elin", "einv", "surk"};
    for (int i = 0; i < 7; i++) if (std::strcmp(s, names[i]) == 0) return i + 1;
    return 0;
}
const int kVdwNParam[8] = {0, 2, 3, 3, 5, 3, 3, 4};   // vdw.cpp:195

int bond_type_by_name(const char* s)
{   // bonds.cpp:158-252
    static const char* names[] = {"harm", "mors", "pdn", "buck", "e612"};
    for (int i = 0; i < 5; i++) if (std::strcmp(s, names[i]) == 0) return i + 1;
    return 0;
}
const int kBondNParam[6] = {0, 2, 4, 5, 3, 5};

int species_by_name(const Model& m, const char* name)
{
    for (int i = 0; i < m.nSpec(); i++) if (m.species[i].name == name) return i;
    return -1;
}

void reject_section(FILE* f, const char* templ, const char* what)
{
    int n = 0;
    if (seek_value(f, templ, &n) && n != 0)
        fail(std::string("out of scope: '") + what + "' section of field.txt is not part of the accelerated hot path");
}

// read_field: sys_init.cpp:174-485
void read_field(const std::string& dir, Model& m)
{
    File file(dir + "/field.txt", "ERROR[001]");
    FILE* f = file.f;
    int n = 0;
    if (!seek_value(f, " spec %d", &n) || n <= 0) fail("ERROR[004] there is no 'spec' section in field.txt");
    if (n > kMaxSpecies) fail("too many species (MX_SPEC = 15, defines.h:14)");
    m.species.resize(n);
    m.charged_spec = 0;
    for (int i = 0; i < n; i++)
    {   // read_spec: sys_init.cpp:83-130
        char name[64], nucl[64];
        Species& s = m.species[i];
        if (std::fscanf(f, "%63s %63s %lf %lf %lf", name, nucl, &s.mass_amu, &s.charge, &s.energy) != 5)
            fail("ERROR[004] malformed 'spec' line " + std::to_string(i + 1));
        s.name = name; s.nucleus = nucl;
        s.mass = s.mass_amu * units::m_scale;
        s.charged = std::fabs(s.charge) < 1.0E-10 ? 0 : 1;
        if (s.charge != 0.0) m.charged_spec = 1;
    }
    reject_section(f, " red-ox %d", "red-ox");
    if (seek_value(f, " frozensp %d", &n))
        for (int i = 0; i < n; i++)
        {
            char name[64];
            if (std::fscanf(f, "%63s", name) != 1) break;
            int j = species_by_name(m, name);
            if (j >= 0) m.species[j].frozen = 1;
            else m.warnings.push_back(std::string("WARNING[b001] unknown atom type in 'frozensp': ") + name);
        }
    // vdw: sys_init.cpp:258-285 + read_vdw vdw.cpp:234-308
    m.pairpots.assign((size_t)m.nSpec() * m.nSpec(), PairPot());
    m.minRvdw = 999999.9; m.maxRvdw = 0.0;
    if (seek_value(f, " vdw %d", &n) && n > 0)
    {
        m.nVdW = n;
        for (int i = 0; i < n; i++)
        {
            char a[64], b[64], c[64];
            double rc, p[5] = {0, 0, 0, 0, 0};
            if (std::fscanf(f, " %63s %63s %63s %lf %lf %lf ", a, b, c, &rc, &p[0], &p[1]) != 6)
                fail("ERROR[006] malformed vdw line " + std::to_string(i + 1));
            int type = vdw_type_by_name(c);
            if (!type) fail(std::string("ERROR[006] unknown potential type (") + c + ") in vdw line " + std::to_string(i + 1));
            for (int k = 2; k < kVdwNParam[type]; k++)
                if (std::fscanf(f, " %lf", &p[k]) != 1) fail("ERROR[006] too few parameters in vdw line " + std::to_string(i + 1));
            int ia = species_by_name(m, a), ib = species_by_name(m, b);
            if (ia < 0 || ib < 0) fail(std::string("ERROR[005] unknown atom type in vdw line: ") + a + " " + b + " " + c);
            PairPot pp = prepare_vdw(type, rc, p);
            if (pp.rcut < m.minRvdw) m.minRvdw = pp.rcut;
            if (pp.rcut > m.maxRvdw) m.maxRvdw = pp.rcut;
            if (m.pairpots[(size_t)ia * m.nSpec() + ib].type)
                m.warnings.push_back(std::string("WARNING[002] pair potential between ") + a + " and " + b + " redeclared");
            m.pairpots[(size_t)ia * m.nSpec() + ib] = pp;
            if (type != AZTOT_VDW_SURK) m.pairpots[(size_t)ib * m.nSpec() + ia] = pp;   // vdw.cpp:303-307
        }
    }
    else
        m.warnings.push_back("WARNING[001] no Van-der-Waals interactions");
    // bond types: sys_init.cpp:289-314 + read_bond bonds.cpp:125-364.  Only constant bonds ('con con') are on the
    // accelerated path; 'mut' / 'br' make the reference's use_bnd = 2 (variable bonds) and are refused.
    m.bondTypes.clear(); m.angleTypes.clear();
    if (seek_value(f, " bonds %d", &n) && n > 0)
        for (int i = 1; i <= n; i++)
        {
            int id; char a[16], b[16], key[16];
            if (std::fscanf(f, "%d %8s %8s %8s", &id, a, b, key) != 4) fail("ERROR[126] malformed bonds line " + std::to_string(i));
            const int ia = species_by_name(m, a), ib = species_by_name(m, b);
            if (ia < 0 || ib < 0) fail(std::string("ERROR[124]: Unknown species in bonds declaration: ") + a + " " + b);
            const int type = bond_type_by_name(key);
            if (!type) fail(std::string("ERROR[126]: Unknown potential type in bonds declaration: ") + key);
            double p[5] = {0, 0, 0, 0, 0};
            for (int k = 0; k < kBondNParam[type]; k++)
                if (std::fscanf(f, " %lf", &p[k]) != 1) fail("ERROR[126] too few parameters in bonds line " + std::to_string(i));
            for (int lim = 0; lim < 2; lim++)
            {
                if (std::fscanf(f, "%8s", key) != 1) fail("ERROR[501] truncated bonds line " + std::to_string(i));
                if (std::strcmp(key, "con") != 0)
                {
                    if (std::strcmp(key, "mut") == 0 || (lim == 1 && std::strcmp(key, "br") == 0))
                        fail("out of scope: variable bonds ('mut'/'br', use_bnd = 2) are not part of the accelerated hot path");
                    fail(std::string(lim ? "ERROR[502]" : "ERROR[501]") + ": Unknown type of bond limit: " + key);
                }
            }
            add_bond_type(m, ia, ib, type, p);
        }
    reject_section(f, " evol_bonds %d", "evol_bonds");
    reject_section(f, " h-bonds %d", "h-bonds");
    // angle types: sys_init.cpp:411-427 + read_angle angles.cpp:78-128
    if (seek_value(f, " angles %d ", &n) && n > 0)
        for (int i = 1; i <= n; i++)
        {
            int id; char a[16], key[16]; double p0, p1;
            if (std::fscanf(f, "%d %8s %8s %lf %lf", &id, a, key, &p0, &p1) != 5) fail("ERROR[012] malformed angles line " + std::to_string(i));
            const int ia = species_by_name(m, a);
            if (ia < 0) fail(std::string("ERROR[011]: Unknown species in angle declaration: ") + a);
            if (std::strcmp(key, "hcos") != 0) fail(std::string("ERROR[012]: Unknown potential type in angle declaration: ") + key);
            add_angle_type(m, ia, 1, p0, p1);
        }
    reject_section(f, " angle_forming %d ", "angle_forming");
    reject_section(f, " linkage %d", "linkage");
    // radii: the integer after the keyword is ignored, nSpec lines follow (sys_init.cpp:468-480)
    if (seek_value(f, " radii %d", &n))
    {
        m.has_radii = 1;
        for (int i = 0; i < m.nSpec(); i++)
        {
            char name[64];
            if (std::fscanf(f, "%63s", name) != 1) fail("ERROR[b018] truncated radii section");
            int j = species_by_name(m, name);
            if (j < 0) fail(std::string("ERROR[b018] wrong species(") + name + ") in radii section");
            if (std::fscanf(f, "%lf %lf %lf", &m.species[j].radA, &m.species[j].radB, &m.species[j].mxEng) != 3)
                fail("ERROR[b018] malformed radii line");
        }
    }
}

// the 'bond_list' / 'angle_list' switches of field.txt and the files they name (read by the reference inside read_sim,
// sys_init.cpp:626-673, i.e. after the atoms): bonds.txt 'N' + N x 'at1 at2 type' (bonds.cpp:25-110),
// angles.txt 'N' + N x 'central lig1 lig2 type' (angles.cpp:22-60); atom indices are 0-based.
void read_bonded_lists(const std::string& dir, Model& m)
{
    File file(dir + "/field.txt", "ERROR[409]");
    int flag = 0;
    if (seek_value(file.f, " bond_list %d", &flag))
    {
        FILE* g = std::fopen((dir + "/bonds.txt").c_str(), "r");
        if (!g) m.warnings.push_back("WARNING[a001] bond list is used, but there is no such file. No bonds are downloaded");
        else
        {
            int n = 0;
            if (std::fscanf(g, "%d", &n) != 1 || n < 0) { std::fclose(g); fail("ERROR[121] malformed bonds.txt"); }
            std::vector<int32_t> a(n), b(n), t(n);
            for (int i = 0; i < n; i++)
                if (std::fscanf(g, "%d %d %d", &a[i], &b[i], &t[i]) != 3) { std::fclose(g); fail("ERROR[121] truncated bonds.txt at line " + std::to_string(i)); }
            std::fclose(g);
            set_bond_list(m, n, a.data(), b.data(), t.data());
        }
    }
    if (seek_value(file.f, " angle_list %d", &flag))
    {
        if (m.angleTypes.empty())
            m.warnings.push_back("WARNING[b006] 'anlge_list' directive is ignored, because there are no angle type defintions");
        else
        {
            FILE* g = std::fopen((dir + "/angles.txt").c_str(), "r");
            if (!g) m.warnings.push_back("WARNING[a002] angle list is used, but there is no such file. No anlges are downloaded");
            else
            {
                int n = 0;
                if (std::fscanf(g, "%d", &n) != 1 || n < 0) { std::fclose(g); fail("ERROR[013] malformed angles.txt"); }
                std::vector<int32_t> c(n), l1(n), l2(n), t(n);
                for (int i = 0; i < n; i++)
                    if (std::fscanf(g, "%d %d %d %d", &c[i], &l1[i], &l2[i], &t[i]) != 4) { std::fclose(g); fail("ERROR[013] truncated angles.txt at line " + std::to_string(i)); }
                std::fclose(g);
                set_angle_list(m, n, c.data(), l1.data(), l2.data(), t.data());
            }
        }
    }
}

// read_atoms_box: sys_init.cpp:487-565 ; read_box: box.cpp:9-28
void read_atoms_box(const std::string& dir, Model& m)
{
    File file(dir + "/atoms.xyz", "ERROR[007]");
    FILE* f = file.f;
    int n = 0, btype = 0;
    if (std::fscanf(f, "%d", &n) != 1 || n <= 0) fail("ERROR[007] atoms.xyz: bad atom count");
    if (std::fscanf(f, "%d", &btype) != 1 || btype != 1) fail("ERROR[008] unknown box type (only 1 = rectangular)");
    if (std::fscanf(f, "%lf %lf %lf", &m.L[0], &m.L[1], &m.L[2]) != 3) fail("ERROR[008] malformed box line");
    m.nAt = n;
    m.types.resize(n); m.x.resize(n); m.y.resize(n); m.z.resize(n);
    m.vx.assign(n, 0.0); m.vy.assign(n, 0.0); m.vz.assign(n, 0.0);
    for (auto& s : m.species) s.number = 0;
    char name[64], last[64] = "";
    int last_id = -1;
    for (int i = 0; i < n; i++)
    {
        if (std::fscanf(f, "%63s %lf %lf %lf", name, &m.x[i], &m.y[i], &m.z[i]) != 4)
            fail("ERROR[009] atoms.xyz: malformed atom line " + std::to_string(i + 1));
        if (last_id < 0 || std::strcmp(name, last) != 0)
        {
            last_id = species_by_name(m, name);
            if (last_id < 0) fail("ERROR[009] unknown atom[" + std::to_string(i + 1) + "] type=" + name + " in atoms.xyz");
            std::strcpy(last, name);
        }
        m.types[i] = last_id;
        m.species[last_id].number++;
    }
}

// read_sim: sys_init.cpp:590-989 (hot-path directives) ; read_tstat temperature.cpp:91-259 ; read_elec elec.cpp:14-67
void read_sim(const std::string& dir, Model& m)
{
    File file(dir + "/control.txt", "ERROR[410]");
    FILE* f = file.f;
    if (!seek_value(f, " timestep %lf ", &m.tSt)) fail("ERROR[411] timestep must be declared in control.txt");
    if (!seek_value(f, " timesim %lf ", &m.tSim))
    {
        if (!seek_value(f, " nstep %d", &m.nSt)) fail("ERROR[412] no 'nstep' or 'timesim' directives in control.txt");
        m.tSim = double(m.nSt * m.tSt);
    }
    else
        m.nSt = (int)(m.tSim / m.tSt);
    double tEq = 0;
    if (!seek_value(f, " timeequil %lf ", &tEq))
        m.nEq = seek_int_or(f, " nequil %d ", 0);
    else
        m.nEq = (int)(tEq / m.tSt);
    m.freqEq = 0;
    if (m.nEq) m.freqEq = seek_int_or(f, " eqfreq %d ", 0);

    // thermostat
    if (!seek_value(f, " temperature %lf ", &m.Temp)) fail("ERROR[404] temperature is not defined in control.txt");
    char s[64];
    if (std::fscanf(f, "%63s", s) != 1) fail("ERROR[405] thermostat type missing");
    if (std::strcmp(s, "none") == 0) m.tstat_type = AZTOT_TSTAT_NONE;
    else if (std::strcmp(s, "nose") == 0)
    {
        m.tstat_type = AZTOT_TSTAT_NOSE;
        if (std::fscanf(f, " %lf ", &m.tau) != 1) fail("ERROR[405] 'nose' needs a relaxation time");
    }
    else if (std::strcmp(s, "radi") == 0)
    {   // the shipped files write 'radi 0.2', which %d reads as 0 and leaves '.2' behind (temperature.cpp:113)
        if (std::fscanf(f, "%d", &m.tstat_step) != 1) fail("ERROR[a002] there is no step parameter for radiative thermostat");
        m.tstat_type = AZTOT_TSTAT_RADI;
    }
    else fail("ERROR[405] unknown thermostat type");

    // electrostatics
    if (!seek_value(f, " elec %63s", s)) fail("ERROR[401] electrostatic calculations are not specified in control.txt");
    if (std::strcmp(s, "none") == 0)
    {
        m.elec_type = AZTOT_ELEC_NONE; m.rReal = 0.0;
        if (m.charged_spec) m.warnings.push_back("WARNING[b003] species have charges but 'elec none': charges ignored");
    }
    else if (std::strcmp(s, "dir") == 0)
    {
        m.elec_type = AZTOT_ELEC_DIRECT;
        if (std::fscanf(f, " %lf ", &m.rReal) != 1) fail("ERROR[404] 'elec dir' needs a cut-off");
    }
    else if (std::strcmp(s, "pme") == 0)
    {
        m.elec_type = AZTOT_ELEC_EWALD;
        if (std::fscanf(f, " %lf %lf %d %d %d", &m.rReal, &m.alpha, &m.ewald_k[0], &m.ewald_k[1], &m.ewald_k[2]) != 5)
            fail("ERROR[404] malformed 'elec pme'");
    }
    else if (std::strcmp(s, "fenn") == 0)
    {
        m.elec_type = AZTOT_ELEC_FENNEL;
        if (std::fscanf(f, " %lf %lf", &m.rReal, &m.alpha) != 2) fail("ERROR[404] malformed 'elec fenn'");
    }
    else fail(std::string("ERROR[404] unknown type of electrostatic calculations: ") + s);
    if (!m.charged_spec && m.elec_type)
    {   // elec.cpp:52-56
        m.warnings.push_back(std::string("WARNING[b004] species have no charges but elec is ") + s + ": switched to none");
        m.elec_type = AZTOT_ELEC_NONE;
    }
    m.r2Real = m.rReal * m.rReal;
    if (!seek_value(f, " permittivity %lf ", &m.eps)) m.eps = 1.0;

    // initial velocities: sys_init.cpp:750-805
    if (!seek_value(f, " init_vel %63s", s)) fail("ERROR[406] no init_vel directive in control.txt");
    if (std::strcmp(s, "zero") == 0) m.init_vel = AZTOT_VEL_ZERO;
    else if (std::strcmp(s, "gaus") == 0) m.init_vel = AZTOT_VEL_GAUSS;
    else if (std::strcmp(s, "const") == 0)
    {
        m.init_vel = AZTOT_VEL_CONST;
        if (std::fscanf(f, "%lf %lf %lf", &m.init_vel_par[0], &m.init_vel_par[1], &m.init_vel_par[2]) != 3)
            fail("ERROR[407] 'init_vel const' needs three components");
    }
    else if (std::strcmp(s, "keng") == 0)
    {
        m.init_vel = AZTOT_VEL_KENG;
        if (std::fscanf(f, "%lf", &m.init_vel_par[0]) != 1) fail("ERROR[407] 'init_vel keng' needs an energy");
    }
    else fail("ERROR[407] unknown value of init_vel directive");

    if (seek_int_or(f, " eJump %d ", 0) != 0) fail("out of scope: electron hopping (eJump) is not part of the accelerated hot path");

    // external field: 'elecfield Ux Uy Uz' (sys_init.cpp:842-849)
    if (seek_value(f, " elecfield %lf ", &m.E[0]))
    {
        if (std::fscanf(f, " %lf %lf ", &m.E[1], &m.E[2]) != 2) { m.E[1] = 0; m.E[2] = 0; }
    }
    else { m.E[0] = m.E[1] = m.E[2] = 0.0; }
    double shiftX = 0;
    if (seek_value(f, " shiftX %lf ", &shiftX)) m.warnings.push_back("WARNING: 'shiftX' (serial-only special purpose push) is ignored");
    if (seek_int_or(f, " reset_vels %d ", 0)) m.warnings.push_back("WARNING: 'reset_vels' is ignored");

    if (seek_value(f, " cell_list %lf ", &m.desired_cell_size)) m.use_clist = 1;   // sys_init.cpp:862-865
    if (!seek_value(f, " stat %d ", &m.stat)) m.stat = 1000;
    if (!seek_value(f, " max_neigh %d ", &m.max_neigh)) m.max_neigh = 50;
}

// read_cuda: cuInit.cu:684-754 (optional file: defaults as the reference's)
void read_cuda(const std::string& dir, Model& m)
{
    FILE* f = std::fopen((dir + "/cuda.txt").c_str(), "r");
    if (!f) { m.warnings.push_back("cuda.txt not found: defaults used"); return; }
    m.nstep_stat = seek_int_or(f, " nstep stat %d", 10);
    m.nthread_a = seek_int_or(f, " nthread a %d", 16);
    m.nthread_b = seek_int_or(f, " nthread b %d", 32);
    std::fclose(f);
}

double rand01(uint64_t seed, uint64_t stream, uint64_t id, uint64_t* draw)
{   // utils.cpp:197-201 keeps only 1e-4 resolution; libc rand() replaced by the counter RNG
    uint32_t r = rng_draw(seed, stream, id, (*draw)++);
    return double(r % 10000) / 10000;
}

double gauss(double stdev, double mean, uint64_t seed, uint64_t id, uint64_t* draw)
{   // md_utils.cpp:7-25 (Frenkel p.579)
    double v1 = 0, v2 = 0, r = 2.0;
    while (r > 1.0 || r == 0.0)
    {
        v1 = 2.0 * rand01(seed, kRngStreamInitVel, id, draw) - 1.0;
        v2 = 2.0 * rand01(seed, kRngStreamInitVel, id, draw) - 1.0;
        r = v1 * v1 + v2 * v2;
    }
    return mean + stdev * (v1 * std::sqrt(-2.0 * std::log(r) / r));
}

double prob4(double x, double y, double theta)
{   // temperature.cpp:18-26
    const double r24 = 1.0 / 24.0, r6 = 1.0 / 6.0;
    double ty = theta * y, ty2 = ty * ty;
    return (1 - x) * std::exp(y * theta) - (r24 * ty2 * ty2 + r6 * ty2 * ty + 0.5 * ty * ty + ty + 1);
}

}  // namespace

void add_bond_type(Model& m, int spec1, int spec2, int type, const double p[5])
{   // read_bond bonds.cpp:125-252: the unit factors applied there (E_scale, r_scale; const.h:39-41) are exactly 1.0
    if (type < 1 || type > 5) fail("ERROR[126]: Unknown potential type in bonds declaration");
    if (spec1 < 0 || spec1 >= m.nSpec() || spec2 < 0 || spec2 >= m.nSpec()) fail("ERROR[124]: Unknown species in bonds declaration");
    BondType b; b.type = type; b.spec1 = spec1; b.spec2 = spec2;
    for (int k = 0; k < 5; k++) b.p[k] = (k < kBondNParam[type]) ? p[k] : 0.0;
    m.bondTypes.push_back(b);
}

void add_angle_type(Model& m, int central, int type, double k, double cos0)
{   // read_angle angles.cpp:78-128
    if (type != 1) fail("ERROR[012]: Unknown potential type in angle declaration");
    if (central < 0 || central >= m.nSpec()) fail("ERROR[011]: Unknown species in angle declaration");
    AngleType a; a.type = 1; a.central = central; a.k = k; a.cos0 = cos0;
    m.angleTypes.push_back(a);
}

void set_bond_list(Model& m, int n, const int32_t* a, const int32_t* b, const int32_t* t)
{
    m.bondA.clear(); m.bondB.clear(); m.bondT.clear();
    for (int i = 0; i < n; i++)
    {
        int at1 = a[i], at2 = b[i];
        const int k = t[i];
        if (k < 1 || k > (int)m.bondTypes.size()) fail("ERROR[121] unknown bond type " + std::to_string(k) + " in bond list, line " + std::to_string(i));
        if (at1 < 0 || at1 >= m.nAt || at2 < 0 || at2 >= m.nAt) fail("ERROR[121] atom index out of range in bond list, line " + std::to_string(i));
        const BondType& bt = m.bondTypes[k - 1];
        if (bt.spec1 == m.types[at1])
        {
            if (bt.spec2 != m.types[at2]) fail("ERROR [121] incorrect type of 2th atom in bond (type: " + std::to_string(k) + ", line: " + std::to_string(i) + ")");
        }
        else if (bt.spec1 == m.types[at2])
        {
            if (bt.spec2 == m.types[at1]) std::swap(at1, at2);      // bonds.cpp:62-67
            else fail("ERROR [122] incorrect type of 1th atom in bond (type: " + std::to_string(k) + ", line: " + std::to_string(i) + ")");
        }
        else fail("ERROR [123] incorrect type of atoms for bond type(" + std::to_string(k) + ") in bond list, line: " + std::to_string(i));
        m.bondA.push_back(at1); m.bondB.push_back(at2); m.bondT.push_back(k);
    }
}

void set_angle_list(Model& m, int n, const int32_t* c, const int32_t* l1, const int32_t* l2, const int32_t* t)
{
    m.angC.clear(); m.angL1.clear(); m.angL2.clear(); m.angT.clear();
    for (int i = 0; i < n; i++)
    {
        if (t[i] < 1 || t[i] > (int)m.angleTypes.size()) fail("ERROR[013] wrong atom type number in angles.txt, line " + std::to_string(i));
        if (c[i] < 0 || c[i] >= m.nAt || l1[i] < 0 || l1[i] >= m.nAt || l2[i] < 0 || l2[i] >= m.nAt)
            fail("ERROR[013] atom index out of range in angle list, line " + std::to_string(i));
        if (m.types[c[i]] != m.angleTypes[t[i] - 1].central)
            fail("ERROR[014] wrong central atom type in angle list (" + std::to_string(i) + " postion)");
        m.angC.push_back(c[i]); m.angL1.push_back(l1[i]); m.angL2.push_back(l2[i]); m.angT.push_back(t[i]);
    }
}

PairPot prepare_vdw(int type, double rcut, const double p[5])
{   // read_vdw: vdw.cpp:261-299 ; scale tables vdw.cpp:209-219 with r_scale = E_scale = 1 (const.h:38-41)
    PairPot pp;
    pp.type = type; pp.rcut = rcut; pp.r2cut = rcut * rcut;
    pp.p0 = p[0]; pp.p1 = p[1]; pp.p2 = p[2]; pp.p3 = p[3]; pp.p4 = p[4];
    switch (type)
    {
    case AZTOT_VDW_LJ:
        pp.p0 *= 4; pp.p3 = 0; pp.p4 = 0;
        pp.p1 = pp.p1 * pp.p1;       // sigma^2
        pp.p2 = 6 * pp.p0;           // 24 epsilon
        break;
    case AZTOT_VDW_BUCK: case AZTOT_VDW_746: case AZTOT_VDW_ELIN: case AZTOT_VDW_EINV:
        pp.p3 = 0; pp.p4 = 0; break;
    case AZTOT_VDW_BHM: break;
    case AZTOT_VDW_SURK: pp.p4 = 0; pp.use_radii = 1; break;
    default: fail("ERROR[006] unknown potential type id " + std::to_string(type));
    }
    return pp;
}

void center_box(Model& m)
{   // box.cpp:337-384 (serial path only: sys_init.cpp:1145); note the shift is half the extent minus L/2
    double mx[3] = {0, 0, 0}, mn[3] = {m.L[0], m.L[1], m.L[2]};
    std::vector<double>* c[3] = {&m.x, &m.y, &m.z};
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < m.nAt; i++)
        {
            double v = (*c[k])[i];
            if (v > mx[k]) mx[k] = v;
            if (v < mn[k]) mn[k] = v;
        }
    for (int k = 0; k < 3; k++)
    {
        double d = 0.5 * (mx[k] - mn[k]) - m.L[k] * 0.5;
        for (int i = 0; i < m.nAt; i++) (*c[k])[i] -= d;
    }
}

void photon_engs(int n, double* engs, double T, uint64_t seed)
{   // photon_engs: temperature.cpp:28-89 - bisection of the Gamma(5, kT) CDF on y in [0, 1] eV
    const double eps = 1e-3; const int limit = 20;
    const double theta = 1.0 / (units::kB * T);
    for (int i = 0; i < n; i++)
    {
        uint64_t draw = 0;
        double a = 0.0, b = 1.0, x, ra, rb;
        do
        {
            x = rand01(seed, kRngStreamTables, (uint64_t)i, &draw);
            ra = prob4(x, 0.0, theta); rb = prob4(x, 1.0, theta);
        } while (ra * rb > 0);
        double y = 0.5, r = prob4(x, y, theta);
        int k = 0;
        while ((r > eps) || (r < -eps))
        {
            if ((r * ra) < 0) { b = y; y = 0.5 * (a + y); }
            else { a = y; y = 0.5 * (y + b); }
            r = prob4(x, y, theta);
            if (++k >= limit) { if (i > 0) y = engs[i - 1]; break; }   // temperature.cpp:76-80; i == 0 keeps y (SURVEY C-21)
        }
        engs[i] = y;
    }
}

void unit_vectors(double* ux, double* uy, double* uz)
{   // temperature.cpp:165-223: 32 phi x 16 theta grid, each with its negative, in three axis permutations
    const int nTh = 16, nPhi = 32;
    const double twopi = 2.0 * units::pi;
    int k = 0;
    for (int perm = 0; perm < 3; perm++)
        for (int i = 0; i < nPhi; i++)
        {
            double phi = (double)i / nPhi * twopi;
            for (int j = 0; j < nTh; j++)
            {
                double theta = (double)j / nTh * units::pi;
                double st = std::sin(theta), ct = std::cos(theta), sp = std::sin(phi), cp = std::cos(phi);
                double a = cp * ct, b = sp * ct, c = st;
                double X, Y, Z;
                if (perm == 0) { X = a; Y = b; Z = c; }
                else if (perm == 1) { X = a; Z = b; Y = c; }
                else { Z = a; Y = b; X = c; }
                ux[k] = X; uy[k] = Y; uz[k] = Z;
                ux[k + 1] = -X; uy[k + 1] = -Y; uz[k + 1] = -Z;
                k += 2;
            }
        }
}

double initial_radius(uint64_t seed, uint64_t id)
{   // init_cuda_tstat: cuTemp.cu:41
    uint64_t draw = 1000;
    return 0.577 + rand01(seed, kRngStreamTables, id, &draw) * 0.0001;
}

void finish_model(Model& m, uint64_t seed)
{
    // prepare_elec: elec.cpp:371-406
    m.kvecs.clear(); m.engElec1 = 0.0;
    if (m.elec_type == AZTOT_ELEC_EWALD)
    {   // 'elec pme' is a plain Ewald sum in the reference (ewald_rec elec.cpp:167-335; GPU twin recip_ewald + ewald_force cuElec.cu:151-382)
        const int kx = m.ewald_k[0], ky = m.ewald_k[1], kz = m.ewald_k[2];
        if (kx < 1 || ky < 1 || kz < 1 || kx > kEwaldKMax || ky > kEwaldKMax || kz > kEwaldKMax)
            fail("ERROR[404] 'elec pme' needs 1 <= kx, ky, kz <= " + std::to_string(kEwaldKMax) + " k-vectors per axis");
        for (int k = 0; k < 3; k++) if (!(m.L[k] > 0)) fail("ERROR[008] box lengths must be positive");
        if (!(m.alpha > 0)) fail("ERROR[404] 'elec pme' needs a positive alpha");
        const double twopi = 2.0 * units::pi, sqrtpi = std::sqrt(units::pi);
        const double ra = 1.0 / m.L[0], rb = 1.0 / m.L[1], rc = 1.0 / m.L[2], rvol = 1.0 / (m.L[0] * m.L[1] * m.L[2]);
        m.daipi2 = 2 * m.alpha / sqrtpi;
        m.el_scale = 2 * twopi * rvol * units::Fcoul_scale / m.eps;          // elec.cpp:380
        m.el_scale2 = 2 * m.el_scale;
        m.mr4a2 = -0.25 / m.alpha / m.alpha;
        // ip1..ip3 come out of prepare_box's general cell-matrix algebra even for a rectangular box (box.cpp:92-151): 1/la, 1/lb,
        // 1/lc up to rounding.  Followed operation by operation: boxes built from a lattice put k-vectors exactly ON the cut-off
        // sphere, and there the last bit decides whether the reference includes them.
        double ip1, ip2, ip3;
        {
            const double la = m.L[0], lb = m.L[1], lc = m.L[2];
            const double axb3 = la * lb, bxc1 = lb * lc, cxa2 = la * lc;
            const double vol = la * lb * lc, det = la * bxc1, rdet = 1.0 / det, rv = 1.0 / vol;
            const double iax = rdet * bxc1, iby = rdet * cxa2, icz = rdet * axb3;
            const double iaxb3 = iax * iby, ibxc1 = iby * icz, icxa2 = iax * icz;
            ip1 = rv / std::sqrt(ibxc1 * ibxc1); ip2 = rv / std::sqrt(icxa2 * icxa2); ip3 = rv / std::sqrt(iaxb3 * iaxb3);
        }
        double rkcut = kx * ip1;
        if (rkcut > ky * ip2) rkcut = ky * ip2;
        if (rkcut > kz * ip3) rkcut = kz * ip3;
        rkcut *= twopi * 1.05;
        m.rkcut2 = rkcut * rkcut;
        // the k-vector list in the order ewald_rec visits it (half space: l >= 0; m >= 0 when l == 0; n >= 1 when l == m == 0);
        // same table the GPU reference builds on the host, cuInit.cu:1017-1046
        int mmin = 0, nmin = 1;
        for (int l = 0; l < kx; l++)
        {
            const double rkx = l * twopi * ra;
            for (int mm = mmin; mm < ky; mm++)
            {
                const double rky = mm * twopi * rb;
                for (int n = nmin; n < kz; n++)
                {
                    const double rkz = n * twopi * rc;
                    const double rk2 = rkx * rkx + rky * rky + rkz * rkz;
                    if (rk2 < m.rkcut2) m.kvecs.push_back(KVec{l, mm, n, rkx, rky, rkz, std::exp(rk2 * m.mr4a2) / rk2});
                }
                nmin = 1 - kz;
            }
            mmin = 1 - ky;
        }
        double sq = 0.0, eng = 0.0;                                            // ewald_const: elec.cpp:144-164
        for (int i = 0; i < m.nAt; i++) { const double q = m.species[m.types[i]].charge; sq += q; eng += q * q; }
        eng *= (-1.0) * m.alpha / sqrtpi;
        const double q = -0.5 * units::pi * (sq * sq / m.alpha / m.alpha) * rvol;
        m.engElec1 = units::Fcoul_scale * (eng + q) / m.eps;
    }
    if (m.tstat_type == AZTOT_TSTAT_NOSE && !(m.tau > 0)) fail("ERROR[405] 'nose' needs a positive relaxation time");
    if (m.elec_type == AZTOT_ELEC_FENNEL)
    {
        const double sqrtpi = std::sqrt(units::pi);
        double aRc = m.alpha * m.rReal;
        m.daipi2 = 2 * m.alpha / sqrtpi;
        m.el_scale = std::erfc(aRc) / m.rReal;
        m.el_scale2 = std::erfc(aRc) / m.r2Real + m.daipi2 * std::exp(-aRc * aRc) / m.rReal;
    }
    if (m.tSt <= 0) fail("ERROR[411] timestep must be positive");
    for (auto& s : m.species)
    {
        if (s.mass <= 0) fail("species '" + s.name + "' has non-positive mass");
        s.rMass_hdt = 0.5 * m.tSt / s.mass;                    // sys_init.cpp:1056-1057
    }
    // maximal cut-off: rReal REPLACES the VdW range when electrostatics are on (sys_init.cpp:1060-1071, SURVEY C-3)
    m.rMax = 0.0;
    if (m.elec_type) m.rMax = m.rReal;
    else if (m.nVdW) m.rMax = m.maxRvdw;
    m.r2Max = m.rMax * m.rMax;
    if (m.elec_type && m.rReal < m.maxRvdw)
        m.warnings.push_back("WARNING: rReal < max VdW cut-off: VdW is truncated at rReal exactly as the reference does (sys_init.cpp:1061)");
    m.degFree = 3 * m.nAt;
    if (m.tstat_type) m.degFree--;                            // sys_init.cpp:1099-1103
    m.revDegFree = (double)(1.0 / m.degFree);
    m.tKin = 0.5 * m.Temp * units::kB * m.degFree;            // sys_init.cpp:1106
    if (m.nEq && m.freqEq <= 0) { m.warnings.push_back("WARNING[003] no t-Scale during equilibration period"); m.nEq = 0; }
    for (int k = 0; k < 3; k++) if (!(m.L[k] > 0)) fail("ERROR[008] box lengths must be positive");
    if (m.rMax > 0 && (2 * m.rMax > m.L[0] || 2 * m.rMax > m.L[1] || 2 * m.rMax > m.L[2]))
        fail("cut-off exceeds half the box: the minimum-image convention (box.cpp:180) would be invalid");

    // initial velocities
    const int N = m.nAt;
    if (m.init_vel == AZTOT_VEL_CONST)
        for (int i = 0; i < N; i++) { m.vx[i] = m.init_vel_par[0]; m.vy[i] = m.init_vel_par[1]; m.vz[i] = m.init_vel_par[2]; }
    else if (m.init_vel == AZTOT_VEL_KENG)
    {   // sys_init.cpp:769-795: |v| from the kinetic energy, direction on a 32 x 32 angular grid
        const double twopi = 2.0 * units::pi;
        for (int i = 0; i < N; i++)
        {
            double vel = std::sqrt(2.0 * m.init_vel_par[0] / m.species[m.types[i]].mass);
            double phi = double(rng_draw(seed, kRngStreamInitVel, (uint64_t)i, 0) % 32) / 32.0 * twopi;
            double theta = double(rng_draw(seed, kRngStreamInitVel, (uint64_t)i, 1) % 32) / 32.0 * twopi;
            double cost = std::cos(theta);
            m.vz[i] = std::sin(theta) * vel;
            m.vy[i] = std::sin(phi) * cost * vel;
            m.vx[i] = std::cos(phi) * cost * vel;
        }
    }
    else if (m.init_vel == AZTOT_VEL_GAUSS)
    {   // gauss_temp: temperature.cpp:262-337
        double cp[3] = {0, 0, 0}, totMass = 0.0;
        for (int i = 0; i < N; i++)
        {
            uint64_t draw = 0;
            m.vx[i] = gauss(0.5, 0.0, seed, (uint64_t)i, &draw);
            m.vy[i] = gauss(0.5, 0.0, seed, (uint64_t)i, &draw);
            m.vz[i] = gauss(0.5, 0.0, seed, (uint64_t)i, &draw);
            double mass = m.species[m.types[i]].mass;
            cp[0] += m.vx[i] * mass; cp[1] += m.vy[i] * mass; cp[2] += m.vz[i] * mass;
            totMass += mass;
        }
        for (int k = 0; k < 3; k++) cp[k] /= totMass;
        double kE = 0.0;
        for (int i = 0; i < N; i++)
        {
            m.vx[i] -= cp[0]; m.vy[i] -= cp[1]; m.vz[i] -= cp[2];
            kE += m.species[m.types[i]].mass * (m.vx[i] * m.vx[i] + m.vy[i] * m.vy[i] + m.vz[i] * m.vz[i]);
        }
        kE *= 0.5;
        double k = std::sqrt(m.tKin / kE);
        for (int i = 0; i < N; i++) { m.vx[i] *= k; m.vy[i] *= k; m.vz[i] *= k; }
    }
}

void init_md(const std::string& dir, Model& m)
{
    read_field(dir, m);
    read_atoms_box(dir, m);
    read_bonded_lists(dir, m);
    read_sim(dir, m);
    read_cuda(dir, m);
}

void model_from_system(const aztot_system& sys, Model& m)
{
    if (sys.n_atoms <= 0 || sys.n_species <= 0 || sys.n_species > kMaxSpecies) fail("bad atom/species count");
    if (!sys.types || !sys.x || !sys.y || !sys.z || !sys.species) fail("null array in aztot_system");
    m.nAt = sys.n_atoms;
    for (int k = 0; k < 3; k++) m.L[k] = sys.box[k];
    m.species.resize(sys.n_species);
    m.charged_spec = 0;
    for (int i = 0; i < sys.n_species; i++)
    {
        const aztot_species& a = sys.species[i];
        Species& s = m.species[i];
        char nm[9]; std::memcpy(nm, a.name, 8); nm[8] = 0;
        s.name = nm; s.nucleus = nm;
        s.mass_amu = a.mass_amu; s.mass = a.mass_amu * units::m_scale; s.charge = a.charge;
        s.charged = std::fabs(s.charge) < 1.0E-10 ? 0 : 1;
        if (s.charge != 0.0) m.charged_spec = 1;
        s.frozen = a.frozen; s.radA = a.radA; s.radB = a.radB; s.mxEng = a.mxEng;
        if (a.radA != 0.0 || a.radB != 0.0) m.has_radii = 1;
    }
    m.types.assign(sys.types, sys.types + sys.n_atoms);
    for (int i = 0; i < sys.n_atoms; i++)
    {
        if (m.types[i] < 0 || m.types[i] >= sys.n_species) fail("ERROR[009] atom type out of range");
        m.species[m.types[i]].number++;
    }
    m.x.assign(sys.x, sys.x + sys.n_atoms); m.y.assign(sys.y, sys.y + sys.n_atoms); m.z.assign(sys.z, sys.z + sys.n_atoms);
    if (sys.vx && sys.vy && sys.vz)
    {
        m.vx.assign(sys.vx, sys.vx + sys.n_atoms); m.vy.assign(sys.vy, sys.vy + sys.n_atoms); m.vz.assign(sys.vz, sys.vz + sys.n_atoms);
    }
    else { m.vx.assign(sys.n_atoms, 0.0); m.vy.assign(sys.n_atoms, 0.0); m.vz.assign(sys.n_atoms, 0.0); }
    m.pairpots.assign((size_t)sys.n_species * sys.n_species, PairPot());
    m.nVdW = sys.n_vdw; m.minRvdw = 999999.9; m.maxRvdw = 0.0;
    for (int k = 0; k < sys.n_vdw; k++)
    {
        const aztot_vdw& v = sys.vdw[k];
        if (v.spec_a < 0 || v.spec_a >= sys.n_species || v.spec_b < 0 || v.spec_b >= sys.n_species) fail("ERROR[005] unknown atom type in vdw entry");
        PairPot pp = prepare_vdw(v.type, v.rcut, v.p);
        if (pp.rcut < m.minRvdw) m.minRvdw = pp.rcut;
        if (pp.rcut > m.maxRvdw) m.maxRvdw = pp.rcut;
        m.pairpots[(size_t)v.spec_a * sys.n_species + v.spec_b] = pp;
        if (v.type != AZTOT_VDW_SURK) m.pairpots[(size_t)v.spec_b * sys.n_species + v.spec_a] = pp;
    }
    const aztot_control& c = sys.control;
    m.tSt = c.timestep; m.nSt = c.nstep; m.tSim = double(c.nstep * c.timestep);
    m.nEq = c.nequil; m.freqEq = c.nequil ? c.eqfreq : 0;
    m.Temp = c.temperature; m.tstat_type = c.tstat_type; m.tau = c.tstat_tau;
    m.elec_type = c.elec_type; m.rReal = c.r_real; m.alpha = c.alpha;
    for (int k = 0; k < 3; k++) m.ewald_k[k] = c.ewald_k[k];
    if (m.elec_type == AZTOT_ELEC_NONE) m.rReal = 0.0;
    if (!m.charged_spec && m.elec_type) m.elec_type = AZTOT_ELEC_NONE;
    m.r2Real = m.rReal * m.rReal;
    m.init_vel = (sys.vx && sys.vy && sys.vz) ? -1 : c.init_vel;    // explicit velocities win
    for (int k = 0; k < 3; k++) { m.init_vel_par[k] = c.init_vel_par[k]; m.E[k] = c.elecfield[k]; }
    m.use_clist = c.use_cell_list; m.desired_cell_size = c.cell_list;
    m.stat = c.stat > 0 ? c.stat : 1000;
}

}  // namespace aztot
