/* TEST INFRASTRUCTURE - CPU oracle for the azTotMD per-step hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (aztotmd_amd/) never links, imports or
 * executes anything in oracle/.
 *
 * This file is a from-scratch fp64 restatement (plain C99) of the reference's serial
 * algorithm for the path
 *     cell-list build -> pair VdW + short-range Coulomb -> velocity Verlet (+ thermostat)
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/src).  Where the serial program has no implementation (radiative
 * thermostat, elin/einv/surk potentials) the CUDA source text is followed in fp64 and the
 * header of the function says so.
 *
 * PARITY PINNING: pair functions, min-image, wrap, linked-cell table/traversal, integrator
 * and energy bookkeeping are pinned against the reference's own compiled translation units
 * (oracle/_ref, built by oracle/Makefile from the sources where they lie) by
 * tests/test_oracle_vs_ref.py and by the committed fixtures in tests/golden/ that the same
 * binary generated (tests/golden/make_golden.py).  The parser/derived-parameter chain is
 * pinned by the known-answer table of SURVEY.md Appendix F (tests/golden/appendix_f.json).
 * The radiative thermostat, elin/einv/surk are "parity unpinned" by the reference (it ships
 * no CPU implementation, no tests and no outputs for them): see DESIGN.md.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- constants: const.h:11-49 */
static const double ORC_PI = 3.14159265359;               /* const.h:11 (truncated on purpose) */
#define ORC_E_SI 1.60217733E-19
#define ORC_Q_SI 1.60217657E-19
#define ORC_KB_SI 1.3806488E-23
#define ORC_E0_SI 8.854187817E-12
#define ORC_AMU_SI 1.6605402E-27
#define ORC_R_SI 1.0E-10
#define ORC_T_SI 1.0E-12

static double c_m_scale(void) { double m_SI = ORC_E_SI * ORC_T_SI * ORC_T_SI / ORC_R_SI / ORC_R_SI; return ORC_AMU_SI / m_SI; }   /* const.h:27,43 */
static double c_kB(void) { return ORC_KB_SI / (1.0 * ORC_E_SI); }                                                                 /* const.h:47 */
static double c_Fcoul(void)
{   /* const.h:28-29,44 */
    double F_SI = ORC_E_SI / ORC_R_SI;
    double Fcoul_SI = 0.25 / ORC_PI / ORC_E0_SI * ORC_Q_SI * ORC_Q_SI / ORC_R_SI / ORC_R_SI;
    return Fcoul_SI / F_SI;
}

enum { VDW_NONE = 0, VDW_LJ = 1, VDW_BUCK = 2, VDW_746 = 3, VDW_BHM = 4, VDW_ELIN = 5, VDW_EINV = 6, VDW_SURK = 7 };  /* vdw.h:15-21 */
enum { ELEC_NONE = 0, ELEC_DIR = 1, ELEC_EWALD = 2, ELEC_FENNEL = 3 };                                                 /* elec.h:8-12 */
enum { TSTAT_NONE = 0, TSTAT_NOSE = 1, TSTAT_RADI = 2 };                                                               /* temperature.h:10-12 */
#define ORC_NUVECT 3072                                                                                                /* cuTemp.h:4 */

typedef struct { int type; int use_radii; double p0, p1, p2, p3, p4, r2cut; } orc_vdw;
typedef struct { int type, spec1, spec2; double p0, p1, p2, p3, p4; } orc_bond;      /* Bond, dataStruct.h:321-339 (constant bonds only) */
typedef struct { int type, central; double p0, p1; } orc_angle;                      /* Angle, dataStruct.h:341-346 */

typedef struct
{
    int N, nSpec;
    double la, lb, lc, ha, hb, hc, ra, rb, rc_;          /* Box: dataStruct.h:205-241 */
    double dt;
    int *types;
    double *x, *y, *z, *vx, *vy, *vz, *fx, *fy, *fz;
    /* species: dataStruct.h:244-291 */
    double *mass, *charge, *rMass_hdt, *radA, *radB, *mxEng; int *charged; int *frozen;
    orc_vdw *vdw;                                        /* nSpec*nSpec, type 0 = none (NULL pointer in the reference) */
    double maxRvdw;
    /* elec: dataStruct.h:349-366 */
    int elec_type; double rReal, r2Real, alpha, el_scale, el_scale2, daipi2;
    int kx, ky, kz; double mr4a2, rkcut2, engElec1, engElec2;      /* Ewald sum: elec.cpp:371-397 */
    double *ew;                                                      /* per-atom work arrays of init_ewald, elec.cpp:69-129 */
    double rMax, r2Max;
    double Ux, Uy, Uz;
    /* thermostat + control */
    int tstat_type; double Temp, tKin; int degFree; int nEq, freqEq;
    double tau, chit, conint, rQmass, qMassTau2;         /* Nose-Hoover: temperature.h:24-31 */
    int use_clist;                                       /* control.txt 'cell_list' present */
    /* linked cells: Sim fields dataStruct.h:88-99 */
    int nHead, cnX, cnY, cnZ, cnYZ; double clX, clY, clZ;
    int *clist, *chead, *neig;                           /* neig[nHead*13] */
    /* energies / counters */
    double engVdW, engElec3, engKin, engTot, engElecField, engTemp, TempNow;
    double mom[6];                                       /* Xn, Xp, Yn, Yp, Zn, Zp: box.cpp:230-295 */
    long long cross[6];
    long long specCross[16 * 6];                         /* per species: specAcBoxNeg/Pos of put_periodic, cuMDfunc.cu:35-106 (msd.dat) */
    long long nDropped;                                  /* pairs dropped by the f^2 > 1e10 rule, integrators.cpp:170 */
    int iStep;                                           /* 1-based index of the last completed step (main.cpp:92) */
    /* radiative thermostat state (cuStruct.h:250-255,335-339,384-385) */
    double *U, *rad, *photons; double *uvx, *uvy, *uvz; int *pid; uint64_t seed;
    double revLight, radiate_frac, radiate_thr, numPi;
    /* bonded terms: Field fields dataStruct.h:389-410 ; index 0 of both type tables is the reserved 'none' */
    int nBdata, nAdata, nBonds, nAngles;
    orc_bond *bdata; orc_angle *adata;
    int *at1, *at2, *bTypes, *centrs, *lig1, *lig2, *angTypes;
    double engBond, engAngle;
} orc_sys;

/* ---------------------------------------------------------------- counter-based RNG (ours: SURVEY C-11/E) */
static uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}
uint32_t orc_rng(uint64_t seed, uint64_t step, uint64_t id, uint64_t draw)
{
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1));
    z = mix64(z ^ (0xD1B54A32D192ED03ULL * (id + 1)));
    z = mix64(z ^ (0x8CB92BA72F3D8DD7ULL * (draw + 1)));
    return (uint32_t)(z >> 32);
}
#define RNG_STREAM_TABLES 0xFFFFFFFFFFFFFFF0ULL   /* "step" value reserved for table generation */

/* ---------------------------------------------------------------- lifecycle */
orc_sys *orc_create(int N, int nSpec, const double *box, const int *types,
                    const double *x, const double *y, const double *z,
                    const double *vx, const double *vy, const double *vz)
{
    orc_sys *s = (orc_sys *)calloc(1, sizeof(orc_sys));
    s->N = N; s->nSpec = nSpec;
    s->la = box[0]; s->lb = box[1]; s->lc = box[2];
    /* prepare_box: box.cpp:73-86 */
    s->ha = s->la * 0.5; s->hb = s->lb * 0.5; s->hc = s->lc * 0.5;
    s->ra = 1.0 / s->la; s->rb = 1.0 / s->lb; s->rc_ = 1.0 / s->lc;
    size_t nb = sizeof(double) * (size_t)N;
    s->types = (int *)malloc(sizeof(int) * (size_t)N); memcpy(s->types, types, sizeof(int) * (size_t)N);
    double **dst[] = { &s->x, &s->y, &s->z, &s->vx, &s->vy, &s->vz, &s->fx, &s->fy, &s->fz, &s->U, &s->rad };
    const double *src[] = { x, y, z, vx, vy, vz, NULL, NULL, NULL, NULL, NULL };
    for (int k = 0; k < 11; k++) { *dst[k] = (double *)calloc((size_t)N, sizeof(double)); if (src[k]) memcpy(*dst[k], src[k], nb); }
    s->pid = (int *)malloc(sizeof(int) * (size_t)N); for (int i = 0; i < N; i++) s->pid[i] = i;
    s->mass = (double *)calloc(nSpec, 8); s->charge = (double *)calloc(nSpec, 8); s->rMass_hdt = (double *)calloc(nSpec, 8);
    s->radA = (double *)calloc(nSpec, 8); s->radB = (double *)calloc(nSpec, 8); s->mxEng = (double *)calloc(nSpec, 8);
    s->charged = (int *)calloc(nSpec, sizeof(int)); s->frozen = (int *)calloc(nSpec, sizeof(int));
    s->vdw = (orc_vdw *)calloc((size_t)nSpec * nSpec, sizeof(orc_vdw));
    s->revLight = 3.33567e-5;      /* cuTemp.cu:225 (100x the physical 1/c: SURVEY C-19, kept) */
    s->radiate_frac = 0.9;         /* cuTemp.cu:639 */
    s->radiate_thr = 1e-4;         /* cuTemp.cu:747 */
    s->numPi = 3.14159;            /* cuTemp.cu:228 */
    s->seed = 12345;
    return s;
}

void orc_free(orc_sys *s)
{
    if (!s) return;
    free(s->types); free(s->x); free(s->y); free(s->z); free(s->vx); free(s->vy); free(s->vz);
    free(s->fx); free(s->fy); free(s->fz); free(s->U); free(s->rad); free(s->pid);
    free(s->mass); free(s->charge); free(s->rMass_hdt); free(s->radA); free(s->radB); free(s->mxEng);
    free(s->charged); free(s->frozen); free(s->vdw);
    free(s->clist); free(s->chead); free(s->neig);
    free(s->photons); free(s->uvx); free(s->uvy); free(s->uvz);
    free(s->ew);
    free(s->bdata); free(s->adata); free(s->at1); free(s->at2); free(s->bTypes);
    free(s->centrs); free(s->lig1); free(s->lig2); free(s->angTypes);
    free(s);
}

/* read_spec: sys_init.cpp:83-130 (mass in amu, charge in e) ; radii section sys_init.cpp:468-480 */
void orc_set_species(orc_sys *s, int i, double mass_amu, double charge, int frozen, double radA, double radB, double mxEng)
{
    s->mass[i] = mass_amu * c_m_scale();
    s->charge[i] = charge * 1.0;
    s->charged[i] = fabs(s->charge[i]) < 1.0E-10 ? 0 : 1;
    s->frozen[i] = frozen;
    s->radA[i] = radA; s->radB[i] = radB; s->mxEng[i] = mxEng;
}

/* read_vdw: vdw.cpp:234-308 ; scale tables vdw.cpp:209-219 (r_scale = E_scale = 1) */
int orc_set_vdw(orc_sys *s, int a, int b, int type, double rc, const double *p)
{
    orc_vdw pp; memset(&pp, 0, sizeof(pp));
    pp.type = type;
    if (rc > s->maxRvdw) s->maxRvdw = rc;
    pp.r2cut = rc * rc;
    pp.p0 = p[0]; pp.p1 = p[1]; pp.p2 = p[2]; pp.p3 = p[3]; pp.p4 = p[4];
    switch (type)
    {
    case VDW_LJ:   pp.p0 *= 4; pp.p2 = 0; pp.p3 = 0; pp.p4 = 0;
                   pp.p1 = pp.p1 * pp.p1; pp.p2 = 6 * pp.p0; break;            /* vdw.cpp:283-288 */
    case VDW_BUCK: pp.p3 = 0; pp.p4 = 0; break;
    case VDW_746:  pp.p3 = 0; pp.p4 = 0; break;
    case VDW_BHM:  break;
    case VDW_ELIN: pp.p3 = 0; pp.p4 = 0; break;
    case VDW_EINV: pp.p3 = 0; pp.p4 = 0; break;
    case VDW_SURK: pp.p4 = 0; pp.use_radii = 1; break;                         /* vdw.cpp:289-299 */
    default: return 0;
    }
    s->vdw[a * s->nSpec + b] = pp;
    if (type != VDW_SURK) s->vdw[b * s->nSpec + a] = pp;                       /* vdw.cpp:303-307 */
    return 1;
}

/* read_elec elec.cpp:14-67 */
void orc_set_elec(orc_sys *s, int type, double rReal, double alpha)
{
    s->elec_type = type; s->rReal = rReal; s->alpha = alpha;
}

void orc_set_nose(orc_sys *s, double tau) { s->tau = tau; }
/* 'elec pme rReal alpha kx ky kz' (read_elec elec.cpp:33-38): numbers of k-vectors per axis, exclusive upper bounds */
void orc_set_ewald(orc_sys *s, int kx, int ky, int kz) { s->kx = kx; s->ky = ky; s->kz = kz; }

void orc_set_control(orc_sys *s, double dt, double T, int tstat_type, int nEq, int freqEq, int use_clist,
                     double Ux, double Uy, double Uz, uint64_t seed)
{
    s->dt = dt; s->Temp = T; s->tstat_type = tstat_type; s->nEq = nEq; s->freqEq = freqEq; s->use_clist = use_clist;
    s->Ux = Ux; s->Uy = Uy; s->Uz = Uz; s->seed = seed;
}

/* ---------------------------------------------------------------- linked cells: cell_list.cpp:24-134 */
static int cell_wrap_index(int x, int y, int z, int mx, int my, int mz)
{   /* cell_index: cell_list.cpp:22-46 */
    int i = x, j = y, k = z;
    if (i >= mx) i = 0; else if (i < 0) i = mx - 1;
    if (j >= my) j = 0; else if (j < 0) j = my - 1;
    if (k >= mz) k = 0; else if (k < 0) k = mz - 1;
    return i * my * mz + j * mz + k;
}

static int init_clist(orc_sys *s, double rCut)
{   /* init_clist: cell_list.cpp:48-134.  Only the "all dimensions full" geometry (:107-134) is restated;
       slab/stack geometries (:138-262) fall back to all-pairs, which visits the same pairs. */
    int nX = (int)floor(s->la / rCut), nY = (int)floor(s->lb / rCut), nZ = (int)floor(s->lc / rCut);
    if (nX == 0) nX = 1; if (nY == 0) nY = 1; if (nZ == 0) nZ = 1;
    s->cnX = nX; s->cnY = nY; s->cnZ = nZ; s->nHead = 0;
    if ((nX < 4) && (nY < 4) && (nZ < 4)) return 0;                              /* :74-75 */
    if (!((nX > 2) && (nY > 2) && (nZ > 2))) return 0;
    int nYZ = nY * nZ; s->cnYZ = nYZ;
    s->clX = s->la / nX; s->clY = s->lb / nY; s->clZ = s->lc / nZ;
    s->nHead = nX * nY * nZ;
    s->clist = (int *)malloc(sizeof(int) * (size_t)s->N);
    s->chead = (int *)malloc(sizeof(int) * (size_t)s->nHead);
    s->neig = (int *)malloc(sizeof(int) * 13 * (size_t)s->nHead);
    static const int off[13][3] = {                                            /* order of :119-133 */
        { 1, 1, 1 }, { 1, 0, 1 }, { 1, -1, 1 }, { 0, 1, 1 }, { 0, 0, 1 }, { 0, -1, 1 }, { -1, 1, 1 }, { -1, 0, 1 }, { -1, -1, 1 },
        { 1, 1, 0 }, { 1, 0, 0 }, { 1, -1, 0 }, { 0, 1, 0 } };
    for (int i = 0; i < nX; i++) for (int j = 0; j < nY; j++) for (int k = 0; k < nZ; k++)
    {
        int ci = i * nYZ + j * nZ + k;
        for (int m = 0; m < 13; m++) s->neig[ci * 13 + m] = cell_wrap_index(i + off[m][0], j + off[m][1], k + off[m][2], nX, nY, nZ);
    }
    return s->nHead;
}

int orc_cells(const orc_sys *s, int *dims) { dims[0] = s->cnX; dims[1] = s->cnY; dims[2] = s->cnZ; return s->nHead; }
const int *orc_neig_table(const orc_sys *s) { return s->neig; }

/* ---------------------------------------------------------------- photon / unit-vector tables */
static double prob4(double x, double y, double theta)
{   /* temperature.cpp:18-26 */
    const double r24 = 1.0 / 24.0, r6 = 1.0 / 6.0;
    double ty = theta * y, ty2 = ty * ty;
    return (1 - x) * exp(y * theta) - (r24 * ty2 * ty2 + r6 * ty2 * ty + 0.5 * ty * ty + ty + 1);
}

static double rand01_ctr(uint64_t seed, uint64_t id, uint64_t *draw)
{   /* rand01: utils.cpp:197-201 (1e-4 resolution kept), libc rand() replaced by the counter RNG */
    uint32_t r = orc_rng(seed, RNG_STREAM_TABLES, id, (*draw)++);
    return (double)(r % 10000) / 10000;
}

void orc_photon_engs(int n, double *engs, double T, uint64_t seed)
{   /* photon_engs: temperature.cpp:28-89 */
    const double eps = 1e-3; const int limit = 20;
    double theta = 1.0 / (c_kB() * T);
    for (int i = 0; i < n; i++)
    {
        uint64_t draw = 0;
        double a = 0.0, b = 1.0, x, ra, rb, y, r;
        do { x = rand01_ctr(seed, (uint64_t)i, &draw); ra = prob4(x, 0.0, theta); rb = prob4(x, 1.0, theta); } while (ra * rb > 0);
        y = 0.5; r = prob4(x, y, theta);
        int k = 0;
        while ((r > eps) || (r < -eps))
        {
            if ((r * ra) < 0) { b = y; y = 0.5 * (a + y); } else { a = y; y = 0.5 * (y + b); }
            r = prob4(x, y, theta);
            k++;
            if (k >= limit) { if (i > 0) y = engs[i - 1]; break; }   /* :76-80 ; i == 0 would read engs[-1] (SURVEY C-21): keep y */
        }
        engs[i] = y;
    }
}

void orc_unit_vectors(double *ux, double *uy, double *uz)
{   /* temperature.cpp:165-223: 32 phi x 16 theta, +/- pairs, three axis permutations */
    const int nTh = 16, nPhi = 32; int k = 0;
    const double twopi = 2.0 * ORC_PI;
    for (int perm = 0; perm < 3; perm++)
        for (int i = 0; i < nPhi; i++)
        {
            double phi = (double)i / nPhi * twopi;
            for (int j = 0; j < nTh; j++)
            {
                double theta = (double)j / nTh * ORC_PI;
                double st = sin(theta), ct = cos(theta), sp = sin(phi), cp = cos(phi);
                double a = cp * ct, b = sp * ct, c = st;     /* (cos t cos p, cos t sin p, sin t) */
                double X, Y, Z;
                if (perm == 0) { X = a; Y = b; Z = c; }        /* :174-178 */
                else if (perm == 1) { X = a; Z = b; Y = c; }   /* :192-196 */
                else { Z = a; Y = b; X = c; }                  /* :210-214 */
                ux[k] = X; uy[k] = Y; uz[k] = Z;
                ux[k + 1] = -X; uy[k + 1] = -Y; uz[k + 1] = -Z;
                k += 2;
            }
        }
}

/* ---------------------------------------------------------------- init_md derived parameters: sys_init.cpp:1053-1112 */
int orc_prepare(orc_sys *s)
{
    int any_charged = 0;
    for (int i = 0; i < s->nSpec; i++) if (s->charge[i] != 0.0) any_charged = 1;      /* sys_init.cpp:210 */
    if (!any_charged && s->elec_type) s->elec_type = ELEC_NONE;                         /* elec.cpp:52-56 */
    if (s->elec_type == ELEC_NONE && 0) s->rReal = 0.0;
    s->r2Real = s->rReal * s->rReal;
    if (s->elec_type == ELEC_FENNEL)
    {   /* prepare_elec: elec.cpp:399-405 */
        double sqrtpi = sqrt(ORC_PI);
        double aRc = s->alpha * s->rReal;
        s->daipi2 = 2 * s->alpha / sqrtpi;
        s->el_scale = erfc(aRc) / s->rReal;
        s->el_scale2 = erfc(aRc) / s->r2Real + s->daipi2 * exp(-aRc * aRc) / s->rReal;
    }
    s->engElec1 = 0.0; s->engElec2 = 0.0;
    if (s->elec_type == ELEC_EWALD)
    {   /* prepare_elec: elec.cpp:377-397 (eps = 1) ; ewald_const :144-164 */
        const double twopi = 2.0 * ORC_PI, sqrtpi = sqrt(ORC_PI), rvol = 1.0 / (s->la * s->lb * s->lc);
        s->daipi2 = 2 * s->alpha / sqrtpi;
        s->el_scale = 2 * twopi * rvol * c_Fcoul() / 1.0;
        s->el_scale2 = 2 * s->el_scale;
        s->mr4a2 = -0.25 / s->alpha / s->alpha;
        /* ip1..ip3 ("perpendicular widths" for the k cut-off) come out of prepare_box's general cell-matrix algebra even for a
           rectangular box (box.cpp:92-151): equal to 1/la, 1/lb, 1/lc up to rounding - restated literally because lattice-built
           boxes put k-vectors EXACTLY on the cut-off sphere, where the last bit decides membership */
        double ip1, ip2, ip3;
        {
            const double la = s->la, lb = s->lb, lc = s->lc;
            const double axb3 = la * lb, bxc1 = lb * lc, cxa2 = la * lc;                 /* box.cpp:92-104 */
            const double vol = la * lb * lc, det = la * bxc1, rdet = 1.0 / det, rv = 1.0 / vol;   /* :110-119 */
            const double iax = rdet * bxc1, iby = rdet * cxa2, icz = rdet * axb3;          /* :123-133 */
            const double iaxb3 = iax * iby, ibxc1 = iby * icz, icxa2 = iax * icz;          /* :136-146 */
            ip1 = rv / sqrt(ibxc1 * ibxc1); ip2 = rv / sqrt(icxa2 * icxa2); ip3 = rv / sqrt(iaxb3 * iaxb3);   /* :149-151 */
        }
        double rkcut = s->kx * ip1;
        if (rkcut > s->ky * ip2) rkcut = s->ky * ip2;
        if (rkcut > s->kz * ip3) rkcut = s->kz * ip3;
        rkcut *= twopi * 1.05;
        s->rkcut2 = rkcut * rkcut;
        double sq = 0.0, eng = 0.0;
        for (int i = 0; i < s->N; i++) { double q = s->charge[s->types[i]]; sq += q; eng += q * q; }
        eng *= (-1.0) * s->alpha / sqrtpi;
        double q = -0.5 * ORC_PI * (sq * sq / s->alpha / s->alpha) * rvol;
        s->engElec1 = c_Fcoul() * (eng + q) / 1.0;
        free(s->ew);
        s->ew = (double *)malloc(sizeof(double) * (size_t)s->N * (size_t)(2 * (2 + s->ky + s->kz) + 4));
    }
    for (int i = 0; i < s->nSpec; i++) s->rMass_hdt[i] = 0.5 * s->dt / s->mass[i];    /* :1056-1057 */
    s->rMax = 0.0;
    if (s->elec_type) s->rMax = s->rReal; else s->rMax = s->maxRvdw;                  /* :1060-1071 */
    s->r2Max = s->rMax * s->rMax;
    s->degFree = 3 * s->N; if (s->tstat_type) s->degFree--;                             /* :1099-1103 */
    s->tKin = 0.5 * s->Temp * c_kB() * s->degFree;                                      /* :1106 */
    if (s->tstat_type == TSTAT_NOSE)
    {   /* sys_init.cpp:1107-1112 ; read_tstat temperature.cpp:103-111 */
        s->rQmass = 0.5 / s->tKin / s->tau / s->tau;
        s->qMassTau2 = 2 * s->tKin;
        s->chit = 0.0; s->conint = 0.0;
    }
    if (s->use_clist) init_clist(s, s->rMax);                                           /* :1130-1133 */
    if (s->tstat_type == TSTAT_RADI)
    {   /* read_tstat temperature.cpp:113-245 ; init_cuda_tstat cuTemp.cu:25-60 */
        s->photons = (double *)malloc(8 * (size_t)s->N);
        orc_photon_engs(s->N, s->photons, s->Temp, s->seed);
        s->uvx = (double *)malloc(8 * ORC_NUVECT); s->uvy = (double *)malloc(8 * ORC_NUVECT); s->uvz = (double *)malloc(8 * ORC_NUVECT);
        orc_unit_vectors(s->uvx, s->uvy, s->uvz);
        for (int i = 0; i < s->N; i++)
        {
            uint64_t draw = 1000;
            s->U[i] = 0.0;
            s->rad[i] = 0.577 + rand01_ctr(s->seed, (uint64_t)i, &draw) * 0.0001;      /* cuTemp.cu:41 */
        }
    }
    return 1;
}

/* center_box: box.cpp:337-384 (serial path only, sys_init.cpp:1145) */
void orc_center_box(orc_sys *s)
{
    double mxx = 0, mxy = 0, mxz = 0, mnx = s->la, mny = s->lb, mnz = s->lc;
    for (int i = 0; i < s->N; i++)
    {
        if (s->x[i] > mxx) mxx = s->x[i]; if (s->x[i] < mnx) mnx = s->x[i];
        if (s->y[i] > mxy) mxy = s->y[i]; if (s->y[i] < mny) mny = s->y[i];
        if (s->z[i] > mxz) mxz = s->z[i]; if (s->z[i] < mnz) mnz = s->z[i];
    }
    double dx = 0.5 * (mxx - mnx) - s->ha, dy = 0.5 * (mxy - mny) - s->hb, dz = 0.5 * (mxz - mnz) - s->hc;
    for (int i = 0; i < s->N; i++) { s->x[i] -= dx; s->y[i] -= dy; s->z[i] -= dz; }
}

/* ---------------------------------------------------------------- pair potentials */
/* Returns f = -(1/r) dU/dr and adds U to *eng.  r is "0 = unknown" as in the reference. */
static double vdw_fer(const orc_vdw *v, double r2, double *r, double radi, double radj, double *eng)
{
    switch (v->type)
    {
    case VDW_LJ:
    {   /* fer_lj: vdw.cpp:16-26 */
        double r2i = 1.0 / r2;
        double sr2 = v->p1 * r2i;
        double sr6 = sr2 * sr2 * sr2;
        *eng += v->p0 * sr6 * (sr6 - 1.0);
        return v->p2 * r2i * sr6 * (2.0 * sr6 - 1.0);
    }
    case VDW_BUCK:
    {   /* fer_buckingham: vdw.cpp:60-70 */
        double r2i = 1.0 / r2, r4i = r2i * r2i;
        if (*r == 0.0) *r = sqrt(r2);
        *eng += v->p0 * exp(-*r / v->p1) - v->p2 * r4i * r2i;
        return v->p0 * exp(-*r / v->p1) / *r / v->p1 - 6.0 * v->p2 * r4i * r4i;
    }
    case VDW_BHM:
    {   /* fer_bhm: vdw.cpp:102-112 */
        double r2i = 1.0 / r2, r4i = r2i * r2i;
        if (*r == 0.0) *r = sqrt(r2);
        *eng += v->p0 * exp(v->p1 * (v->p2 - *r)) - v->p3 * r4i * r2i - v->p4 * r4i * r4i;
        return v->p0 * v->p1 * exp(v->p1 * (v->p2 - *r)) / *r - 6.0 * v->p3 * r4i * r4i - 8.0 * v->p4 * r4i * r4i * r2i;
    }
    case VDW_746:
    {   /* fer_746: vdw.cpp:144-157 */
        double r2i = 1.0 / r2, r4i = r2i * r2i, ri;
        if (*r == 0.0) ri = sqrt(r2i); else ri = 1.0 / *r;
        *eng += r4i * (v->p0 * r2i * ri - v->p1 - v->p2 * r2i);
        return r4i * r2i * (7.0 * v->p0 * r2i * ri - 4.0 * v->p1 - 6.0 * v->p2 * r2i);
    }
    case VDW_ELIN:
    {   /* cu_fer_elin: cuVdW.cu:162-171 (fp64 here; no serial implementation, vdw.cpp:204-207) */
        if (*r == 0.0) *r = sqrt(r2);
        *eng += v->p0 * exp(-*r / v->p1) + v->p2 * *r;
        return v->p0 * exp(-*r / v->p1) / *r / v->p1 - v->p2 / *r;
    }
    case VDW_EINV:
    {   /* cu_fer_einv: cuVdW.cu:200-208 */
        if (*r == 0.0) *r = sqrt(r2);
        *eng += v->p0 * exp(-*r / v->p1) - v->p2 / *r;
        return v->p0 * exp(-*r / v->p1) / *r / v->p1 - v->p2 / *r / r2;
    }
    case VDW_SURK:
    {   /* surk_pot: cuVdW.cu:236-257 */
        double c2ir_sum = v->p1 / (v->p2 * radi + v->p3 * radj);
        double r_prod = radi * radj;
        double C1ab2 = r_prod * r_prod * v->p0;
        double r6 = r2 * r2 * r2;
        double rr = sqrt(r2);
        double ir6 = 1.0 / r6, ir = 1.0 / rr;
        *eng += r_prod * ir6 * (C1ab2 * ir - c2ir_sum);
        return r_prod * ir6 / r2 * (7.0 * C1ab2 * ir - 6.0 * c2ir_sum);
    }
    }
    return 0.0;
}

static void pair_elec(orc_sys *s, int it, int jt, double r2, double r, double *force)
{   /* r is passed BY VALUE in the reference (elec.h:16) so VdW still sees r = 0 afterwards */
    if (s->elec_type == ELEC_NONE) return;                                  /* none_elec: elec.cpp:447 */
    if (!s->charged[it] || !s->charged[jt]) return;
    if (s->elec_type == ELEC_DIR)
    {   /* direct_coul: elec.cpp:415-428 */
        double kqq = s->charge[it] * s->charge[jt] * c_Fcoul();
        if (r == 0) r = sqrt(r2);
        s->engElec3 += kqq / r;
        *force += kqq / r / r2;
    }
    else if (s->elec_type == ELEC_FENNEL)
    {   /* fennel: elec.cpp:430-444 */
        if (r == 0) r = sqrt(r2);
        double ir = 1.0 / r;
        double kqq = s->charge[it] * s->charge[jt] * c_Fcoul();
        double ar = s->alpha * r;
        double erfcar = erfc(ar);
        s->engElec3 += kqq * (erfcar * ir - s->el_scale + s->el_scale2 * (r - s->rReal));
        *force += kqq * ir * ((erfcar / r2 + s->daipi2 * exp(-ar * ar) * ir) - s->el_scale2);
    }
    else if (s->elec_type == ELEC_EWALD)
    {   /* direct_ewald -> coul_iter: elec.cpp:408-413, 344-369 (real-space part only) */
        double kqq = s->charge[it] * s->charge[jt] * c_Fcoul();
        if (r == 0) r = sqrt(r2);
        double ar = s->alpha * r, erfcar = erfc(ar);
        s->engElec3 += kqq * erfcar / r;
        *force += kqq / r / r2 * (erfcar + 2 * ar / sqrt(ORC_PI) * exp(-ar * ar));
    }
}

/* known-answer entry points for the unit tests */
double orc_vdw_pair(int type, double rc, const double *p, double r2, double radi, double radj, double *eng_out)
{
    orc_sys *s = orc_create(0, 1, (double[]){ 1, 1, 1 }, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    orc_set_vdw(s, 0, 0, type, rc, p);
    double r = 0.0, e = 0.0;
    double f = vdw_fer(&s->vdw[0], r2, &r, radi, radj, &e);
    *eng_out = e; orc_free(s); return f;
}
double orc_coul_pair(int type, double rReal, double alpha, double qi, double qj, double r2, double *eng_out)
{
    orc_sys *s = orc_create(0, 2, (double[]){ 1, 1, 1 }, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    orc_set_species(s, 0, 1.0, qi, 0, 0, 0, 0); orc_set_species(s, 1, 1.0, qj, 0, 0, 0, 0);
    orc_set_elec(s, type, rReal, alpha); s->dt = 1; orc_prepare(s);
    double f = 0.0; pair_elec(s, 0, 1, r2, 0.0, &f);
    *eng_out = s->engElec3; orc_free(s); return f;
}
void orc_elec_consts(const orc_sys *s, double *out) { out[0] = s->el_scale; out[1] = s->el_scale2; out[2] = s->daipi2; out[3] = s->rMax; out[4] = s->tKin; out[5] = c_kB(); out[6] = c_m_scale(); out[7] = c_Fcoul(); }

/* ---------------------------------------------------------------- box */
static void delta_periodic(const orc_sys *s, double *dx, double *dy, double *dz)
{   /* box.cpp:180-207 */
    if (*dx > s->ha) *dx -= s->la; else if (*dx < -s->ha) *dx += s->la;
    if (*dy > s->hb) *dy -= s->lb; else if (*dy < -s->hb) *dy += s->lb;
    if (*dz > s->hc) *dz -= s->lc; else if (*dz < -s->hc) *dz += s->lc;
}

static void put_periodic(orc_sys *s, int i)
{   /* box.cpp:230-295.  One deviation, documented (SURVEY C-6): a coordinate that lands on
       x >= L after the shift (incl. the serial path's surviving x == L) is set to 0 as the
       GPU path does (cuMDfunc.cu:64-69), so the cell index stays in range. */
    double m = s->mass[s->types[i]];
    if (s->x[i] < 0) { s->x[i] += ((int)(-s->x[i] * s->ra) + 1) * s->la; s->cross[0]++; s->specCross[s->types[i] * 6 + 0]++; s->mom[0] += m * (-s->vx[i]); }
    else if (s->x[i] > s->la) { s->x[i] -= ((int)(s->x[i] * s->ra)) * s->la; s->cross[1]++; s->specCross[s->types[i] * 6 + 1]++; s->mom[1] += m * s->vx[i]; }
    if (s->y[i] < 0) { s->y[i] += ((int)(-s->y[i] * s->rb) + 1) * s->lb; s->cross[2]++; s->specCross[s->types[i] * 6 + 2]++; s->mom[2] += m * (-s->vy[i]); }
    else if (s->y[i] > s->lb) { s->y[i] -= ((int)(s->y[i] * s->rb)) * s->lb; s->cross[3]++; s->specCross[s->types[i] * 6 + 3]++; s->mom[3] += m * s->vy[i]; }
    if (s->z[i] < 0) { s->z[i] += ((int)(-s->z[i] * s->rc_) + 1) * s->lc; s->cross[4]++; s->specCross[s->types[i] * 6 + 4]++; s->mom[4] += m * (-s->vz[i]); }
    else if (s->z[i] > s->lc) { s->z[i] -= ((int)(s->z[i] * s->rc_)) * s->lc; s->cross[5]++; s->specCross[s->types[i] * 6 + 5]++; s->mom[5] += m * s->vz[i]; }
    if (s->x[i] >= s->la) s->x[i] = 0.0;
    if (s->y[i] >= s->lb) s->y[i] = 0.0;
    if (s->z[i] >= s->lc) s->z[i] = 0.0;
}

/* ---------------------------------------------------------------- integrators.cpp */
static void clear_force(orc_sys *s)
{   /* clear_force: integrators.cpp:17-39 with the GPU path's 3-component field (cuMDfunc.cu:478);
       the serial 'shiftX' special-purpose push (:33-36) is out of scope */
    for (int i = 0; i < s->N; i++)
    {
        double q = s->charge[s->types[i]];
        s->fx[i] = -q * s->Ux; s->fy[i] = -q * s->Uy; s->fz[i] = -q * s->Uz;
    }
}

static void reset_chars(orc_sys *s) { s->engVdW = 0.0; s->engElec3 = 0.0; s->engElecField = 0.0; s->engTemp = 0.0; s->engBond = 0.0; s->engAngle = 0.0; }   /* integrators.cpp:42-60 */

static void pair_inter(orc_sys *s, int i, int j)
{   /* pair_inter: integrators.cpp:139-185 ; sqr_distance_proj box.cpp:327-335 */
    double dx = s->x[i] - s->x[j], dy = s->y[i] - s->y[j], dz = s->z[i] - s->z[j];
    delta_periodic(s, &dx, &dy, &dz);
    double r2 = dx * dx + dy * dy + dz * dz;
    if (r2 <= s->r2Max)
    {
        int it = s->types[i], jt = s->types[j];
        double r = 0.0, f = 0.0;
        pair_elec(s, it, jt, r2, r, &f);
        const orc_vdw *v = &s->vdw[it * s->nSpec + jt];
        if (v->type != VDW_NONE)
            if (r2 <= v->r2cut)
                f += vdw_fer(v, r2, &r, s->rad[i], s->rad[j], &s->engVdW);
        if (f * f > 1e10) { s->nDropped++; return; }                        /* :170-174 */
        s->fx[i] += f * dx; s->fx[j] -= f * dx;
        s->fy[i] += f * dy; s->fy[j] -= f * dy;
        s->fz[i] += f * dz; s->fz[j] -= f * dz;
    }
}

static void all_pairs(orc_sys *s)
{   /* integrators.cpp:278-289 */
    for (int i = 0; i < s->N - 1; i++) for (int j = i + 1; j < s->N; j++) pair_inter(s, i, j);
}

static void cell_list_forces(orc_sys *s)
{   /* cell_list: integrators.cpp:238-276 */
    for (int iC = 0; iC < s->nHead; iC++)
    {
        int i = s->chead[iC];
        while (i >= 0)
        {
            int j = s->clist[i];
            while (j >= 0) { pair_inter(s, i, j); j = s->clist[j]; }
            for (int jC = 0; jC < 13; jC++)
            {
                j = s->chead[s->neig[iC * 13 + jC]];
                while (j >= 0) { pair_inter(s, i, j); j = s->clist[j]; }
            }
            i = s->clist[i];
        }
    }
}

static void build_clist(orc_sys *s)
{   /* the binning half of integrate1_clst: integrators.cpp:343-345, 368-371 ; cell_index_sim cell_list.cpp:12-20 */
    for (int c = 0; c < s->nHead; c++) s->chead[c] = -1;
    for (int i = 0; i < s->N; i++)
    {
        int a = (int)(s->x[i] / s->clX), b = (int)(s->y[i] / s->clY), c3 = (int)(s->z[i] / s->clZ);
        int c = a * s->cnYZ + b * s->cnZ + c3;
        s->clist[i] = s->chead[c]; s->chead[c] = i;
    }
}

static double tstat_nose(orc_sys *s)
{   /* tstat_nose: temperature.cpp:339-360 */
    s->chit += s->dt * (s->engKin - s->tKin) * s->rQmass;
    double scale = 1 - s->dt * s->chit;
    for (int i = 0; i < s->N; i++) { s->vx[i] *= scale; s->vy[i] *= scale; s->vz[i] *= scale; }
    double kinE = s->engKin * scale * scale;
    s->conint += s->dt * s->chit * s->qMassTau2;
    s->chit += s->dt * (kinE - s->tKin) * s->rQmass;
    return kinE;
}

static void integrate1(orc_sys *s)
{   /* integrate1 / integrate1_clst: integrators.cpp:292-376 ; frozen species as cuMDfunc.cu:415-420 */
    double tSt = s->dt;
    if (s->tstat_type == TSTAT_NOSE) tstat_nose(s);        /* :305-306 / :340-341 - the returned energy is discarded there */
    if (s->nHead) for (int c = 0; c < s->nHead; c++) s->chead[c] = -1;
    for (int i = 0; i < s->N; i++)
    {
        int t = s->types[i];
        double rM = s->rMass_hdt[t], q = s->charge[t];
        s->vx[i] += rM * s->fx[i]; s->vy[i] += rM * s->fy[i]; s->vz[i] += rM * s->fz[i];
        if (!s->frozen[t]) { s->x[i] += s->vx[i] * tSt; s->y[i] += s->vy[i] * tSt; s->z[i] += s->vz[i] * tSt; }
        put_periodic(s, i);
        if (s->nHead)
        {
            int a = (int)(s->x[i] / s->clX), b = (int)(s->y[i] / s->clY), c3 = (int)(s->z[i] / s->clZ);
            int c = a * s->cnYZ + b * s->cnZ + c3;
            s->clist[i] = s->chead[c]; s->chead[c] = i;
        }
        s->engElecField += q * (s->x[i] * s->Ux + s->y[i] * s->Uy + s->z[i] * s->Uz);   /* :374 ; cuMDfunc.cu:476 */
    }
}

static void integrate2(orc_sys *s, int tScale)
{   /* integrate2: integrators.cpp:486-531 ; scaling rule: serial :511-522, with the radiative
       factor c = 0.25 and the engKin == 0 guard of temp_scale (cuTemp.cu:84-94,109-113) */
    double tempA = 0.0;
    for (int i = 0; i < s->N; i++)
    {
        int t = s->types[i]; double rM = s->rMass_hdt[t];
        s->vx[i] += rM * s->fx[i]; s->vy[i] += rM * s->fy[i]; s->vz[i] += rM * s->fz[i];
        tempA += (s->vx[i] * s->vx[i] + s->vy[i] * s->vy[i] + s->vz[i] * s->vz[i]) * s->mass[t];
    }
    s->engKin = 0.5 * tempA;
    if (tScale && s->engKin != 0.0)
    {
        double c = (s->tstat_type == TSTAT_RADI) ? 0.25 : 1.0;
        double k = sqrt(c * s->tKin / s->engKin);
        for (int i = 0; i < s->N; i++) { s->vx[i] *= k; s->vy[i] *= k; s->vz[i] *= k; }
        s->engKin = s->tKin;
    }
    if (s->tstat_type == TSTAT_NOSE) s->engKin = tstat_nose(s);        /* integrators.cpp:525-529 */
}

static void calc_chars(orc_sys *s)
{   /* calc_chars: integrators.cpp:63-73 (engElec1/2 and engOwn are 0 on this path) */
    s->TempNow = 2.0 * s->engKin * (double)(1.0 / s->degFree) * (1.0 / c_kB());
    s->engTot = s->engElecField + s->engVdW + (s->engElec1 + s->engElec2 + s->engElec3) + s->engKin + s->engBond + s->engAngle;
}

/* ---------------------------------------------------------------- radiative thermostat (cuTemp.cu, fp64 restatement) */
static void get_angled_vector(const double v[3], double cos_phi, double theta, double out[3])
{   /* get_angled_vector: cuTemp.cu:395-453 */
    double l1 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double v1[3] = { v[0] / l1, v[1] / l1, v[2] / l1 }, v2[3], v3[3];
    if (v1[0] != 0.0) { v2[1] = 1.0; v2[2] = 1.0; v2[0] = -(v1[1] * v2[1] + v1[2] * v2[2]) / v1[0]; }
    else if (v1[1] != 0.0) { v2[0] = 1.0; v2[2] = 1.0; v2[1] = -(v1[2] * v2[2]) / v1[1]; }
    else { v2[0] = 1.0; v2[1] = 0.0; v2[2] = 0.0; }
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
    v3[1] = -v1[0] * v2[2] + v1[2] * v2[0];
    v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    double l2 = sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    double l3 = sqrt(v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2]);
    for (int k = 0; k < 3; k++) { v2[k] /= l2; v3[k] /= l3; }
    double sinPhi = sqrt(1 - cos_phi * cos_phi), sinTh = sin(theta), cosTh = cos(theta);
    for (int k = 0; k < 3; k++) out[k] = v1[k] * cos_phi + sinPhi * (cosTh * v2[k] + sinTh * v3[k]);
}

static void tstat_radi(orc_sys *s, uint64_t step)
{   /* tstat_radi9: cuTemp.cu:689-773 with the fixes listed in SURVEY Appendix C-9..C-12 / E */
    for (int i = 0; i < s->N; i++)
    {
        uint64_t id = (uint64_t)s->pid[i];
        double m = s->mass[s->types[i]];
        double pe = s->photons[(id + step) % (uint64_t)s->N];                 /* C-12 restated */
        /* draw 0 is taken and discarded in the reference (:738) */
        {   /* adsorb_rand_photon: cuTemp.cu:484-507 */
            uint32_t rnd = orc_rng(s->seed, step, id, 1) % ORC_NUVECT;       /* rand_uvect :230-234 */
            double v02 = s->vx[i] * s->vx[i] + s->vy[i] * s->vy[i] + s->vz[i] * s->vz[i];
            double ermc = pe * s->revLight / m;
            s->vx[i] += ermc * s->uvx[rnd]; s->vy[i] += ermc * s->uvy[rnd]; s->vz[i] += ermc * s->uvz[rnd];
            double v12 = s->vx[i] * s->vx[i] + s->vy[i] * s->vy[i] + s->vz[i] * s->vz[i];
            s->U[i] += pe + 0.5 * m * (v02 - v12);
        }
        if (s->U[i] > s->radiate_thr)
        {   /* radiate_photon3: cuTemp.cu:631-685 */
            double u0 = s->U[i];
            double v[3] = { s->vx[i], s->vy[i], s->vz[i] };
            double v02 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], v0 = sqrt(v02);
            double ph = s->radiate_frac * u0;
            double ermc = ph * s->revLight / m;
            double d[3];
            if (v0 == 0.0)
            {   /* direction undefined in the reference; use a preset unit vector */
                uint32_t rnd = orc_rng(s->seed, step, id, 2) % ORC_NUVECT;
                d[0] = s->uvx[rnd]; d[1] = s->uvy[rnd]; d[2] = s->uvz[rnd];
            }
            else
            {
                double ermcv0 = ermc / v0;
                if (ermcv0 >= 1.0) { d[0] = -v[0] / v0; d[1] = -v[1] / v0; d[2] = -v[2] / v0; }   /* C-10 fix: cos_phi = -1 */
                else
                {
                    uint32_t r1 = orc_rng(s->seed, step, id, 2) % 2048;
                    double cos_phi = (double)r1 / 1024.0 * (1.0 - ermcv0);
                    cos_phi -= 1.0;
                    uint32_t r2 = orc_rng(s->seed, step, id, 3) % 2048;
                    double theta = (double)r2 / 1024.0 * s->numPi;
                    get_angled_vector(v, cos_phi, theta, d);
                }
            }
            s->vx[i] += ermc * d[0]; s->vy[i] += ermc * d[1]; s->vz[i] += ermc * d[2];
            double v12 = s->vx[i] * s->vx[i] + s->vy[i] * s->vy[i] + s->vz[i] * s->vz[i];
            s->U[i] -= (ph + 0.5 * m * (v12 - v02));
        }
        {   /* radius: cuTemp.cu:757-759 */
            int tp = s->types[i];
            double restrE = s->U[i] < s->mxEng[tp] ? s->U[i] : s->mxEng[tp];
            s->rad[i] = s->radA[tp] / (s->radB[tp] - restrE);
        }
        s->engTemp += s->U[i];
    }
}

/* ewald_rec: elec.cpp:167-335.  Reciprocal part of the Ewald sum over l in [0,kx), m in (-ky,ky), n in (-kz,kz) (half space),
   rk2 < rkcut2.  Adds to the forces, leaves engElec2.  Array names follow the reference; storage is one block. */
static void ewald_rec(orc_sys *s)
{
    const int Nat = s->N, kx = s->kx, ky = s->ky, kz = s->kz;
    const double twopi = 2.0 * ORC_PI;
    double *elc = s->ew, *els = elc + 2 * (size_t)Nat;                     /* [i*2 + {0,1}] */
    double *emc = els + 2 * (size_t)Nat, *ems = emc + (size_t)ky * Nat;    /* [i*ky + m] */
    double *enc = ems + (size_t)ky * Nat, *ens = enc + (size_t)kz * Nat;   /* [i*kz + n] */
    double *lmc = ens + (size_t)kz * Nat, *lms = lmc + Nat, *ckc = lms + Nat, *cks = ckc + Nat;
    int mmin = 0, nmin = 1;
    double eng = 0.0;
    for (int i = 0; i < Nat; i++)
    {
        elc[i * 2] = 1.0; els[i * 2] = 0.0;
        emc[(size_t)i * ky] = 1.0; ems[(size_t)i * ky] = 0.0;              /* init_ewald: elec.cpp:105-107,121-123 */
        enc[(size_t)i * kz] = 1.0; ens[(size_t)i * kz] = 0.0;
        double a;
        a = twopi * s->x[i] * s->ra; els[i * 2 + 1] = sin(a); elc[i * 2 + 1] = cos(a);                           /* sincos, :205-207 */
        a = twopi * s->y[i] * s->rb; if (ky > 1) { ems[(size_t)i * ky + 1] = sin(a); emc[(size_t)i * ky + 1] = cos(a); }
        a = twopi * s->z[i] * s->rc_; if (kz > 1) { ens[(size_t)i * kz + 1] = sin(a); enc[(size_t)i * kz + 1] = cos(a); }
    }
    for (int l = 2; l < ky; l++)
        for (int i = 0; i < Nat; i++)
        {
            double *c = emc + (size_t)i * ky, *sn = ems + (size_t)i * ky;
            c[l] = c[l - 1] * c[1] - sn[l - 1] * sn[1];
            sn[l] = sn[l - 1] * c[1] + c[l - 1] * sn[1];
        }
    for (int l = 2; l < kz; l++)
        for (int i = 0; i < Nat; i++)
        {
            double *c = enc + (size_t)i * kz, *sn = ens + (size_t)i * kz;
            c[l] = c[l - 1] * c[1] - sn[l - 1] * sn[1];
            sn[l] = sn[l - 1] * c[1] + c[l - 1] * sn[1];
        }
    for (int l = 0; l < kx; l++)
    {
        const double rkx = l * twopi * s->ra;
        if (l == 1)
            for (int i = 0; i < Nat; i++) { elc[i * 2] = elc[i * 2 + 1]; els[i * 2] = els[i * 2 + 1]; }
        else if (l > 1)
            for (int i = 0; i < Nat; i++)
            {
                double x = elc[i * 2];
                elc[i * 2] = x * elc[i * 2 + 1] - els[i * 2] * els[i * 2 + 1];
                els[i * 2] = els[i * 2] * elc[i * 2 + 1] + x * els[i * 2 + 1];
            }
        for (int m = mmin; m < ky; m++)
        {
            const double rky = m * twopi * s->rb;
            if (m >= 0)
                for (int i = 0; i < Nat; i++)
                {
                    const double ec = emc[(size_t)i * ky + m], es = ems[(size_t)i * ky + m];
                    lmc[i] = elc[i * 2] * ec - els[i * 2] * es;
                    lms[i] = els[i * 2] * ec + es * elc[i * 2];
                }
            else
                for (int i = 0; i < Nat; i++)
                {
                    const double ec = emc[(size_t)i * ky - m], es = ems[(size_t)i * ky - m];
                    lmc[i] = elc[i * 2] * ec + els[i * 2] * es;
                    lms[i] = els[i * 2] * ec - es * elc[i * 2];
                }
            for (int n = nmin; n < kz; n++)
            {
                const double rkz = n * twopi * s->rc_;
                const double rk2 = rkx * rkx + rky * rky + rkz * rkz;
                if (rk2 < s->rkcut2)
                {
                    double sumC = 0, sumS = 0;
                    if (n >= 0)
                        for (int i = 0; i < Nat; i++)
                        {
                            const double ch = s->charge[s->types[i]];
                            const double ec = enc[(size_t)i * kz + n], es = ens[(size_t)i * kz + n];
                            ckc[i] = ch * (lmc[i] * ec - lms[i] * es);
                            cks[i] = ch * (lms[i] * ec + lmc[i] * es);
                            sumC += ckc[i]; sumS += cks[i];
                        }
                    else
                        for (int i = 0; i < Nat; i++)
                        {
                            const double ch = s->charge[s->types[i]];
                            const double ec = enc[(size_t)i * kz - n], es = ens[(size_t)i * kz - n];
                            ckc[i] = ch * (lmc[i] * ec + lms[i] * es);
                            cks[i] = ch * (lms[i] * ec - lmc[i] * es);
                            sumC += ckc[i]; sumS += cks[i];
                        }
                    const double akk = exp(rk2 * s->mr4a2) / rk2;
                    eng += akk * (sumC * sumC + sumS * sumS);
                    for (int i = 0; i < Nat; i++)
                    {
                        double x = akk * (cks[i] * sumC - ckc[i] * sumS);
                        x *= s->el_scale2;
                        s->fx[i] += rkx * x; s->fy[i] += rky * x; s->fz[i] += rkz * x;
                    }
                }
            }
            nmin = 1 - kz;
        }
        mmin = 1 - ky;
    }
    s->engElec2 = s->el_scale * eng;
}

/* ---------------------------------------------------------------- bonds.cpp / angles.cpp (constant bonds, hcos angles) */
/* read_bond: bonds.cpp:125-364.  n types, ids 1..n (index 0 reserved, sys_init.cpp:293-295).  type: 1 harm (k r0),
   2 mors (D a r0 C), 3 pdn (D a r0 C E), 4 buck (A ro C), 5 e612 (A ro C D F).  All unit factors of const.h:39-41
   are exactly 1.0 (E_scale, r_scale), so the stored parameters equal the raw ones bit for bit; only the
   'con con' tail (mnEx = mxEx = 0) is in scope. */
int orc_set_bond_types(orc_sys *s, int n, const int *spec1, const int *spec2, const int *type, const double *p)
{
    free(s->bdata);
    s->nBdata = n + 1;
    s->bdata = (orc_bond *)calloc((size_t)n + 1, sizeof(orc_bond));
    for (int i = 1; i <= n; i++)
    {
        orc_bond *b = &s->bdata[i];
        if (type[i - 1] < 1 || type[i - 1] > 5) return -1;
        if (spec1[i - 1] < 0 || spec1[i - 1] >= s->nSpec || spec2[i - 1] < 0 || spec2[i - 1] >= s->nSpec) return -2;   /* ERROR[124] */
        b->type = type[i - 1]; b->spec1 = spec1[i - 1]; b->spec2 = spec2[i - 1];
        const double *q = p + 5 * (size_t)(i - 1);
        b->p0 = q[0]; b->p1 = q[1]; b->p2 = q[2]; b->p3 = q[3]; b->p4 = q[4];
    }
    return 0;
}

/* read_angle: angles.cpp:78-128 ('hcos' k cos0 ; ids 1..n, index 0 reserved sys_init.cpp:414-420) */
int orc_set_angle_types(orc_sys *s, int n, const int *central, const int *type, const double *p)
{
    free(s->adata);
    s->nAdata = n + 1;
    s->adata = (orc_angle *)calloc((size_t)n + 1, sizeof(orc_angle));
    for (int i = 1; i <= n; i++)
    {
        if (type[i - 1] != 1) return -1;                                  /* ERROR[012] */
        if (central[i - 1] < 0 || central[i - 1] >= s->nSpec) return -2;  /* ERROR[011] */
        s->adata[i].type = 1; s->adata[i].central = central[i - 1];
        s->adata[i].p0 = p[2 * (size_t)(i - 1)]; s->adata[i].p1 = p[2 * (size_t)(i - 1) + 1];
    }
    return 0;
}

/* read_bondlist: bonds.cpp:25-110 (bonds.txt: 'N' then 'at1 at2 type', atoms 0-based).  The pair is turned so that
   at1 carries the type's spec1 (:51-79); a species mismatch is ERROR[121..123] -> negative return here. */
int orc_set_bond_list(orc_sys *s, int n, const int *a, const int *b, const int *t)
{
    free(s->at1); free(s->at2); free(s->bTypes);
    s->nBonds = n;
    s->at1 = (int *)malloc(sizeof(int) * (size_t)(n + 1)); s->at2 = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    s->bTypes = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; i++)
    {
        int i1 = a[i], i2 = b[i], k = t[i];
        if (k < 1 || k >= s->nBdata || i1 < 0 || i1 >= s->N || i2 < 0 || i2 >= s->N) return -4;
        const orc_bond *bt = &s->bdata[k];
        if (bt->spec1 == s->types[i1]) { if (bt->spec2 != s->types[i2]) return -1; }
        else if (bt->spec1 == s->types[i2]) { if (bt->spec2 == s->types[i1]) { int w = i1; i1 = i2; i2 = w; } else return -2; }
        else return -3;
        s->at1[i] = i1; s->at2[i] = i2; s->bTypes[i] = k;
    }
    return 0;
}

/* read_anglelist: angles.cpp:22-60 (angles.txt: 'N' then 'central lig1 lig2 type') */
int orc_set_angle_list(orc_sys *s, int n, const int *c, const int *l1, const int *l2, const int *t)
{
    free(s->centrs); free(s->lig1); free(s->lig2); free(s->angTypes);
    s->nAngles = n;
    s->centrs = (int *)malloc(sizeof(int) * (size_t)(n + 1)); s->lig1 = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    s->lig2 = (int *)malloc(sizeof(int) * (size_t)(n + 1)); s->angTypes = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; i++)
    {
        if (t[i] < 1 || t[i] >= s->nAdata) return -1;                                    /* ERROR[013] */
        if (c[i] < 0 || c[i] >= s->N || l1[i] < 0 || l1[i] >= s->N || l2[i] < 0 || l2[i] >= s->N) return -4;
        if (s->types[c[i]] != s->adata[t[i]].central) return -2;                         /* ERROR[014] */
        s->centrs[i] = c[i]; s->lig1[i] = l1[i]; s->lig2[i] = l2[i]; s->angTypes[i] = t[i];
    }
    return 0;
}

static double bond_iter(double r2, const orc_bond *b, double *eng)
{   /* bond_iter: bonds.cpp:731-787 ; returns -(1/r) dU/dr */
    double r, x, y, irn, ir2;
    switch (b->type)
    {
    case 1:
        r = sqrt(r2); x = r - b->p1;
        *eng += 0.5 * b->p0 * x * x;
        return -b->p0 / r * x;
    case 2:
        r = sqrt(r2); x = r - b->p2; x = exp(-b->p1 * x); y = 1 - x;
        *eng += b->p0 * y * y - b->p3;
        return -2.0 * b->p0 * b->p1 * x * y / r;
    case 3:
        r = sqrt(r2); x = r - b->p2; x = exp(-b->p1 * x); y = 1 - x;
        ir2 = 1.0 / r2; irn = ir2 * ir2; irn = irn * irn * irn;
        *eng += b->p0 * y * y - b->p3 - b->p4 * irn;
        return -2.0 * b->p0 * b->p1 * x * y / r - 12.0 * b->p4 * irn * ir2;
    case 4:
        r = sqrt(r2); ir2 = 1.0 / r2; irn = ir2 * ir2;
        *eng += b->p0 * exp(-r / b->p1) - b->p2 * irn * ir2;
        return b->p0 * exp(-r / b->p1) / r / b->p1 - 6.0 * b->p2 * irn * irn;
    case 5:
        r = sqrt(r2); ir2 = 1.0 / r2; irn = ir2 * ir2;
        *eng += b->p0 * exp(-r / b->p1) - b->p2 * irn * ir2 - b->p3 * irn * irn - b->p4 * irn * irn * irn;
        return b->p0 * exp(-r / b->p1) / r / b->p1 - 6.0 * b->p2 * irn * irn - 8.0 * b->p3 * irn * irn * ir2
               - 12.0 * b->p4 * irn * irn * irn * ir2;
    }
    return 0.0;
}

static void exec_bondlist(orc_sys *s)
{   /* exec_bondlist: bonds.cpp:1069-1218 with mnEx = mxEx = 0 (constant bonds) */
    double eng = 0.0;
    for (int i = 0; i < s->nBonds; i++)
    {
        int ia = s->at1[i], ja = s->at2[i];
        double dx = s->x[ia] - s->x[ja], dy = s->y[ia] - s->y[ja], dz = s->z[ia] - s->z[ja];   /* sqr_distance_proj, box.cpp:327-335 */
        delta_periodic(s, &dx, &dy, &dz);
        double r2 = dx * dx + dy * dy + dz * dz;
        double f = bond_iter(r2, &s->bdata[s->bTypes[i]], &eng);
        s->fx[ia] += f * dx; s->fx[ja] -= f * dx;
        s->fy[ia] += f * dy; s->fy[ja] -= f * dy;
        s->fz[ia] += f * dz; s->fz[ja] -= f * dz;
    }
    s->engBond = eng;
}

static void exec_anglelist(orc_sys *s)
{   /* angle_iter + exec_anglelist: angles.cpp:179-242 */
    double eng = 0.0;
    for (int i = 0; i < s->nAngles; i++)
    {
        int c = s->centrs[i], l1 = s->lig1[i], l2 = s->lig2[i];
        const orc_angle *ang = &s->adata[s->angTypes[i]];
        double k = ang->p0, cos0 = ang->p1;
        double xij = s->x[l1] - s->x[c], yij = s->y[l1] - s->y[c], zij = s->z[l1] - s->z[c];
        delta_periodic(s, &xij, &yij, &zij);
        double r2ij = xij * xij + yij * yij + zij * zij;
        double rij = sqrt(r2ij);
        double xik = s->x[l2] - s->x[c], yik = s->y[l2] - s->y[c], zik = s->z[l2] - s->z[c];
        delta_periodic(s, &xik, &yik, &zik);
        double r2ik = xik * xik + yik * yik + zik * zik;
        double rik = sqrt(r2ik);
        double cos_th = (xij * xik + yij * yik + zij * zik) / rij / rik;
        double dCos = cos_th - cos0;
        double c1 = -k * dCos;
        double c2 = 1.0 / rij / rik;
        s->fx[c] += -c1 * (xik * c2 + xij * c2 - cos_th * (xij / r2ij + xik / r2ik));
        s->fy[c] += -c1 * (yik * c2 + yij * c2 - cos_th * (yij / r2ij + yik / r2ik));
        s->fz[c] += -c1 * (zik * c2 + zij * c2 - cos_th * (zij / r2ij + zik / r2ik));
        s->fx[l1] += c1 * (xik * c2 - cos_th * xij / r2ij);
        s->fy[l1] += c1 * (yik * c2 - cos_th * yij / r2ij);
        s->fz[l1] += c1 * (zik * c2 - cos_th * zij / r2ij);
        s->fx[l2] += c1 * (xij * c2 - cos_th * xik / r2ik);
        s->fy[l2] += c1 * (yij * c2 - cos_th * yik / r2ik);
        s->fz[l2] += c1 * (zij * c2 - cos_th * zik / r2ik);
        eng += 0.5 * k * dCos * dCos;
    }
    s->engAngle = eng;
}

/* single-term known answers for the tests */
double orc_bond_pair(int type, const double *p, double r2, double *eng_out)
{
    orc_bond b; b.type = type; b.spec1 = b.spec2 = 0; b.p0 = p[0]; b.p1 = p[1]; b.p2 = p[2]; b.p3 = p[3]; b.p4 = p[4];
    *eng_out = 0.0;
    return bond_iter(r2, &b, eng_out);
}

/* ---------------------------------------------------------------- public stepping API */
/* mode 0: all pairs (integrators.cpp:278), 1: linked cells (integrators.cpp:238; falls back to 0 if no table);
   mode | 2: without the bonded terms (the state init_serial leaves, sys_init.cpp:1181-1184: all_pairs only) */
void orc_forces(orc_sys *s, int mode)
{
    reset_chars(s);
    clear_force(s);
    if (s->elec_type == ELEC_EWALD) ewald_rec(s);                          /* add_elec: sys_init.cpp:1183 / main.cpp:99 */
    if ((mode & 1) && s->nHead) { build_clist(s); cell_list_forces(s); } else all_pairs(s);
    if (!(mode & 2))
    {
        if (s->nBonds) exec_bondlist(s);                                   /* main.cpp:101-104 */
        if (s->nAngles) exec_anglelist(s);
    }
}

void orc_step(orc_sys *s, int nsteps)
{   /* loop body of main.cpp:89-142 ; thermostat placement as main.cu:370-382 */
    for (int n = 0; n < nsteps; n++)
    {
        s->iStep++;
        reset_chars(s);
        integrate1(s);
        clear_force(s);
        if (s->elec_type == ELEC_EWALD) ewald_rec(s);                      /* sim->add_elec, main.cpp:99 */
        if (s->nHead) cell_list_forces(s); else all_pairs(s);
        if (s->nBonds) exec_bondlist(s);                                   /* main.cpp:101-104 */
        if (s->nAngles) exec_anglelist(s);
        int tScale = (s->iStep <= s->nEq) && s->freqEq > 0 && ((s->iStep % s->freqEq) == 0);   /* main.cpp:110-119 */
        integrate2(s, tScale);
        if (s->tstat_type == TSTAT_RADI) tstat_radi(s, (uint64_t)s->iStep);
        calc_chars(s);
    }
}

/* stage entry points (for kernel-by-kernel parity tests) */
void orc_stage_integrate1(orc_sys *s) { reset_chars(s); integrate1(s); }
void orc_stage_integrate2(orc_sys *s, int tScale) { integrate2(s, tScale); calc_chars(s); }
void orc_stage_tstat(orc_sys *s, long long step) { s->engTemp = 0.0; tstat_radi(s, (uint64_t)step); }

void orc_get_state(const orc_sys *s, double *x, double *y, double *z, double *vx, double *vy, double *vz,
                   double *fx, double *fy, double *fz, double *U, double *rad)
{
    size_t nb = 8 * (size_t)s->N;
    if (x) memcpy(x, s->x, nb); if (y) memcpy(y, s->y, nb); if (z) memcpy(z, s->z, nb);
    if (vx) memcpy(vx, s->vx, nb); if (vy) memcpy(vy, s->vy, nb); if (vz) memcpy(vz, s->vz, nb);
    if (fx) memcpy(fx, s->fx, nb); if (fy) memcpy(fy, s->fy, nb); if (fz) memcpy(fz, s->fz, nb);
    if (U) memcpy(U, s->U, nb); if (rad) memcpy(rad, s->rad, nb);
}
void orc_set_vel(orc_sys *s, const double *vx, const double *vy, const double *vz)
{ size_t nb = 8 * (size_t)s->N; memcpy(s->vx, vx, nb); memcpy(s->vy, vy, nb); memcpy(s->vz, vz, nb); }
void orc_set_forces(orc_sys *s, const double *fx, const double *fy, const double *fz)
{ size_t nb = 8 * (size_t)s->N; memcpy(s->fx, fx, nb); memcpy(s->fy, fy, nb); memcpy(s->fz, fz, nb); }
void orc_set_thermo(orc_sys *s, const double *U, const double *rad)
{ size_t nb = 8 * (size_t)s->N; if (U) memcpy(s->U, U, nb); if (rad) memcpy(s->rad, rad, nb); }
const double *orc_photons(const orc_sys *s) { return s->photons; }

/* out[0..21]: engVdW, engElec3, engKin, engTot, engElecField, engTemp, Temp, mom[6], nDropped, iStep, tKin, chit, conint, engBond, engAngle,
   engElec1 (Ewald constant), engElec2 (Ewald reciprocal) */
void orc_get_stats(const orc_sys *s, double *out)
{
    out[0] = s->engVdW; out[1] = s->engElec3; out[2] = s->engKin; out[3] = s->engTot; out[4] = s->engElecField;
    out[5] = s->engTemp; out[6] = s->TempNow;
    for (int k = 0; k < 6; k++) out[7 + k] = s->mom[k];
    out[13] = (double)s->nDropped; out[14] = (double)s->iStep; out[15] = s->tKin;
    out[16] = s->chit; out[17] = s->conint; out[18] = s->engBond; out[19] = s->engAngle; out[20] = s->engElec1; out[21] = s->engElec2;
}
void orc_get_cross(const orc_sys *s, long long *out) { for (int k = 0; k < 6; k++) out[k] = s->cross[k]; }
void orc_get_species_cross(const orc_sys *s, long long *out) { for (int k = 0; k < 6 * s->nSpec; k++) out[k] = s->specCross[k]; }
