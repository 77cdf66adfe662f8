// TEST INFRASTRUCTURE - never linked into, imported by or shipped with the product.
//
// ref_driver: a small harness of OUR OWN that is linked against the reference's own
// serial-path translation units, compiled where they lie under /root/reference/src
// (box.cpp, vdw.cpp, elec.cpp, cell_list.cpp, integrators.cpp, temperature.cpp, bonds.cpp, angles.cpp -
// see oracle/Makefile).
// No reference source is copied here; this file only *calls* the reference through the
// declarations in its headers (-I/root/reference/src).
//
// What it does: fills the reference's host model structs (Sim/Field/Atoms/Box/Elec/TStat,
// dataStruct.h:40-416, temperature.h:15) from a binary "case" file, then replays the loop
// body of the serial program (main.cpp:89-142): reset_chars -> integrator1 -> clear_force ->
// forcefield -> exec_bondlist -> exec_anglelist -> integrate2 -> calc_chars, dumping x/v/f/energies at the requested steps.
//
// What is NOT the reference here (restated by us, cited): the input-file scanning and the
// derived-parameter preparation of sys_init.cpp:1036-1187 / vdw.cpp:261-299, because
// utils.cpp (find_*, min/max, rand01) needs <conio.h> and fscanf_s and is therefore
// unbuildable in this image (see DESIGN.md "Oracle"). Everything arithmetic on the hot path
// (pair functions, min-image, wrap, linked cells + half-shell table, traversal, integrator,
// energy bookkeeping) is the reference's compiled code.
//
// Case-file layout: see oracle/casefile.py (single source of truth for both sides).

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
#include <time.h>

#include "const.h"
#include "dataStruct.h"
#include "temperature.h"
#include "vdw.h"
#include "elec.h"
#include "box.h"
#include "cell_list.h"
#include "integrators.h"
#include "bonds.h"
#include "angles.h"

// NOT the reference: ewald_rec (elec.cpp:205-207) calls the reference's own helper sincos(), which lives in utils.cpp - the one
// translation unit that cannot be compiled here (<conio.h>).  Its body (utils.cpp:65-80) is `s = sin(arg); c = cos(arg);`
// (the fsincos asm is commented out there); this harness restates exactly that, so the Ewald pin is "the reference's compiled
// ewald_rec / ewald_const / coul_iter + two libm calls made from here".  No header, library or tool is faked.
void sincos(double arg, double& s, double& c) { s = sin(arg); c = cos(arg); }

static void rd(FILE* f, void* p, size_t n)
{
    if (fread(p, 1, n, f) != n) { fprintf(stderr, "ref_driver: short read\n"); exit(2); }
}
static int rd_i(FILE* f) { int32_t v; rd(f, &v, 4); return v; }
static double rd_d(FILE* f) { double v; rd(f, &v, 8); return v; }

static void wr_arr(FILE* f, const double* a, int n) { fwrite(a, 8, n, f); }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_driver case.bin out.bin [time]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    char magic[4]; rd(f, magic, 4);
    if (memcmp(magic, "AZTC", 4) != 0) { fprintf(stderr, "bad magic\n"); return 2; }
    int version = rd_i(f);
    int N = rd_i(f), nSpec = rd_i(f), nVdw = rd_i(f);
    double L[3]; rd(f, L, 24);
    double dt = rd_d(f);
    int nsteps = rd_i(f);
    int elec_type = rd_i(f); double rReal = rd_d(f), alpha = rd_d(f);
    int use_clist = rd_i(f), do_center = rd_i(f), init_forces = rd_i(f);
    int nEq = rd_i(f), freqEq = rd_i(f); double T = rd_d(f); int tstat_type = rd_i(f);
    double Ux = rd_d(f);
    double tau = (version >= 2) ? rd_d(f) : 0.0;
    int ndump = rd_i(f);
    int* dump = (int*)malloc(sizeof(int) * (ndump + 1));
    for (int i = 0; i < ndump; i++) dump[i] = rd_i(f);

    Elec* elec = (Elec*)calloc(1, sizeof(Elec));
    TStat* tstat = (TStat*)calloc(1, sizeof(TStat));
    Box* box = (Box*)calloc(1, sizeof(Box));
    Field* field = (Field*)calloc(1, sizeof(Field));
    Atoms* atm = (Atoms*)calloc(1, sizeof(Atoms));
    Sim* sim = (Sim*)calloc(1, sizeof(Sim));

    // ---- species (what read_spec, sys_init.cpp:83-130, leaves behind) ----
    field->nSpec = nSpec;
    field->species = (Spec*)calloc(nSpec, sizeof(Spec));
    field->charged_spec = 0;
    for (int i = 0; i < nSpec; i++)
    {
        Spec* s = &field->species[i];
        snprintf(s->name, 8, "S%d", i);
        s->mass = rd_d(f) * m_scale;          // sys_init.cpp:108
        s->charge = rd_d(f) * q_scale;        // sys_init.cpp:109
        s->charged = fabs(s->charge) < 1.0E-10 ? 0 : 1;   // sys_init.cpp:112-115
        if (s->charge != 0.0) field->charged_spec = 1;   // sys_init.cpp:210
    }

    // ---- pair potentials (what read_vdw, vdw.cpp:234-308, leaves behind) ----
    field->nVdW = nVdw;
    field->pairpots = (VdW*)calloc(nVdw > 0 ? nVdw : 1, sizeof(VdW));
    field->vdws = (VdW***)malloc(sizeof(VdW**) * nSpec);
    for (int i = 0; i < nSpec; i++) field->vdws[i] = (VdW**)calloc(nSpec, sizeof(VdW*));
    field->minRvdw = 999999.9; field->maxRvdw = 0.0;     // sys_init.cpp:258-259
    const double r4s = r_scale * r_scale * r_scale * r_scale, r6s = r4s * r_scale * r_scale, r8s = r4s * r4s;
    for (int k = 0; k < nVdw; k++)
    {
        int a = rd_i(f), b = rd_i(f), type = rd_i(f);
        double rc = rd_d(f), p[5]; rd(f, p, 40);
        VdW pp; memset(&pp, 0, sizeof(pp));
        rc *= r_scale;                                   // vdw.cpp:261
        if (rc < field->minRvdw) field->minRvdw = rc;
        if (rc > field->maxRvdw) field->maxRvdw = rc;
        pp.r2cut = rc * rc; pp.type = type;
        pp.p0 = p[0]; pp.p1 = p[1]; pp.p2 = p[2]; pp.p3 = p[3]; pp.p4 = p[4];
        switch (type)                                    // vdw.cpp:210-219 scale tables, :283-288 LJ prefactors
        {
        case lj_type:
            pp.p0 *= 4 * E_scale; pp.p1 *= r_scale; pp.p2 = 0; pp.p3 = 0; pp.p4 = 0;
            pp.p1 = pp.p1 * pp.p1; pp.p2 = 6 * pp.p0;
            pp.eng = e_lj; pp.eng_r = er_lj; pp.feng = fe_lj; pp.feng_r = fer_lj; break;
        case bh_type:
            pp.p0 *= E_scale; pp.p1 *= r_scale; pp.p2 *= r6s * E_scale; pp.p3 = 0; pp.p4 = 0;
            pp.eng = e_buckingham; pp.eng_r = er_buckingham; pp.feng = fe_buckingham; pp.feng_r = fer_buckingham; break;
        case CuCl_type:
            pp.p0 *= E_scale * r_scale * r6s; pp.p1 *= E_scale * r4s; pp.p2 *= E_scale * r6s; pp.p3 = 0; pp.p4 = 0;
            pp.eng = e_746; pp.eng_r = er_746; pp.feng = fe_746; pp.feng_r = fer_746; break;
        case BHM_type:
            pp.p0 *= E_scale; pp.p1 *= 1.0 / r_scale; pp.p2 *= r_scale; pp.p3 *= E_scale * r6s; pp.p4 *= E_scale * r8s;
            pp.eng = e_bhm; pp.eng_r = er_bhm; pp.feng = fe_bhm; pp.feng_r = fer_bhm; break;
        default:
            fprintf(stderr, "ref_driver: vdw type %d has no serial implementation (vdw.cpp:204-207)\n", type); return 3;
        }
        pp.use_radii = 0;
        field->pairpots[k] = pp;
        field->vdws[a][b] = &field->pairpots[k];
        field->vdws[b][a] = &field->pairpots[k];         // vdw.cpp:303-307
    }
    field->maxR2vdw = field->maxRvdw * field->maxRvdw;  // sys_init.cpp:279

    // ---- atoms (read_atoms_box, sys_init.cpp:487-565) ----
    atm->nAt = N;
    atm->types = (int*)malloc(sizeof(int) * N);
    double** arrs[] = { &atm->xs, &atm->ys, &atm->zs, &atm->vxs, &atm->vys, &atm->vzs, &atm->fxs, &atm->fys, &atm->fzs,
                        &atm->x0s, &atm->y0s, &atm->z0s, &atm->vx0, &atm->vy0, &atm->vz0 };
    for (size_t i = 0; i < sizeof(arrs) / sizeof(arrs[0]); i++) *arrs[i] = (double*)calloc(N, sizeof(double));
    atm->nBonds = (int*)calloc(N, sizeof(int));
    atm->parents = (int*)malloc(sizeof(int) * N);
    { int32_t* t = (int32_t*)malloc(4 * N); rd(f, t, 4 * (size_t)N); for (int i = 0; i < N; i++) { atm->types[i] = t[i]; atm->parents[i] = -1; } free(t); }
    rd(f, atm->xs, 8 * (size_t)N); rd(f, atm->ys, 8 * (size_t)N); rd(f, atm->zs, 8 * (size_t)N);
    rd(f, atm->vxs, 8 * (size_t)N); rd(f, atm->vys, 8 * (size_t)N); rd(f, atm->vzs, 8 * (size_t)N);

    // ---- bonded terms (what read_bond bonds.cpp:125-364, read_angle angles.cpp:78-128, read_bondlist bonds.cpp:25-110
    //      and read_anglelist angles.cpp:22-60 leave behind; constant 'con con' bonds only) ----
    field->nBdata = 0; field->nAdata = 0; field->nBonds = 0; field->nAngles = 0;
    if (version >= 3)
    {
        int nbt = rd_i(f);
        field->nBdata = nbt ? nbt + 1 : 0;                       // [0] reserved as 'empty bond', sys_init.cpp:294
        field->bdata = (Bond*)calloc(nbt + 1, sizeof(Bond));
        for (int i = 1; i <= nbt; i++)
        {
            Bond* b = &field->bdata[i];
            b->spec1 = rd_i(f); b->spec2 = rd_i(f); b->type = rd_i(f);
            double p[5]; rd(f, p, 40);
            b->hatom = -1; b->evol = 0; b->number = 0; b->mnEx = 0; b->mxEx = 0;   // bonds.cpp:144-148,262-266,285-287
            const double r2s = r_scale * r_scale, r6 = r2s * r2s * r2s;
            switch (b->type)                                     // unit handling of bonds.cpp:158-252 (all factors are 1.0)
            {
            case 1: b->p0 = p[0] * E_scale / r2s; b->p1 = p[1] * r_scale; break;
            case 2: b->p0 = p[0] * E_scale; b->p1 = p[1] / r2s; b->p2 = p[2] * r_scale; b->p3 = p[3] * E_scale; break;
            case 3: b->p0 = p[0] * E_scale; b->p1 = p[1] / r2s; b->p2 = p[2] * r_scale; b->p3 = p[3] * E_scale; b->p4 = p[4] * E_scale; break;
            case 4: b->p0 = p[0] * E_scale; b->p1 = p[1] * r_scale; b->p2 = p[2] * E_scale * r6; break;
            case 5: b->p0 = p[0] * E_scale; b->p1 = p[1] * r_scale; b->p2 = p[2] * E_scale * r6;
                    b->p3 = p[3] * E_scale * r6 * r2s; b->p4 = p[4] * E_scale * r6 * r6; break;
            default: fprintf(stderr, "ref_driver: unknown bond potential %d\n", b->type); return 3;
            }
        }
        int nat = rd_i(f);
        field->nAdata = nat ? nat + 1 : 0;                       // sys_init.cpp:414
        field->adata = (Angle*)calloc(nat + 1, sizeof(Angle));
        for (int i = 1; i <= nat; i++)
        {
            Angle* a = &field->adata[i];
            a->central = rd_i(f); a->type = rd_i(f);
            a->p0 = rd_d(f) * E_scale; a->p1 = rd_d(f);          // angles.cpp:113-119
        }
        int nb = rd_i(f);
        field->nBonds = nb;
        field->at1 = (int*)malloc(sizeof(int) * (nb + 1)); field->at2 = (int*)malloc(sizeof(int) * (nb + 1));
        field->bTypes = (int*)malloc(sizeof(int) * (nb + 1));
        for (int i = 0; i < nb; i++)
        {
            int at1 = rd_i(f), at2 = rd_i(f), k = rd_i(f);
            if (k < 1 || k >= field->nBdata) { fprintf(stderr, "ref_driver: bond %d has unknown type %d\n", i, k); return 3; }
            Bond* bt = &field->bdata[k];
            if (bt->spec1 == atm->types[at1]) { if (bt->spec2 != atm->types[at2]) { fprintf(stderr, "ERROR [121]\n"); return 3; } }
            else if (bt->spec1 == atm->types[at2])
            {
                if (bt->spec2 == atm->types[at1]) { int w = at1; at1 = at2; at2 = w; }   // bonds.cpp:62-67: turn the bond
                else { fprintf(stderr, "ERROR [122]\n"); return 3; }
            }
            else { fprintf(stderr, "ERROR [123]\n"); return 3; }
            bt->number++;
            field->at1[i] = at1; field->at2[i] = at2; field->bTypes[i] = k;
        }
        int na = rd_i(f);
        field->nAngles = na;
        field->centrs = (int*)malloc(sizeof(int) * (na + 1)); field->lig1 = (int*)malloc(sizeof(int) * (na + 1));
        field->lig2 = (int*)malloc(sizeof(int) * (na + 1)); field->angTypes = (int*)malloc(sizeof(int) * (na + 1));
        for (int i = 0; i < na; i++)
        {
            field->centrs[i] = rd_i(f); field->lig1[i] = rd_i(f); field->lig2[i] = rd_i(f);
            int x = rd_i(f);
            if (!(x && x < field->nAdata)) { fprintf(stderr, "ERROR[013]\n"); return 3; }
            field->angTypes[i] = x;
            if (atm->types[field->centrs[i]] != field->adata[x].central) { fprintf(stderr, "ERROR[014]\n"); return 3; }
        }
    }
    int ewk[3] = {0, 0, 0};
    if (version >= 4) { ewk[0] = rd_i(f); ewk[1] = rd_i(f); ewk[2] = rd_i(f); }
    fclose(f);

    // ---- box (read_box box.cpp:9-28 -> prepare_box) ----
    box->type = tpBoxRect; box->la = L[0]; box->lb = L[1]; box->lc = L[2];
    prepare_box(box);                                    // reference code

    // ---- elec (read_elec elec.cpp:14-67, prepare_elec :371-406) ----
    elec->type = elec_type; elec->rReal = rReal * r_scale; elec->alpha = alpha; elec->eps = 1.0;
    if (!field->charged_spec && elec->type) elec->type = tpElecNone;   // elec.cpp:52-56
    if (elec->type == tpElecNone) { /* rReal kept as given; unused */ }
    elec->r2Real = elec->rReal * elec->rReal;
    elec->kx = ewk[0]; elec->ky = ewk[1]; elec->kz = ewk[2];                         // read_elec elec.cpp:36
    if (elec->type == tpElecEwald && (ewk[0] < 1 || ewk[1] < 2 || ewk[2] < 2)) { fprintf(stderr, "ref_driver: Ewald needs kx >= 1, ky, kz >= 2\n"); return 3; }
    init_elec(elec, box, sim, atm);                      // reference code (Ewald work arrays), sys_init.cpp:1048
    prepare_elec(atm, field, elec, sim, box);            // reference code (Fennel / Ewald constants, ewald_const)

    // ---- sim: derived parameters (init_md, sys_init.cpp:1053-1112) ----
    sim->tSt = dt; sim->nSt = nsteps; sim->nEq = nEq; sim->freqEq = freqEq;
    sim->Ux = Ux; sim->Uy = 0; sim->Uz = 0; sim->shiftX = 0.0; sim->shiftVal = 0.0;
    sim->ejtype = 0; sim->eJump = 0; sim->use_bnd = field->nBonds ? 1 : 0; sim->use_angl = field->nAngles ? 1 : 0;
    for (int i = 0; i < nSpec; i++) field->species[i].rMass_hdt = 0.5 * sim->tSt / field->species[i].mass;
    sim->rMax = 0.0;
    if (elec->type) sim->rMax = elec->rReal; else if (field->nVdW) sim->rMax = field->maxRvdw;
    sim->r2Max = sim->rMax * sim->rMax;
    tstat->type = tstat_type; tstat->Temp = T; sim->tTemp = T;
    sim->degFree = 3 * N - 0; if (tstat->type) sim->degFree--;
    sim->revDegFree = (double)(1.0 / sim->degFree);
    tstat->tKin = 0.5 * sim->tTemp * kB * sim->degFree;
    if (tstat->type == tpTermNose)
    {   // read_tstat temperature.cpp:103-111 ; init_md sys_init.cpp:1107-1112
        tstat->tau = tau; tstat->chit = 0.0; tstat->conint = 0.0;
        tstat->qMass = 2 * tstat->tKin * tstat->tau * tstat->tau;
        tstat->rQmass = 0.5 / tstat->tKin / tstat->tau / tstat->tau;
        tstat->qMassTau2 = 2 * tstat->tKin;
    }

    // ---- init_serial (sys_init.cpp:1122-1187) ----
    sim->nHead = 0;
    if (use_clist) init_clist(atm, sim, box, sim->rMax);  // reference code; leaves nHead = 0 if < 4 cells per axis
    if (do_center) center_box(atm, box);                  // reference code
    if (sim->nHead) { sim->integrator1 = integrate1_clst; sim->forcefield = cell_list; }
    else { sim->integrator1 = integrate1; sim->forcefield = all_pairs; }
    sim->pair = pair_inter;
    sim->pair_elec = pair_elecs[elec->type];
    sim->add_elec = add_elecs[elec->type];                    // ewald_rec for 'pme' (reference code), elec.h:46
    reset_chars(sim);
    clear_force(atm, field->species, sim, box);
    if (init_forces) { sim->add_elec(atm, field, elec, box, sim); all_pairs(atm, field, elec, box, sim); }   // sys_init.cpp:1181-1184

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    int32_t hdr[4] = { N, sim->nHead, sim->cnX * 1000000 + sim->cnY * 1000 + sim->cnZ, ndump };
    fwrite(hdr, 4, 4, o);

    int idump = 0;
    double tSim = 0.0;
    auto dump_now = [&](int step)
    {
        int32_t s = step; fwrite(&s, 4, 1, o);
        double e[16] = { sim->engVdW, sim->engElec3, sim->engKin, sim->engTot, sim->engElecField, sim->Temp,
                         box->momXn, box->momXp, box->momYn, box->momYp, box->momZn, box->momZp,
                         sim->engBond, sim->engAngle, sim->engElec1, sim->engElec2 };
        fwrite(e, 8, 16, o);
        wr_arr(o, atm->xs, N); wr_arr(o, atm->ys, N); wr_arr(o, atm->zs, N);
        wr_arr(o, atm->vxs, N); wr_arr(o, atm->vys, N); wr_arr(o, atm->vzs, N);
        wr_arr(o, atm->fxs, N); wr_arr(o, atm->fys, N); wr_arr(o, atm->fzs, N);
    };
    if (idump < ndump && dump[idump] == 0) { dump_now(0); idump++; }

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int iSt = 0;
    while (iSt < sim->nSt)                                   // main.cpp:89-142
    {
        iSt++;
        reset_chars(sim);
        sim->integrator1(atm, field->species, sim, box, tstat);
        clear_force(atm, field->species, sim, box);
        sim->add_elec(atm, field, elec, box, sim);
        sim->forcefield(atm, field, elec, box, sim);
        if (field->nBonds) exec_bondlist(atm, field, sim, box);          // reference code, main.cpp:101-104
        if (field->nAngles) exec_anglelist(atm, field, sim, box);
        if (iSt > sim->nEq)
            integrate2(atm, field->species, sim, 0, tstat);
        else
        {
            if ((iSt % sim->freqEq) == 0) integrate2(atm, field->species, sim, 1, tstat);
            else integrate2(atm, field->species, sim, 0, tstat);
        }
        calc_chars(sim, tSim);
        if (idump < ndump && dump[idump] == iSt) { dump_now(iSt); idump++; }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fclose(o);
    double wall = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    printf("{\"steps\": %d, \"wall_s\": %.6f, \"n_atoms\": %d, \"n_cells\": %d, \"engVdW\": %.17g, \"engTot\": %.17g}\n",
           nsteps, wall, N, sim->nHead, sim->engVdW, sim->engTot);
    return 0;
}
