// aztotmd - command-line driver with the reference program's surface (main.cu:239-463): reads atoms.xyz, field.txt,
// control.txt, cuda.txt from a directory (default: the cwd, as the reference), runs `nstep` steps of the hot path on
// the GPU and writes the reference's result files for this path:
//   stat.dat        header + one row every `stat` steps          (start_stat / copy_stat, cuStat.cu:300-330, 40-71)
//   revcon.xyz      final configuration in atoms.xyz format     (out_atoms, out_md.cpp:65-87; main.cu:436)
//   velocities.dat  per-species |v|, vx, vy, vz table            (out_velocities, out_md.cpp:126-190; main.cu:445)
//   tchars.dat      thermal energies and radii (radiative thermostat only)  (out_thermalchar, main.cu:51-118)
// Unlike the reference, atoms are written in their ORIGINAL order (the reference writes them cell-sorted, SURVEY C-20).
// Everything goes through the C ABI of include/aztot.h.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/aztot.h"

static void die(const char* what)
{
    std::fprintf(stderr, "FATAL ERROR: %s: %s\n", what, aztot_last_error());
    std::exit(1);
}

static double q1(const aztot_model* m, const char* key)
{
    double v = 0.0;
    if (aztot_model_query(m, key, &v, 1) < 1) die(key);
    return v;
}

int main(int argc, char** argv)
{
    std::string dir = ".", out = ".";
    int device = 0, nstep_override = -1;
    for (int i = 1; i < argc; i++)
    {
        std::string a = argv[i];
        if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--nstep" && i + 1 < argc) nstep_override = std::atoi(argv[++i]);
        else if (a == "-h" || a == "--help") { std::printf("usage: aztotmd [input-dir] [--out dir] [--device n] [--nstep n]\n"); return 0; }
        else dir = a;
    }
    std::printf("azTotMD hot path on MI355X (%s)\n", aztot_version());
    const std::time_t t0 = std::time(nullptr);
    aztot_model* model = nullptr;
    if (aztot_init_md(dir.c_str(), &model) != AZTOT_OK) die("SYSTEM CAN'T BE INITIALIZED");
    const int N = (int)q1(model, "n_atoms"), nSpec = (int)q1(model, "n_species");
    const int nStep = nstep_override >= 0 ? nstep_override : (int)q1(model, "nstep");
    const int stat = std::max(1, (int)q1(model, "stat"));
    const bool radi = (int)q1(model, "tstat_type") == AZTOT_TSTAT_RADI;
    double box[3];
    aztot_model_query(model, "box", box, 3);
    std::vector<std::string> names(nSpec);
    for (int i = 0; i < nSpec; i++) { char b[16]; aztot_model_species_name(model, i, b, 16); names[i] = b; }

    aztot_options opt;
    aztot_default_options(&opt);
    opt.device = device;
    opt.initial_forces = 0;                         // the GPU program starts from F = 0 (sys_init.cpp:551-553)
    aztot_md* md = nullptr;
    if (aztot_init_device(model, &opt, &md) != AZTOT_OK) die("DEVICE CAN'T BE INITIALIZED");
    std::printf("MD long %d timesteps of %f ps, %d atoms\n", nStep, q1(model, "dt"), N);

    FILE* sf = std::fopen((out + "/stat.dat").c_str(), "w");
    if (!sf) { std::perror("stat.dat"); return 1; }
    // columns follow start_stat (cuStat.cu:300-330): engBnd / engAngle appear when field.txt declares bond / angle types
    double nbd[4] = {0, 0, 0, 0};
    aztot_model_query(model, "n_bonded", nbd, 4);
    const bool hasB = nbd[0] > 0, hasA = nbd[1] > 0;
    std::fprintf(sf, "time\tstep\tengTot\tengKin\tengVdW\tengCoul1\tengCoul2%s%s%s\tmomPx\tmomNx\tmomPy\tmomNy\tmomPz\tmomNz\tpress\n", radi ? "\tengTerm" : "",
                 hasB ? "\tengBnd" : "", hasA ? "\tengAngle" : "");
    std::fprintf(sf, "time, ps\tstep, n\tengTot, eV\tengKin, eV\tengVdW, eV\tengCoul1, eV\tengCoul2, eV%s%s%s"
                     "\tmomPx, eVps/A\tmomNx, eVps/A\tmomPy, eVps/A\tmomNy, eVps/A\tmomPz, eVps/A\tmomNz, eVps/A\tpress, atm\n", radi ? "\tengTerm, eV" : "",
                 hasB ? "\tengBnd, eV" : "", hasA ? "\tengAngle, eV" : "");
    // msd.dat: despite its name the reference writes the per-species wall-crossing counters there (start_stat cuStat.cu:345-350,
    // init_cuda_stat :278-288: specAcBoxPos/Neg x, y, z per species), one row per statistics step
    FILE* mf = std::fopen((out + "/msd.dat").c_str(), "w");
    if (!mf) { std::perror("msd.dat"); return 1; }
    std::fprintf(mf, "time\tstep");
    for (int j = 0; j < nSpec; j++) std::fprintf(mf, "\t%s_px\tnx\tpy\tny\tpz\tnz", names[j].c_str());
    std::fprintf(mf, "\n");
    std::vector<int64_t> crossings(6 * (size_t)nSpec);
    aztot_stats st;
    for (int done = 0; done < nStep;)
    {
        const int n = std::min(stat, nStep - done);
        if (aztot_step(md, n) != AZTOT_OK) die("step");
        done += n;
        if (aztot_get_stats(md, &st) != AZTOT_OK) die("stats");
        std::fprintf(sf, "%f\t%d\t%f\t%f\t%f\t%f\t%f", st.time, (int)st.step, st.engTot, st.engKin, st.engVdW, st.engCoul, st.engCoulRec);
        if (radi) std::fprintf(sf, "\t%f", st.engTemp);
        if (hasB) std::fprintf(sf, "\t%f", st.engBond);
        if (hasA) std::fprintf(sf, "\t%f", st.engAngle);
        std::fprintf(sf, "\t%f\t%f\t%f\t%f\t%f\t%f\t%f\n", st.posMom[0], st.negMom[0], st.posMom[1], st.negMom[1], st.posMom[2], st.negMom[2], st.pressure);
        if (aztot_species_crossings(md, crossings.data(), (int)crossings.size()) != AZTOT_OK) die("species crossings");
        std::fprintf(mf, "%f\t%d", st.time, (int)st.step);
        for (int j = 0; j < nSpec; j++)        // our slots are Xn, Xp, Yn, Yp, Zn, Zp; the file wants px nx py ny pz nz
            std::fprintf(mf, "\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld", (long long)crossings[6 * j + 1], (long long)crossings[6 * j + 0],
                         (long long)crossings[6 * j + 3], (long long)crossings[6 * j + 2], (long long)crossings[6 * j + 5], (long long)crossings[6 * j + 4]);
        std::fprintf(mf, "\n");
        std::printf("time=%f(%d) Tot=%f Kin=%f VdW=%f Coul=%f T=%f P=%f\n", st.time, (int)st.step, st.engTot, st.engKin, st.engVdW, st.engCoul, st.temperature, st.pressure);
    }
    std::fclose(sf);
    std::fclose(mf);

    std::vector<double> x(N), y(N), z(N), vx(N), vy(N), vz(N), U(N), rad(N);
    std::vector<int32_t> types(N);
    aztot_state s = {};
    s.n_atoms = N; s.x = x.data(); s.y = y.data(); s.z = z.data(); s.vx = vx.data(); s.vy = vy.data(); s.vz = vz.data();
    s.U = U.data(); s.radius = rad.data(); s.types = types.data();
    if (aztot_md_to_host(md, &s) != AZTOT_OK) die("md_to_host");

    if (FILE* f = std::fopen((out + "/revcon.xyz").c_str(), "w"))
    {
        std::fprintf(f, "%d\n%d %f %f %f\n", N, 1, box[0], box[1], box[2]);
        for (int i = 0; i < N; i++) std::fprintf(f, "%s\t%f\t%f\t%f\n", names[types[i]].c_str(), x[i], y[i], z[i]);
        std::fclose(f);
    }
    // per-species column tables (out_velocities / out_thermalchar layout)
    std::vector<std::vector<int>> bySpec(nSpec);
    size_t mx = 0;
    for (int i = 0; i < N; i++) bySpec[types[i]].push_back(i);
    for (auto& v : bySpec) mx = std::max(mx, v.size());
    if (FILE* f = std::fopen((out + "/velocities.dat").c_str(), "w"))
    {
        std::fprintf(f, "No");
        for (int j = 0; j < nSpec; j++) std::fprintf(f, "\t%s\tx\ty\tz", names[j].c_str());
        std::fprintf(f, "\n");
        for (size_t r = 0; r < mx; r++)
        {
            std::fprintf(f, "%zu", r + 1);
            for (int j = 0; j < nSpec; j++)
                if (r < bySpec[j].size())
                {
                    const int i = bySpec[j][r];
                    std::fprintf(f, "\t%f\t%f\t%f\t%f", std::sqrt(vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]), vx[i], vy[i], vz[i]);
                }
                else std::fprintf(f, "\t\t\t\t");
            std::fprintf(f, "\n");
        }
        std::fclose(f);
    }
    if (radi)
        if (FILE* f = std::fopen((out + "/tchars.dat").c_str(), "w"))
        {
            std::fprintf(f, "No");
            for (int j = 0; j < nSpec; j++) std::fprintf(f, "\t%s_eng\t%s_rad", names[j].c_str(), names[j].c_str());
            std::fprintf(f, "\n");
            for (size_t r = 0; r < mx; r++)
            {
                std::fprintf(f, "%zu", r + 1);
                for (int j = 0; j < nSpec; j++)
                    if (r < bySpec[j].size()) std::fprintf(f, "\t%f\t%f", U[bySpec[j][r]], rad[bySpec[j][r]]);
                    else std::fprintf(f, "\t\t");
                std::fprintf(f, "\n");
            }
            std::fclose(f);
        }
    aztot_free_device(md);
    aztot_free_md(model);
    const int spent = (int)(std::time(nullptr) - t0);
    std::printf("The program's just finished correctly, the running time: %d s\n", spent);
    return 0;
}
