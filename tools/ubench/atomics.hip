// build: hipcc -O3 --offload-arch=gfx950 atomics.hip -o atomics   (run: tools/ubench/atomics on an MI355X)
// Micro-benchmark: cost of scattering half-shell pair forces with fp64 global atomics on MI355X.
// 42^3 cells x 14 atoms; one wave per cell adds to the atoms of its 13 forward neighbour cells + itself (3 components),
// the access pattern a Newton-3 version of the pair kernel would have.  Variants: device-scope hardware atomics,
// plain (non-atomic, wrong but shows the store cost) read-modify-write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NC = 42, APC = 14;

template <int VARIANT>
__global__ __launch_bounds__(64) void k_scatter(double* fx, double* fy, double* fz, int nCells)
{
    const int per = (nCells + 7) >> 3;
    const int cr = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (cr >= nCells) return;
    const int cz = cr % NC, cy = (cr / NC) % NC, cx = cr / (NC * NC);
    const int lane = threadIdx.x;
    // 14 cells: own + 13 forward; entries = 14 cells x 14 atoms = 196 -> 4 rounds of 64 lanes (last partial)
    for (int r = 0; r < 4; r++)
    {
        const int e = r * 64 + lane;
        if (e >= 14 * APC) break;
        const int n = e / APC, a = e - n * APC;
        int ox, oy, oz;
        if (n == 0) { ox = 0; oy = 0; oz = 0; }
        else if (n == 1) { ox = 0; oy = 0; oz = 1; }
        else if (n < 5) { ox = 0; oy = 1; oz = n - 3; }
        else { ox = 1; oy = (n - 5) / 3 - 1; oz = (n - 5) % 3 - 1; }
        const int nx = (cx + ox) % NC, ny = (cy + oy + NC) % NC, nz = (cz + oz + NC) % NC;
        const int j = ((nx * NC + ny) * NC + nz) * APC + a;
        const double v = 1e-3 * (lane + 1);
        if (VARIANT == 0) { unsafeAtomicAdd(&fx[j], v); unsafeAtomicAdd(&fy[j], v); unsafeAtomicAdd(&fz[j], v); }
        else if (VARIANT == 1) { atomicAdd(&fx[j], v); atomicAdd(&fy[j], v); atomicAdd(&fz[j], v); }
        else if (VARIANT == 2) { fx[j] += v; fy[j] += v; fz[j] += v; }
        else
        {
            __hip_atomic_fetch_add(&fx[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&fy[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&fz[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

int main()
{
    const int nCells = NC * NC * NC, N = nCells * APC;
    double *fx, *fy, *fz;
    CK(hipMalloc(&fx, 8 * N)); CK(hipMalloc(&fy, 8 * N)); CK(hipMalloc(&fz, 8 * N));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 8 * ((nCells + 7) / 8);
    const char* names[] = {"unsafeAtomicAdd (device scope, hw)", "atomicAdd (default)", "plain rmw (racy)", "workgroup-scope atomic"};
    for (int v = 0; v < 4; v++)
    {
        CK(hipMemset(fx, 0, 8 * N)); CK(hipMemset(fy, 0, 8 * N)); CK(hipMemset(fz, 0, 8 * N));
        float best = 1e9;
        for (int rep = 0; rep < 12; rep++)
        {
            CK(hipEventRecord(a));
            if (v == 0) hipLaunchKernelGGL(k_scatter<0>, dim3(grid), dim3(64), 0, 0, fx, fy, fz, nCells);
            if (v == 1) hipLaunchKernelGGL(k_scatter<1>, dim3(grid), dim3(64), 0, 0, fx, fy, fz, nCells);
            if (v == 2) hipLaunchKernelGGL(k_scatter<2>, dim3(grid), dim3(64), 0, 0, fx, fy, fz, nCells);
            if (v == 3) hipLaunchKernelGGL(k_scatter<3>, dim3(grid), dim3(64), 0, 0, fx, fy, fz, nCells);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 2 && ms < best) best = ms;
        }
        std::vector<double> h(N);
        CK(hipMemcpy(h.data(), fx, 8 * N, hipMemcpyDeviceToHost));
        double s = 0; for (double x : h) s += x;
        printf("%-40s %8.1f us   checksum %.6f (atomics per launch: %d)\n", names[v], best * 1e3, s / 12.0, nCells * 14 * APC * 3);
    }
    return 0;
}
