// Accuracy of the raw gfx950 v_rcp_f64 / v_rsq_f64 and of the refinement steps fast_rcp / fast_rsqrt build on them (kernels.hip.h):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/rcp_accuracy.hip -o tools/ubench/rcp_accuracy && tools/ubench/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k(const double* x, double* raw, double* n1, double* c3, double* rsraw, double* rs1, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rcp(v);
    raw[i] = y;
    const double e = fma(-v, y, 1.0);
    n1[i] = fma(y, e, y);                       // one Newton step: 2 FMAs
    c3[i] = fma(y, fma(e, e, e), y);            // the cubic step fast_rcp uses: 3 FMAs
    double r = __builtin_amdgcn_rsq(v);
    rsraw[i] = r;
    const double h = fma(-v * r, r, 1.0);       // 1 - v r^2
    rs1[i] = fma(0.5 * r, h, r);                // one Newton step
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < n; i++)
    {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x[i] = 1.0 + 99.0 * (double)(s >> 11) / 9007199254740992.0;       // squared distances of a liquid: 1 .. 100 A^2
    }
    double *dx, *d[5];
    hipMalloc(&dx, sizeof(double) * n);
    for (auto& p : d) hipMalloc(&p, sizeof(double) * n);
    hipMemcpy(dx, x.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], d[4], n);
    std::vector<double> o(n);
    const char* names[5] = {"v_rcp_f64 raw", "rcp + 1 Newton step (2 FMA)", "rcp + cubic step (3 FMA)", "v_rsq_f64 raw", "rsq + 1 Newton step"};
    for (int q = 0; q < 5; q++)
    {
        hipMemcpy(o.data(), d[q], sizeof(double) * n, hipMemcpyDeviceToHost);
        long double worst = 0;
        for (int i = 0; i < n; i++)
        {
            const long double exact = (q < 3) ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
            const long double rel = fabsl(((long double)o[i] - exact) / exact);
            if (rel > worst) worst = rel;
        }
        printf("%-30s max relative error %.3Le (%.2Lf ulp of 2^-53)\n", names[q], worst, worst / 1.1102230246251565e-16L);
    }
    return 0;
}
