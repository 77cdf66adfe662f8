// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the instruction classes the pair kernel is made of, on gfx950,
// at 1..8 waves per SIMD, and whether v_mfma_f64_16x16x4_f64 overlaps with fp64 VALU work of OTHER waves on the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates && tools/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kUnroll = 16;

// KIND: 0 v_fma_f64, 1 v_add_f64, 2 v_mul_f64, 3 v_alignbit_b32, 4 v_fma_f32, 5 v_pk_fma_f32, 6 v_rcp_f64, 7 mfma f64 16x16x4, 8 ds_read_b64,
//       9 v_cndmask_b32, 10 v_max_f64, 11 v_cmp_le_f64 (to vcc), 12 half the waves MFMA / half v_fma_f64, 13 v_max3_f32, 14 v_cvt_f32_f64
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(int iters, double* out, long long* cyc)
{
    __shared__ double lds[1024];
    lds[threadIdx.x] = threadIdx.x * 0.5; lds[threadIdx.x + 256] = 1.0; lds[threadIdx.x + 512] = 2.0; lds[threadIdx.x + 768] = 3.0;
    __syncthreads();
    double a[kUnroll], b = 1.0000001 + threadIdx.x * 1e-9, c = 0.9999999;
    float fa[kUnroll]; float fb = 1.0000001f, fc = 0.999f;
    unsigned ua[kUnroll];
    typedef float float2_t __attribute__((ext_vector_type(2)));
    float2_t pa[kUnroll], pb = {1.0000001f, 1.0000002f}, pc = {0.5f, 0.25f};
    double4_t acc[4];
    for (int k = 0; k < kUnroll; k++) { a[k] = 1.0 + k * 0.001 + threadIdx.x * 1e-6; fa[k] = 1.0f + k; ua[k] = k * 7919u + threadIdx.x; pa[k] = float2_t{1.0f + k, 2.0f + k}; }
    for (int k = 0; k < 4; k++) acc[k] = double4_t{0, 0, 0, 0};
    const int wave = threadIdx.x >> 6;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int k = 0; k < kUnroll; k++)
        {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            else if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            else if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            else if (KIND == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(ua[k]) : "v"(ua[(k + 1) % kUnroll]));
            else if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[k]) : "v"(fb), "v"(fc));
            else if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[k]) : "v"(pb), "v"(pc));
            else if (KIND == 6) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
            else if (KIND == 7) acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b, acc[k & 3], 0, 0, 0);
            else if (KIND == 8) { double v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)((threadIdx.x & 255) * 8 + (k & 3) * 2048))); a[k] = v; }
            else if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ua[k]) : "v"(ua[(k + 1) % kUnroll]));
            else if (KIND == 10) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            else if (KIND == 11) asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(a[k]), "v"(c) : "vcc");
            else if (KIND == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ua[k]) : "v"(ua[(k + 5) % kUnroll]));
            else if (KIND == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ua[k]) : "v"(ua[(k + 5) % kUnroll]));
            else if (KIND == 17) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(ua[k]));
            else if (KIND == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ua[k]) : "v"(ua[15]));
            else if (KIND == 19) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(ua[k]) : "s"(0x55555555u));
            else if (KIND == 20) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(fa[k]), "v"(fc) : "vcc");
            else if (KIND == 21) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(fa[k]) : "v"(fc));
            else if (KIND == 22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(fa[k]) : "v"(fb));
            else if (KIND == 23) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(ua[k]));
            else if (KIND == 24) asm volatile("v_ffbh_u32 %0, %0" : "+v"(ua[k]));
            else if (KIND == 25) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(ua[k]) : "v"(ua[15]));
            else if (KIND == 26) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(ua[k]) : "v"(ua[15]));
            else if (KIND == 13) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(fa[k]) : "v"(fb), "v"(fc));
            else if (KIND == 14) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[k]) : "v"(a[k]));
        }
        if (KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < kUnroll; k++) s += a[k] + fa[k] + ua[k] + pa[k].x + pa[k].y;
    for (int k = 0; k < 4; k++) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// MFMA + VALU in ONE wave's stream: per iteration 1 v_mfma_f64_16x16x4_f64 followed by NV independent v_fma_f64 (same wave);
// all waves run the same program.  Shows how much fp64 VALU work hides behind the 64-cycle matrix instruction.
template <int NV, int NMF>
__global__ __launch_bounds__(256) void k_mix(int iters, double* out)
{
    double a[16], b = 1.0000001 + threadIdx.x * 1e-9, c = 0.9999999;
    for (int k = 0; k < 16; k++) a[k] = 1.0 + k * 0.001 + threadIdx.x * 1e-6;
    double4_t acc[4];
    for (int k = 0; k < 4; k++) acc[k] = double4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
#pragma unroll
            for (int m = 0; m < NMF; m++) acc[(q + m) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b, acc[(q + m) & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(q * 4 + k) & 15]) : "v"(b), "v"(c));
        }
    }
    double s = 0;
    for (int k = 0; k < 16; k++) s += a[k];
    for (int k = 0; k < 4; k++) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
// the same with the f32-input matrix instruction (MT 0: v_mfma_f32_16x16x4_f32, 256 results) or the f16 one (MT 1: v_mfma_f32_16x16x32_f16)
template <int NV, int NMF, int MT>
__global__ __launch_bounds__(256) void k_mix32(int iters, double* out)
{
    double a[16], b = 1.0000001 + threadIdx.x * 1e-9, c = 0.9999999;
    for (int k = 0; k < 16; k++) a[k] = 1.0 + k * 0.001 + threadIdx.x * 1e-6;
    float4_t acc[4];
    for (int k = 0; k < 4; k++) acc[k] = float4_t{0, 0, 0, 0};
    float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
    half8_t ha, hb;
    for (int k = 0; k < 8; k++) { ha[k] = (_Float16)(0.01f * k + threadIdx.x * 1e-3f); hb[k] = (_Float16)(0.5f - 0.01f * k); }
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
#pragma unroll
            for (int m = 0; m < NMF; m++)
            {
                if (MT == 0) acc[(q + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[(q + m) & 3], 0, 0, 0);
                else acc[(q + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[(q + m) & 3], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NV; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(q * 4 + k) & 15]) : "v"(b), "v"(c));
        }
    }
    double s = 0;
    for (int k = 0; k < 16; k++) s += a[k];
    for (int k = 0; k < 4; k++) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NMF, int MT>
int run_mix32(int wavesPerSimd)
{
    const int blocks = 256 * wavesPerSimd, iters = 1000;
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix32<NV, NMF, MT>), dim3(blocks), dim3(256), 0, 0, 10, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix32<NV, NMF, MT>), dim3(blocks), dim3(256), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double groups = (double)iters * 4 * wavesPerSimd;
    printf("mix32: %d %s + %2d v_fma_f64 per group, waves/SIMD %d : %9.3f us -> %.1f ns per group per SIMD (= %.1f cycles at 2.4 GHz)\n", NMF,
           MT == 0 ? "mfma_f32_16x16x4_f32" : "mfma_f32_16x16x32_f16", NV, wavesPerSimd, ms * 1e3, ms * 1e6 / groups, ms * 1e6 / groups * 2.4);
    CK(hipFree(out));
    return 0;
}

template <int NV, int NMF>
int run_mix(int wavesPerSimd)
{
    const int blocks = 256 * wavesPerSimd, iters = 1000;
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<NV, NMF>), dim3(blocks), dim3(256), 0, 0, 10, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NV, NMF>), dim3(blocks), dim3(256), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double groups = (double)iters * 4 * wavesPerSimd;      // (NMF MFMA + NV fma) groups per SIMD
    printf("mix: %d MFMA_f64 + %2d v_fma_f64 per group, waves/SIMD %d : %9.3f us -> %.1f ns per group per SIMD (= %.1f cycles at 2.4 GHz)\n", NMF, NV, wavesPerSimd, ms * 1e3,
           ms * 1e6 / groups, ms * 1e6 / groups * 2.4);
    CK(hipFree(out));
    return 0;
}

template <int KIND>
int run(const char* name, int wavesPerSimd)
{
    // 256 CUs; a 256-thread block = 4 waves = one per SIMD; wavesPerSimd blocks per CU
    const int blocks = 256 * wavesPerSimd, iters = 2000;
    double* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256)); CK(hipMalloc(&cyc, sizeof(long long) * blocks));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, 10, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, iters, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // s_memtime ticks at 100 MHz on gfx9; report both the tick-based and the wall-based figure
    const double instrPerSimd = (double)iters * kUnroll * wavesPerSimd;
    printf("%-34s waves/SIMD %d : %8.3f us  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz) [memtime ticks/block %.0f]\n", name, wavesPerSimd,
           ms * 1e3, ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4, avg);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1)
    {   // only the matrix / fp64-vector co-execution part
        for (int w : {1, 4, 8})
        {
            run_mix32<0, 1, 0>(w); run_mix32<4, 1, 0>(w); run_mix32<8, 1, 0>(w); run_mix32<16, 1, 0>(w); run_mix32<16, 2, 0>(w); run_mix32<16, 0, 0>(w);
            run_mix32<0, 1, 1>(w); run_mix32<4, 1, 1>(w); run_mix32<8, 1, 1>(w); run_mix32<16, 1, 1>(w); run_mix32<16, 2, 1>(w);
        }
        return 0;
    }
    for (int w : {1, 4, 8})
    {
        run<0>("v_fma_f64", w); run<1>("v_add_f64", w); run<2>("v_mul_f64", w); run<10>("v_max_f64", w); run<11>("v_cmp_le_f64", w); run<6>("v_rcp_f64", w);
        run<3>("v_alignbit_b32", w); run<9>("v_cndmask_b32", w); run<4>("v_fma_f32", w); run<13>("v_max3_f32", w); run<14>("v_cvt_f32_f64", w); run<5>("v_pk_fma_f32", w);
        run<7>("v_mfma_f64_16x16x4_f64", w); run<8>("ds_read_b64", w);
        run<15>("v_add_u32", w); run<16>("v_and_b32", w); run<17>("v_lshlrev_b32", w); run<18>("v_cndmask_b32 (indep)", w); run<19>("v_mbcnt_lo", w);
        run<20>("v_cmp_gt_f32", w); run<21>("v_sub_f32", w); run<22>("v_mul_f32", w); run<23>("v_bfe_u32", w); run<24>("v_ffbh_u32", w); run<25>("v_mad_u32_u24", w);
        run<26>("v_lshl_add_u32", w);
        run_mix<0, 1>(w); run_mix<4, 1>(w); run_mix<8, 1>(w); run_mix<12, 1>(w); run_mix<16, 1>(w); run_mix<16, 0>(w); run_mix<8, 0>(w);
    }
    return 0;
}
