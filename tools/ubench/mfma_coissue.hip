// Round 6 microbenchmark: does the f64 matrix pipe (v_mfma_f64_4x4x4_4b_f64 / v_mfma_f64_16x16x4_f64) run BESIDE the f64 vector ALU on gfx950,
// or do the two share their multipliers?  Stage D's f64 element is issue-bound on the vector ALU (0.88-0.91 of the f64 issue rate); its three
// moment updates (v_add_f64 + 2 v_fma_f64 per element) could be handed to the matrix pipe as same-lane "diagonal" products (A = a, B = 1 | a | b:
// D[i][i] accumulates the lane's own product) IF the matrix instructions cost the vector ALU nothing.  Four waves per SIMD, as in the kernel.
//   fma     : NV v_fma_f64 per step                                  (the vector-ALU yardstick)
//   mfma4   : NM v_mfma_f64_4x4x4_4b_f64 per step                    (the matrix pipe alone)
//   mfma16  : NM v_mfma_f64_16x16x4_f64 per step
//   mix*    : both in one wave's instruction stream, NV : NM          (co-issue inside a wave / across the waves of a SIMD)
//   split   : waves 0,2 of a SIMD issue only vector work, waves 1,3 only matrix work
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define ITER 2000

template <int NV, int NM, int KIND, bool SPLIT>
__global__ void __launch_bounds__(256) k_mix(double *out, double seed)
{
    double a[8], acc4[4];
    v4d acc16[2];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
    for (int i = 0; i < 4; ++i) acc4[i] = 0.0;
    acc16[0] = acc16[1] = v4d{0, 0, 0, 0};
    const double b0 = 1.0000001, b1 = 1e-9;
    const int wave = threadIdx.x >> 6;
    const bool do_v = !SPLIT || (blockIdx.x & 1) == 0, do_m = !SPLIT || (blockIdx.x & 1) == 1;   // (SPLIT: whole workgroups alternate; 2 + 2 waves per SIMD)
    (void)wave;
    for (int it = 0; it < ITER; ++it) {
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(b0), "v"(b1));
        }
        if (do_m) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (KIND == 4) acc4[i & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b0, acc4[i & 3], 0, 0, 0);
                else acc16[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b0, acc16[i & 1], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc4[i];
    s += acc16[0][0] + acc16[1][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// roles by wave: in a 1024-thread workgroup wave w runs on SIMD w % 4; waves with (w >> 2) & 1 == 0 issue only vector work, the others only
// matrix work; two workgroups per CU -> 4 vector + 4 matrix waves per SIMD.  mode 0: both roles, 1: the matrix waves exit at once, 2: the
// vector waves exit at once.  INDEP: the matrix operands are registers no vector instruction writes.
template <int NV, int NM, int KIND>
__global__ void __launch_bounds__(1024) k_roles(double *out, double seed, int mode)
{
    double a[8], acc4[4];
    v4d acc16[2];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
    for (int i = 0; i < 4; ++i) acc4[i] = 0.0;
    acc16[0] = acc16[1] = v4d{0, 0, 0, 0};
    const double b0 = 1.0000001, b1 = 1e-9, c0 = seed * 0.5 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const bool mrole = (wave >> 2) & 1;
    if ((mode == 1 && mrole) || (mode == 2 && !mrole)) return;
    if (!mrole) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(b0), "v"(b1));
        }
    } else {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (KIND == 4) acc4[i & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(c0, b0, acc4[i & 3], 0, 0, 0);
                else acc16[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, b0, acc16[i & 1], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc4[i];
    s += acc16[0][0] + acc16[1][1];
    out[(blockIdx.x * 1024 + threadIdx.x) % (256 * 4 * 256)] = s;
}

// one stream, independent operands: NV vector + NM matrix instructions per step, the matrix operands never written by the vector ones
template <int NV, int NM, int KIND>
__global__ void __launch_bounds__(256) k_indep(double *out, double seed)
{
    double a[8], acc4[4];
    v4d acc16[2];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
    for (int i = 0; i < 4; ++i) acc4[i] = 0.0;
    acc16[0] = acc16[1] = v4d{0, 0, 0, 0};
    const double b0 = 1.0000001, b1 = 1e-9, c0 = seed * 0.5 + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < (NV > NM ? NV : NM); ++i) {
            if (i < NV) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(b0), "v"(b1));
            if (i * NM / (NV > NM ? NV : NM) != (i + 1) * NM / (NV > NM ? NV : NM)) {      // NM matrix instructions spread evenly among the NV vector ones
                const int j = i * NM / (NV > NM ? NV : NM);
                if (KIND == 4) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc4[j & 3]) : "v"(c0), "v"(b0));
                else acc16[j & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, b0, acc16[j & 1], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc4[i];
    s += acc16[0][0] + acc16[1][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K> static double run_roles(K kern, const char *name, int mode, double *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 2;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, out, 1.0, mode);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, out, 1.0, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-60s mode %d  %8.3f ms\n", name, mode, best);
    return best;
}

template <typename K> static double run(K kern, const char *name, int nv, int nm, double *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;                                // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 4 waves x ITER steps; microseconds per step of ONE wave's work when four share the SIMD
    const double ns_per_step = best * 1e6 / (4.0 * ITER);
    printf("%-34s NV=%2d NM=%2d  %8.3f ms   %7.2f ns per wave-step", name, nv, nm, best, ns_per_step);
    if (nv && !nm) printf("   = %.2f ns per v_fma_f64", ns_per_step / nv);
    if (nm && !nv) printf("   = %.2f ns per mfma", ns_per_step / nm);
    printf("\n");
    return best;
}

int main()
{
    double *out; hipMalloc(&out, 256 * 4 * 256 * sizeof(double));
    printf("# 256 CUs x 4 workgroups x 4 waves; times are whole launches of %d steps\n", ITER);
    const double f15 = run(k_mix<15, 0, 4, false>, "fma only", 15, 0, out);
    const double f12 = run(k_mix<12, 0, 4, false>, "fma only", 12, 0, out);
    const double m3 = run(k_mix<0, 3, 4, false>, "mfma 4x4x4 only", 0, 3, out);
    const double m1 = run(k_mix<0, 1, 4, false>, "mfma 4x4x4 only", 0, 1, out);
    const double M3 = run(k_mix<0, 3, 16, false>, "mfma 16x16x4 only", 0, 3, out);
    const double M1 = run(k_mix<0, 1, 16, false>, "mfma 16x16x4 only", 0, 1, out);
    const double x1 = run(k_mix<12, 3, 4, false>, "mix 4x4x4, same stream", 12, 3, out);
    const double x2 = run(k_mix<15, 1, 4, false>, "mix 4x4x4, same stream", 15, 1, out);
    const double x3 = run(k_mix<12, 3, 16, false>, "mix 16x16x4, same stream", 12, 3, out);
    const double x4 = run(k_mix<12, 1, 16, false>, "mix 16x16x4, same stream", 12, 1, out);
    const double s1 = run(k_mix<24, 6, 4, true>, "split workgroups 4x4x4", 24, 6, out);
    const double s2 = run(k_mix<24, 2, 16, true>, "split workgroups 16x16x4", 24, 2, out);
    const double sv = run(k_mix<24, 0, 4, true>, "split, vector half alone", 24, 0, out);
    const double sm = run(k_mix<0, 6, 4, true>, "split, matrix half alone (4x4x4)", 0, 6, out);
    printf("# co-issue test: mix(12 fma + 3 mfma4) = %.3f ms; sum of the parts %.3f, max of the parts %.3f\n", x1, f12 + m3, f12 > m3 ? f12 : m3);
    printf("# co-issue test: mix(15 fma + 1 mfma4) = %.3f ms; sum of the parts %.3f, max of the parts %.3f\n", x2, f15 + m1, f15 > m1 ? f15 : m1);
    printf("# co-issue test: mix(12 fma + 3 mfma16) = %.3f ms; sum %.3f, max %.3f\n", x3, f12 + M3, f12 > M3 ? f12 : M3);
    printf("# co-issue test: mix(12 fma + 1 mfma16) = %.3f ms; sum %.3f, max %.3f\n", x4, f12 + M1, f12 > M1 ? f12 : M1);
    printf("# split workgroups (2 vector + 2 matrix waves per SIMD): both %.3f ms (4x4x4) / %.3f (16x16x4); vector half alone %.3f, matrix half alone %.3f\n", s1, s2, sv, sm);
    printf("# ---- one stream, matrix operands independent of the vector instructions\n");
    const double i1 = run(k_indep<12, 3, 4>, "indep mix 4x4x4", 12, 3, out);
    const double i2 = run(k_indep<15, 3, 4>, "indep mix 4x4x4", 15, 3, out);
    const double i3 = run(k_indep<15, 0, 4>, "indep fma only", 15, 0, out);
    const double i4 = run(k_indep<0, 3, 4>, "indep mfma4 only", 0, 3, out);
    printf("# indep: mix(12+3) %.3f, mix(15+3) %.3f ms; fma15 alone %.3f, mfma4 x3 alone %.3f\n", i1, i2, i3, i4);
    printf("# ---- roles by wave, 4 vector + 4 matrix waves per SIMD (two 1024-thread workgroups per CU)\n");
    run_roles(k_roles<15, 3, 4>, "15 fma | 3 mfma4 per step: both", 0, out);
    run_roles(k_roles<15, 3, 4>, "15 fma | 3 mfma4 per step: vector waves only", 1, out);
    run_roles(k_roles<15, 3, 4>, "15 fma | 3 mfma4 per step: matrix waves only", 2, out);
    run_roles(k_roles<15, 6, 4>, "15 fma | 6 mfma4 per step: both", 0, out);
    run_roles(k_roles<15, 6, 4>, "15 fma | 6 mfma4 per step: matrix waves only", 2, out);
    run_roles(k_roles<15, 1, 16>, "15 fma | 1 mfma16 per step: both", 0, out);
    run_roles(k_roles<15, 1, 16>, "15 fma | 1 mfma16 per step: matrix waves only", 2, out);
    return 0;
}
