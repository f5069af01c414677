// f64 edition of valu_issue.hip (round 3): issue cost of the instruction classes the f64 stage-D element is made of, and of the
// element itself (literal partial-sqrt rule of speedboosted.pyx:372-378 + the three moment updates) in the forms the kernel can take.
// Shader cycles from s_memtime per wave, 4 resident waves per SIMD like the kernel; clocks per wave64 instruction (classes) and per
// ELEMENT (element kernels: one element per lane = 64 pair-genes per wave step).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue_f64 valu_issue_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include <algorithm>
#define ITER 500
#define REP 4
#define KD(name, body)                                                                                    \
    __global__ void __launch_bounds__(256) name(double *out, unsigned long long *cyc, double seed)        \
    {                                                                                                     \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        double b0 = 1.0001, b1 = 0.9999;                                                                  \
        float f0 = (float)seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < ITER; ++i) {                                                                  \
            _Pragma("unroll") for (int r = 0; r < REP; ++r) { body }                                      \
        }                                                                                                 \
        asm volatile("s_nop 0" ::: "memory");                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7); \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                 \
    }
#define A8(op) op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7)
#define F8(op) op(f0, a0) op(f1, a1) op(f2, a2) op(f3, a3) op(f4, a4) op(f5, a5) op(f6, a6) op(f7, a7)
#define DADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b1));
#define DADDABS(x) asm volatile("v_add_f64 %0, |%0|, %1" : "+v"(x) : "v"(b1));
#define DMUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b1));
#define DFMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b0), "v"(b1));
#define DRSQ(x) asm volatile("v_rsq_f64_e32 %0, %0" : "+v"(x));
#define DCMP(x) asm volatile("v_cmp_lt_f64_e64 vcc, |%0|, %1" : : "v"(x), "v"(b1) : "vcc");
#define DCND(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(f0) : "v"(f1));
#define CVT32(f, d) asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(f) : "v"(d));
#define CVT64(f, d) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d) : "v"(f));
#define LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, -1" : "+v"(x));
KD(k_dadd, A8(DADD) A8(DADD))
KD(k_daddabs, A8(DADDABS) A8(DADDABS))
KD(k_dmul, A8(DMUL) A8(DMUL))
KD(k_dfma, A8(DFMA) A8(DFMA))
KD(k_drsq, A8(DRSQ) A8(DRSQ))
KD(k_dcmp, A8(DCMP) A8(DCMP))
KD(k_cvt32, F8(CVT32) F8(CVT32))
KD(k_cvt64, F8(CVT64) F8(CVT64))
KD(k_ldexp, A8(LDEXP) A8(LDEXP))

// ---- element kernels: NE independent elements per lane and step, the kernel's own arithmetic (compiled, not asm)
__device__ __forceinline__ double sqrt_lib_iter(double x)           // what k_cdc_partial_grouped<double> runs today (coldeltacor.hip sqrt_normal_f64)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double sqrt_f32seed(double x)            // candidate: f32 reciprocal square root as the seed (2^-23), one coupled
{                                                                    // Goldschmidt step (2^-45) and one residual correction (< 1 ulp)
    const float yf = __builtin_amdgcn_rsqf((float)x);
    const double y = (double)yf, h = (double)(0.5f * yf);
    double g = x * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double sqrt_f32seed2(double x)           // candidate with the second correction (bitwise the correctly rounded root almost always)
{
    const float yf = __builtin_amdgcn_rsqf((float)x);
    const double y = (double)yf;
    double h = (double)(0.5f * yf);
    double g = x * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double sqrt_f32prod(double x)            // round 4 (what the kernel runs now): the f32 PRODUCT x_f * rsq(x_f) as a 24-bit seed
{                                                                    // of the root - its square is exact in f64, so the residual x - s0^2 is exact -
    const float xf = (float)x;                                       // and two Newton corrections with h = rsq / 2 (exponent decrement, an integer
    const float yf = __builtin_amdgcn_rsqf(xf);                      // op): 2^-22 -> 2^-44 -> below half an ulp.  One f64 multiply less than the
    const double s0 = (double)(xf * yf);                             // Goldschmidt form with one correction, and closer to the correctly rounded root
    const double y = (double)yf;
    const double h = __hiloint2double(__double2hiint(y) - 0x00100000, __double2loint(y));
    double d = fma(-s0, s0, x);
    const double s1 = fma(d, h, s0);
    d = fma(-s1, s1, x);
    return fma(d, h, s1);
}
__device__ __forceinline__ double sqrt_seedzero(double x, bool discard)   // round 4, second step (what the kernel runs now): the zero rule selects the ARGUMENT of
{                                                                          // v_rsq_f32 (+inf -> y_f = 0 -> s0 = h = 0 -> the root is an exact 0: one v_cndmask, not two);
    const float xf0 = (float)x;                                            // h = y_f / 2 halved in f32 (an exponent decrement would turn the zero into -inf)
    const float xf = discard ? __builtin_inff() : xf0;
    const float yf = __builtin_amdgcn_rsqf(xf);
    const double s0 = (double)(xf0 * yf), h = (double)(0.5f * yf);
    double d = fma(-s0, s0, x);
    const double s1 = fma(d, h, s0);
    d = fma(-s1, s1, x);
    return fma(d, h, s1);
}
__device__ __forceinline__ double sqrt_seedzero_f64prod(double x, bool discard)   // measured and not kept: s0 = x y, h = y / 2 as f64 products of the one converted
{                                                                                  // seed (17 instructions per element, two more on the f64 multiplier)
    const float xf = discard ? __builtin_inff() : (float)x;
    const double y = (double)__builtin_amdgcn_rsqf(xf);
    const double s0 = x * y, h = 0.5 * y;
    double d = fma(-s0, s0, x);
    const double s1 = fma(d, h, s0);
    d = fma(-s1, s1, x);
    return fma(d, h, s1);
}
template <int MODE> __device__ __forceinline__ double elem(double t, double psc)
{
    if (MODE == 5) return copysign(sqrt_seedzero(fabs(t) + psc, fabs(t) < 1e-16), t);
    if (MODE == 6) return copysign(sqrt_seedzero_f64prod(fabs(t) + psc, fabs(t) < 1e-16), t);
    const double a = fabs(t) + psc;
    const double s = MODE == 0 ? sqrt_lib_iter(a) : (MODE == 1 ? sqrt_f32seed(a) : (MODE == 2 ? sqrt_f32seed2(a) : (MODE == 4 ? sqrt_f32prod(a) : sqrt(a))));
    return (fabs(t) < 1e-16) ? 0.0 : copysign(s, t);
}
template <int MODE>
__global__ void __launch_bounds__(256) k_elem(double *out, unsigned long long *cyc, double seed)
{
    constexpr int NE = 4;
    double x[NE], e[NE], b[NE], sA[2] = {0, 0}, sAA[2] = {0, 0}, sAb[2] = {0, 0};
    for (int k = 0; k < NE; ++k) { x[k] = seed * (threadIdx.x + k + 1); e[k] = 0.37 * (k + 1); b[k] = 0.11 * (threadIdx.x + k); }
    const double psc = seed * 1e-10;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER * REP; ++i) {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const double a = elem<MODE>(x[k] - e[k], psc);
            sA[k & 1] += a;
            sAA[k & 1] = fma(a, a, sAA[k & 1]);
            sAb[k & 1] = fma(a, b[k], sAb[k & 1]);
            x[k] += 0.5;                                    // keeps the compiler from hoisting the element out of the loop
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = sA[0] + sA[1] + sAA[0] + sAA[1] + sAb[0] + sAb[1];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// accuracy of the candidates against the library square root over a sweep of magnitudes
__global__ void k_sqrt_err(double *maxrel, int n)
{
    double m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    unsigned long long off = 0, off4 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = exp2(-60.0 + 120.0 * (double)i / n) * (1.0 + 1e-3 * (i % 997));
        const double s = sqrt(x);
        m1 = fmax(m1, fabs(sqrt_f32seed(x) - s) / s);
        m2 = fmax(m2, fabs(sqrt_f32seed2(x) - s) / s);
        const double p = sqrt_f32prod(x);
        m3 = fmax(m3, fabs(p - s) / s);
        off += p != s;
        const double q = sqrt_seedzero(x, false);
        m4 = fmax(m4, fabs(q - s) / s);
        off4 += q != s;
        if (sqrt_seedzero(x, true) != 0.0) off4 += 1ull << 40;          // a discarded element must be an exact zero
    }
    atomicMax((unsigned long long *)&maxrel[4], (unsigned long long)__double_as_longlong(m4));
    atomicAdd((unsigned long long *)&maxrel[5], off4);
    atomicMax((unsigned long long *)&maxrel[0], (unsigned long long)__double_as_longlong(m1));
    atomicMax((unsigned long long *)&maxrel[1], (unsigned long long)__double_as_longlong(m2));
    atomicMax((unsigned long long *)&maxrel[2], (unsigned long long)__double_as_longlong(m3));
    atomicAdd((unsigned long long *)&maxrel[3], off);
}

template <typename F> static void run(const char *name, F k, int wps, double units_per_iter, const char *unit)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * wps;
    double *out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    unsigned long long *cyc; hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, cyc, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, 256>>>(out, cyc, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], mx = (double)h.back();
    const double per_simd = (double)ITER * REP * units_per_iter * wps;
    printf("%-44s w/SIMD=%d  wall %7.3f ms  -> %6.2f clk per %s per SIMD (med), %6.2f (max; tick rate %.0f MHz)\n", name, wps, ms, med / per_simd, unit,
           mx / per_simd, mx / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}
int main()
{
    const int w = 4;
    run("v_add_f64", k_dadd, w, 16, "instr"); run("v_add_f64 |v|,v", k_daddabs, w, 16, "instr"); run("v_mul_f64", k_dmul, w, 16, "instr");
    run("v_fma_f64", k_dfma, w, 16, "instr"); run("v_rsq_f64", k_drsq, w, 16, "instr"); run("v_cmp_lt_f64 |v|,v", k_dcmp, w, 16, "instr");
    run("v_cvt_f32_f64", k_cvt32, w, 16, "instr"); run("v_cvt_f64_f32", k_cvt64, w, 16, "instr"); run("v_ldexp_f64", k_ldexp, w, 16, "instr");
    run("f64 element, v_rsq_f64 + lib iteration (today)", k_elem<0>, w, 4, "element");
    run("f64 element, f32 seed + 1 step + 1 correction", k_elem<1>, w, 4, "element");
    run("f64 element, f32 seed + 1 step + 2 corrections", k_elem<2>, w, 4, "element");
    run("f64 element, f32 product seed + 2 Newton corrections (round 4)", k_elem<4>, w, 4, "element");
    run("f64 element, zero rule on the seed's argument (round 4, second step: the kernel's)", k_elem<5>, w, 4, "element");
    run("f64 element, the same with s0 = x y and h = y / 2 as f64 products (not kept)", k_elem<6>, w, 4, "element");
    run("f64 element, library sqrt()", k_elem<3>, w, 4, "element");
    double *mr; hipMalloc(&mr, 48); hipMemset(mr, 0, 48);
    k_sqrt_err<<<1024, 256>>>(mr, 1 << 26);
    double h[6]; hipMemcpy(h, mr, 48, hipMemcpyDeviceToHost);
    unsigned long long noff, noff4; memcpy(&noff, &h[3], 8); memcpy(&noff4, &h[5], 8);
    printf("max relative error against sqrt() over 2^26 arguments in [2^-60, 2^60]: f32 seed + 1 correction %.3g, + 2 corrections %.3g, f32 product seed + 2 Newton "
           "corrections %.3g with %llu results not equal to sqrt() bit for bit (ulp = 1.1e-16); the kernel's element (zero rule on the seed's argument) %.3g with %llu "
           "(2^40 x the discarded elements that are not an exact 0 would show here)\n", h[0], h[1], h[2], noff, h[4], noff4);
    return 0;
}
