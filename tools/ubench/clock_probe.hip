// Does the shader clock hold 2.4 GHz under a chip-wide VALU+transcendental load?  Runs the stage-D inner-loop
// instruction mix for ~tens of ms on every CU and reports shader-clock cycles (s_memtime) per 100 MHz
// wall-clock tick (s_memrealtime), i.e. the effective frequency, plus clocks per element pair.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k_mix(float *out, long long *clk, int iters, float seed)
{
    f2 x[4], e[4], b[4], sA = {0, 0}, sAA = {0, 0}, sAb = {0, 0};
    for (int i = 0; i < 4; ++i) { x[i] = (f2){seed + threadIdx.x + i, seed * 3 + i}; e[i] = (f2){seed * 2 + i, threadIdx.x * 0.5f + i}; b[i] = (f2){0.25f * i, 0.125f}; }
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f2 t = x[i] - e[i];
                f2 a;
                if (MODE == 0) {        // packed form (med3 sign)
                    f2 q = t * 0x1p54f, sg;
                    sg.x = __builtin_amdgcn_fmed3f(q.x, -1.f, 1.f); sg.y = __builtin_amdgcn_fmed3f(q.y, -1.f, 1.f);
                    f2 u = __builtin_elementwise_fma(t, sg, (f2){1e-10f, 1e-10f});
                    f2 s; s.x = __builtin_amdgcn_sqrtf(u.x); s.y = __builtin_amdgcn_sqrtf(u.y);
                    a = s * sg;
                } else if (MODE == 1) { // scalar form (cmp / bfi / cndmask)
                    float sx = __builtin_amdgcn_sqrtf(fabsf(t.x) + 1e-10f), sy = __builtin_amdgcn_sqrtf(fabsf(t.y) + 1e-10f);
                    a.x = fabsf(t.x) < 1e-16f ? 0.f : copysignf(sx, t.x);
                    a.y = fabsf(t.y) < 1e-16f ? 0.f : copysignf(sy, t.y);
                } else {                // no transform
                    a = t;
                }
                sA += a; sAA = __builtin_elementwise_fma(a, a, sAA); sAb = __builtin_elementwise_fma(a, b[i], sAb);
                x[i] += (f2){1e-3f, -1e-3f};
            }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = sA.x + sA.y + sAA.x + sAA.y + sAb.x + sAb.y;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int MODE> static void run(const char *name, int blocks, int iters)
{
    float *out; long long *clk, *h = new long long[2 * blocks];
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_mix<MODE><<<blocks, 256>>>(out, clk, 10, 1.f); hipDeviceSynchronize();
    hipEventRecord(e0); k_mix<MODE><<<blocks, 256>>>(out, clk, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost);
    double sc = 0, wc = 0; for (int i = 0; i < blocks; ++i) { sc += h[2 * i]; wc += h[2 * i + 1]; }
    const double pairs = (double)iters * 12;                 // element PAIRS per lane
    printf("%-28s blocks=%5d %8.2f ms  shader clk/wall tick = %6.2f (x100 MHz)  %6.2f shader-clk per 2 elements per wave (4 waves/SIMD: x1/4 per SIMD)  %6.2f ns-based clk@2.4\n",
           name, blocks, ms, sc / wc, sc / blocks / pairs, ms * 1e-3 * 2.4e9 / pairs);
    hipFree(out); hipFree(clk); delete[] h;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int blocks : {cus * 4, cus / 8 * 4}) {       // whole chip vs 1/8 of it (4 waves per SIMD either way)
        run<0>("packed med3 form", blocks, 60000);
        run<1>("scalar cmp/bfi/cndmask form", blocks, 60000);
        run<2>("no transform", blocks, 60000);
    }
    return 0;
}
