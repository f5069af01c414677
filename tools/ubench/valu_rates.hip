// VALU issue-rate microbenchmark for gfx950: clocks per wave64 instruction with 4 waves per SIMD resident.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITER 2000
#define K(name, body)                                                                                   \
    __global__ void __launch_bounds__(256) name(float *out, float seed)                                 \
    {                                                                                                   \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = 1.0001f, b1 = 0.9999f;                                                               \
        typedef float f2 __attribute__((ext_vector_type(2)));                                           \
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};                    \
        for (int i = 0; i < ITER; ++i) {                                                                \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { body }                                \
        }                                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y; \
    }
#define A8(op) op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7)
#define P8(op) op(p0) op(p1) op(p2) op(p3) op(p0) op(p1) op(p2) op(p3)
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b0), "v"(b1));
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define RSQ(x) asm volatile("v_rsq_f32 %0, %0" : "+v"(x));
#define MED3(x) asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(x));
#define BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b0), "v"(b1));
#define CND(x) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x));
#define CMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(b1) : "vcc");
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(q));
#define PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(q));
#define PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(q));
#define MULE32(x) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define SUBE32(x) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define FMAC(x) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(b0), "v"(b1));
#define MULCLAMP(x) asm volatile("v_mul_f32_e64 %0, |%0|, %1 clamp" : "+v"(x) : "v"(b1));
#define FMAS(x) asm volatile("v_fma_f32 %0, s4, %0, |%1|" : "+v"(x) : "v"(b1) : "s4");
#define ANDB(x) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define MAXF(x) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define MOVB(x) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x) : "v"(b1));
#define ADDDPPQ(x) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define ADDDPPB(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
#define ADDDPPM(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
#define CNDE32(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(b1));
#define ADD3(x) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(x) : "v"(b0), "v"(b1));
#define MIXS(x) asm volatile("v_sqrt_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(a7) : "v"(b0), "v"(b1));
K(k_fma, A8(FMA))
K(k_add, A8(ADD))
K(k_sqrt, A8(SQRT))
K(k_rsq, A8(RSQ))
K(k_med3, A8(MED3))
K(k_bfi, A8(BFI))
K(k_cnd, A8(CND))
K(k_cmp, A8(CMP))
K(k_pkfma, P8(PKFMA))
K(k_pkadd, P8(PKADD))
K(k_pkmul, P8(PKMUL))
K(k_mule32, A8(MULE32))
K(k_sube32, A8(SUBE32))
K(k_fmac, A8(FMAC))
K(k_mulclamp, A8(MULCLAMP))
K(k_fmas, A8(FMAS))
K(k_andb, A8(ANDB))
K(k_maxf, A8(MAXF))
K(k_movb, A8(MOVB))
K(k_adddppq, A8(ADDDPPQ))
K(k_adddppb, A8(ADDDPPB))
K(k_adddppm, A8(ADDDPPM))
K(k_cnde32, A8(CNDE32))
K(k_add3, A8(ADD3))
K(k_mix_sqrt_3fma, MIXS(a0) MIXS(a1) MIXS(a2) MIXS(a3) MIXS(a4) MIXS(a5) MIXS(a6) MIXS(a0))
template <typename F> static void run(const char *name, F k, int waves_per_simd, double instr_per_rep_unit)
{
    float *out; hipMalloc(&out, 4 << 20);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * waves_per_simd;      // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, 256>>>(out, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double clk = p.clockRate * 1e3;                            // Hz
    const double instr = (double)ITER * REP * instr_per_rep_unit * waves_per_simd;   // wave-instructions per SIMD
    printf("%-18s waves/SIMD=%d  %8.3f ms  %6.2f clk/wave-instr (at %.2f GHz nominal)\n", name, waves_per_simd, ms, ms * 1e-3 * clk / instr, clk * 1e-9);
    hipFree(out);
}
int main()
{
    for (int w : {4}) {
        run("v_fma_f32", k_fma, w, 1); run("v_add_f32", k_add, w, 1); run("v_sqrt_f32", k_sqrt, w, 1); run("v_rsq_f32", k_rsq, w, 1);
        run("v_med3_f32", k_med3, w, 1); run("v_bfi_b32", k_bfi, w, 1); run("v_cndmask_b32", k_cnd, w, 1); run("v_cmp_lt_f32", k_cmp, w, 1);
        run("v_pk_fma_f32", k_pkfma, w, 1); run("v_pk_add_f32", k_pkadd, w, 1); run("v_pk_mul_f32", k_pkmul, w, 1);
        run("v_mul_f32_e32", k_mule32, w, 1); run("v_sub_f32_e32", k_sube32, w, 1); run("v_fmac_f32_e32", k_fmac, w, 1);
        run("v_mul |x| clamp e64", k_mulclamp, w, 1); run("v_fma s,v,|v|", k_fmas, w, 1); run("v_and_b32", k_andb, w, 1);
        run("v_max_f32", k_maxf, w, 1); run("v_mov_b32", k_movb, w, 1); run("v_add_dpp quad_perm", k_adddppq, w, 1);
        run("v_add_dpp row_bcast15", k_adddppb, w, 1); run("v_add_dpp row_mirror", k_adddppm, w, 1);
        run("v_cndmask_e32 vcc", k_cnde32, w, 1); run("v_add_f32 d=a+b", k_add3, w, 1);
        run("sqrt+3fma (x4)", k_mix_sqrt_3fma, w, 4);
    }
    return 0;
}
