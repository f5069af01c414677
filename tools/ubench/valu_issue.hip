// VALU issue-rate microbenchmark for gfx950, second edition (round 2): settles "2 clk or 4.3 clk per wave64 v_fma_f32".
// Differences from valu_rates.hip: (1) cycles are SHADER cycles read with s_memtime inside every wave (DVFS-proof; the first
// edition multiplied wall time by the nominal 2.4 GHz), wall time is reported beside them, which also gives the effective
// clock under this load; (2) 16 independent destination registers per wave, so a dependent-issue latency of up to 16 issue
// slots is covered by one wave alone; (3) swept over 1, 2, 4 and 8 resident waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITER 1000
#define REP 4          // 16-instruction groups per loop iteration
#define K(name, body)                                                                                     \
    __global__ void __launch_bounds__(256) name(float *out, unsigned long long *cyc, float seed)          \
    {                                                                                                     \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float a8 = a0 + 8, a9 = a0 + 9, aa = a0 + 10, ab = a0 + 11, ac = a0 + 12, ad = a0 + 13, ae = a0 + 14, af = a0 + 15;     \
        float b0 = 1.0001f, b1 = 0.9999f;                                                                 \
        typedef float f2 __attribute__((ext_vector_type(2)));                                             \
        typedef float f4 __attribute__((ext_vector_type(4))); f4 m0 = {a0, a1, a2, a3}, m1 = m0, m2 = m0; f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a8, a9}, p5 = {aa, ab}, p6 = {ac, ad}, p7 = {ae, af}, q = {b0, b1}; \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < ITER; ++i) {                                                                  \
            _Pragma("unroll") for (int r = 0; r < REP; ++r) { body }                                      \
        }                                                                                                 \
        asm volatile("s_nop 0" ::: "memory");                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + aa + ab + ac + ad + ae + af +   \
            m0.x + m1.y + m2.z + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                 \
    }
#define A16(op) op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7) op(a8) op(a9) op(aa) op(ab) op(ac) op(ad) op(ae) op(af)
#define P16(op) op(p0) op(p1) op(p2) op(p3) op(p4) op(p5) op(p6) op(p7) op(p0) op(p1) op(p2) op(p3) op(p4) op(p5) op(p6) op(p7)
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b0), "v"(b1));
#define FMAC(x) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(b0), "v"(b1));
#define FMAS(x) asm volatile("v_fma_f32 %0, s4, %0, |%1|" : "+v"(x) : "v"(b1) : "s4");
#define ADD(x) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define MULCLAMP(x) asm volatile("v_mul_f32_e64 %0, |%0|, %1 clamp" : "+v"(x) : "v"(b1));
#define SQRT(x) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(x));
#define BFI(x) asm volatile("v_bfi_b32 %0, s4, %0, %1" : "+v"(x) : "v"(b1) : "s4");
#define BFIV(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b0), "v"(b1));
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(q));
#define PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(q));
#define PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(q));
#define ADDDPP(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
#define MOVB(x) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x) : "v"(b1));
// the element transform + moment updates of k_cdc_partial_grouped for TWO elements, in the instruction mix and order the
// compiler emits (coldeltacor ISA, profiles/r02_cdc_grouped_isa.txt): pk sub, 2 x (mul|.|clamp, fma, sqrt, bfi), pk_add,
// 2 x pk_fma = 12 instructions per 2 elements; x, y carry the two dependent 4-instruction chains, pa..pd the packed ops.
#define MIX(pa, pb, pc, pd, x, y) PKADD(pa) MULCLAMP(x) MULCLAMP(y) FMAS(x) FMAS(y) SQRT(x) SQRT(y) BFI(x) BFI(y) PKADD(pb) PKFMA(pc) PKFMA(pd)
#define E8 MIX(p0, p1, p2, p3, a0, a1) MIX(p4, p5, p6, p7, a2, a3) MIX(p0, p1, p2, p3, a4, a5) MIX(p4, p5, p6, p7, a6, a7) \
           MIX(p0, p1, p2, p3, a8, a9) MIX(p4, p5, p6, p7, aa, ab) MIX(p0, p1, p2, p3, ac, ad) MIX(p4, p5, p6, p7, ae, af)
#define FMA_ABS(x) asm volatile("v_fma_f32 %0, %1, %0, |%2|" : "+v"(x) : "v"(b0), "v"(b1));
#define FMA_S(x) asm volatile("v_fma_f32 %0, s4, %0, %1" : "+v"(x) : "v"(b1) : "s4");

#define MED3N(x) asm volatile("v_med3_f32 %0, %0, -%0, %1" : "+v"(x) : "v"(b1));
#define ANDOR(x) asm volatile("v_and_or_b32 %0, %1, s4, %0" : "+v"(x) : "v"(b1) : "s4");
#define ANDORV(x) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(x) : "v"(b1), "v"(b0));
#define XORB(x) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define ADDABS(x) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(x) : "v"(b1));
#define ADDABS_S(x) asm volatile("v_add_f32_e64 %0, |%0|, s4" : "+v"(x) : : "s4");
#define MULLIT(x) asm volatile("v_mul_f32_e32 %0, 0x71800000, %0" : "+v"(x));
#define CMPS(x) asm volatile("v_cmp_lt_f32_e64 s[6:7], %0, %1" : : "v"(x), "v"(b1) : "s6", "s7");
#define CNDS(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[6:7]" : "+v"(x) : "v"(b1) : "s6", "s7");
#define RDLANE(x) asm volatile("v_readlane_b32 s6, %0, 63" : : "v"(x) : "s6");
#define MFMA1(t) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(t) : "v"(b0), "v"(b1));
#define ADD16M(t) A16(ADD) MFMA1(t)
#define ADD16M3 A16(ADD) MFMA1(m0) MFMA1(m1) MFMA1(m2)
// candidate element mix: pk sub, 2 x (add|t|+psc, sqrt, mul 2^100, med3) , pk_add, 2 pk_fma
#define MIXB(pa, pb, pc, pd, x, y) PKADD(pa) ADDABS(x) ADDABS(y) SQRT(x) SQRT(y) MULLIT(x) MULLIT(y) MED3N(x) MED3N(y) PKADD(pb) PKFMA(pc) PKFMA(pd)
#define E8B MIXB(p0, p1, p2, p3, a0, a1) MIXB(p4, p5, p6, p7, a2, a3) MIXB(p0, p1, p2, p3, a4, a5) MIXB(p4, p5, p6, p7, a6, a7) \
            MIXB(p0, p1, p2, p3, a8, a9) MIXB(p4, p5, p6, p7, aa, ab) MIXB(p0, p1, p2, p3, ac, ad) MIXB(p4, p5, p6, p7, ae, af)
// current mix with the fma's psc in a VGPR instead of an SGPR
#define FMAV(x) asm volatile("v_fma_f32 %0, %1, %0, |%2|" : "+v"(x) : "v"(b0), "v"(b1));
#define MIXC(pa, pb, pc, pd, x, y) PKADD(pa) MULCLAMP(x) MULCLAMP(y) FMAV(x) FMAV(y) SQRT(x) SQRT(y) BFIV(x) BFIV(y) PKADD(pb) PKFMA(pc) PKFMA(pd)
#define E8C MIXC(p0, p1, p2, p3, a0, a1) MIXC(p4, p5, p6, p7, a2, a3) MIXC(p0, p1, p2, p3, a4, a5) MIXC(p4, p5, p6, p7, a6, a7) \
            MIXC(p0, p1, p2, p3, a8, a9) MIXC(p4, p5, p6, p7, aa, ab) MIXC(p0, p1, p2, p3, ac, ad) MIXC(p4, p5, p6, p7, ae, af)
#define PSWAP32(x) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(x), "+v"(b0));
#define PSWAP16(x) asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(x), "+v"(b0));
#define CNDVCC(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(b1));
#define DPPSHR(x) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));
// round 2, second edition: the no-pseudocount element (csrc/coldeltacor.hip, VCY_RULES_PARTIAL_NOPSC): sub, rsq|t|, legacy mul, add, 2 fmac
#define RSQA(x) asm volatile("v_rsq_f32_e64 %0, |%0|" : "+v"(x));
#define MULLEG(x) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define SUBV(x) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x) : "v"(b1));
#define MIXD(x1, y1, x2, y2) SUBV(x1) SUBV(x2) asm volatile("v_rsq_f32_e64 %0, |%1|" : "=v"(y1) : "v"(x1)); asm volatile("v_rsq_f32_e64 %0, |%1|" : "=v"(y2) : "v"(x2)); \
    asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(y1) : "v"(x1)); asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(y2) : "v"(x2)); \
    asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(ac) : "v"(y1)); asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(ad) : "v"(y2)); \
    asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(ae) : "v"(y1)); asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(af) : "v"(y2)); \
    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(aa) : "v"(y1), "v"(b0)); asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(ab) : "v"(y2), "v"(b0));
#define E8D MIXD(a0, a4, a1, a5) MIXD(a2, a6, a3, a7) MIXD(a0, a8, a1, a9) MIXD(a2, a4, a3, a5)
// the same element, four interleaved (sub x4, rsq x4, mul x4, add x4, fmac x8) and eight interleaved
#define RSQ2(y, x) asm volatile("v_rsq_f32_e64 %0, |%1|" : "=v"(y) : "v"(x));
#define MULL2(y, x) asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(y) : "v"(x));
#define ADD2(s, y) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(s) : "v"(y));
#define FMAC2(s, y) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(s) : "v"(y));
#define FMAC3(s, y) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s) : "v"(y), "v"(b0));
#define MIXD4 SUBV(a0) SUBV(a1) SUBV(a2) SUBV(a3) RSQ2(a4, a0) RSQ2(a5, a1) RSQ2(a6, a2) RSQ2(a7, a3) MULL2(a4, a0) MULL2(a5, a1) MULL2(a6, a2) MULL2(a7, a3) \
    ADD2(ac, a4) ADD2(ad, a5) ADD2(ac, a6) ADD2(ad, a7) FMAC2(ae, a4) FMAC2(af, a5) FMAC2(ae, a6) FMAC2(af, a7) FMAC3(aa, a4) FMAC3(ab, a5) FMAC3(aa, a6) FMAC3(ab, a7)
#define MIXD8 SUBV(a0) SUBV(a1) SUBV(a2) SUBV(a3) SUBV(p0.x) SUBV(p0.y) SUBV(p1.x) SUBV(p1.y) \
    RSQ2(a4, a0) RSQ2(a5, a1) RSQ2(a6, a2) RSQ2(a7, a3) RSQ2(p2.x, p0.x) RSQ2(p2.y, p0.y) RSQ2(p3.x, p1.x) RSQ2(p3.y, p1.y) \
    MULL2(a4, a0) MULL2(a5, a1) MULL2(a6, a2) MULL2(a7, a3) MULL2(p2.x, p0.x) MULL2(p2.y, p0.y) MULL2(p3.x, p1.x) MULL2(p3.y, p1.y) \
    ADD2(ac, a4) ADD2(ad, a5) ADD2(ac, a6) ADD2(ad, a7) ADD2(ac, p2.x) ADD2(ad, p2.y) ADD2(ac, p3.x) ADD2(ad, p3.y) \
    FMAC2(ae, a4) FMAC2(af, a5) FMAC2(ae, a6) FMAC2(af, a7) FMAC2(ae, p2.x) FMAC2(af, p2.y) FMAC2(ae, p3.x) FMAC2(af, p3.y) \
    FMAC3(aa, a4) FMAC3(ab, a5) FMAC3(aa, a6) FMAC3(ab, a7) FMAC3(aa, p2.x) FMAC3(ab, p2.y) FMAC3(aa, p3.x) FMAC3(ab, p3.y)
K(k_elem_d4, MIXD4 MIXD4)
K(k_elem_d8, MIXD8)
K(k_rsqa, A16(RSQA))
K(k_mulleg, A16(MULLEG))
K(k_subv, A16(SUBV))
K(k_elem_d, E8D)
K(k_pswap32, A16(PSWAP32))
K(k_pswap16, A16(PSWAP16))
K(k_cndvcc, A16(CNDVCC))
K(k_dppshr, A16(DPPSHR))
K(k_fma, A16(FMA))
K(k_fmac, A16(FMAC))
K(k_fmas, A16(FMAS))
K(k_add, A16(ADD))
K(k_mulclamp, A16(MULCLAMP))
K(k_sqrt, A16(SQRT))
K(k_bfi, A16(BFI))
K(k_bfiv, A16(BFIV))
K(k_pkfma, P16(PKFMA))
K(k_pkadd, P16(PKADD))
K(k_pkmul, P16(PKMUL))
K(k_adddpp, A16(ADDDPP))
K(k_movb, A16(MOVB))
K(k_elem2, E8)
K(k_fma_abs, A16(FMA_ABS))
K(k_fma_s, A16(FMA_S))
K(k_med3n, A16(MED3N))
K(k_andor, A16(ANDOR))
K(k_andorv, A16(ANDORV))
K(k_xorb, A16(XORB))
K(k_addabs, A16(ADDABS))
K(k_addabs_s, A16(ADDABS_S))
K(k_mullit, A16(MULLIT))
K(k_cmps, A16(CMPS))
K(k_cnds, A16(CNDS))
K(k_rdlane, A16(RDLANE))
K(k_add16m, ADD16M(m0))
K(k_add16m3, ADD16M3)
K(k_elem2b, E8B)
K(k_elem2c, E8C)

template <typename F> static void run(const char *name, F k, int wps, double instr_per_group)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * wps;      // 256 threads = 4 waves = 1 per SIMD
    float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned long long *cyc; hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, cyc, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, 256>>>(out, cyc, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], mx = (double)h.back();
    const double instr_wave = (double)ITER * REP * instr_per_group;          // wave-instructions issued by ONE wave
    const double per_simd = instr_wave * wps;                                // ... by all waves of one SIMD
    // s_memtime ticks: the guide says shader cycles; the ratio to wall time (effective MHz) is printed so this can be checked
    printf("%-22s w/SIMD=%d  wall %7.3f ms  wave cycles med %9.0f max %9.0f  -> %5.2f clk per wave-instr per SIMD (med), %5.2f (wall @ tick rate %.0f MHz)\n",
           name, wps, ms, med, mx, med / per_simd, mx / per_simd, mx / (ms * 1e3));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : {4}) {
        run("v_fma_f32 (3 vgpr)", k_fma, w, 16); run("v_fmac_f32_e32", k_fmac, w, 16); run("v_fma s,v,|v|", k_fmas, w, 16);
        run("v_add_f32", k_add, w, 16); run("v_mul |x| clamp", k_mulclamp, w, 16); run("v_sqrt_f32", k_sqrt, w, 16);
        run("v_bfi_b32 s,v,v", k_bfi, w, 16); run("v_bfi_b32 v,v,v", k_bfiv, w, 16);
        run("v_pk_fma_f32", k_pkfma, w, 16); run("v_pk_add_f32", k_pkadd, w, 16); run("v_pk_mul_f32", k_pkmul, w, 16);
        run("v_add_f32_dpp", k_adddpp, w, 16); run("v_mov_b32", k_movb, w, 16);
        run("cdc element pair x8", k_elem2, w, 8 * 12);
        run("v_med3 v,-v,v", k_med3n, w, 16); run("v_and_or v,s,v", k_andor, w, 16); run("v_and_or v,v,v", k_andorv, w, 16);
        run("v_xor_b32", k_xorb, w, 16); run("v_add |v|,v", k_addabs, w, 16); run("v_add |v|,s", k_addabs_s, w, 16);
        run("v_mul lit,v", k_mullit, w, 16); run("v_cmp_e64 sgpr", k_cmps, w, 16); run("v_cndmask_e64 sgpr", k_cnds, w, 16);
        run("v_readlane", k_rdlane, w, 16);
        run("v_permlane32_swap", k_pswap32, w, 16); run("v_permlane16_swap", k_pswap16, w, 16); run("v_cndmask_e32 vcc", k_cndvcc, w, 16);
        run("v_add_dpp row_shr:4", k_dppshr, w, 16);
        run("16 v_add + 1 mfma16x16x4f32", k_add16m, w, 16); run("16 v_add + 3 mfma (3 acc)", k_add16m3, w, 16);
        run("cdc mix B (add,sqrt,mul,med3)", k_elem2b, w, 8 * 12); run("cdc mix C (vgpr psc fma)", k_elem2c, w, 8 * 12);
        run("v_rsq_f32 |v|", k_rsqa, w, 16); run("v_mul_legacy_f32", k_mulleg, w, 16); run("v_sub_f32", k_subv, w, 16);
        run("cdc no-psc element x8 (6 instr each)", k_elem_d, w, 8 * 6);
        run("same, 4 elements interleaved", k_elem_d4, w, 8 * 6); run("same, 8 elements interleaved", k_elem_d8, w, 8 * 6);
        printf("\n");
    }
    return 0;
}
