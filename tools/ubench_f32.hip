// tools/ubench_f32.hip — issue cost of the float-domain forms the round-4 K1 (csrc/mtq_f32dom.hip) is built from (not part of the product).
// Same harness shape as tools/ubench.hip: ITER x 8 independent chains per lane, 2 and 4 waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f32.hip -o build/ubench_f32 && build/ubench_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 8192

#define KERNEL(name, T, INIT, BODY)                                                     \
    __global__ __launch_bounds__(256) void k_##name(T *out, T seed, unsigned long long *cyc) \
    {                                                                                   \
        T a0 = seed + (T)threadIdx.x, a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3;   \
        T a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7;              \
        T b = seed + (T)3; INIT;                                                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < ITER; ++i) {                                                \
            BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7)     \
        }                                                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        if ((threadIdx.x & 63) == 0) { atomicAdd(cyc, t1 - t0); atomicAdd(cyc + 1, r1 - r0); } \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;    \
    }

#define B_ADD(a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_SUB(a) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_ADDABS(a) asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a) : "v"(b));
#define B_ADDNEG(a) asm volatile("v_add_f32 %0, -%0, %1" : "+v"(a) : "v"(b));
#define B_FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define B_FMANEG(a) asm volatile("v_fma_f32 %0, -%0, %1, |%1|" : "+v"(a) : "v"(b));
#define B_FMAC(a) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a) : "v"(b));
#define B_MUL(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MAX(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MIN(a) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MAXABS(a) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(a) : "v"(b));
#define B_MAX3(a) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_MAX3ABS(a) asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|" : "+v"(a) : "v"(b), "v"(c));
#define B_MED3(a) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_MAXIMUM3(a) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_CMPVCC(a) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define B_CMPABSS(a) asm volatile("v_cmp_lt_f32 s[20:21], |%0|, %1\n\ts_or_b64 s[22:23], s[22:23], s[20:21]" : : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "scc");
#define B_CMPCND(a) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
#define B_CND(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
#define B_ANDLIT(a) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a));
#define B_LSHL16(a) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a) : "v"(b));
#define B_LSHLDEP(a) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
#define B_LSHRDEP(a) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a));
#define B_FMA3(a) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define B_FMAC2(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_PKFMA3(a) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define B_MOVSDWA(a) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(a) : "v"(b));
#define B_MIN3ABS(a) asm volatile("v_min3_f32 %0, |%0|, |%1|, |%2|" : "+v"(a) : "v"(b), "v"(c));
#define B_XOR(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_BFI(a) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_PKADD(a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKADDNEG(a) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(b));
#define B_PKMUL(a) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKFMA(a) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define B_CVTF64(a) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(fb));
#define B_FMAF64(a) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a) : "v"(b));
#define B_ADDF64(a) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_ADDF64ABS(a) asm volatile("v_add_f64 %0, %0, |%1|" : "+v"(a) : "v"(b));
#define B_DOT2BF(a) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a) : "v"(ub), "v"(uc));
#define B_DOT2CBF(a) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a) : "v"(ub), "v"(uc));
#define B_DOT2F16(a) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a) : "v"(ub), "v"(uc));
#define B_PKADDF16(a) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKFMAF16(a) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define B_PKMAXF16(a) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_FMAMIX(a) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a) : "v"(ub), "v"(uc));
#define B_CVTI32(a) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a));
#define B_RNDNE(a) asm volatile("v_rndne_f32 %0, %0" : "+v"(a));
#define B_LDEXP(a) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a) : "v"(ub));
#define B_FREXPE(a) asm volatile("v_frexp_exp_i32_f32 %0, %0" : "+v"(a));
// mixed streams: do a full-rate and a slow-class instruction overlap, or do their costs add?
#define B_MIX_FMA_DOT(a) asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_dot2_u32_u16 %1, %3, %3, %1" : "+v"(a), "+v"(ua) : "v"(b), "v"(ub));
#define B_MIX_FMA_F64(a) asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_add_f64 %1, %1, %3" : "+v"(a), "+v"(da) : "v"(b), "v"(db));
#define B_MIX_FMA_MED3(a) asm volatile("v_fma_f32 %0, %0, %2, %2\n\tv_med3_f32 %1, %1, %2, %3" : "+v"(a), "+v"(fa) : "v"(b), "v"(c));
#define B_MIX_ADD_AND(a) asm volatile("v_add_f32 %0, %0, %2\n\tv_and_b32 %1, %1, %3" : "+v"(a), "+v"(ua) : "v"(b), "v"(ub));

KERNEL(add_f32, float, , B_ADD) KERNEL(sub_f32, float, , B_SUB) KERNEL(add_f32_abs, float, , B_ADDABS) KERNEL(add_f32_neg, float, , B_ADDNEG)
KERNEL(fma_f32, float, , B_FMA) KERNEL(fma_f32_negabs, float, , B_FMANEG) KERNEL(fmac_f32, float, , B_FMAC) KERNEL(mul_f32, float, , B_MUL)
KERNEL(max_f32, float, , B_MAX) KERNEL(min_f32, float, , B_MIN) KERNEL(max_f32_abs, float, , B_MAXABS)
KERNEL(max3_f32, float, float c = b * 7.f, B_MAX3) KERNEL(max3_f32_abs, float, float c = b * 7.f, B_MAX3ABS) KERNEL(med3_f32, float, float c = b * 7.f, B_MED3)
KERNEL(maximum3_f32, float, float c = b * 7.f, B_MAXIMUM3)
KERNEL(cmp_lt_f32_vcc, float, , B_CMPVCC) KERNEL(cmp_abs_sgpr_s_or, float, , B_CMPABSS) KERNEL(cmp_cndmask, float, , B_CMPCND) KERNEL(cndmask, float, , B_CND)
KERNEL(and_literal, uint32_t, , B_ANDLIT) KERNEL(lshlrev16, uint32_t, , B_LSHL16) KERNEL(xor_b32, uint32_t, , B_XOR) KERNEL(bfi_b32, uint32_t, uint32_t c = b * 7u, B_BFI)
KERNEL(lshl_dep, uint32_t, , B_LSHLDEP) KERNEL(lshr_dep, uint32_t, , B_LSHRDEP) KERNEL(fma_f32_acc, float, float c = b * 7.f, B_FMA3) KERNEL(fmac_f32_acc, float, float c = b * 7.f, B_FMAC2)
KERNEL(pk_fma_f32_acc, double, double c = b * 7., B_PKFMA3) KERNEL(mov_sdwa_w1, uint32_t, , B_MOVSDWA) KERNEL(min3_f32_abs, float, float c = b * 7.f, B_MIN3ABS)
KERNEL(pk_add_f32, double, , B_PKADD) KERNEL(pk_add_f32_neg, double, , B_PKADDNEG) KERNEL(pk_mul_f32, double, , B_PKMUL) KERNEL(pk_fma_f32, double, , B_PKFMA)
KERNEL(cvt_f64_f32, double, float fb = (float)threadIdx.x, B_CVTF64) KERNEL(fma_f64, double, , B_FMAF64) KERNEL(add_f64, double, , B_ADDF64) KERNEL(add_f64_abs, double, , B_ADDF64ABS)
KERNEL(dot2_f32_bf16, float, uint32_t ub = threadIdx.x * 0x10001u; uint32_t uc = 0x3f803f80u, B_DOT2BF)
KERNEL(dot2c_f32_bf16, float, uint32_t ub = threadIdx.x * 0x10001u; uint32_t uc = 0x3f803f80u, B_DOT2CBF)
KERNEL(dot2_f32_f16, float, uint32_t ub = threadIdx.x * 0x10001u; uint32_t uc = 0x3c003c00u, B_DOT2F16)
KERNEL(pk_add_f16, uint32_t, , B_PKADDF16) KERNEL(pk_fma_f16, uint32_t, , B_PKFMAF16) KERNEL(pk_max_f16, uint32_t, , B_PKMAXF16)
KERNEL(fma_mix_f32, float, uint32_t ub = threadIdx.x * 0x10001u; uint32_t uc = 0x3c003c00u, B_FMAMIX)
KERNEL(cvt_i32_f32, float, , B_CVTI32) KERNEL(rndne_f32, float, , B_RNDNE) KERNEL(ldexp_f32, float, uint32_t ub = threadIdx.x & 3, B_LDEXP) KERNEL(frexp_exp, float, , B_FREXPE)
KERNEL(mix_fma_dot2, float, uint32_t ua = threadIdx.x; uint32_t ub = 77u, B_MIX_FMA_DOT)
KERNEL(mix_fma_addf64, float, double da = threadIdx.x; double db = 3.0, B_MIX_FMA_F64)
KERNEL(mix_fma_med3, float, float fa = threadIdx.x; float c = b * 7.f, B_MIX_FMA_MED3)
KERNEL(mix_add_and, float, uint32_t ua = threadIdx.x; uint32_t ub = 77u, B_MIX_ADD_AND)

template <typename T, typename K>
static void run(const char *name, K kern, int waves_per_simd, int insts_per_body)
{
    int blocks = 256 * waves_per_simd;
    T *out;
    unsigned long long *cyc, hc[2];
    hipMalloc(&out, sizeof(T) * 256 * blocks);
    hipMalloc(&cyc, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, (T)1, cyc);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipMemset(cyc, 0, 16);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, (T)1, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    double bodies = (double)blocks * 4 * ITER * 8;
    double per_cu_per_us = bodies / 256.0 / (best * 1e3);
    // in-kernel: shader cycles a wave spent per body, divided by the waves sharing its SIMD = issue cycles per body per SIMD at the
    // clock the chip actually held; that clock = shader ticks / (100 MHz ticks) * 100 MHz
    const double waves = (double)blocks * 4, cyc_body = (double)hc[0] / waves / (ITER * 8.0) / waves_per_simd;
    const double ghz = hc[1] ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0;
    printf("%-22s w/simd=%d  %8.3f ms  => %5.2f cyc/body/SIMD @2.4GHz nominal | in-kernel %5.2f cyc/body/SIMD at %4.2f GHz (%d VALU inst per body)\n", name, waves_per_simd, best,
           4 * 2400.0 / per_cu_per_us, cyc_body, ghz, insts_per_body);
    hipFree(cyc);
    hipFree(out);
}

#define RUN(name, T) for (int w : {2, 4}) run<T>(#name, k_##name, w, 1);
#define RUNM(name, T) for (int w : {2, 4}) run<T>(#name, k_##name, w, 2);

int main()
{
    RUN(add_f32, float) RUN(sub_f32, float) RUN(add_f32_abs, float) RUN(add_f32_neg, float) RUN(fma_f32, float) RUN(fma_f32_negabs, float)
    RUN(fmac_f32, float) RUN(mul_f32, float) RUN(max_f32, float) RUN(min_f32, float) RUN(max_f32_abs, float) RUN(max3_f32, float)
    RUN(max3_f32_abs, float) RUN(med3_f32, float) RUN(maximum3_f32, float) RUN(cmp_lt_f32_vcc, float) RUN(cmp_abs_sgpr_s_or, float)
    RUNM(cmp_cndmask, float) RUN(cndmask, float) RUN(and_literal, uint32_t) RUN(lshlrev16, uint32_t) RUN(xor_b32, uint32_t) RUN(bfi_b32, uint32_t)
    RUN(lshl_dep, uint32_t) RUN(lshr_dep, uint32_t) RUN(fma_f32_acc, float) RUN(fmac_f32_acc, float) RUN(pk_fma_f32_acc, double) RUN(mov_sdwa_w1, uint32_t) RUN(min3_f32_abs, float)
    RUN(pk_add_f32, double) RUN(pk_add_f32_neg, double) RUN(pk_mul_f32, double) RUN(pk_fma_f32, double)
    RUN(cvt_f64_f32, double) RUN(fma_f64, double) RUN(add_f64, double) RUN(add_f64_abs, double)
    RUN(dot2_f32_bf16, float) RUN(dot2c_f32_bf16, float) RUN(dot2_f32_f16, float) RUN(pk_add_f16, uint32_t) RUN(pk_fma_f16, uint32_t)
    RUN(pk_max_f16, uint32_t) RUN(fma_mix_f32, float) RUN(cvt_i32_f32, float) RUN(rndne_f32, float) RUN(ldexp_f32, float) RUN(frexp_exp, float)
    RUNM(mix_fma_dot2, float) RUNM(mix_fma_addf64, float) RUNM(mix_fma_med3, float) RUNM(mix_add_and, float)
    return 0;
}
