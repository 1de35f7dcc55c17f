// tools/ubench.hip — VALU instruction-throughput calibration for the K1 design (not part of the product).
// Each kernel issues ITER x 8 independent instances of one instruction per lane; 8 waves per SIMD worth of blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define ITER 8192
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

#define KERNEL(name, T, INIT, BODY)                                                     \
    __global__ __launch_bounds__(256) void k_##name(T *out, T seed)                     \
    {                                                                                   \
        T a0 = seed + (T)threadIdx.x, a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3;   \
        T a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7;              \
        T b = seed + (T)3; INIT;                                                        \
        for (int i = 0; i < ITER; ++i) {                                                \
            BODY(a0) BODY(a1) BODY(a2) BODY(a3) BODY(a4) BODY(a5) BODY(a6) BODY(a7)     \
        }                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;    \
    }

#define B_ADDU32(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKADD(a) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKSHR(a) asm volatile("v_pk_lshrrev_b16 %0, %1, %0" : "+v"(a) : "v"(b));
#define B_PKMUL(a) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKMAX(a) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKMAD(a) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define B_DOT2U(a) asm volatile("v_dot2_u32_u16 %0, %1, %1, %0" : "+v"(a) : "v"(b));
#define B_DOT2I(a) asm volatile("v_dot2_i32_i16 %0, %1, %1, %0" : "+v"(a) : "v"(b));
#define B_DOT4U(a) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(a) : "v"(b));
#define B_SAD16(a) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define B_SAD8(a) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
#define B_ANDOR(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_ADD3(a) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_PERM(a) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_BFE(a) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a));
#define B_MULLO(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MAD24(a) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(a) : "v"(b));
#define B_DPPMOV(a) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
#define B_DPPADD(a) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
#define B_ADDF32(a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_FMAF32(a) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define B_PKFMAF32(a) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define B_ADDF64(a) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_FMAF64(a) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define B_MULF64(a) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_LDEXPF64(a) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(ib));
#define B_CVTF64I32(a) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a) : "v"(ib));
#define B_CVTF64U32(a) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a) : "v"(ib));
#define B_CVTF64F32(a) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(ib));
#define B_MADU64(a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a) : "v"(ib) : "vcc");
#define B_LSHL64(a) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a) : "v"(ib));
#define B_ADDCO(a) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
#define B_CVTF32U32(a) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a));
#define B_CMPSEL(a) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");


#define B_AND(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_OR(a) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_XOR(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_LSHL(a) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));
#define B_LSHRV(a) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "v"(b));
#define B_ASHR(a) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a));
#define B_SUB(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MAXU(a) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MINI(a) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_CNDMASK(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : );
#define B_MULF32(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_FMACF32(a) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a) : "v"(b));
#define B_MAXF32(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MULU24(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_DOT2C(a) asm volatile("v_dot2c_i32_i16 %0, %1, %1" : "+v"(a) : "v"(b));
#define B_DOT4C(a) asm volatile("v_dot4c_i32_i8 %0, %1, %1" : "+v"(a) : "v"(b));
#define B_PKFMAC16(a) asm volatile("v_pk_fmac_f16 %0, %1, %1" : "+v"(a) : "v"(b));
#define B_ADDSDWA(a) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a) : "v"(b));
#define B_ADDU16(a) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_MAX3(a) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_LSHLOR(a) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
#define B_LSHLADD(a) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b));
#define B_BFI(a) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c));
#define B_ALIGNBIT(a) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a) : "v"(b));
#define B_FFBH(a) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));
#define B_MOV(a) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
#define B_CVTF32BF(a) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_ADDDPPSHR(a) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
#define B_PKADDF32(a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKMULF32(a) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKSUBI16(a) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_PKASHR(a) asm volatile("v_pk_ashrrev_i16 %0, 3, %0" : "+v"(a));
#define B_PKMIN(a) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define B_DOT2F32BF16(a) asm volatile("v_dot2_f32_bf16 %0, %1, %1, %0" : "+v"(a) : "v"(b));
#define B_SWAP32(a) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
KERNEL(add_u32, uint32_t, , B_ADDU32)
KERNEL(pk_add_u16, uint32_t, , B_PKADD)
KERNEL(pk_lshrrev_b16, uint32_t, , B_PKSHR)
KERNEL(pk_mul_lo_u16, uint32_t, , B_PKMUL)
KERNEL(pk_max_u16, uint32_t, , B_PKMAX)
KERNEL(pk_mad_u16, uint32_t, , B_PKMAD)
KERNEL(dot2_u32_u16, uint32_t, , B_DOT2U)
KERNEL(dot2_i32_i16, uint32_t, , B_DOT2I)
KERNEL(dot4_u32_u8, uint32_t, , B_DOT4U)
KERNEL(sad_u16, uint32_t, uint32_t c = b * 7u, B_SAD16)
KERNEL(sad_u8, uint32_t, uint32_t c = b * 7u, B_SAD8)
KERNEL(and_or_b32, uint32_t, uint32_t c = b * 7u, B_ANDOR)
KERNEL(add3_u32, uint32_t, uint32_t c = b * 7u, B_ADD3)
KERNEL(perm_b32, uint32_t, uint32_t c = 0x07060100u, B_PERM)
KERNEL(bfe_u32, uint32_t, , B_BFE)
KERNEL(mul_lo_u32, uint32_t, , B_MULLO)
KERNEL(mad_u32_u24, uint32_t, , B_MAD24)
KERNEL(mov_dpp, uint32_t, , B_DPPMOV)
KERNEL(add_u32_dpp, uint32_t, , B_DPPADD)
KERNEL(cvt_f32_u32, uint32_t, , B_CVTF32U32)
KERNEL(cmp_cndmask, uint32_t, , B_CMPSEL)
KERNEL(add_co_addc_pair, uint32_t, , B_ADDCO)
KERNEL(add_f32, float, , B_ADDF32)
KERNEL(fma_f32, float, , B_FMAF32)
KERNEL(pk_fma_f32, double, , B_PKFMAF32)
KERNEL(add_f64, double, , B_ADDF64)
KERNEL(fma_f64, double, , B_FMAF64)
KERNEL(mul_f64, double, , B_MULF64)
KERNEL(ldexp_f64, double, int ib = (int)threadIdx.x & 3, B_LDEXPF64)
KERNEL(cvt_f64_i32, double, int ib = (int)threadIdx.x, B_CVTF64I32)
KERNEL(cvt_f64_u32, double, int ib = (int)threadIdx.x, B_CVTF64U32)
KERNEL(cvt_f64_f32, double, float ib = (float)threadIdx.x, B_CVTF64F32)
KERNEL(mad_u64_u32, uint64_t, uint32_t ib = threadIdx.x, B_MADU64)
KERNEL(lshlrev_b64, uint64_t, uint32_t ib = threadIdx.x & 1, B_LSHL64)


KERNEL(and_b32, uint32_t, , B_AND) KERNEL(or_b32, uint32_t, , B_OR) KERNEL(xor_b32, uint32_t, , B_XOR)
KERNEL(lshlrev_b32_imm, uint32_t, , B_LSHL) KERNEL(lshrrev_b32_v, uint32_t, , B_LSHRV) KERNEL(ashrrev_i32, uint32_t, , B_ASHR)
KERNEL(sub_u32, uint32_t, , B_SUB) KERNEL(max_u32, uint32_t, , B_MAXU) KERNEL(min_i32, uint32_t, , B_MINI)
KERNEL(cndmask, uint32_t, , B_CNDMASK) KERNEL(mul_f32, float, , B_MULF32) KERNEL(fmac_f32, float, , B_FMACF32)
KERNEL(max_f32, float, , B_MAXF32) KERNEL(mul_u32_u24, uint32_t, , B_MULU24) KERNEL(dot2c_i32_i16, uint32_t, , B_DOT2C)
KERNEL(dot4c_i32_i8, uint32_t, , B_DOT4C) KERNEL(pk_fmac_f16, uint32_t, , B_PKFMAC16) KERNEL(add_u32_sdwa, uint32_t, , B_ADDSDWA)
KERNEL(add_u16, uint32_t, , B_ADDU16) KERNEL(max3_u32, uint32_t, uint32_t c = b * 7u, B_MAX3) KERNEL(lshl_or_b32, uint32_t, , B_LSHLOR)
KERNEL(lshl_add_u32, uint32_t, , B_LSHLADD) KERNEL(bfi_b32, uint32_t, uint32_t c = b * 7u, B_BFI) KERNEL(alignbit_b32, uint32_t, , B_ALIGNBIT)
KERNEL(ffbh_u32, uint32_t, , B_FFBH) KERNEL(mov_b32, uint32_t, , B_MOV) KERNEL(cvt_pk_bf16_f32, float, , B_CVTF32BF)
KERNEL(add_f32_dpp, float, , B_ADDDPPSHR) KERNEL(pk_add_f32, double, , B_PKADDF32) KERNEL(pk_mul_f32, double, , B_PKMULF32)
KERNEL(pk_sub_i16, uint32_t, , B_PKSUBI16) KERNEL(pk_ashrrev_i16, uint32_t, , B_PKASHR) KERNEL(pk_min_u16, uint32_t, , B_PKMIN)
KERNEL(dot2_f32_bf16, float, , B_DOT2F32BF16) KERNEL(permlane32_swap, uint32_t, , B_SWAP32)

template <typename T, typename K>
static void run(const char *name, K kern, int waves_per_simd, int insts_per_body = 1)
{
    int blocks = 256 * waves_per_simd; // 256 CUs x (waves_per_simd*4 waves / 4 waves per block)
    T *out;
    hipMalloc(&out, sizeof(T) * 256 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, (T)1);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, (T)1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double insts = (double)blocks * 4 /*waves*/ * ITER * 8 * insts_per_body; // wave-instructions
    double per_cu_per_us = insts / 256.0 / (best * 1e3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: (4 SIMD * 2400 cycles/us) / per_cu_per_us
    printf("%-20s w/simd=%d  %8.3f ms  %7.1f wave-inst/us/CU  => %5.2f cyc/inst/SIMD @2.4GHz\n", name, waves_per_simd, best,
           per_cu_per_us, 4 * 2400.0 / per_cu_per_us);
    hipFree(out);
}

#define RUN(name, T) for (int w : {1, 2, 4, 8}) run<T>(#name, k_##name, w);
#define RUN2(name, T) for (int w : {2, 8}) run<T>(#name, k_##name, w);

int main()
{
    RUN(add_u32, uint32_t)
    RUN2(pk_add_u16, uint32_t) RUN2(pk_lshrrev_b16, uint32_t) RUN2(pk_mul_lo_u16, uint32_t) RUN2(pk_max_u16, uint32_t)
    RUN2(pk_mad_u16, uint32_t) RUN2(dot2_u32_u16, uint32_t) RUN2(dot2_i32_i16, uint32_t) RUN2(dot4_u32_u8, uint32_t)
    RUN2(sad_u16, uint32_t) RUN2(sad_u8, uint32_t) RUN2(and_or_b32, uint32_t) RUN2(add3_u32, uint32_t) RUN2(perm_b32, uint32_t)
    RUN2(bfe_u32, uint32_t) RUN2(mul_lo_u32, uint32_t) RUN2(mad_u32_u24, uint32_t) RUN2(mov_dpp, uint32_t) RUN2(add_u32_dpp, uint32_t)
    RUN2(cvt_f32_u32, uint32_t) RUN2(cmp_cndmask, uint32_t) RUN2(add_co_addc_pair, uint32_t)
    RUN2(add_f32, float) RUN2(fma_f32, float) RUN2(pk_fma_f32, double)
    RUN(add_f64, double) RUN2(fma_f64, double) RUN2(mul_f64, double) RUN2(ldexp_f64, double)
    RUN2(cvt_f64_i32, double) RUN2(cvt_f64_u32, double) RUN2(cvt_f64_f32, double) RUN2(mad_u64_u32, uint64_t) RUN2(lshlrev_b64, uint64_t)
    RUN2(and_b32, uint32_t) RUN2(or_b32, uint32_t) RUN2(xor_b32, uint32_t) RUN2(lshlrev_b32_imm, uint32_t) RUN2(lshrrev_b32_v, uint32_t)
    RUN2(ashrrev_i32, uint32_t) RUN2(sub_u32, uint32_t) RUN2(max_u32, uint32_t) RUN2(min_i32, uint32_t) RUN2(cndmask, uint32_t)
    RUN2(mul_f32, float) RUN2(fmac_f32, float) RUN2(max_f32, float) RUN2(mul_u32_u24, uint32_t) RUN2(dot2c_i32_i16, uint32_t)
    RUN2(dot4c_i32_i8, uint32_t) RUN2(pk_fmac_f16, uint32_t) RUN2(add_u32_sdwa, uint32_t) RUN2(add_u16, uint32_t) RUN2(max3_u32, uint32_t)
    RUN2(lshl_or_b32, uint32_t) RUN2(lshl_add_u32, uint32_t) RUN2(bfi_b32, uint32_t) RUN2(alignbit_b32, uint32_t) RUN2(ffbh_u32, uint32_t)
    RUN2(mov_b32, uint32_t) RUN2(cvt_pk_bf16_f32, float) RUN2(add_f32_dpp, float) RUN2(pk_add_f32, double) RUN2(pk_mul_f32, double)
    RUN2(pk_sub_i16, uint32_t) RUN2(pk_ashrrev_i16, uint32_t) RUN2(pk_min_u16, uint32_t) RUN2(dot2_f32_bf16, float) RUN2(permlane32_swap, uint32_t)
    return 0;
}
