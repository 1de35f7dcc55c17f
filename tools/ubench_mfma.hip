// tools/ubench_mfma.hip — can the idle matrix pipe take K1's dot-product accumulations? (round 2)
// K1 spends 4 of its ~14 per-format instructions per element pair on v_dot2 / v_sad accumulations.  v_mfma_i32_4x4x4_16B_i8 gives
// every lane the dot product of its own 4 bytes of A and B on the diagonal of its 4x4 block (D[vgpr lane%4]), i.e. a dot4 on
// the matrix pipe.  This measures what such an instruction costs the VALU stream it would have to be interleaved with:
//   valu   : 8 independent chains of v_dot2_u32_u16
//   mfma   : 8 independent accumulators of v_mfma_i32_4x4x4_16B_i8
//   mixed  : one v_dot2 and one v_mfma alternating (the VALU work that remains + the accumulations moved to the matrix pipe)
// at 1, 2 and 3 waves per SIMD.  If mixed ≈ max(valu, mfma) the matrix pipe is free capacity; if mixed ≈ valu + mfma it is not.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma_bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 4096
typedef int v4i __attribute__((ext_vector_type(4)));

#define DOT2(a) asm volatile("v_dot2_u32_u16 %0, %1, %1, %0" : "+v"(a) : "v"(b));
#define MFMA(c) asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %1, %0" : "+v"(c) : "v"(b));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed * 3u + threadIdx.x;
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) { DOT2(a0) DOT2(a1) DOT2(a2) DOT2(a3) DOT2(a4) DOT2(a5) DOT2(a6) DOT2(a7) }
        if (MODE == 1) { MFMA(c0) MFMA(c1) MFMA(c2) MFMA(c3) MFMA(c4) MFMA(c5) MFMA(c6) MFMA(c7) }
        if (MODE == 2) { DOT2(a0) MFMA(c0) DOT2(a1) MFMA(c1) DOT2(a2) MFMA(c2) DOT2(a3) MFMA(c3) DOT2(a4) MFMA(c4) DOT2(a5) MFMA(c5) DOT2(a6) MFMA(c6) DOT2(a7) MFMA(c7) }
        if (MODE == 3) { DOT2(a0) DOT2(a1) DOT2(a2) MFMA(c0) DOT2(a3) DOT2(a4) DOT2(a5) MFMA(c1) DOT2(a6) DOT2(a7) DOT2(a0) MFMA(c2) DOT2(a1) DOT2(a2) DOT2(a3) MFMA(c3) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0.x + c1.y + c2.z + c3.w + c4.x + c5.y + c6.z + c7.w;
}

template <int MODE>
static float run(int blocks_per_cu, uint32_t *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 7u + r);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    return best;
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const double clk = 2.4e9;
    printf("cycles per wave-instruction per SIMD at 2.4 GHz (8 instructions per loop body for valu / mfma, 16 for mixed 1:1, 12+4 for mixed 3:1)\n");
    for (int w = 1; w <= 3; ++w) {   // blocks of 4 waves per CU = waves per SIMD
        const float v = run<0>(w, out), m = run<1>(w, out), x = run<2>(w, out), y = run<3>(w, out);
        const double per = clk * 1e-3 / ((double)ITER * w);   // ms → cycles per loop iteration per wave-slot
        printf("%d wave(s)/SIMD: valu %.2f cyc/instr | mfma_i32_4x4x4_16B_i8 %.2f cyc/instr | mixed 1:1 %.2f cyc per (dot2+mfma) pair [sum of parts %.2f] | mixed 3:1 %.2f cyc per (3 dot2 + 1 mfma) [sum %.2f]\n", w,
               v * per / 8, m * per / 8, x * per / 8, (v + m) * per / 8, y * per / 4, (3 * v / 8 * 4 + m / 8 * 4) * per / 4);
    }
    return 0;
}
