// tools/ubench_lut.hip — feasibility of a table-driven K1 (round 2): per element ONE index computation and two 16-byte LDS
// look-ups whose packed integer fields are summed with plain 32-bit adds, instead of ~37 VALU lane-ops per element of
// format arithmetic.  The table contents are arbitrary here (throughput only): the op mix per element is what a real
// implementation would issue — index (shared exponent, d, m7), 8 dword accumulations (6 unsigned-field dwords, 2 signed via
// xor/sub), two running maxima, and a per-group epilogue of ~20 int→f64 conversions.  Prints elements/s next to the shipped
// kernel's 0.87 T elements/s (0.62 ms per 32 x 4096²).   hipcc --offload-arch=gfx950 -O3 tools/ubench_lut.hip -o /tmp/ubench_lut
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ uint32_t as_u32(us2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_lshr(uint32_t v, uint32_t sh) { uint32_t r; asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh), "v"(v)); return r; }

constexpr int kRows = 9 * 128;   // (d = 0..8) x m7; row 8 = the all-zero entry of values that quantise to 0 everywhere
constexpr int kEntryBytes = 32;

template <int LOOKUPS>
__global__ __launch_bounds__(256, 2) void lut_k1(const uint4 *__restrict__ x, int64_t groups, const uint4 *__restrict__ table, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < kRows * kEntryBytes / 16; i += 256) reinterpret_cast<uint4 *>(lds)[i] = table[i];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    double acc[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) acc[s] = 0.0;
    float mxf = 0.0f;
    uint32_t special = 0u;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += stride) {
        const uint4 lo = x[2 * g], hi = x[2 * g + 1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t ab[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ab[i] = w[i] & 0x7FFF7FFFu;
        us2 m01 = __builtin_elementwise_max(as_us2(ab[0]), as_us2(ab[1])), m23 = __builtin_elementwise_max(as_us2(ab[2]), as_us2(ab[3]));
        us2 m45 = __builtin_elementwise_max(as_us2(ab[4]), as_us2(ab[5])), m67 = __builtin_elementwise_max(as_us2(ab[6]), as_us2(ab[7]));
        const uint32_t mxp = as_u32(__builtin_elementwise_max(__builtin_elementwise_max(m01, m23), __builtin_elementwise_max(m45, m67)));
        const uint32_t E = max(mxp & 0xFFFFu, mxp >> 16) >> 7;
        const uint32_t Ep = E | (E << 16);
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, s0 = 0, s1 = 0, mx0 = 0, mx1 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t e = pk_lshr(ab[i], 0x00070007u);
            const uint32_t dd = Ep - e;
            const uint32_t d = as_u32(__builtin_elementwise_min(as_us2(dd), as_us2(0x00080008u)));
            special |= ((dd + 0x00780078u) & 0x00800080u) >> (i & 7);                       // elements with d >= 8: the sparse b / tail path
            const uint32_t off2 = ((ab[i] & 0x007F007Fu) | (d << 7)) << 5;                  // byte offsets of both entries
            const uint32_t off[2] = {off2 & 0xFFFFu, off2 >> 16};
            const uint32_t sm[2] = {(uint32_t)((int32_t)(w[i] << 16) >> 31), (uint32_t)((int32_t)w[i] >> 31)};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 A = *reinterpret_cast<const uint4 *>(lds + off[h]);
                a0 += A.x; a1 += A.y; a2 += A.z; a3 += A.w;
                if (LOOKUPS >= 2) {
                    const uint4 B = *reinterpret_cast<const uint4 *>(lds + off[h] + 16);
                    a4 += B.x; a5 += B.y;
                    s0 += (B.z ^ sm[h]) - sm[h];
                    s1 += (B.w ^ sm[h]) - sm[h];
                    mx1 = max(mx1, B.y);
                } else {
                    s0 += (A.z ^ sm[h]) - sm[h];
                    s1 += (A.w ^ sm[h]) - sm[h];
                }
                mx0 = as_u32(__builtin_elementwise_max(as_us2(mx0), as_us2(A.w)));
            }
        }
        // epilogue of the group: the integer sums unpacked, scaled by the group's exponent, added to the float64 partials
        const int e1 = (int)E - 148;
        const uint32_t f[14] = {a0 & 0xFFFFu, a0 >> 16, a1 & 0xFFFFFu, a1 >> 20, a2, a3 & 0x3FFFFu, a3 >> 18, a4, a5 & 0xFFFFu, a5 >> 16,
                                s0 & 0xFFFu, s0 >> 12, s1 & 0xFFFFFu, s1 >> 20};
#pragma unroll
        for (int s = 0; s < 14; ++s) acc[s] = acc[s] + __builtin_ldexp((double)f[s], e1 + s);
        mxf = fmaxf(mxf, (float)max(mx0 & 0xFFFFu, max(mx0 >> 16, mx1)));
    }
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < 14; ++s) t += acc[s];
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = t + (double)mxf + (double)special;
}

int main()
{
    const int64_t elems = 32ll * 4096 * 4096, groups = elems / 16;
    std::vector<uint16_t> h((size_t)elems);
    uint32_t s = 12345u;
    for (auto &v : h) {   // N(0, 0.02^2)-like bf16 values: sum of uniforms, sign + exponent spread of a Gaussian
        float u = 0.f;
        for (int k = 0; k < 4; ++k) { s = s * 1664525u + 1013904223u; u += (float)(s >> 8) / 16777216.f - 0.5f; }
        const float f = u * 0.02f * 1.7f;
        uint32_t b; memcpy(&b, &f, 4);
        v = (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
    }
    uint16_t *dx; uint4 *dt; double *dout;
    hipMalloc(&dx, elems * 2); hipMemcpy(dx, h.data(), elems * 2, hipMemcpyHostToDevice);
    std::vector<uint32_t> tab(kRows * kEntryBytes / 4);
    for (size_t i = 0; i < tab.size(); ++i) { s = s * 1664525u + 1013904223u; tab[i] = (s >> 12) & 0x000F00FFu; }
    hipMalloc(&dt, tab.size() * 4); hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    int cus = 256;
    const int blocks = cus * 2;
    hipMalloc(&dout, (size_t)blocks * 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lut_k1<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kRows * kEntryBytes);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lut_k1<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kRows * kEntryBytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 2; variant >= 1; --variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (variant == 2) hipLaunchKernelGGL(lut_k1<2>, dim3(blocks), dim3(256), kRows * kEntryBytes, 0, reinterpret_cast<const uint4 *>(dx), groups, dt, dout);
            else hipLaunchKernelGGL(lut_k1<1>, dim3(blocks), dim3(256), kRows * kEntryBytes, 0, reinterpret_cast<const uint4 *>(dx), groups, dt, dout);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("table-driven skeleton, %d x 16-byte look-ups per element: %.3f ms per 32 x 4096^2 = %.3f T elements/s = %.3f of the bf16 HBM-read roofline (shipped K1: 0.62 ms, 0.87 T, 0.216)\n",
               variant, best, elems / (best * 1e-3) / 1e12, elems * 2.0 / (best * 1e-3) / 8e12);
    }
    return 0;
}
