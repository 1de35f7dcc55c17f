// dependent-chain latency of a few VALU ops, one wave per block: ticks (s_memtime) per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void chain(double *out, unsigned long long *ticks, double seed, int blocks_active)
{
    double s = seed + threadIdx.x, t = seed * 3 + threadIdx.x, u = seed * 5, v = seed * 7;
    float f = (float)seed;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) s = s + t;                                  // dependent add_f64
            if (KIND == 1) { s = s + t; u = u + v; }                   // two independent chains
            if (KIND == 2) f = f + 1.5f;                               // dependent add_f32
            if (KIND == 3) s = __builtin_fma(s, t, u);                 // dependent fma_f64
            if (KIND == 4) { s = s + t; u = u + v; t = t + 1.0; v = v + 1.0; }   // four chains (two pairs)
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = s + u + f + t + v;
}
int main()
{
    double *out; unsigned long long *ticks, h[1024];
    hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&ticks, 1024 * 8);
    const char *names[5] = {"add_f64 dependent", "add_f64 x2 chains", "add_f32 dependent", "fma_f64 dependent", "add_f64 x4 chains"};
    for (int blocks : {1, 128, 1024}) {
        for (int kind = 0; kind < 5; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(chain<0>, dim3(blocks), dim3(64), 0, 0, out, ticks, 1.0, blocks);
                if (kind == 1) hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(64), 0, 0, out, ticks, 1.0, blocks);
                if (kind == 2) hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(64), 0, 0, out, ticks, 1.0, blocks);
                if (kind == 3) hipLaunchKernelGGL(chain<3>, dim3(blocks), dim3(64), 0, 0, out, ticks, 1.0, blocks);
                if (kind == 4) hipLaunchKernelGGL(chain<4>, dim3(blocks), dim3(64), 0, 0, out, ticks, 1.0, blocks);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, ticks, blocks * 8, hipMemcpyDeviceToHost);
            const int per = kind == 1 ? 2 : (kind == 4 ? 4 : 1);
            printf("%4d blocks  %-20s %6.2f ticks per instruction (block 0), %6.2f per chain step\n", blocks, names[kind], (double)h[0] / (256.0 * 16 * per), (double)h[0] / (256.0 * 16));
        }
    }
    // wall-clock calibration of the tick: one long kernel
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, out, ticks, 1.0, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost);
    printf("50 launches %.3f ms; one launch %llu ticks -> upper bound %.2f G ticks/s\n", ms, h[0], 50.0 * h[0] / (ms * 1e-3) / 1e9);
    return 0;
}
