#include <chrono>
#include <cstdio>
#include <vector>
#include "../../include/mtq.h"
using clk = std::chrono::steady_clock;
int main() {
    mtq_rng *r; mtq_rng_create(&r, 123);
    std::vector<int64_t> out(16384);
    const int reps = 2000;
    auto t0 = clk::now();
    for (int i = 0; i < reps; ++i) mtq_rng_permutation(r, 16384, out.data());
    auto t1 = clk::now();
    printf("permutation(16384): %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
    t0 = clk::now();
    for (int i = 0; i < reps; ++i) mtq_rng_permutation(r, 2514, out.data());
    t1 = clk::now();
    printf("permutation(2514): %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
    return 0;
}
