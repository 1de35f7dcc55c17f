#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/mtq.h"
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main(int argc, char **argv)
{
    FILE *f = fopen("/tmp/mtq_stats4096.bin", "rb");
    const int64_t T = 16384; const int rec = 22;
    std::vector<double> st(T * rec);
    if (fread(st.data(), 8, st.size(), f) != st.size()) return 1;
    fclose(f);
    const int reps = 20;
    std::vector<std::vector<double>> copies(reps, st);
    int fm[4] = {0, 1, 2, 3};
    std::vector<int8_t> map(T); int64_t counts[4]; double out[9];
    // whole run
    auto t0 = clk::now();
    for (int r = 0; r < reps; ++r) mtq_greedy_run(copies[r].data(), T, 0xF, fm, 4, 0, 0.999, 4096.0 * 4096.0, 123, map.data(), counts, out);
    auto t1 = clk::now();
    printf("greedy_run: %.3f ms/tensor  counts %ld %ld %ld %ld\n", ms(t0, t1) / reps, counts[0], counts[1], counts[2], counts[3]);
    // parts
    double tc = 0, tp[4] = {0, 0, 0, 0}, tr = 0, tcol = 0;
    for (int r = 0; r < reps; ++r) {
        auto a = clk::now();
        mtq_greedy *g; mtq_greedy_create(&g, copies[r].data(), T, 0xF, 0, 0.999, 4096.0 * 4096.0, 0);
        auto b = clk::now(); tc += ms(a, b);
        mtq_rng *rng; mtq_rng_create(&rng, 123);
        std::vector<uint8_t> fixed(T); std::vector<int64_t> cand(T), perm(T), order(T);
        for (int k = 0; k < 4; ++k) {
            auto c0 = clk::now();
            mtq_greedy_fixed(g, fixed.data()); int64_t n = 0; for (int64_t t = 0; t < T; ++t) if (!fixed[t]) cand[n++] = t;
            if (!n) break;
            mtq_rng_permutation(rng, n, perm.data()); for (int64_t i = 0; i < n; ++i) order[i] = cand[perm[i]];
            auto c1 = clk::now(); tr += ms(c0, c1);
            mtq_greedy_pass(g, k, order.data(), n);
            auto c2 = clk::now(); tp[k] += ms(c1, c2);
        }
        auto d0 = clk::now();
        mtq_greedy_assignment(g, map.data()); mtq_columns_from_stats(copies[r].data(), T, 0xF, map.data(), 4096.0 * 4096.0, out);
        auto d1 = clk::now(); tcol += ms(d0, d1);
        mtq_greedy_destroy(g); mtq_rng_destroy(rng);
    }
    printf("create %.3f  rng+order %.3f  pass %.3f %.3f %.3f %.3f  columns %.3f (ms/tensor)\n", tc / reps, tr / reps, tp[0] / reps, tp[1] / reps, tp[2] / reps, tp[3] / reps, tcol / reps);
    return 0;
}
