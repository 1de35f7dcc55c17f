#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/mtq.h"
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main(int argc, char **argv)
{
    const int64_t Tfull = 16384; const int rec = 11;
    FILE *f = fopen("/tmp/mtq_slim4096.bin", "rb");
    std::vector<double> st(Tfull * rec);
    if (fread(st.data(), 8, st.size(), f) != st.size()) return 1;
    fclose(f);
    const uint32_t mask = 0xE | MTQ_MASK_BF16_IDENTITY | MTQ_MASK_SLIM;
    for (int64_t T : {16384L, 2048L}) {
        const int reps = 30;
        double tc = 0, tp[4] = {0, 0, 0, 0}, tr = 0; int64_t nn[4] = {0,0,0,0};
        int64_t counts[4];
        for (int r = 0; r < reps; ++r) {
            std::vector<double> copy(st.begin(), st.begin() + T * rec);
            auto a = clk::now();
            mtq_greedy *g; if (mtq_greedy_create(&g, copy.data(), T, mask, 0, 0.999, 1024.0 * T, 0)) { printf("create failed %s\n", mtq_last_error()); return 1; }
            auto b = clk::now(); tc += us(a, b);
            mtq_rng *rng; mtq_rng_create(&rng, 123);
            std::vector<uint8_t> fixed(T); std::vector<int64_t> cand(T), perm(T), order(T);
            for (int k = 0; k < 4; ++k) {
                auto c0 = clk::now();
                mtq_greedy_fixed(g, fixed.data()); int64_t n = 0; for (int64_t t = 0; t < T; ++t) if (!fixed[t]) cand[n++] = t;
                if (!n) break;
                mtq_rng_permutation(rng, n, perm.data()); for (int64_t i = 0; i < n; ++i) order[i] = cand[perm[i]];
                auto c1 = clk::now(); tr += us(c0, c1);
                mtq_greedy_pass(g, k, order.data(), n);
                auto c2 = clk::now(); tp[k] += us(c1, c2); nn[k] = n;
            }
            mtq_greedy_counts(g, counts);
            mtq_greedy_destroy(g); mtq_rng_destroy(rng);
        }
        printf("T=%ld: create %.1f us  rng+order %.1f us  passes:", T, tc / reps, tr / reps);
        for (int k = 0; k < 4; ++k) printf("  [n=%ld %.1f us = %.2f ns/decision]", nn[k], tp[k] / reps, nn[k] ? 1e3 * tp[k] / reps / nn[k] : 0.0);
        printf("  counts %ld %ld %ld %ld\n", counts[0], counts[1], counts[2], counts[3]);
    }
    return 0;
}
