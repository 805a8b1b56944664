#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "../../kmertools_amd/csrc/host/computers.hpp"
using namespace kthost;
int main(int argc, char **argv) {
    const uint64_t n = 100000, bins = 136;
    const int t = atoi(argv[1]), mode = atoi(argv[2]);
    std::vector<double> rows(n * bins);
    for (uint64_t i = 0; i < n * bins; i++) rows[i] = (double)((i * 2654435761ull >> 7) % 5) / 147.0;
    std::vector<std::string> pieces(t);
    std::vector<std::vector<char>> raw(t);
    for (int rep = 0; rep < 4; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int w = 0; w < t; w++) pool.emplace_back([&, w] {
            const uint64_t lo = n * w / t, hi = n * (w + 1) / t;
            if (mode == 0) {  // std::string append
                std::string s; s.swap(pieces[w]); s.clear(); s.reserve((hi - lo) * 1225);
                char buf[FIXED6_BUF];
                for (uint64_t r = lo; r < hi; r++) { for (uint64_t i = 0; i < bins; i++) { if (i) s += ' '; s.append(buf, format_fixed6(buf, rows[r * bins + i])); } s += '\n'; }
                s.swap(pieces[w]);
            } else if (mode == 1) {  // format only
                char buf[FIXED6_BUF]; size_t acc = 0;
                for (uint64_t r = lo; r < hi; r++) for (uint64_t i = 0; i < bins; i++) acc += format_fixed6(buf, rows[r * bins + i]) + buf[3];
                if (acc == 42) puts("x");
            } else {  // raw char buffer
                raw[w].resize((hi - lo) * 1225 + 400);
                char *q = raw[w].data();
                for (uint64_t r = lo; r < hi; r++) { for (uint64_t i = 0; i < bins; i++) { if (i) *q++ = ' '; q += format_fixed6(q, rows[r * bins + i]); } *q++ = '\n'; }
            }
        });
        for (auto &th : pool) th.join();
        printf("t=%d mode=%d rep %d: %.3f s\n", t, mode, rep, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
}
