// latency.cpp — single-query latency through the C-ABI from a C++ host (what the
// drop-in Recommender pays per recommendByIndex call), without any Python in the loop.
//   g++ -O2 -std=c++17 -Iinclude tools/latency.cpp -Lspotify_recommender_amd -lmi355rec \
//       -Wl,-rpath,'$ORIGIN/../spotify_recommender_amd' -o tools/latency
//   tools/latency [rows=10000000] [topn=100] [queries=2000]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mi355rec.h"

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int topn = argc > 2 ? atoi(argv[2]) : 100;
    const int queries = argc > 3 ? atoi(argv[3]) : 2000;
    std::vector<float> feats(static_cast<size_t>(n) * 12);
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> dist(0.0f, 1.0f);
    for (float& v : feats) v = dist(gen);
    mi355rec_t* h = nullptr;
    if (mi355rec_create(feats.data(), n, 12, 0, 0, &h) != MI355REC_OK) {
        std::fprintf(stderr, "create failed: %s\n", mi355rec_last_global_error());
        return 1;
    }
    std::vector<int64_t> idx(topn);
    std::vector<float> score(topn);
    int count = 0;
    std::vector<double> us;
    for (int k = 0; k < queries + 50; ++k) {
        const int64_t row = (static_cast<int64_t>(k) * 7919) % n;
        const auto t0 = std::chrono::steady_clock::now();
        if (mi355rec_query_row_topn(h, row, topn, idx.data(), score.data(), &count) != MI355REC_OK) {
            std::fprintf(stderr, "query failed: %s\n", mi355rec_last_error(h));
            return 1;
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (k >= 50) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(us.begin(), us.end());
    std::printf("{\"rows\": %lld, \"topn\": %d, \"queries\": %d, \"p50_us\": %.1f, \"p90_us\": %.1f, \"p99_us\": %.1f, \"min_us\": %.1f}\n",
                (long long)n, topn, queries, us[us.size() / 2], us[us.size() * 9 / 10], us[us.size() * 99 / 100], us.front());
    mi355rec_destroy(h);
    return 0;
}
