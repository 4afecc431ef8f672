// lanes.cpp — a stream of single queries from a C++ host (no Python in the loop), dealt over L lanes of one handle
// (mi355rec_create_lane / mi355rec_own_stream, include/mi355rec.h): queries per second for L = 1 .. max_lanes.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/lanes.cpp -Lspotify_recommender_amd -lmi355rec \
//       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/spotify_recommender_amd -o /tmp/lanes
//   /tmp/lanes [rows=10000000] [topn=100] [queries=3000] [max_lanes=3]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "mi355rec_diag.h"

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int topn = argc > 2 ? atoi(argv[2]) : 100;
    const int queries = argc > 3 ? atoi(argv[3]) : 3000;
    const int max_lanes = argc > 4 ? atoi(argv[4]) : 3;
    std::vector<float> feats(static_cast<size_t>(n) * 12);
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> dist(0.0f, 1.0f);
    for (float& v : feats) v = dist(gen);
    mi355rec_t* first = nullptr;
    if (mi355rec_create(feats.data(), n, 12, 0, 0, &first) != MI355REC_OK) {
        std::fprintf(stderr, "create: %s\n", mi355rec_last_error(nullptr));
        return 1;
    }
    std::vector<mi355rec_t*> lanes{first};
    for (int l = 1; l < max_lanes; ++l) {
        mi355rec_t* ln = nullptr;
        if (mi355rec_create_lane(first, &ln) != MI355REC_OK) {
            std::fprintf(stderr, "create_lane: %s\n", mi355rec_last_error(first));
            return 1;
        }
        lanes.push_back(ln);
    }
    mi355rec_key_t* d_keys = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_keys), sizeof(mi355rec_key_t) * topn * 64) != hipSuccess) return 1;
    std::vector<mi355rec_key_t> ref(topn), got(topn);
    std::printf("{\"rows\": %lld, \"topn\": %d, \"queries\": %d", (long long)n, topn, queries);
    for (int L = 1; L <= max_lanes; ++L) {
        auto run = [&](int q0, int q1) {
            for (int k = q0; k < q1; ++k) {
                mi355rec_t* h = lanes[k % L];
                const int64_t row = (static_cast<int64_t>(k) * 7919) % n;
                if (mi355rec_enqueue_row_keys_streamed(h, row, topn, d_keys + static_cast<size_t>(k % 64) * topn, mi355rec_own_stream(h)) != MI355REC_OK) {
                    std::fprintf(stderr, "enqueue: %s\n", mi355rec_last_error(h));
                    std::exit(1);
                }
            }
            for (int l = 0; l < L; ++l) mi355rec_enqueue_flush(lanes[l], mi355rec_own_stream(lanes[l]));
            for (int l = 0; l < L; ++l) (void)hipStreamSynchronize(static_cast<hipStream_t>(mi355rec_own_stream(lanes[l])));
        };
        run(0, 64);
        const auto t0 = std::chrono::steady_clock::now();
        run(64, 64 + queries);
        const auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / queries;
        // the last query of the run against the synchronous call of the first handle
        const int last = 64 + queries - 1;
        (void)hipMemcpy(got.data(), d_keys + static_cast<size_t>(last % 64) * topn, sizeof(mi355rec_key_t) * topn, hipMemcpyDeviceToHost);
        std::vector<int64_t> idx(topn);
        std::vector<float> sc(topn);
        int count = 0;
        mi355rec_query_row_topn(first, (static_cast<int64_t>(last) * 7919) % n, topn, idx.data(), sc.data(), &count);
        bool same = count == topn;
        for (int i = 0; i < count && same; ++i) same = mi355rec_key_row(got[i]) == idx[i];
        std::printf(", \"lanes_%d\": {\"us_per_query\": %.2f, \"queries_per_s\": %.0f, \"last_matches_sync_call\": %s}", L, us, 1e6 / us,
                    same ? "true" : "false");
    }
    std::printf("}\n");
    for (size_t l = lanes.size(); l-- > 0;) mi355rec_destroy(lanes[l]);
    (void)hipFree(d_keys);
    return 0;
}
