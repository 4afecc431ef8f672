// latency.cpp — single-query latency from a C++ host, without any Python in the loop:
//   (a) the C-ABI call directly                    mi355rec_query_row_topn
//   (b) the drop-in class the reference's main.cpp uses   Recommender::recommendByIndex
//       (include/Recommender.h over mi355rec_create_sharded; Recommender.cu:275-318)
//   (c) the sharded handle's synchronous call      mi355rec_sharded_query_row_topn
// (b) and (c) must cost what (a) costs on a one-GPU box (one shard forwards to its handle).
//   g++ -O2 -std=c++17 -Iinclude tools/latency.cpp spotify_recommender_amd/csrc/Recommender.cpp \
//       spotify_recommender_amd/csrc/DataManager.cpp -Lspotify_recommender_amd -lmi355rec \
//       -Wl,-rpath,'$ORIGIN/../spotify_recommender_amd' -o tools/latency
//   tools/latency [rows=10000000] [topn=100] [queries=2000] [virtual_shards=0] [replica_mode=-1]
//   (replica_mode >= 0: mi355rec_set_replica on the single-device handle before (a) is measured — 1 = the fp32 rows)
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "Recommender.h"
#include "mi355rec_diag.h"

namespace {

struct Pct {
    double p50, p90, p99, min;
};

Pct measure(int queries, int64_t n, const std::function<bool(int64_t)>& one) {
    std::vector<double> us;
    for (int k = 0; k < queries + 50; ++k) {
        const int64_t row = (static_cast<int64_t>(k) * 7919) % n;
        const auto t0 = std::chrono::steady_clock::now();
        if (!one(row)) std::exit(1);
        const auto t1 = std::chrono::steady_clock::now();
        if (k >= 50) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(us.begin(), us.end());
    return {us[us.size() / 2], us[us.size() * 9 / 10], us[us.size() * 99 / 100], us.front()};
}

std::string g_json;   // the one JSON line, printed last (the Recommender talks on stdout while it initialises)

void add(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_json += buf;
}

void print(const char* name, const Pct& p, bool last) {
    add("\"%s\": {\"p50_us\": %.1f, \"p90_us\": %.1f, \"p99_us\": %.1f, \"min_us\": %.1f}%s", name, p.p50, p.p90, p.p99, p.min,
        last ? "" : ", ");
}

}  // namespace

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int topn = argc > 2 ? atoi(argv[2]) : 100;
    const int queries = argc > 3 ? atoi(argv[3]) : 2000;
    const int vshards = argc > 4 ? atoi(argv[4]) : 0;
    const int replica_mode = argc > 5 ? atoi(argv[5]) : -1;
    std::vector<float> feats(static_cast<size_t>(n) * 12);
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> dist(0.0f, 1.0f);
    for (float& v : feats) v = dist(gen);
    std::vector<int64_t> idx(topn);
    std::vector<float> score(topn);
    int count = 0;

    add("{\"rows\": %lld, \"topn\": %d, \"queries\": %d, ", (long long)n, topn, queries);

    // (a) the single-device handle
    std::vector<int64_t> direct_first;
    {
        mi355rec_t* h = nullptr;
        if (mi355rec_create(feats.data(), n, 12, 0, 0, &h) != MI355REC_OK) {
            std::fprintf(stderr, "create failed: %s\n", mi355rec_last_global_error());
            return 1;
        }
        if (replica_mode >= 0 && mi355rec_set_replica(h, replica_mode) != MI355REC_OK) {
            std::fprintf(stderr, "set_replica failed: %s\n", mi355rec_last_error(h));
            return 1;
        }
        add("\"replica_mode\": %d, ", replica_mode);
        const Pct p = measure(queries, n, [&](int64_t row) {
            if (mi355rec_query_row_topn(h, row, topn, idx.data(), score.data(), &count) == MI355REC_OK) return true;
            std::fprintf(stderr, "query failed: %s\n", mi355rec_last_error(h));
            return false;
        });
        mi355rec_query_row_topn(h, 4242 % n, topn, idx.data(), score.data(), &count);
        direct_first.assign(idx.begin(), idx.begin() + count);
        print("c_abi_query_row_topn", p, false);
        mi355rec_destroy(h);
    }

    // (c) the sharded handle: every visible GPU, or `vshards` virtual shards of device 0
    {
        mi355rec_sharded_t* h = nullptr;
        int rc;
        if (vshards > 1) {
            std::vector<int> devs(vshards, 0);
            rc = mi355rec_create_sharded_on(feats.data(), n, 12, devs.data(), vshards, &h);
        } else {
            rc = mi355rec_create_sharded(feats.data(), n, 12, 0, &h);
        }
        if (rc != MI355REC_OK) {
            std::fprintf(stderr, "create_sharded failed: %s\n", mi355rec_sharded_last_error(nullptr));
            return 1;
        }
        int shards = 0;
        mi355rec_sharded_info(h, &shards, nullptr, nullptr, nullptr, nullptr);
        const Pct p = measure(queries, n, [&](int64_t row) {
            if (mi355rec_sharded_query_row_topn(h, row, topn, idx.data(), score.data(), &count) == MI355REC_OK) return true;
            std::fprintf(stderr, "sharded query failed: %s\n", mi355rec_sharded_last_error(h));
            return false;
        });
        add("\"shards\": %d, ", shards);
        print("sharded_query_row_topn", p, false);
        mi355rec_sharded_destroy(h);
    }

    // (b) the drop-in class
    {
        std::vector<std::string> ids(static_cast<size_t>(n)), names(static_cast<size_t>(n));
        for (int64_t i = 0; i < n; ++i) {
            ids[i] = "t" + std::to_string(i);
            names[i] = "s" + std::to_string(i);
        }
        Recommender rec;
        if (!rec.initialize(feats, ids, names)) return 1;
        std::vector<int> out;
        const Pct p = measure(queries, n, [&](int64_t row) {
            out = rec.recommendByIndex(static_cast<int>(row), topn);
            return !out.empty();
        });
        out = rec.recommendByIndex(static_cast<int>(4242 % n), topn);
        bool same = out.size() == direct_first.size();
        for (size_t i = 0; same && i < out.size(); ++i) same = out[i] == direct_first[i];
        print("recommender_recommend_by_index", p, false);
        add("\"recommender_matches_c_abi\": %s}", same ? "true" : "false");
    }
    std::printf("%s\n", g_json.c_str());
    return 0;
}
