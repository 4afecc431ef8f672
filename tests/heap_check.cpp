// Replays the reference's top-N loop (Recommender.cu:293-315) with the REAL
// libstdc++ std::priority_queue and the Recommendation comparator semantics of
// Recommender.h:12-22, so tests can pin oracle_topn_heap (a C restatement of
// libstdc++'s heap algorithms) against the genuine container.
// usage: heap_check <n> <exclude> <topn> < scores(float32 binary)   -> indices, one per line
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <vector>

struct Rec {
    int idx;
    float sim;
    bool operator<(const Rec& o) const { return sim > o.sim; }
};

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    long n = atol(argv[1]), exclude = atol(argv[2]), topn = atol(argv[3]);
    std::vector<float> s(n);
    if (fread(s.data(), sizeof(float), n, stdin) != (size_t)n) return 3;
    std::priority_queue<Rec> heap;
    for (long i = 0; i < n; ++i) {
        if (i == exclude) continue;
        Rec r{(int)i, s[i]};
        if (heap.size() < (size_t)topn) heap.push(r);
        else if (r.sim > heap.top().sim) { heap.pop(); heap.push(r); }
    }
    std::vector<int> out;
    while (!heap.empty()) { out.push_back(heap.top().idx); heap.pop(); }
    std::reverse(out.begin(), out.end());
    for (int v : out) printf("%d\n", v);
    return 0;
}
