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

#ifdef HEAP_CHECK_REFERENCE_HEADER
// Build container only (-I /root/reference): the reference's OWN Recommendation
// struct and comparator (Recommender.h:12-22), not a restatement.
#include "Recommender.h"
typedef Recommendation Rec;
static inline int rec_idx(const Rec& r) { return r.songIndex; }
static inline float rec_sim(const Rec& r) { return r.similarity; }
#else
// Where /root/reference is absent (the GPU box): the same 3 lines restated.
struct Rec {
    int songIndex;
    float similarity;
    Rec(int i, float s) : songIndex(i), similarity(s) {}
    bool operator<(const Rec& o) const { return similarity > o.similarity; }
};
static inline int rec_idx(const Rec& r) { return r.songIndex; }
static inline float rec_sim(const Rec& r) { return r.similarity; }
#endif

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    long n = atol(argv[1]), exclude = atol(argv[2]), topn = atol(argv[3]);
    std::vector<float> s(n);
    if (fread(s.data(), sizeof(float), n, stdin) != (size_t)n) return 3;
    std::priority_queue<Rec> heap;
    for (long i = 0; i < n; ++i) {
        if (i == exclude) continue;
        Rec r((int)i, s[i]);
        if (heap.size() < (size_t)topn) heap.push(r);
        else if (rec_sim(r) > rec_sim(heap.top())) { heap.pop(); heap.push(r); }
    }
    std::vector<int> out;
    while (!heap.empty()) { out.push_back(rec_idx(heap.top())); heap.pop(); }
    std::reverse(out.begin(), out.end());
    for (int v : out) printf("%d\n", v);
    return 0;
}
