// Recommender.h — the reference's public class API over the MI355X engine.
//
// Drop-in for the reference's Recommender.h:28-83: same class name, same
// public member signatures and the public `Recommendation` struct
// (Recommender.h:12-22), so the reference's main.cpp compiles against this
// header unchanged.  The private part is an opaque pointer: all device state
// lives behind the C-ABI in include/mi355rec.h.
//
// Behavioural contract:
//  - initialize(): false for an empty song list (Recommender.cu:103-106).  On a
//    host WITHOUT a HIP device it does what the reference does (Recommender.cu:
//    117-127,176-181): prints the reference's fallback lines, serves the
//    catalogue from the product's own CPU backend (csrc/cpu_backend.cpp, behind
//    mi355rec_create_placed) and returns true with isGPUEnabled() == false.  On a
//    host WITH a device nothing ever falls back: a failed HIP call is an error.
//  - recommendByIndex(): ids of the topN most cosine-similar songs, best
//    first, the query excluded by index; min(topN, N-1) results; {} plus the
//    reference's stderr text for an uninitialised object or a bad index.
//    topN <= 0 returns {} (the reference crashes).
//    Order inside runs of exactly equal scores is ascending index (the
//    reference's is a heap artefact).
//  - recommend()/recommendByName(): the reference's lookup rules
//    (Recommender.cu:320-354): exact id; name = case-insensitive exact match
//    first, else first case-insensitive substring match.
#ifndef RECOMMENDER_H
#define RECOMMENDER_H

#include <map>
#include <string>
#include <vector>

#include "Song.h"

struct Recommendation {
    int songIndex;
    float similarity;

    Recommendation() : songIndex(-1), similarity(0.0f) {}
    Recommendation(int idx, float sim) : songIndex(idx), similarity(sim) {}

    // inverted on purpose, as in the reference: a std::priority_queue of
    // Recommendation keeps the LOWEST similarity on top
    bool operator<(const Recommendation& other) const { return similarity > other.similarity; }
};

class Recommender {
public:
    Recommender();
    ~Recommender();
    Recommender(const Recommender&) = delete;
    Recommender& operator=(const Recommender&) = delete;

    bool initialize(const std::vector<Song>& songs);

    std::vector<int> recommend(const std::string& trackId, int topN);
    std::vector<int> recommendByName(const std::string& trackName, int topN);
    std::vector<int> recommendByIndex(int songIndex, int topN);

    bool isInitialized() const;
    bool isGPUEnabled() const;
    int getSongCount() const;

    // Extension: the same initialisation from what DataManager::loadCatalogue read —
    // the row-major N x 12 matrix goes to the GPUs as it is, ids / names feed the
    // lookups; no vector<Song> is ever built (the reference deep-copies it,
    // Recommender.cu:109, then flattens it, :162-167).
    bool initialize(const std::vector<float>& features, const std::vector<std::string>& trackIds,
                    const std::vector<std::string>& trackNames);

    // Extensions (not in the reference): the scores of the last
    // recommendByIndex result, and the full score vector of one query row
    // (what the reference's private calculateSimilarities produced).
    const std::vector<float>& lastScores() const;
    bool similarities(int songIndex, std::vector<float>& out);

    struct Impl;   // opaque: defined in Recommender.cpp

private:
    Impl* impl_;
};

#endif  // RECOMMENDER_H
