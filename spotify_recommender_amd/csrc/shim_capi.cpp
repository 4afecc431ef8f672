// shim_capi.cpp — plain-C handles onto the C++ drop-in classes so that the
// parity tests (ctypes) can drive Recommender / DataManager exactly as
// main.cpp does.  Test/tooling surface only; applications use the classes.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "DataManager.h"
#include "Recommender.h"
#include "Song.h"

namespace {
struct Catalogue {
    std::vector<Song> songs;
    std::map<int, std::string> genres;
    Recommender rec;
};
int64_t copyOut(const std::string& s, char* out, int64_t cap) {
    const int64_t n = static_cast<int64_t>(s.size()) < cap ? static_cast<int64_t>(s.size()) : cap;
    if (n > 0) std::memcpy(out, s.data(), static_cast<size_t>(n));
    return static_cast<int64_t>(s.size());
}
}  // namespace

extern "C" {

int shim_preprocess(const char* csv, const char* out) { return DataManager::preprocessData(csv, out) ? 1 : 0; }

void* shim_load(const char* bin) {
    Catalogue* c = new Catalogue();
    if (!DataManager::loadData(bin, c->songs, c->genres)) {
        delete c;
        return nullptr;
    }
    return c;
}

void* shim_from_matrix(const float* feats, int64_t n) {
    Catalogue* c = new Catalogue();
    c->songs.resize(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) {
        c->songs[i].track_id = "id" + std::to_string(i);
        c->songs[i].track_name = "Song " + std::to_string(i);
        std::memcpy(c->songs[i].features, feats + i * FEATURE_COUNT, sizeof(float) * FEATURE_COUNT);
    }
    return c;
}

// The CLI's query-mode path: DataManager::loadCatalogue + the matrix overload of
// Recommender::initialize.  Songs are only materialised one at a time (readSong).
struct FastCatalogue {
    DataManager::Catalogue cat;
    Recommender rec;
};

void* shim_fast_load(const char* bin) {
    FastCatalogue* c = new FastCatalogue();
    if (!DataManager::loadCatalogue(bin, c->cat)) {
        delete c;
        return nullptr;
    }
    return c;
}
int shim_fast_initialize(void* h) {
    FastCatalogue* c = static_cast<FastCatalogue*>(h);
    return c->rec.initialize(c->cat.features, c->cat.trackIds, c->cat.trackNames) ? 1 : 0;
}
void shim_fast_free(void* h) { delete static_cast<FastCatalogue*>(h); }
int64_t shim_fast_song_count(void* h) { return static_cast<int64_t>(static_cast<FastCatalogue*>(h)->cat.size()); }
int64_t shim_fast_recommend(void* h, const char* id, int topn, int* out, int64_t cap) {
    FastCatalogue* c = static_cast<FastCatalogue*>(h);
    const std::vector<int> r = c->rec.recommend(id, topn);
    for (size_t i = 0; i < r.size() && static_cast<int64_t>(i) < cap; ++i) out[i] = r[i];
    return static_cast<int64_t>(r.size());
}
int64_t shim_fast_recommend_by_name(void* h, const char* name, int topn, int* out, int64_t cap) {
    FastCatalogue* c = static_cast<FastCatalogue*>(h);
    const std::vector<int> r = c->rec.recommendByName(name, topn);
    for (size_t i = 0; i < r.size() && static_cast<int64_t>(i) < cap; ++i) out[i] = r[i];
    return static_cast<int64_t>(r.size());
}
// which: 0 id, 1 name, 2 artists, 3 genre name — read back from the file through readSong
int64_t shim_fast_song_string(void* h, int64_t i, int which, char* out, int64_t cap) {
    FastCatalogue* c = static_cast<FastCatalogue*>(h);
    Song s;
    if (i < 0 || !DataManager::readSong(c->cat, static_cast<size_t>(i), s)) return -1;
    if (which == 3) return copyOut(c->cat.genreMap[s.genre_id], out, cap);
    return copyOut(which == 0 ? s.track_id : which == 1 ? s.track_name : s.artists, out, cap);
}
// writes a synthetic songs_data.bin (Song::serialize) straight from a feature matrix: test fixture
int shim_write_synthetic_bin(const char* path, const float* feats, int64_t n, int genres) {
    std::ofstream out(path, std::ios::binary);
    if (!out.is_open()) return 0;
    const size_t numSongs = static_cast<size_t>(n), numGenres = static_cast<size_t>(genres);
    out.write(reinterpret_cast<const char*>(&numSongs), sizeof numSongs);
    out.write(reinterpret_cast<const char*>(&numGenres), sizeof numGenres);
    for (int g = 0; g < genres; ++g) {
        const std::string name = "genre-" + std::to_string(g);
        const size_t len = name.size();
        out.write(reinterpret_cast<const char*>(&g), sizeof g);
        out.write(reinterpret_cast<const char*>(&len), sizeof len);
        out.write(name.data(), static_cast<std::streamsize>(len));
    }
    Song s;
    for (int64_t i = 0; i < n; ++i) {
        s.track_id = "id" + std::to_string(i);
        s.track_name = "Song " + std::to_string(i);
        s.artists = "Artist " + std::to_string(i % 977);
        s.genre_id = static_cast<int>(i % genres);
        std::memcpy(s.features, feats + i * FEATURE_COUNT, sizeof(float) * FEATURE_COUNT);
        s.serialize(out);
    }
    return out.good() ? 1 : 0;
}

void shim_free(void* h) { delete static_cast<Catalogue*>(h); }
int64_t shim_song_count(void* h) { return static_cast<int64_t>(static_cast<Catalogue*>(h)->songs.size()); }
int64_t shim_genre_count(void* h) { return static_cast<int64_t>(static_cast<Catalogue*>(h)->genres.size()); }

int shim_song_features(void* h, int64_t i, float* out12, int* genre) {
    Catalogue* c = static_cast<Catalogue*>(h);
    if (i < 0 || i >= static_cast<int64_t>(c->songs.size())) return 0;
    std::memcpy(out12, c->songs[i].features, sizeof(float) * FEATURE_COUNT);
    *genre = c->songs[i].genre_id;
    return 1;
}

int64_t shim_song_string(void* h, int64_t i, int which, char* out, int64_t cap) {
    Catalogue* c = static_cast<Catalogue*>(h);
    if (i < 0 || i >= static_cast<int64_t>(c->songs.size())) return -1;
    const Song& s = c->songs[i];
    return copyOut(which == 0 ? s.track_id : which == 1 ? s.track_name : s.artists, out, cap);
}

int64_t shim_genre_name(void* h, int id, char* out, int64_t cap) {
    Catalogue* c = static_cast<Catalogue*>(h);
    const auto it = c->genres.find(id);
    return it == c->genres.end() ? -1 : copyOut(it->second, out, cap);
}


int shim_initialize(void* h) { Catalogue* c = static_cast<Catalogue*>(h); return c->rec.initialize(c->songs) ? 1 : 0; }
int shim_is_initialized(void* h) { return static_cast<Catalogue*>(h)->rec.isInitialized() ? 1 : 0; }
int shim_is_gpu_enabled(void* h) { return static_cast<Catalogue*>(h)->rec.isGPUEnabled() ? 1 : 0; }
int shim_get_song_count(void* h) { return static_cast<Catalogue*>(h)->rec.getSongCount(); }

static int64_t giveBack(Catalogue* c, const std::vector<int>& r, int* out, float* scores, int64_t cap) {
    const int64_t n = static_cast<int64_t>(r.size()) < cap ? static_cast<int64_t>(r.size()) : cap;
    for (int64_t i = 0; i < n; ++i) {
        out[i] = r[i];
        if (scores) scores[i] = c->rec.lastScores()[i];
    }
    return static_cast<int64_t>(r.size());
}

int64_t shim_recommend_by_index(void* h, int idx, int topn, int* out, float* scores, int64_t cap) {
    Catalogue* c = static_cast<Catalogue*>(h);
    return giveBack(c, c->rec.recommendByIndex(idx, topn), out, scores, cap);
}
int64_t shim_recommend(void* h, const char* id, int topn, int* out, float* scores, int64_t cap) {
    Catalogue* c = static_cast<Catalogue*>(h);
    return giveBack(c, c->rec.recommend(id, topn), out, scores, cap);
}
int64_t shim_recommend_by_name(void* h, const char* name, int topn, int* out, float* scores, int64_t cap) {
    Catalogue* c = static_cast<Catalogue*>(h);
    return giveBack(c, c->rec.recommendByName(name, topn), out, scores, cap);
}
int shim_similarities(void* h, int idx, float* out_n) {
    Catalogue* c = static_cast<Catalogue*>(h);
    std::vector<float> v;
    if (!c->rec.similarities(idx, v)) return 0;
    std::memcpy(out_n, v.data(), v.size() * sizeof(float));
    return 1;
}

}  // extern "C"
