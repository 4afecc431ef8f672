// ref_dm_driver.cpp — C-ABI driver around the REFERENCE's own DataManager
// (compiled in place from /root/reference by oracle/Makefile into
// oracle/_ref/libref_dm.so; test infrastructure only, never shipped).
//
// It lets tests/ pin our songs_data.bin reader/writer and CSV preprocessing
// against what the reference's code really writes and reads
// (DataManager.cpp:94-361 preprocessData, :363-409 loadData, Song.h:35-77).
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "DataManager.h"
#include "Song.h"

extern "C" {

// Runs the reference's preprocessData (CSV -> songs_data.bin). 1 = ok.
int ref_dm_preprocess(const char* csv_path, const char* out_path) {
    return DataManager::preprocessData(csv_path, out_path) ? 1 : 0;
}

static std::vector<Song> g_songs;
static std::map<int, std::string> g_genres;

// Runs the reference's loadData and keeps the result for the getters below.
// Returns the number of songs, or -1 on failure.
int64_t ref_dm_load(const char* bin_path) {
    g_songs.clear();
    g_genres.clear();
    if (!DataManager::loadData(bin_path, g_songs, g_genres)) return -1;
    return (int64_t)g_songs.size();
}

int64_t ref_dm_genre_count() { return (int64_t)g_genres.size(); }

// Copies song i's features (12 floats) and genre id.
int ref_dm_song_features(int64_t i, float* out12, int* genre_id) {
    if (i < 0 || i >= (int64_t)g_songs.size()) return 0;
    std::memcpy(out12, g_songs[i].features, sizeof(float) * FEATURE_COUNT);
    *genre_id = g_songs[i].genre_id;
    return 1;
}

// which: 0 track_id, 1 track_name, 2 artists.  Returns length, copies up to cap.
int64_t ref_dm_song_string(int64_t i, int which, char* out, int64_t cap) {
    if (i < 0 || i >= (int64_t)g_songs.size()) return -1;
    const std::string& s = which == 0 ? g_songs[i].track_id
                         : which == 1 ? g_songs[i].track_name
                                      : g_songs[i].artists;
    int64_t n = (int64_t)s.size() < cap ? (int64_t)s.size() : cap;
    std::memcpy(out, s.data(), (size_t)n);
    return (int64_t)s.size();
}

int64_t ref_dm_genre_name(int id, char* out, int64_t cap) {
    auto it = g_genres.find(id);
    if (it == g_genres.end()) return -1;
    int64_t n = (int64_t)it->second.size() < cap ? (int64_t)it->second.size() : cap;
    std::memcpy(out, it->second.data(), (size_t)n);
    return (int64_t)it->second.size();
}

int ref_dm_sizeof_song() { return (int)sizeof(Song); }

}  // extern "C"
