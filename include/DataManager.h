// DataManager.h — CSV preprocessing and songs_data.bin I/O of the drop-in.
// Public surface = the reference's DataManager.h:16-37.  Not part of the
// accelerated path (one-time string / file work); kept so that main.cpp's three
// modes work and the file format stays byte-compatible.
#ifndef DATAMANAGER_H
#define DATAMANAGER_H

#include <cstdint>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "Song.h"

class DataManager {
public:
    // CSV -> validated, min-max normalised songs -> binary file
    // (reference DataManager.cpp:94-361).  Genre ids are assigned in order of
    // first appearance among valid rows, which is what the reference produces
    // with OMP_NUM_THREADS=1 (with more threads its ids are nondeterministic,
    // SURVEY.md App. B9).
    static bool preprocessData(const std::string& csvPath, const std::string& outputPath);

    // Binary file -> songs + genre id->name map (reference DataManager.cpp:363-409).
    static bool loadData(const std::string& binaryPath, std::vector<Song>& songs,
                         std::map<int, std::string>& genreMap);

    // Extension used by the engine-side loaders: the same file straight into a
    // row-major n x 12 float matrix plus ids / names (no vector<Song> copy).
    static bool loadFeatureMatrix(const std::string& binaryPath, std::vector<float>& features,
                                  std::vector<std::string>& trackIds,
                                  std::vector<std::string>& trackNames);

    // What the query modes of the CLI need, in ONE pass over a read-only mapping
    // (replaces loadData + Recommender::initialize's copies, reference
    // DataManager.cpp:363-409 and Recommender.cu:109,162-167): the feature matrix the
    // engine uploads as is, ids and names for the lookups, the genre map, and the file
    // offset of every record so that the few songs that get PRINTED can be read back
    // in full (readSong) without ever materialising vector<Song>.
    struct Catalogue {
        std::string path;
        std::vector<float> features;            // row-major n x FEATURE_COUNT
        std::vector<std::string> trackIds;
        std::vector<std::string> trackNames;
        std::vector<uint64_t> recordOffsets;    // byte offset of song i in the file
        std::map<int, std::string> genreMap;
        // readSong's file handle: created by loadCatalogue (shared by copies of the catalogue), opened on first use
        // and kept (one open per catalogue, not per printed song); the mutex makes readSong safe to call from
        // several threads, on one catalogue or on copies of it
        struct Reader {
            std::mutex lock;
            std::ifstream in;
        };
        std::shared_ptr<Reader> reader;
        size_t size() const { return trackIds.size(); }
    };
    static bool loadCatalogue(const std::string& binaryPath, Catalogue& out);
    static bool readSong(const Catalogue& catalogue, size_t index, Song& out);

private:
    static std::vector<std::string> parseCSVLine(const std::string& line);
    static std::string trim(const std::string& str);
    static bool isValidNumber(const std::string& str);
};

#endif  // DATAMANAGER_H
