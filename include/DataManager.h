// DataManager.h — CSV preprocessing and songs_data.bin I/O of the drop-in.
// Public surface = the reference's DataManager.h:16-37.  Not part of the
// accelerated path (one-time string / file work); kept so that main.cpp's three
// modes work and the file format stays byte-compatible.
#ifndef DATAMANAGER_H
#define DATAMANAGER_H

#include <map>
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

private:
    static std::vector<std::string> parseCSVLine(const std::string& line);
    static std::string trim(const std::string& str);
    static bool isValidNumber(const std::string& str);
};

#endif  // DATAMANAGER_H
