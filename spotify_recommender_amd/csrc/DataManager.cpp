// DataManager.cpp — CSV preprocessing + songs_data.bin I/O of the drop-in.
// Behavioural restatement of the reference's DataManager.cpp (cited per
// function); serial on purpose: the reference's OpenMP parse assigns genre ids
// nondeterministically with more than one thread and races on a
// vector<bool>, so the deterministic single-thread result
// is the one reproduced.  Pinned byte-for-byte against the reference's own
// code by tests/test_datamanager.py.
#include "DataManager.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <stdexcept>

namespace {

std::string stripBom(const std::string& s) {  // DataManager.cpp:14-22
    if (s.size() >= 3 && static_cast<unsigned char>(s[0]) == 0xEF &&
        static_cast<unsigned char>(s[1]) == 0xBB && static_cast<unsigned char>(s[2]) == 0xBF)
        return s.substr(3);
    return s;
}

int keyToNumber(const std::string& key) {  // DataManager.cpp:25-43
    std::string k = key;
    for (char& c : k) c = static_cast<char>(::toupper(c));
    static const char* const names[12][2] = {
        {"C", nullptr},  {"C#", "DB"}, {"D", nullptr},  {"D#", "EB"}, {"E", nullptr},  {"F", nullptr},
        {"F#", "GB"},    {"G", nullptr}, {"G#", "AB"},  {"A", nullptr}, {"A#", "BB"},  {"B", nullptr}};
    for (int i = 0; i < 12; ++i)
        for (const char* n : names[i])
            if (n && k == n) return i;
    return -1;
}

int modeToNumber(const std::string& mode) {  // DataManager.cpp:46-54
    std::string m = mode;
    for (char& c : m) c = static_cast<char>(::tolower(c));
    if (m == "major" || m == "1") return 1;
    if (m == "minor" || m == "0") return 0;
    return -1;
}

const char* const kFeatureCols[FEATURE_COUNT - 1] = {  // DataManager.cpp:156-159
    "danceability", "energy", "key", "loudness", "mode", "speechiness",
    "acousticness", "instrumentalness", "liveness", "valence", "tempo"};

}  // namespace

std::string DataManager::trim(const std::string& str) {  // DataManager.cpp:57-62
    const size_t first = str.find_first_not_of(" \t\r\n");
    if (first == std::string::npos) return "";
    const size_t last = str.find_last_not_of(" \t\r\n");
    return str.substr(first, last - first + 1);
}

bool DataManager::isValidNumber(const std::string& str) {  // DataManager.cpp:64-69
    if (str.empty()) return false;
    char* end = nullptr;
    std::strtod(str.c_str(), &end);
    return end != str.c_str() && *end == '\0';
}

// Every '"' toggles the quoted state and is dropped; commas split only outside
// quotes; fields are trimmed (DataManager.cpp:72-92, including its handling of
// doubled quotes).
std::vector<std::string> DataManager::parseCSVLine(const std::string& line) {
    std::vector<std::string> fields;
    std::string cur;
    bool quoted = false;
    for (char c : line) {
        if (c == '"') quoted = !quoted;
        else if (c == ',' && !quoted) {
            fields.push_back(trim(cur));
            cur.clear();
        } else cur += c;
    }
    fields.push_back(trim(cur));
    return fields;
}

bool DataManager::preprocessData(const std::string& csvPath, const std::string& outputPath) {
    std::cout << "Starting data preprocessing from: " << csvPath << std::endl;
    std::ifstream csv(csvPath);
    if (!csv.is_open()) {
        std::cerr << "Error: Could not open CSV file: " << csvPath << std::endl;
        return false;
    }
    std::string headerLine;
    if (!std::getline(csv, headerLine)) {
        std::cerr << "Error: Empty CSV file" << std::endl;
        return false;
    }
    const std::vector<std::string> headers = parseCSVLine(stripBom(headerLine));
    std::map<std::string, int> col;
    for (size_t i = 0; i < headers.size(); ++i) col[headers[i]] = static_cast<int>(i);

    const char* const required[] = {"track_id", "track_name", "artists", "danceability", "energy",
                                    "key", "loudness", "mode", "speechiness", "acousticness",
                                    "instrumentalness", "liveness", "valence", "tempo", "track_genre"};
    for (const char* name : required) {  // DataManager.cpp:121-132
        if (!col.count(name)) {
            std::cerr << "Error: Required column '" << name << "' not found in CSV" << std::endl;
            return false;
        }
    }

    std::vector<std::string> lines;
    for (std::string line; std::getline(csv, line);)
        if (!line.empty()) lines.push_back(line);
    csv.close();
    std::cout << "Read " << lines.size() << " data rows from CSV" << std::endl;
    std::cout << "Parsing and validating songs..." << std::endl;

    const int idCol = col["track_id"], nameCol = col["track_name"], artistCol = col["artists"];
    const int genreCol = col["track_genre"];
    int featCol[FEATURE_COUNT - 1];
    for (int j = 0; j < FEATURE_COUNT - 1; ++j) featCol[j] = col[kFeatureCols[j]];

    std::vector<Song> songs;                 // valid songs, in file order
    std::vector<float> raw;                  // (FEATURE_COUNT-1) raw values per valid song
    std::map<std::string, int> genreToId;    // name -> id (ids by first appearance)
    for (const std::string& line : lines) {  // DataManager.cpp:169-252
        const std::vector<std::string> f = parseCSVLine(line);
        if (f.size() < headers.size()) continue;
        Song song;
        song.track_id = f[idCol];
        song.track_name = f[nameCol];
        song.artists = f[artistCol];
        bool valid = !(song.track_id.empty() || song.track_name.empty());
        float values[FEATURE_COUNT - 1] = {};
        for (int j = 0; j < FEATURE_COUNT - 1 && valid; ++j) {
            const std::string& text = f[featCol[j]];
            const std::string name = kFeatureCols[j];
            int special = -1;
            if (name == "key") special = keyToNumber(text);
            else if (name == "mode") special = modeToNumber(text);
            if (special >= 0) {
                values[j] = static_cast<float>(special);
            } else if (isValidNumber(text)) {
                try {
                    values[j] = std::stof(text);
                } catch (const std::exception&) {  // out-of-range text: the reference would abort
                    valid = false;
                }
            } else {
                valid = false;
            }
        }
        const std::string& genre = f[genreCol];
        if (genre.empty()) valid = false;
        if (!valid) continue;
        auto it = genreToId.find(genre);
        if (it == genreToId.end()) it = genreToId.emplace(genre, static_cast<int>(genreToId.size())).first;
        song.genre_id = it->second;
        songs.push_back(song);
        raw.insert(raw.end(), values, values + FEATURE_COUNT - 1);
    }

    std::cout << "Valid songs: " << songs.size() << " out of " << lines.size() << std::endl;
    std::cout << "Unique genres: " << genreToId.size() << std::endl;
    if (songs.empty()) {
        std::cerr << "Error: No valid songs found in CSV" << std::endl;
        return false;
    }

    float lo[FEATURE_COUNT - 1], hi[FEATURE_COUNT - 1];  // DataManager.cpp:270-280
    for (int j = 0; j < FEATURE_COUNT - 1; ++j) {
        lo[j] = std::numeric_limits<float>::max();
        hi[j] = std::numeric_limits<float>::lowest();
    }
    for (size_t i = 0; i < songs.size(); ++i)
        for (int j = 0; j < FEATURE_COUNT - 1; ++j) {
            lo[j] = std::min(lo[j], raw[i * (FEATURE_COUNT - 1) + j]);
            hi[j] = std::max(hi[j], raw[i * (FEATURE_COUNT - 1) + j]);
        }

    std::cout << "Normalizing features..." << std::endl;
    const int genreDen = std::max(1, static_cast<int>(genreToId.size()) - 1);
    for (size_t i = 0; i < songs.size(); ++i) {  // DataManager.cpp:288-301
        for (int j = 0; j < FEATURE_COUNT - 1; ++j) {
            const float range = hi[j] - lo[j];
            songs[i].features[j] = range > 0.0001f ? (raw[i * (FEATURE_COUNT - 1) + j] - lo[j]) / range : 0.5f;
        }
        songs[i].features[FEATURE_COUNT - 1] = static_cast<float>(songs[i].genre_id) / genreDen;
    }

    std::cout << "Writing binary data to: " << outputPath << std::endl;
    std::ofstream out(outputPath, std::ios::binary);
    if (!out.is_open()) {
        std::cerr << "Error: Could not create output file: " << outputPath << std::endl;
        return false;
    }
    const size_t numSongs = songs.size(), numGenres = genreToId.size();  // DataManager.cpp:322-327
    out.write(reinterpret_cast<const char*>(&numSongs), sizeof numSongs);
    out.write(reinterpret_cast<const char*>(&numGenres), sizeof numGenres);
    for (const auto& g : genreToId) {  // name-sorted (std::map order), DataManager.cpp:329-337
        const int id = g.second;
        const size_t len = g.first.size();
        out.write(reinterpret_cast<const char*>(&id), sizeof id);
        out.write(reinterpret_cast<const char*>(&len), sizeof len);
        out.write(g.first.data(), static_cast<std::streamsize>(len));
    }
    for (const Song& s : songs) s.serialize(out);
    out.close();

    std::cout << "Preprocessing complete! Saved " << numSongs << " songs to binary file." << std::endl;
    std::cout << "\nGenre Mapping:" << std::endl;
    std::vector<std::pair<int, std::string>> byId;
    for (const auto& g : genreToId) byId.push_back({g.second, g.first});
    std::sort(byId.begin(), byId.end());
    for (const auto& g : byId) std::cout << "  ID " << g.first << ": " << g.second << std::endl;
    return true;
}

namespace {

// Header + genre table (DataManager.cpp:375-394).  Lengths are validated: the
// reference trusts them.
bool readHeader(std::ifstream& in, size_t& numSongs, std::map<int, std::string>* genreMap) {
    size_t numGenres = 0;
    in.read(reinterpret_cast<char*>(&numSongs), sizeof numSongs);
    in.read(reinterpret_cast<char*>(&numGenres), sizeof numGenres);
    if (!in || numSongs > (size_t(1) << 32) || numGenres > (size_t(1) << 24)) return false;
    for (size_t i = 0; i < numGenres; ++i) {
        int id = 0;
        size_t len = 0;
        in.read(reinterpret_cast<char*>(&id), sizeof id);
        in.read(reinterpret_cast<char*>(&len), sizeof len);
        if (!in || len > (size_t(1) << 20)) return false;
        std::string name(len, '\0');
        if (len) in.read(&name[0], static_cast<std::streamsize>(len));
        if (genreMap) (*genreMap)[id] = name;
    }
    return static_cast<bool>(in);
}

}  // namespace

bool DataManager::loadData(const std::string& binaryPath, std::vector<Song>& songs,
                           std::map<int, std::string>& genreMap) {
    std::cout << "Loading preprocessed data from: " << binaryPath << std::endl;
    std::ifstream in(binaryPath, std::ios::binary);
    if (!in.is_open()) {
        std::cerr << "Error: Could not open binary file: " << binaryPath << std::endl;
        return false;
    }
    size_t numSongs = 0;
    genreMap.clear();
    if (!readHeader(in, numSongs, &genreMap)) {
        std::cerr << "Error: Corrupt binary file: " << binaryPath << std::endl;
        return false;
    }
    // The header is not trusted (the reference trusts it, DataManager.cpp:375-402):
    // every song occupies at least 3 lengths + genre id + 12 features, so a count
    // the remaining bytes cannot hold is a corrupt file, not a 650 GB resize.
    const std::streampos body = in.tellg();
    in.seekg(0, std::ios::end);
    const std::streampos fileEnd = in.tellg();
    in.seekg(body);
    const size_t minRecord = 3 * sizeof(size_t) + sizeof(int) + FEATURE_COUNT * sizeof(float);
    if (!in || body < 0 || fileEnd < body ||
        numSongs > static_cast<size_t>(fileEnd - body) / minRecord) {
        std::cerr << "Error: Corrupt binary file: " << binaryPath << std::endl;
        return false;
    }
    songs.clear();
    songs.resize(numSongs);
    for (size_t i = 0; i < numSongs; ++i) {
        songs[i].deserialize(in);
        if (!in) {
            std::cerr << "Error: Truncated binary file: " << binaryPath << std::endl;
            songs.clear();
            return false;
        }
    }
    std::cout << "Loaded " << numSongs << " songs and " << genreMap.size() << " genres." << std::endl;
    return true;
}

// Same bytes, one pass over a read-only mapping: no iostream call per field and
// no intermediate vector<Song> (the reference's loadData + initialize copy the
// catalogue three times, DataManager.cpp:396-402 and Recommender.cu:109,162-167).
namespace {

bool walkBinary(const std::string& binaryPath, std::vector<float>& features, std::vector<std::string>& trackIds,
                std::vector<std::string>& trackNames, std::vector<uint64_t>* offsets,
                std::map<int, std::string>* genreMap) {
    const int fd = ::open(binaryPath.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size < 16) {
        ::close(fd);
        return false;
    }
    const size_t size = static_cast<size_t>(st.st_size);
    void* map = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) return false;
    (void)::madvise(map, size, MADV_SEQUENTIAL);
    const unsigned char* const begin = static_cast<const unsigned char*>(map);
    const unsigned char* p = begin;
    const unsigned char* const end = p + size;
    bool ok = true;
    auto take = [&](void* dst, size_t n) {
        if (!ok || static_cast<size_t>(end - p) < n) { ok = false; return; }
        std::memcpy(dst, p, n);
        p += n;
    };
    auto takeString = [&](std::string* dst) {
        size_t len = 0;
        take(&len, sizeof len);
        if (!ok || static_cast<size_t>(end - p) < len) { ok = false; return; }
        if (dst) dst->assign(reinterpret_cast<const char*>(p), len);
        p += len;
    };
    size_t numSongs = 0, numGenres = 0;
    take(&numSongs, sizeof numSongs);
    take(&numGenres, sizeof numGenres);
    if (!ok || numSongs > (size_t(1) << 32) || numGenres > (size_t(1) << 24)) ok = false;
    if (genreMap) genreMap->clear();
    for (size_t g = 0; ok && g < numGenres; ++g) {
        int id = 0;
        take(&id, sizeof id);
        std::string name;
        takeString(genreMap ? &name : nullptr);
        if (ok && genreMap) (*genreMap)[id] = name;
    }
    if (ok) {
        // every song needs at least 3 lengths + genre + features
        if (numSongs > size / (3 * sizeof(size_t) + sizeof(int) + FEATURE_COUNT * sizeof(float))) ok = false;
    }
    if (ok) {
        features.resize(numSongs * FEATURE_COUNT);
        trackIds.resize(numSongs);
        trackNames.resize(numSongs);
        if (offsets) offsets->resize(numSongs);
    }
    for (size_t i = 0; ok && i < numSongs; ++i) {
        if (offsets) (*offsets)[i] = static_cast<uint64_t>(p - begin);
        takeString(&trackIds[i]);
        takeString(&trackNames[i]);
        takeString(nullptr);  // artists
        int genre = 0;
        take(&genre, sizeof genre);
        take(&features[i * FEATURE_COUNT], FEATURE_COUNT * sizeof(float));
    }
    ::munmap(map, size);
    if (!ok) {
        features.clear();
        trackIds.clear();
        trackNames.clear();
        if (offsets) offsets->clear();
        if (genreMap) genreMap->clear();
    }
    return ok;
}

}  // namespace

bool DataManager::loadFeatureMatrix(const std::string& binaryPath, std::vector<float>& features,
                                    std::vector<std::string>& trackIds,
                                    std::vector<std::string>& trackNames) {
    return walkBinary(binaryPath, features, trackIds, trackNames, nullptr, nullptr);
}

bool DataManager::loadCatalogue(const std::string& binaryPath, Catalogue& out) {
    std::cout << "Loading preprocessed data from: " << binaryPath << std::endl;
    out.path = binaryPath;
    out.reader = std::make_shared<Catalogue::Reader>();   // opened by the first readSong
    if (!walkBinary(binaryPath, out.features, out.trackIds, out.trackNames, &out.recordOffsets, &out.genreMap)) {
        std::cerr << "Error: Could not read binary file: " << binaryPath << std::endl;
        return false;
    }
    std::cout << "Loaded " << out.size() << " songs and " << out.genreMap.size() << " genres." << std::endl;
    return true;
}

bool DataManager::readSong(const Catalogue& catalogue, size_t index, Song& out) {
    if (index >= catalogue.recordOffsets.size()) return false;
    if (!catalogue.reader) {   // a catalogue that did not come from loadCatalogue: a stream of this call's own
        std::ifstream in(catalogue.path, std::ios::binary);
        if (!in.is_open()) return false;
        in.seekg(static_cast<std::streamoff>(catalogue.recordOffsets[index]));
        out.deserialize(in);
        return static_cast<bool>(in);
    }
    Catalogue::Reader& r = *catalogue.reader;
    std::lock_guard<std::mutex> hold(r.lock);   // position and state of the shared stream belong to one caller at a time
    if (!r.in.is_open()) r.in.open(catalogue.path, std::ios::binary);
    if (!r.in.is_open()) return false;
    r.in.clear();
    r.in.seekg(static_cast<std::streamoff>(catalogue.recordOffsets[index]));
    out.deserialize(r.in);
    return static_cast<bool>(r.in);
}
