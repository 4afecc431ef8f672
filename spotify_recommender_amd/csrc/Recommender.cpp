// Recommender.cpp — host shim: the reference's Recommender class on top of the
// C-ABI (include/mi355rec.h).  Replaces Recommender.cu:80-372; every method
// names the reference lines whose observable behaviour it keeps.
#include "Recommender.h"

#include <algorithm>
#include <cctype>
#include <iostream>
#include <unordered_map>

#include "mi355rec_diag.h"   // (the core + mi355rec_sharded_note / _placement for the messages)

struct Recommender::Impl {
    bool initialized = false;
    bool gpuEnabled = false;
    int numSongs = 0;
    mi355rec_sharded_t* engine = nullptr;   // the catalogue on the node's GPUs (placement: include/mi355rec.h) — or, without any, on the CPU backend
    int numDevices = 0;

    // Only what the lookups need is kept (the reference deep-copies every Song,
    // Recommender.cu:109).  `byId` maps a track id to its FIRST row, which is
    // what the reference's linear scan returns; `lowerNames` is lowered once
    // instead of once per row per query (Recommender.cu:340-351).
    std::vector<std::string> lowerNames;
    std::unordered_map<std::string, int> byId;

    std::vector<int64_t> idxBuf;
    std::vector<float> scoreBuf;
    std::vector<float> lastScores;
};

namespace {

std::string toLower(const std::string& str) {  // Recommender.cu:329-334
    std::string result = str;
    for (char& c : result) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return result;
}

}  // namespace

Recommender::Recommender() : impl_(new Impl()) {}

Recommender::~Recommender() {  // Recommender.cu:86-98
    if (impl_->engine) mi355rec_sharded_destroy(impl_->engine);
    delete impl_;
}

namespace {

// Shared tail of the two initialize() overloads: lookup tables + upload.
bool startEngine(Recommender::Impl* impl, const float* matrix, size_t n);

}  // namespace

bool Recommender::initialize(const std::vector<Song>& songs) {  // Recommender.cu:100-182
    std::cout << "Initializing GPU-accelerated recommender..." << std::endl;
    if (songs.empty()) {  // :103-106
        std::cerr << "Error: Empty song database" << std::endl;
        return false;
    }
    // AoS -> row-major N x 12 (the reference's staging matrix, :162-167)
    std::vector<float> matrix(songs.size() * FEATURE_COUNT);
    impl_->lowerNames.clear();
    impl_->lowerNames.reserve(songs.size());
    impl_->byId.clear();
    impl_->byId.reserve(songs.size() * 2);
    for (size_t i = 0; i < songs.size(); ++i) {
        std::copy(songs[i].features, songs[i].features + FEATURE_COUNT, matrix.begin() + i * FEATURE_COUNT);
        impl_->lowerNames.push_back(toLower(songs[i].track_name));
        impl_->byId.emplace(songs[i].track_id, static_cast<int>(i));  // keeps the first
    }
    return startEngine(impl_, matrix.data(), songs.size());
}

bool Recommender::initialize(const std::vector<float>& features, const std::vector<std::string>& trackIds,
                             const std::vector<std::string>& trackNames) {
    std::cout << "Initializing GPU-accelerated recommender..." << std::endl;
    if (trackIds.empty()) {
        std::cerr << "Error: Empty song database" << std::endl;
        return false;
    }
    if (features.size() != trackIds.size() * FEATURE_COUNT || trackNames.size() != trackIds.size()) {
        std::cerr << "Error: feature matrix / id / name sizes disagree" << std::endl;
        return false;
    }
    impl_->lowerNames.clear();
    impl_->lowerNames.reserve(trackIds.size());
    impl_->byId.clear();
    impl_->byId.reserve(trackIds.size() * 2);
    for (size_t i = 0; i < trackIds.size(); ++i) {
        impl_->lowerNames.push_back(toLower(trackNames[i]));
        impl_->byId.emplace(trackIds[i], static_cast<int>(i));
    }
    return startEngine(impl_, features.data(), trackIds.size());
}

namespace {

bool startEngine(Recommender::Impl* impl, const float* matrix, size_t n) {
    if (impl->engine) {
        mi355rec_sharded_destroy(impl->engine);
        impl->engine = nullptr;
    }
    impl->initialized = false;
    impl->gpuEnabled = false;
    impl->numSongs = static_cast<int>(n);
    // The reference pins device 0 (Recommender.cu:124).  Here the library places the catalogue itself: one device up
    // to 7.9 M rows, row-sharded over as many as keep 4 M rows per shard beyond that (one process, one stream per
    // device, xGMI peer stores or one RCCL all-gather of the per-shard top-N keys: include/mi355rec.h, "PLACEMENT").
    // Without any HIP device the handle is served by the product's own CPU backend, as the reference falls back to
    // its CPU loop (:117-127,176-181).
    const int rc = mi355rec_create_placed(matrix, static_cast<int64_t>(n), FEATURE_COUNT, nullptr, /*n_devices=*/0,
                                          MI355REC_PLACEMENT_AUTO, &impl->engine);
    if (rc != MI355REC_OK) {   // a device is there but could not be used (or memory ran out): say why and give up
        std::cerr << "[GPU Disabled] " << mi355rec_sharded_last_error(nullptr) << std::endl;
        std::cerr << "Error: the recommender could not be initialized" << std::endl;
        return false;
    }
    impl->initialized = true;
    if (mi355rec_sharded_placement(impl->engine) == MI355REC_PLACEMENT_CPU) {
        // the reference's own lines for this case (Recommender.cu:120-121,177)
        std::cerr << "[GPU Disabled] HIP runtime not available: no HIP device visible" << std::endl;
        std::cerr << "Falling back to CPU similarity computation." << std::endl;
        std::cout << "Operating in CPU fallback mode (cosine similarity on CPU)." << std::endl;
        impl->gpuEnabled = false;
        impl->numDevices = 0;
        return true;
    }
    mi355rec_sharded_info(impl->engine, &impl->numDevices, nullptr, nullptr, nullptr, nullptr);
    impl->gpuEnabled = true;
    // the reference's stdout line, byte for byte (Recommender.cu:172); the placement note goes to stderr
    std::cout << "Successfully initialized with " << impl->numSongs << " songs on GPU" << std::endl;
    if (impl->numDevices > 1) {
        std::cerr << "[mi355rec] catalogue row-sharded over " << impl->numDevices << " devices" << std::endl;
        const char* note = mi355rec_sharded_note(impl->engine);
        if (note && note[0]) std::cerr << "[mi355rec] " << note << std::endl;
    }
    return true;
}

}  // namespace

std::vector<int> Recommender::recommendByIndex(int songIndex, int topN) {  // Recommender.cu:275-318
    if (!impl_->initialized) {
        std::cerr << "Error: Recommender not initialized" << std::endl;
        return {};
    }
    if (songIndex < 0 || songIndex >= impl_->numSongs) {
        std::cerr << "Error: Invalid song index: " << songIndex << std::endl;
        return {};
    }
    if (topN <= 0) {
        std::cerr << "Error: topN must be positive" << std::endl;
        return {};
    }
    // The reference's heap never holds more than N-1 entries (Recommender.cu:296-305):
    // a larger topN returns N-1 results, and must not size any buffer.
    if (topN > impl_->numSongs - 1) topN = impl_->numSongs - 1;
    if (topN == 0) return {};  // a one-song catalogue has nothing to recommend
    impl_->idxBuf.assign(static_cast<size_t>(topN), -1);
    impl_->scoreBuf.assign(static_cast<size_t>(topN), 0.0f);
    int count = 0;
    const int rc = mi355rec_sharded_query_row_topn(impl_->engine, songIndex, topN, impl_->idxBuf.data(),
                                                   impl_->scoreBuf.data(), &count);
    if (rc != MI355REC_OK) {
        std::cerr << "Error: " << mi355rec_sharded_last_error(impl_->engine) << std::endl;
        return {};
    }
    std::vector<int> results(static_cast<size_t>(count));
    for (int i = 0; i < count; ++i) results[i] = static_cast<int>(impl_->idxBuf[i]);
    impl_->lastScores.assign(impl_->scoreBuf.begin(), impl_->scoreBuf.begin() + count);
    return results;
}

std::vector<int> Recommender::recommend(const std::string& trackId, int topN) {  // :356-363
    const auto it = impl_->byId.find(trackId);
    if (it == impl_->byId.end()) {
        std::cerr << "Error: Song with track_id '" << trackId << "' not found" << std::endl;
        return {};
    }
    return recommendByIndex(it->second, topN);
}

std::vector<int> Recommender::recommendByName(const std::string& trackName, int topN) {  // :365-372
    const std::string needle = toLower(trackName);
    int index = -1;
    for (size_t i = 0; i < impl_->lowerNames.size(); ++i) {  // exact pass, :340-344
        if (impl_->lowerNames[i] == needle) {
            index = static_cast<int>(i);
            break;
        }
    }
    if (index < 0) {
        for (size_t i = 0; i < impl_->lowerNames.size(); ++i) {  // substring pass, :347-351
            if (impl_->lowerNames[i].find(needle) != std::string::npos) {
                index = static_cast<int>(i);
                break;
            }
        }
    }
    if (index < 0) {
        std::cerr << "Error: Song with name '" << trackName << "' not found" << std::endl;
        return {};
    }
    return recommendByIndex(index, topN);
}

bool Recommender::isInitialized() const { return impl_->initialized; }
bool Recommender::isGPUEnabled() const { return impl_->gpuEnabled; }
int Recommender::getSongCount() const { return impl_->numSongs; }

const std::vector<float>& Recommender::lastScores() const { return impl_->lastScores; }

bool Recommender::similarities(int songIndex, std::vector<float>& out) {  // Recommender.cu:184-254
    if (!impl_->initialized || songIndex < 0 || songIndex >= impl_->numSongs) {
        std::cerr << "Error: Invalid query index or recommender not initialized" << std::endl;
        return false;
    }
    out.resize(static_cast<size_t>(impl_->numSongs));
    return mi355rec_sharded_scores_row(impl_->engine, songIndex, out.data()) == MI355REC_OK;
}
