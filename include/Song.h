// Song.h — row record + songs_data.bin record format of the drop-in.
//
// Same public surface as the reference's Song.h (Song.h:12 FEATURE_COUNT,
// :21-32 struct Song, :35-54 serialize, :57-77 deserialize) so code written
// against the reference compiles unchanged.  The byte format is kept exactly:
// per song   u64 len + bytes  (track_id, track_name, artists — in that order),
//            i32 genre_id, 12 x f32 features; native little-endian, no padding.
// Pinned by tests/test_datamanager.py against files written by the reference's
// own DataManager (tests/golden/sample_songs_data.bin).
#ifndef SONG_H
#define SONG_H

#include <cstddef>
#include <cstdint>
#include <istream>
#include <ostream>
#include <string>
#include <vector>

// Audio features per track (danceability, energy, key, loudness, mode,
// speechiness, acousticness, instrumentalness, liveness, valence, tempo,
// genre_id) — the K of the N x K catalogue matrix.
const int FEATURE_COUNT = 12;

struct Song {
    std::string track_id;
    std::string track_name;
    std::string artists;
    int genre_id;
    float features[FEATURE_COUNT];  // min-max normalised to [0, 1]

    Song() : genre_id(-1) {
        for (float& f : features) f = 0.0f;
    }

    void serialize(std::ostream& out) const {
        putString(out, track_id);
        putString(out, track_name);
        putString(out, artists);
        out.write(reinterpret_cast<const char*>(&genre_id), sizeof genre_id);
        out.write(reinterpret_cast<const char*>(features), sizeof features);
    }

    void deserialize(std::istream& in) {
        getString(in, track_id);
        getString(in, track_name);
        getString(in, artists);
        in.read(reinterpret_cast<char*>(&genre_id), sizeof genre_id);
        in.read(reinterpret_cast<char*>(features), sizeof features);
    }

private:
    static void putString(std::ostream& out, const std::string& s) {
        const size_t len = s.size();  // 8 bytes on the reference's x86-64 target
        out.write(reinterpret_cast<const char*>(&len), sizeof len);
        out.write(s.data(), static_cast<std::streamsize>(len));
    }

    static void getString(std::istream& in, std::string& s) {
        size_t len = 0;
        in.read(reinterpret_cast<char*>(&len), sizeof len);
        // The reference trusts the length; a corrupt file
        // must not make us allocate terabytes.
        if (!in || len > (size_t(1) << 30)) {
            in.setstate(std::ios::failbit);
            s.clear();
            return;
        }
        s.resize(len);
        if (len) in.read(&s[0], static_cast<std::streamsize>(len));
    }
};

static_assert(sizeof(size_t) == 8, "songs_data.bin stores 8-byte lengths");

#endif  // SONG_H
