// pcm_sink.h -- PCM outputs of the host driver: raw stdout, WAV, AIFF, AIFC, FLAC.
// Counterpart of the reference's writers (`-o S|W|A|C|F`, src/main.rs:98-105,206-214; README.md:6-8).
// The engine hands interleaved little-endian frames (16 / 24-bit container / f32); each sink adapts.
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

namespace d2dhost {

enum class OutputType { Stdout, Aiff, Aifc, Wav, Flac };

class PcmSink {
   public:
    virtual ~PcmSink() {}
    virtual std::string write(const uint8_t* frames, size_t bytes) = 0;   // "" or error
    virtual std::string close() = 0;
};

// bit_depth 16/20/24 (integer, 20 in a 24-bit container) or 32 (float).  `id3` = the source's ID3v2
// tag to carry over (may be null or empty): WAV 'id3 ' chunk, AIFF/AIFC 'ID3 ' chunk, FLAC
// VORBIS_COMMENT + PICTURE blocks (id3_tag.h); the raw stdout stream has no place for it.
std::string open_sink(OutputType type, const std::string& path, uint32_t channels, uint32_t rate, uint32_t bit_depth,
                      PcmSink** out, const std::vector<uint8_t>* id3 = nullptr);
const char* output_extension(OutputType t);

}  // namespace d2dhost
