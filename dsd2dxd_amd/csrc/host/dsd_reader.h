// dsd_reader.h -- DSD inputs for the host driver: DSF and DFF containers, raw .dsd files, stdin.
//
// Counterpart of the reference's dsd-reader / dsf-meta / dff-meta crates (Cargo.lock:233-249,267-274;
// sources absent from the reference checkout).  What a container fixes overrides the command-line
// layout flags (README.md:103-105): block size, channel count, DSD rate, bit order, planar vs
// interleaved.  Layouts were measured on the reference's fixtures:
//   DSF  : 'DSD ' chunk (28 B) + 'fmt ' chunk (52 B) + 'data' chunk; planar blocks of
//          block_size bytes per channel; bits/sample 1 = LSB-first, 8 = MSB-first; the last block
//          group is padded, `sample_count` says how many bits per channel are real
//   DFF  : big-endian IFF ('FRM8' ... 'DSD '): FVER, PROP/SND (FS, CHNL, CMPR), 'DSD ' data chunk,
//          byte-interleaved, MSB-first; chunks padded to even length; ID3 may follow (and may be
//          truncated: id3_test/dff/1kHz_stereo_i_brokenid3.dff)
//   .dsd : headerless; layout from the caller's flags
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

namespace d2dhost {

enum class DsdFileFormat { Dsf, Dff, Raw, Stdin, Unknown };

struct DsdInfo {
    DsdFileFormat format = DsdFileFormat::Unknown;
    uint32_t channels = 2;
    uint32_t dsd_rate = 1;            // multiple of 2.8224 MHz
    uint32_t sample_rate = 2822400;
    bool planar = false;
    bool msb_first = true;
    uint32_t block_size = 4096;       // bytes per channel per block (planar)
    uint64_t bytes_per_channel = 0;   // real payload per channel (0 = unknown, stdin)
    uint64_t sample_count = 0;        // bits per channel (DSF)
    uint64_t data_offset = 0;         // file offset of the payload
    uint64_t data_bytes = 0;          // payload bytes as stored (incl. padding)
    uint64_t metadata_offset = 0;     // DSF ID3 pointer / DFF 'ID3 ' chunk offset, 0 = none
    bool metadata_truncated = false;  // tag claims more bytes than the file holds
    std::string warning;
};

DsdFileFormat format_from_path(const std::string& path);     // by extension, like DsdFileFormat::from(&path)
inline bool is_container(DsdFileFormat f) { return f == DsdFileFormat::Dsf || f == DsdFileFormat::Dff; }

// Parse a container header.  Returns "" or an error message.
std::string probe(const std::string& path, DsdInfo& info);

class DsdSource {
   public:
    ~DsdSource();
    // `info` for raw/stdin inputs comes from the caller; for containers it is filled by probe()
    std::string open(const std::string& path, const DsdInfo& info);
    // Next chunk in the engine's call layout: up to `cap` bytes per channel (for planar inputs a whole
    // number of blocks, except the stream's last call which may end with one short block).
    // Returns bytes per channel delivered, 0 at end, -1 on error.
    long read(uint8_t* dst, size_t cap_bytes_per_channel);
    const DsdInfo& info() const { return info_; }

   private:
    FILE* f_ = nullptr;
    bool is_stdin_ = false;
    DsdInfo info_;
    uint64_t done_ = 0;               // bytes per channel delivered so far
    std::vector<uint8_t> tmp_;
};

// files under the given paths ending in .dsf/.dff/.dsd (case-insensitive); directories are walked
// only with `recurse` (README.md:97-101), like rdsd2pcm::find_dsd_files (src/main.rs:275)
std::string find_dsd_files(const std::vector<std::string>& paths, bool recurse, std::vector<std::string>& out);

}  // namespace d2dhost
