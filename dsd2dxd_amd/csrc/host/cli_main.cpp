// dsd2dxd_amd -- a small driver over include/rdsd2pcm.hpp with the reference CLI's flags
// (/root/reference/src/main.rs:40-133) for the conversion-relevant subset, plus:
//   probe  <file>...      print what the container readers see, as JSON (no GPU needed)
//   levels [opts] files   peak dBFS per file and overall (src/bin/dsd_levels/main.rs)
//   tags [-a RATE] <file>...   the source's ID3v2 tag as the converted file would carry it, as JSON
//                         (text frames under their Vorbis names; -a applies the album suffix; no GPU needed)
// It is not the reference's CLI (logging, progress bars, Rayon pool are out of scope).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <cmath>
#include <string>
#include <vector>

#include "../../../include/rdsd2pcm.hpp"
#include "dsd_reader.h"
#include "id3_tag.h"

using namespace rdsd2pcm;
static std::atomic<bool> CANCEL_FLAG{false};

static int probe_main(int argc, char** argv) {
    printf("[");
    for (int i = 0; i < argc; ++i) {
        d2dhost::DsdInfo o;
        std::string err = d2dhost::probe(argv[i], o);
        printf("%s{\"path\": \"%s\", \"error\": \"%s\", \"format\": \"%s\", \"channels\": %u, \"dsd_rate\": %u, \"planar\": %s, "
               "\"msb_first\": %s, \"block_size\": %u, \"bytes_per_channel\": %llu, \"sample_count\": %llu, \"data_offset\": %llu, "
               "\"data_bytes\": %llu, \"metadata_offset\": %llu, \"metadata_truncated\": %s, \"warning\": \"%s\"}",
               i ? ", " : "", argv[i], err.c_str(), o.format == d2dhost::DsdFileFormat::Dsf ? "dsf" : o.format == d2dhost::DsdFileFormat::Dff ? "dff" : "other",
               o.channels, o.dsd_rate, o.planar ? "true" : "false", o.msb_first ? "true" : "false", o.block_size,
               (unsigned long long)o.bytes_per_channel, (unsigned long long)o.sample_count, (unsigned long long)o.data_offset,
               (unsigned long long)o.data_bytes, (unsigned long long)o.metadata_offset, o.metadata_truncated ? "true" : "false", o.warning.c_str());
    }
    printf("]\n");
    return 0;
}

static std::string json_escape(const std::string& s) {
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); o += b; }
        else o += (char)c;
    }
    return o;
}

static int tags_main(int argc, char** argv) {
    uint32_t append_rate = 0;
    int i = 0;
    if (argc >= 2 && !strcmp(argv[0], "-a")) { append_rate = (uint32_t)strtoul(argv[1], 0, 10); i = 2; }
    printf("[");
    for (bool first = true; i < argc; ++i, first = false) {
        d2dhost::DsdInfo o;
        std::string err = d2dhost::probe(argv[i], o), warn;
        std::vector<uint8_t> tag;
        if (err.empty()) err = d2dhost::read_source_tag(argv[i], o, tag, warn);
        bool album_edited = false;
        if (!tag.empty() && append_rate) album_edited = d2dhost::append_to_album(tag, d2dhost::album_rate_suffix(append_rate));
        std::vector<std::pair<std::string, std::string>> fields;
        std::vector<d2dhost::TagPicture> pics;
        d2dhost::tag_to_vorbis(tag, fields, pics);
        printf("%s{\"path\": \"%s\", \"error\": \"%s\", \"warning\": \"%s\", \"tag_bytes\": %zu, \"album_edited\": %s, \"pictures\": %zu, \"fields\": {",
               first ? "" : ", ", json_escape(argv[i]).c_str(), json_escape(err).c_str(), json_escape(warn).c_str(), tag.size(),
               album_edited ? "true" : "false", pics.size());
        for (size_t k = 0; k < fields.size(); ++k)
            printf("%s\"%s\": \"%s\"", k ? ", " : "", json_escape(fields[k].first).c_str(), json_escape(fields[k].second).c_str());
        printf("}}");
    }
    printf("]\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "probe")) return probe_main(argc - 2, argv + 2);
    if (argc >= 2 && !strcmp(argv[1], "tags")) return tags_main(argc - 2, argv + 2);
    bool levels = false;
    int ai = 1;
    if (argc >= 2 && !strcmp(argv[1], "levels")) { levels = true; ai = 2; }
    else if (argc >= 2 && !strcmp(argv[1], "convert")) ai = 2;
    // defaults of the reference CLI (src/main.rs:50-110)
    std::string out_dir; bool have_out_dir = false;
    size_t channels = 2, bit_depth = 24; char fmt = 'I', filt = 'E', endian = 'M', dither = 0, output = 'S';
    uint32_t block = 4096, rate = 352800, inrate = 1; double level = 0.0; bool append = false, recurse = false, quiet = false;
    int device = 0; unsigned long long seed = 0; uint32_t tap_bits = 0;
    std::vector<std::string> files;
    for (; ai < argc; ++ai) {
        std::string a = argv[ai];
        auto val = [&]() -> const char* { if (ai + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++ai]; };
        if (a == "-p" || a == "--path") { out_dir = val(); have_out_dir = true; }
        else if (a == "-c" || a == "--channels") channels = strtoul(val(), 0, 10);
        else if (a == "-f" || a == "--fmt") fmt = val()[0];
        else if (a == "-b" || a == "--bitdepth") bit_depth = strtoul(val(), 0, 10);
        else if (a == "-t" || a == "--filttype") filt = val()[0];
        else if (a == "-e" || a == "--endianness") endian = val()[0];
        else if (a == "-s" || a == "--bs") block = (uint32_t)strtoul(val(), 0, 10);
        else if (a == "-d" || a == "--dither") dither = val()[0];
        else if (a == "-r" || a == "--rate") rate = (uint32_t)strtoul(val(), 0, 10);
        else if (a == "-i" || a == "--inrate") inrate = (uint32_t)strtoul(val(), 0, 10);
        else if (a == "-o" || a == "--output") output = val()[0];
        else if (a == "-l" || a == "--level") level = atof(val());
        else if (a.rfind("--level=", 0) == 0) level = atof(a.c_str() + 8);
        else if (a == "-a" || a == "--append") append = true;
        else if (a == "-R" || a == "--recurse") recurse = true;
        else if (a == "-q" || a == "--quiet") quiet = true;
        else if (a == "-v" || a == "--verbose") {}
        else if (a == "--device") device = atoi(val());
        else if (a == "--seed") seed = strtoull(val(), 0, 10);
        else if (a == "--tap-bits") tap_bits = (uint32_t)strtoul(val(), 0, 10);
        else files.push_back(a);
    }
    if (files.empty()) files.push_back("-");
    try {
        // char -> enum exactly as src/main.rs:165-214 (unknown endianness/filter/output fall back, unknown dither/format fail)
        if (!dither) dither = bit_depth == 32 ? 'F' : 'T';
        DitherType dt;
        switch (tolower(dither)) { case 't': dt = DitherType::TPDF; break; case 'r': dt = DitherType::Rectangular; break;
            case 'f': dt = DitherType::FPD; break; case 'x': dt = DitherType::None; break;
            case 'n': dt = DitherType::NoiseShaped; break;                 // extension: noise-shaped TPDF
            default: throw std::runtime_error("Invalid dither type; must be T, R, F, or X"); }
        FmtType ft;
        switch (tolower(fmt)) { case 'i': ft = FmtType::Interleaved; break; case 'p': ft = FmtType::Planar; break;
            default: throw std::runtime_error("Invalid format; must be I (interleaved) or P (planar)"); }
        Endianness en = tolower(endian) == 'l' ? Endianness::LsbFirst : Endianness::MsbFirst;
        FilterType fl = toupper(filt) == 'X' ? FilterType::XLD : toupper(filt) == 'D' ? FilterType::Dsd2Pcm : toupper(filt) == 'C' ? FilterType::Chebyshev : FilterType::Equiripple;
        OutputType ot = tolower(output) == 'a' ? OutputType::Aiff : tolower(output) == 'c' ? OutputType::Aifc : tolower(output) == 'w' ? OutputType::Wav : tolower(output) == 'f' ? OutputType::Flac : OutputType::Stdout;
        bool has_stdin = false;
        std::vector<std::string> paths;
        for (auto& f : files) { if (f == "-") has_stdin = true; else paths.push_back(f); }
        std::vector<std::string> expanded = find_dsd_files(paths, recurse);
        if (has_stdin) expanded.insert(expanded.begin(), "-");
        float overall = -INFINITY;
        for (auto& path : expanded) {
            const bool is_stdin = path == "-";
            std::optional<std::string> od; if (have_out_dir) od = out_dir;
            std::optional<std::string> ip; if (!is_stdin) ip = path;
            if (levels) {
                Rdsd2Pcm lib = Rdsd2Pcm::new_level_check(rate, ip, ft, en, channels, block, inrate);
                lib.set_device(device);
                float db = lib.check_level(CANCEL_FLAG);
                printf("%s: %.4f dBFS\n", lib.file_name().c_str(), db);
                if (!std::isnan(db) && db > overall) overall = db;          // dsd_levels/main.rs:184-202 skips NaN
                continue;
            }
            Rdsd2Pcm lib = (!is_stdin && DsdFileFormat::from(path).is_container())
                               ? Rdsd2Pcm::from_container(bit_depth, ot, level, rate, od, dt, fl, append, ".", path)
                               : Rdsd2Pcm::create(bit_depth, ot, level, rate, od, dt, ft, en, inrate, block, channels, fl, append, ".", ip);
            lib.set_device(device); lib.set_seed(seed); lib.set_tap_bits(tap_bits);
            lib.do_conversion(CANCEL_FLAG);
            if (!quiet && !lib.warnings().empty()) fprintf(stderr, "WARNING: %s: %s\n", lib.file_name().c_str(), lib.warnings().c_str());
            if (!quiet && lib.audio_seconds() > 0)
                fprintf(stderr, "DSP speed for %s: %.2fx\n", lib.file_name().c_str(), lib.audio_seconds() / std::max(lib.dsp_seconds(), 1e-9));
        }
        if (levels) printf("Highest peak: %.4f dBFS\n", overall);
    } catch (const std::exception& ex) {
        fprintf(stderr, "ERROR: %s\n", ex.what());
        return 1;
    }
    return 0;
}
