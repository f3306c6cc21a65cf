#include "dsd_reader.h"

#include <dirent.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>

namespace d2dhost {

static std::string lower_ext(const std::string& p) {
    size_t d = p.find_last_of('.');
    if (d == std::string::npos) return "";
    std::string e = p.substr(d + 1);
    for (auto& c : e) c = (char)tolower((unsigned char)c);
    return e;
}

DsdFileFormat format_from_path(const std::string& path) {
    if (path == "-") return DsdFileFormat::Stdin;
    const std::string e = lower_ext(path);
    if (e == "dsf") return DsdFileFormat::Dsf;
    if (e == "dff") return DsdFileFormat::Dff;
    if (e == "dsd") return DsdFileFormat::Raw;
    return DsdFileFormat::Unknown;
}

static uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
static uint64_t be64(const uint8_t* p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
static uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

static uint64_t file_size(FILE* f) {
    long cur = ftell(f);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, cur, SEEK_SET);
    return n < 0 ? 0 : (uint64_t)n;
}

static std::string probe_dsf(FILE* f, DsdInfo& o) {
    uint8_t h[28 + 52 + 12];
    if (fread(h, 1, sizeof(h), f) != sizeof(h)) return "DSF: file too short for its headers";
    if (memcmp(h, "DSD ", 4) || le64(h + 4) != 28) return "DSF: bad 'DSD ' chunk";
    const uint64_t total = le64(h + 12), meta = le64(h + 20);
    const uint8_t* m = h + 28;
    if (memcmp(m, "fmt ", 4) || le64(m + 4) != 52) return "DSF: bad 'fmt ' chunk";
    if (le32(m + 12) != 1) return "DSF: unsupported format version";
    if (le32(m + 16) != 0) return "DSF: unsupported format id (not raw DSD)";
    o.channels = le32(m + 24);
    o.sample_rate = le32(m + 28);
    const uint32_t bps = le32(m + 32);
    o.sample_count = le64(m + 36);
    o.block_size = le32(m + 44);
    if (o.channels < 1 || o.channels > 64) return "DSF: bad channel count";
    if (bps != 1 && bps != 8) return "DSF: bits per sample must be 1 or 8";
    if (o.block_size == 0) return "DSF: zero block size";
    if (o.sample_rate % 2822400u) return "DSF: sampling frequency is not a multiple of 2822400";
    o.dsd_rate = o.sample_rate / 2822400u;
    o.msb_first = bps == 8;
    o.planar = true;
    const uint8_t* dch = h + 80;
    if (memcmp(dch, "data", 4)) return "DSF: 'data' chunk not where expected";
    o.data_offset = 92;
    const uint64_t fsz = file_size(f);
    uint64_t stored = le64(dch + 4) >= 12 ? le64(dch + 4) - 12 : 0;
    if (o.data_offset + stored > fsz) { stored = fsz - o.data_offset; o.warning = "DSF: data chunk longer than the file; using what is there"; }
    o.data_bytes = stored;
    const uint64_t per_ch_stored = stored / o.channels;
    uint64_t real = (o.sample_count + 7) / 8;
    if (real == 0 || real > per_ch_stored) real = per_ch_stored;
    o.bytes_per_channel = real;
    o.metadata_offset = meta;
    if (meta && (meta >= fsz || total > fsz)) {          // id3_test/1kHz_mono_brokenid3.dsf: header total > file size
        o.metadata_truncated = true;
        if (o.warning.empty()) o.warning = "DSF: header size fields exceed the file (damaged tag?); audio is intact";
    }
    return "";
}

static std::string probe_dff(FILE* f, DsdInfo& o) {
    uint8_t h[16];
    if (fread(h, 1, 16, f) != 16) return "DFF: file too short";
    if (memcmp(h, "FRM8", 4) || memcmp(h + 12, "DSD ", 4)) return "DFF: not a FRM8/DSD file";
    const uint64_t fsz = file_size(f);
    uint64_t pos = 16;
    bool have_fs = false, have_ch = false, have_data = false;
    o.planar = false; o.msb_first = true; o.block_size = 1;
    while (pos + 12 <= fsz) {
        uint8_t ch[12];
        fseek(f, (long)pos, SEEK_SET);
        if (fread(ch, 1, 12, f) != 12) break;
        uint64_t sz = be64(ch + 4);
        const uint64_t body = pos + 12;
        // a chunk cannot be longer than what is left of the file: clamp before any arithmetic on it (a size near 2^64
        // would wrap `body + sz` back into the file and walk the same chunks for ever)
        const bool overlong = sz > fsz - body;
        const uint64_t sz_claimed = sz;
        if (overlong) sz = fsz - body;
        if (!memcmp(ch, "PROP", 4)) {
            uint8_t t[4];
            if (fread(t, 1, 4, f) != 4 || memcmp(t, "SND ", 4)) return "DFF: PROP chunk is not SND";
            uint64_t p = body + 4, pend = body + sz;
            while (p + 12 <= pend && p + 12 <= fsz) {
                uint8_t sc[12];
                fseek(f, (long)p, SEEK_SET);
                if (fread(sc, 1, 12, f) != 12) break;
                uint64_t ssz = be64(sc + 4);
                if (ssz > pend - (p + 12)) ssz = pend - (p + 12);      // same clamp for the sub-chunks
                if (!memcmp(sc, "FS  ", 4)) {
                    uint8_t v[4];
                    if (fread(v, 1, 4, f) == 4) { o.sample_rate = be32(v); have_fs = true; }
                } else if (!memcmp(sc, "CHNL", 4)) {
                    uint8_t v[2];
                    if (fread(v, 1, 2, f) == 2) { o.channels = be16(v); have_ch = true; }
                } else if (!memcmp(sc, "CMPR", 4)) {
                    uint8_t v[4];
                    if (fread(v, 1, 4, f) == 4 && memcmp(v, "DSD ", 4)) return "DFF: compressed (DST) audio is not supported";
                }
                p += 12 + ssz + (ssz & 1);
            }
        } else if (!memcmp(ch, "DSD ", 4)) {
            o.data_offset = body;
            const uint64_t stored = sz;
            if (overlong) o.warning = "DFF: data chunk longer than the file; using what is there";
            o.data_bytes = stored;
            have_data = true;
        } else if (!memcmp(ch, "ID3 ", 4)) {
            o.metadata_offset = pos;
            if (overlong) { o.metadata_truncated = true; if (o.warning.empty()) o.warning = "DFF: ID3 chunk is truncated; audio is intact"; }
        } else if (!memcmp(ch, "DST ", 4)) {
            return "DFF: compressed (DST) audio is not supported";
        }
        (void)sz_claimed;
        if (overlong) break;                                   // nothing can follow a chunk that runs past the end
        pos = body + sz + (sz & 1);
    }
    if (!have_fs || !have_ch || !have_data) return "DFF: missing FS, CHNL or DSD chunk";
    if (o.channels < 1 || o.channels > 64) return "DFF: bad channel count";
    if (o.sample_rate % 2822400u) return "DFF: sample rate is not a multiple of 2822400";
    o.dsd_rate = o.sample_rate / 2822400u;
    o.bytes_per_channel = o.data_bytes / o.channels;
    o.sample_count = o.bytes_per_channel * 8;
    return "";
}

std::string probe(const std::string& path, DsdInfo& info) {
    info = DsdInfo{};
    info.format = format_from_path(path);
    if (!is_container(info.format)) return "not a DSF/DFF container: " + path;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    std::string err = info.format == DsdFileFormat::Dsf ? probe_dsf(f, info) : probe_dff(f, info);
    fclose(f);
    if (err.empty() && info.dsd_rate != 1 && info.dsd_rate != 2 && info.dsd_rate != 4 && info.dsd_rate != 8)
        err = "unsupported DSD rate";
    return err;
}

DsdSource::~DsdSource() {
    if (f_ && !is_stdin_) fclose(f_);
}

std::string DsdSource::open(const std::string& path, const DsdInfo& info) {
    info_ = info;
    done_ = 0;
    if (path == "-" || info.format == DsdFileFormat::Stdin) {
        f_ = stdin; is_stdin_ = true;
        info_.bytes_per_channel = 0;
        return "";
    }
    f_ = fopen(path.c_str(), "rb");
    if (!f_) return "cannot open " + path;
    if (info_.format == DsdFileFormat::Raw) {
        info_.data_offset = 0;
        info_.data_bytes = file_size(f_);
        info_.bytes_per_channel = info_.data_bytes / info_.channels;
    }
    fseek(f_, (long)info_.data_offset, SEEK_SET);
    return "";
}

long DsdSource::read(uint8_t* dst, size_t cap) {
    const uint32_t C = info_.channels;
    const uint32_t B = info_.planar ? info_.block_size : 1;
    if (cap == 0) return 0;
    size_t want = cap;
    if (B > 1) { want = cap / B * B; if (want == 0) want = B; }
    if (info_.bytes_per_channel) {
        const uint64_t left = info_.bytes_per_channel - done_;
        if (left == 0) return 0;
        if (want > left) want = (size_t)left;
    }
    if (B <= 1) {   // byte interleaved: the file order is the call layout
        size_t got = fread(dst, 1, want * C, f_);
        got = got / C;
        done_ += got;
        return (long)got;
    }
    // planar: the file stores whole (padded) block groups; deliver [ch0 n][ch1 n].. per group where the
    // stream's last group may be short (the engine's layout rule for a short final block)
    size_t delivered = 0;
    uint8_t* out = dst;
    tmp_.resize((size_t)B * C);
    while (delivered < want) {
        const size_t n = std::min<size_t>(B, want - delivered);          // valid bytes per channel in this group
        size_t got = fread(tmp_.data(), 1, (size_t)B * C, f_);
        if (got < (size_t)B * C) {
            // raw planar input ending mid-group: what is there is split evenly (headerless files have no padding rule)
            if (info_.format != DsdFileFormat::Dsf && got >= C) {
                const size_t m = got / C;
                for (uint32_t c = 0; c < C; ++c) memcpy(out + (size_t)c * m, tmp_.data() + (size_t)c * m, m);
                delivered += m;
            }
            break;
        }
        for (uint32_t c = 0; c < C; ++c) memcpy(out + (size_t)c * n, tmp_.data() + (size_t)c * B, n);
        out += n * C;
        delivered += n;
    }
    done_ += delivered;
    return (long)delivered;
}

static bool is_dir(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

static void walk(const std::string& dir, bool recurse, std::vector<std::string>& out) {
    DIR* d = opendir(dir.c_str());
    if (!d) return;
    std::vector<std::string> names;
    while (dirent* e = readdir(d)) {
        std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        names.push_back(n);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    for (auto& n : names) {
        std::string p = dir + "/" + n;
        if (is_dir(p)) { if (recurse) walk(p, recurse, out); }
        else if (format_from_path(p) != DsdFileFormat::Unknown) out.push_back(p);
    }
}

std::string find_dsd_files(const std::vector<std::string>& paths, bool recurse, std::vector<std::string>& out) {
    for (auto& p : paths) {
        if (is_dir(p)) { if (recurse) walk(p, true, out); }
        else {
            struct stat st;
            if (stat(p.c_str(), &st) != 0) return "No such file: " + p;
            if (format_from_path(p) == DsdFileFormat::Unknown) return "Unsupported input file type: " + p;
            out.push_back(p);
        }
    }
    return "";
}

}  // namespace d2dhost
