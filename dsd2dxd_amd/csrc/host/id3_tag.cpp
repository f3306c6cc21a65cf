#include "id3_tag.h"

#include <dirent.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>

namespace d2dhost {

namespace {

uint32_t syncsafe(const uint8_t* p) { return ((uint32_t)(p[0] & 0x7F) << 21) | ((uint32_t)(p[1] & 0x7F) << 14) | ((uint32_t)(p[2] & 0x7F) << 7) | (p[3] & 0x7F); }
void put_syncsafe(uint8_t* p, uint32_t v) { p[0] = (v >> 21) & 0x7F; p[1] = (v >> 14) & 0x7F; p[2] = (v >> 7) & 0x7F; p[3] = v & 0x7F; }
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = (uint8_t)v; }

struct Frame {
    std::string id;                  // 4 characters (3 in ID3v2.2)
    uint8_t flags[2] = {0, 0};
    size_t body = 0, size = 0;       // payload position inside the tag, payload length
};

// Walks the frames of a padding-free or padded tag.  Returns false when the structure does not parse.
bool walk_frames(const std::vector<uint8_t>& tag, std::vector<Frame>& frames, size_t& frames_end) {
    frames.clear();
    if (tag.size() < 10 || memcmp(tag.data(), "ID3", 3)) return false;
    const uint8_t ver = tag[3], hflags = tag[5];
    if (ver < 2 || ver > 4) return false;
    size_t pos = 10;
    const size_t end = std::min(tag.size(), (size_t)10 + syncsafe(tag.data() + 6));
    if (hflags & 0x40) {             // extended header: skip it
        if (pos + 4 > end) return false;
        const uint32_t xs = ver == 4 ? syncsafe(tag.data() + pos) : be32(tag.data() + pos) + 4;
        if (xs < 4 || pos + xs > end) return false;
        pos += xs;
    }
    const size_t hl = ver == 2 ? 6 : 10;
    while (pos + hl <= end && tag[pos] != 0) {
        Frame f;
        const uint8_t* h = tag.data() + pos;
        if (ver == 2) { f.id.assign((const char*)h, 3); f.size = ((size_t)h[3] << 16) | ((size_t)h[4] << 8) | h[5]; }
        else {
            f.id.assign((const char*)h, 4);
            f.size = ver == 4 ? syncsafe(h + 4) : be32(h + 4);
            f.flags[0] = h[8]; f.flags[1] = h[9];
        }
        for (char c : f.id) if (!((c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'))) return false;
        f.body = pos + hl;
        if (f.body + f.size > end) return false;
        frames.push_back(f);
        pos = f.body + f.size;
    }
    frames_end = pos;
    return true;
}

void utf8_put(std::string& s, uint32_t cp) {
    if (cp < 0x80) s += (char)cp;
    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
}

// Decodes ONE string in the given ID3 text encoding starting at p (stops at its terminator or at n);
// *used = bytes consumed including the terminator.
std::string decode_text(uint8_t enc, const uint8_t* p, size_t n, size_t* used = nullptr) {
    std::string out;
    size_t i = 0;
    if (enc == 0 || enc == 3) {
        while (i < n && p[i]) { if (enc == 0) utf8_put(out, p[i]); else out += (char)p[i]; ++i; }
        if (i < n) ++i;
    } else {
        bool be = enc == 2;
        if (enc == 1 && n >= 2) {
            if (p[0] == 0xFE && p[1] == 0xFF) { be = true; i = 2; }
            else if (p[0] == 0xFF && p[1] == 0xFE) { be = false; i = 2; }
        }
        while (i + 1 < n) {
            uint32_t u = be ? ((uint32_t)p[i] << 8) | p[i + 1] : ((uint32_t)p[i + 1] << 8) | p[i];
            i += 2;
            if (u == 0) break;
            if (u >= 0xD800 && u < 0xDC00 && i + 1 < n) {
                const uint32_t lo = be ? ((uint32_t)p[i] << 8) | p[i + 1] : ((uint32_t)p[i + 1] << 8) | p[i];
                if (lo >= 0xDC00 && lo < 0xE000) { u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00); i += 2; }
            }
            utf8_put(out, u);
        }
    }
    if (used) *used = i;
    return out;
}

bool has_image_ext(const std::string& name) {
    const size_t d = name.find_last_of('.');
    if (d == std::string::npos) return false;
    std::string e = name.substr(d + 1);
    for (char& c : e) c = (char)tolower((unsigned char)c);
    static const char* exts[] = {"jpg", "jpeg", "png", "gif", "bmp", "tif", "tiff", "webp"};
    for (const char* x : exts) if (e == x) return true;
    return false;
}

}  // namespace

std::string read_source_tag(const std::string& path, const DsdInfo& info, std::vector<uint8_t>& tag, std::string& warning) {
    tag.clear();
    if (!info.metadata_offset || !is_container(info.format)) return "";
    if (info.metadata_truncated) { warning = "the source's ID3 tag is damaged (it claims more bytes than the file holds); not copied"; return ""; }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    const uint64_t at = info.metadata_offset + (info.format == DsdFileFormat::Dff ? 12 : 0);
    uint8_t h[10];
    std::string err;
    if (fseek(f, (long)at, SEEK_SET) != 0 || fread(h, 1, 10, f) != 10 || memcmp(h, "ID3", 3)) {
        warning = "no ID3v2 tag where the container points; nothing copied";
    } else {
        const uint32_t sz = syncsafe(h + 6);
        const size_t total = 10 + (size_t)sz + ((h[5] & 0x10) ? 10 : 0);      // + footer (v2.4)
        tag.resize(total);
        memcpy(tag.data(), h, 10);
        if (fread(tag.data() + 10, 1, total - 10, f) != total - 10) {
            tag.clear();
            warning = "the source's ID3 tag is damaged (it claims more bytes than the file holds); not copied";
        } else {
            // drop the padding when the tag is one this code can re-frame; otherwise keep it verbatim
            std::vector<Frame> frames; size_t fend = 0;
            const bool plain = !(h[5] & 0xD0);                                  // no unsync / ext header / footer
            if (plain && walk_frames(tag, frames, fend) && fend >= 10) {
                tag.resize(fend);
                put_syncsafe(tag.data() + 6, (uint32_t)(fend - 10));
            }
        }
    }
    fclose(f);
    return err;
}

std::string album_rate_suffix(uint32_t rate) {
    char b[32];
    if (rate % 1000 == 0) snprintf(b, sizeof(b), " [%uK]", rate / 1000);
    else snprintf(b, sizeof(b), " [%u.%uK]", rate / 1000, (rate % 1000) / 100);
    return b;
}

bool append_to_album(std::vector<uint8_t>& tag, const std::string& suffix) {
    std::vector<Frame> frames; size_t fend = 0;
    if (tag.size() < 10 || (tag[5] & 0xD0) || !walk_frames(tag, frames, fend)) return false;
    const uint8_t ver = tag[3];
    for (const Frame& f : frames) {
        if (f.id != (ver == 2 ? "TAL" : "TALB")) continue;
        if (f.size < 1) return false;
        if (ver >= 3 && (f.flags[1] & (ver == 3 ? 0xC0 : 0x0F))) return false;   // compressed / encrypted / unsync / length
        const uint8_t enc = tag[f.body];
        // the text without its trailing terminator(s)
        size_t tl = f.size - 1;
        const uint8_t* t = tag.data() + f.body + 1;
        if (enc == 1 || enc == 2) { while (tl >= 2 && t[tl - 1] == 0 && t[tl - 2] == 0) tl -= 2; }
        else while (tl >= 1 && t[tl - 1] == 0) tl -= 1;
        std::vector<uint8_t> add;
        if (enc == 1 || enc == 2) {
            bool be = enc == 2;
            if (enc == 1 && tl >= 2 && t[0] == 0xFE && t[1] == 0xFF) be = true;
            for (unsigned char c : suffix) { if (be) { add.push_back(0); add.push_back(c); } else { add.push_back(c); add.push_back(0); } }
        } else add.assign(suffix.begin(), suffix.end());
        const size_t new_size = 1 + tl + add.size();
        std::vector<uint8_t> out(tag.begin(), tag.begin() + f.body + 1 + tl);
        out.insert(out.end(), add.begin(), add.end());
        out.insert(out.end(), tag.begin() + f.body + f.size, tag.end());
        uint8_t* fh = out.data() + f.body - (ver == 2 ? 6 : 10);
        if (ver == 2) { fh[3] = (uint8_t)(new_size >> 16); fh[4] = (uint8_t)(new_size >> 8); fh[5] = (uint8_t)new_size; }
        else if (ver == 4) put_syncsafe(fh + 4, (uint32_t)new_size);
        else put_be32(fh + 4, (uint32_t)new_size);
        put_syncsafe(out.data() + 6, (uint32_t)(out.size() - 10));
        tag.swap(out);
        return true;
    }
    return false;
}

void tag_to_vorbis(const std::vector<uint8_t>& tag, std::vector<std::pair<std::string, std::string>>& fields,
                   std::vector<TagPicture>& pictures) {
    fields.clear(); pictures.clear();
    std::vector<Frame> frames; size_t fend = 0;
    if (tag.size() < 10 || (tag[5] & 0x80) || !walk_frames(tag, frames, fend)) return;
    const uint8_t ver = tag[3];
    static const struct { const char* v3; const char* v2; const char* field; } map[] = {
        {"TIT2", "TT2", "TITLE"}, {"TPE1", "TP1", "ARTIST"}, {"TALB", "TAL", "ALBUM"}, {"TPE2", "TP2", "ALBUMARTIST"},
        {"TCOM", "TCM", "COMPOSER"}, {"TCON", "TCO", "GENRE"}, {"TDRC", "", "DATE"}, {"TYER", "TYE", "DATE"},
        {"TRCK", "TRK", "TRACKNUMBER"}, {"TPOS", "TPA", "DISCNUMBER"}, {"TPE3", "TP3", "CONDUCTOR"}, {"TCOP", "TCR", "COPYRIGHT"},
        {"TSRC", "TRC", "ISRC"}, {"TPUB", "TPB", "ORGANIZATION"},
    };
    for (const Frame& f : frames) {
        if (ver >= 3 && (f.flags[1] & (ver == 3 ? 0xC0 : 0x0F))) continue;
        const uint8_t* b = tag.data() + f.body;
        if (f.id[0] == 'T' && f.id != "TXXX" && f.id != "TXX" && f.size >= 1) {
            for (const auto& m : map) {
                if (f.id != (ver == 2 ? m.v2 : m.v3)) continue;
                std::string val = decode_text(b[0], b + 1, f.size - 1);
                if (val.empty()) break;
                const std::string field = m.field;
                const size_t slash = val.find('/');
                if ((field == "TRACKNUMBER" || field == "DISCNUMBER") && slash != std::string::npos) {
                    fields.push_back({field, val.substr(0, slash)});
                    fields.push_back({field == "TRACKNUMBER" ? "TRACKTOTAL" : "DISCTOTAL", val.substr(slash + 1)});
                } else fields.push_back({field, val});
                break;
            }
        } else if ((f.id == "COMM" || f.id == "COM") && f.size >= 5) {
            size_t used = 0;
            decode_text(b[0], b + 4, f.size - 4, &used);                       // short description
            std::string val = decode_text(b[0], b + 4 + used, f.size - 4 - used);
            if (!val.empty()) fields.push_back({"COMMENT", val});
        } else if (f.id == "APIC" && f.size >= 4) {
            TagPicture p;
            size_t i = 1, used = 0;
            p.mime = decode_text(0, b + i, f.size - i, &used); i += used;
            if (i >= f.size) continue;
            p.type = b[i++];
            p.description = decode_text(b[0], b + i, f.size - i, &used); i += used;
            if (i >= f.size) continue;
            p.data.assign(b + i, b + f.size);
            if (p.mime.find('/') == std::string::npos) p.mime = "image/" + p.mime;
            pictures.push_back(std::move(p));
        } else if (f.id == "PIC" && f.size >= 6) {                              // ID3v2.2: 3-character image format
            TagPicture p;
            std::string fmt((const char*)b + 1, 3);
            for (char& c : fmt) c = (char)tolower((unsigned char)c);
            p.mime = fmt == "jpg" ? "image/jpeg" : "image/" + fmt;
            p.type = b[4];
            size_t used = 0;
            p.description = decode_text(b[0], b + 5, f.size - 5, &used);
            if (5 + used >= f.size) continue;
            p.data.assign(b + 5 + used, b + f.size);
            pictures.push_back(std::move(p));
        }
    }
}

int copy_artwork(const std::string& from_dir, const std::string& to_dir) {
    if (from_dir == to_dir) return 0;
    DIR* d = opendir(from_dir.c_str());
    if (!d) return 0;
    int copied = 0;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (!has_image_ext(name)) continue;
        const std::string src = from_dir + "/" + name, dst = to_dir + "/" + name;
        struct stat ss, ds;
        if (stat(src.c_str(), &ss) != 0 || !S_ISREG(ss.st_mode)) continue;
        if (stat(dst.c_str(), &ds) == 0 && ds.st_size == ss.st_size) continue;
        FILE* in = fopen(src.c_str(), "rb");
        if (!in) continue;
        FILE* out = fopen(dst.c_str(), "wb");
        if (!out) { fclose(in); continue; }
        std::vector<uint8_t> buf(1 << 16);
        size_t n;
        bool ok = true;
        while ((n = fread(buf.data(), 1, buf.size(), in)) > 0) if (fwrite(buf.data(), 1, n, out) != n) { ok = false; break; }
        fclose(in);
        if (fclose(out) != 0) ok = false;
        if (ok) ++copied;
    }
    closedir(d);
    return copied;
}

}  // namespace d2dhost
