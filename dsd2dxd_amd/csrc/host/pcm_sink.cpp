#include "pcm_sink.h"

#include "id3_tag.h"

#include <math.h>

#include <algorithm>
#include <thread>
#include <string.h>

namespace d2dhost {

const char* output_extension(OutputType t) {
    switch (t) {
        case OutputType::Aiff: return "aif";
        case OutputType::Aifc: return "aifc";
        case OutputType::Wav: return "wav";
        case OutputType::Flac: return "flac";
        default: return "pcm";
    }
}

static void put_le32(uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
static void put_le16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put_be32(uint8_t* p, uint32_t v) { p[3] = v; p[2] = v >> 8; p[1] = v >> 16; p[0] = v >> 24; }
static void put_be16(uint8_t* p, uint16_t v) { p[1] = (uint8_t)v; p[0] = (uint8_t)(v >> 8); }

namespace {

struct RawSink : PcmSink {
    FILE* f;
    explicit RawSink(FILE* fp) : f(fp) {}
    std::string write(const uint8_t* p, size_t n) override { return fwrite(p, 1, n, f) == n ? "" : "short write"; }
    std::string close() override { fflush(f); return ""; }
};

struct WavSink : PcmSink {
    FILE* f; uint32_t ch, rate, bits; uint64_t data = 0; size_t hdr = 0;
    std::vector<uint8_t> id3;
    std::string begin() {
        const uint32_t cont = bits == 16 ? 16 : (bits == 32 ? 32 : 24);
        const bool ext = bits == 20 || ch > 2;                      // valid-bits / channel mask need EXTENSIBLE
        uint8_t h[68]; memset(h, 0, sizeof(h));
        memcpy(h, "RIFF", 4); memcpy(h + 8, "WAVE", 4); memcpy(h + 12, "fmt ", 4);
        const uint32_t fmtsz = ext ? 40 : 16;
        put_le32(h + 16, fmtsz);
        put_le16(h + 20, ext ? 0xFFFE : (bits == 32 ? 3 : 1));
        put_le16(h + 22, (uint16_t)ch); put_le32(h + 24, rate);
        put_le32(h + 28, rate * ch * cont / 8); put_le16(h + 32, (uint16_t)(ch * cont / 8)); put_le16(h + 34, (uint16_t)cont);
        size_t o = 36;
        if (ext) {
            put_le16(h + 36, 22); put_le16(h + 38, (uint16_t)(bits == 32 ? 32 : bits)); put_le32(h + 40, 0);
            static const uint8_t guid_tail[14] = {0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};
            put_le16(h + 44, bits == 32 ? 3 : 1); memcpy(h + 46, guid_tail, 14);
            o = 60;
        }
        memcpy(h + o, "data", 4); o += 8;
        hdr = o;
        return fwrite(h, 1, o, f) == o ? "" : "short write";
    }
    std::string write(const uint8_t* p, size_t n) override {
        // RIFF sizes are 32-bit: refuse at the first write that cannot be described, not at close() after hours of output
        if (data + n + hdr + 8 + id3.size() + 2 > 0xFFFFFFFFull) return "WAV output exceeds 4 GiB (RIFF size fields are 32-bit); choose FLAC, a lower rate or depth";
        data += n; return fwrite(p, 1, n, f) == n ? "" : "short write";
    }
    std::string close() override {
        if (data & 1) fputc(0, f);
        uint64_t tail = 0;
        if (!id3.empty()) {                                          // 'id3 ' chunk behind the audio
            uint8_t c[8]; memcpy(c, "id3 ", 4); put_le32(c + 4, (uint32_t)id3.size());
            fwrite(c, 1, 8, f); fwrite(id3.data(), 1, id3.size(), f);
            if (id3.size() & 1) fputc(0, f);
            tail = 8 + id3.size() + (id3.size() & 1);
        }
        if (data + hdr + tail > 0xFFFFFFFFull) { fclose(f); return "WAV output exceeds 4 GiB"; }
        uint8_t v[4];
        put_le32(v, (uint32_t)(hdr - 8 + data + (data & 1) + tail)); fseek(f, 4, SEEK_SET); fwrite(v, 1, 4, f);
        put_le32(v, (uint32_t)data); fseek(f, (long)hdr - 4, SEEK_SET); fwrite(v, 1, 4, f);
        return fclose(f) == 0 ? "" : "close failed";
    }
};

// 80-bit IEEE extended, big-endian (AIFF sample rate)
static void put_ext80(uint8_t* p, double v) {
    memset(p, 0, 10);
    if (v <= 0) return;
    int e; double m = frexp(v, &e);                   // v = m * 2^e, 0.5 <= m < 1
    uint64_t mant = (uint64_t)ldexp(m, 64);
    uint16_t ex = (uint16_t)(e - 1 + 16383);
    put_be16(p, ex);
    for (int i = 0; i < 8; ++i) p[2 + i] = (uint8_t)(mant >> (56 - 8 * i));
}

struct AiffSink : PcmSink {
    FILE* f; uint32_t ch, rate, bits; bool aifc; uint64_t data = 0; size_t comm_frames_at = 0, ssnd_size_at = 0;
    std::vector<uint8_t> tmp, id3;
    std::string begin() {
        const uint32_t cont = bits == 16 ? 16 : (bits == 32 ? 32 : 24);
        std::vector<uint8_t> h;
        auto app = [&](const void* s, size_t n) { h.insert(h.end(), (const uint8_t*)s, (const uint8_t*)s + n); };
        uint8_t b[16];
        app("FORM\0\0\0\0", 8); app(aifc ? "AIFC" : "AIFF", 4);
        if (aifc) { app("FVER", 4); put_be32(b, 4); app(b, 4); put_be32(b, 0xA2805140u); app(b, 4); }
        app("COMM", 4);
        const char* ctype = bits == 32 ? "fl32" : "NONE";
        const char* cname = bits == 32 ? "\x0CIEEE 32-bit\0" : "\x0Enot compressed\0";      // pascal strings, even total
        const size_t cname_len = bits == 32 ? 14 : 16;
        put_be32(b, (uint32_t)(18 + (aifc ? 4 + cname_len : 0))); app(b, 4);
        put_be16(b, (uint16_t)ch); app(b, 2);
        comm_frames_at = h.size(); put_be32(b, 0); app(b, 4);
        put_be16(b, (uint16_t)(bits == 20 ? 20 : cont)); app(b, 2);
        uint8_t e80[10]; put_ext80(e80, (double)rate); app(e80, 10);
        if (aifc) { app(ctype, 4); app(cname, cname_len); }
        app("SSND", 4); ssnd_size_at = h.size(); put_be32(b, 0); app(b, 4); put_be32(b, 0); app(b, 4); put_be32(b, 0); app(b, 4);
        return fwrite(h.data(), 1, h.size(), f) == h.size() ? "" : "short write";
    }
    std::string write(const uint8_t* p, size_t n) override {           // little-endian samples -> big-endian
        const size_t sb = bits == 16 ? 2 : (bits == 32 ? 4 : 3);
        if (data + n + 128 + id3.size() > 0xFFFFFFFFull) return "AIFF output exceeds 4 GiB (FORM size fields are 32-bit); choose FLAC, a lower rate or depth";
        tmp.resize(n);
        for (size_t i = 0; i + sb <= n; i += sb)
            for (size_t k = 0; k < sb; ++k) tmp[i + k] = p[i + sb - 1 - k];
        data += n;
        return fwrite(tmp.data(), 1, n, f) == n ? "" : "short write";
    }
    std::string close() override {
        if (data & 1) fputc(0, f);
        if (!id3.empty()) {                                          // 'ID3 ' chunk behind the audio
            uint8_t c[8]; memcpy(c, "ID3 ", 4); put_be32(c + 4, (uint32_t)id3.size());
            fwrite(c, 1, 8, f); fwrite(id3.data(), 1, id3.size(), f);
            if (id3.size() & 1) fputc(0, f);
        }
        const size_t sb = bits == 16 ? 2 : (bits == 32 ? 4 : 3);
        long end = ftell(f);
        if ((uint64_t)end > 0xFFFFFFFFull) { fclose(f); return "AIFF output exceeds 4 GiB"; }
        uint8_t v[4];
        put_be32(v, (uint32_t)(end - 8)); fseek(f, 4, SEEK_SET); fwrite(v, 1, 4, f);
        put_be32(v, (uint32_t)(data / (sb * ch))); fseek(f, (long)comm_frames_at, SEEK_SET); fwrite(v, 1, 4, f);
        put_be32(v, (uint32_t)(data + 8)); fseek(f, (long)ssnd_size_at, SEEK_SET); fwrite(v, 1, 4, f);
        return fclose(f) == 0 ? "" : "close failed";
    }
};

// MD5 (RFC 1321) of the unencoded audio for STREAMINFO
struct Md5 {
    uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
    uint64_t len = 0; uint8_t buf[64]; size_t fill = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
            0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
            0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
            0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
            0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
            0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = p[4 * i] | (p[4 * i + 1] << 8) | (p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B; B = B + rol(A + f + K[i] + m[g], S[i]); A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            const size_t take = std::min(n, 64 - fill);
            memcpy(buf + fill, p, take); fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8]; for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
        update(l, 8);
        const uint32_t v[4] = {a, b, c, d};
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint8_t)(v[i] >> (8 * k));
    }
};

// FLAC with fixed-order-2 prediction and one Rice partition per subframe (valid, modest compression;
// the reference uses the flac-codec crate, Cargo.lock:299-307).  Integer depths only.  Frames are
// independent of each other, so a batch of them is encoded on as many threads and written in order.
struct FlacFrameEnc {
    std::vector<uint8_t> out;
    std::vector<int32_t> res;
    uint64_t bitacc = 0; int bitn = 0;
    static const uint8_t* crc8_table() {
        static uint8_t t[256]; static bool done = false;
        if (!done) { for (int i = 0; i < 256; ++i) { uint8_t c = (uint8_t)i; for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1)); t[i] = c; } done = true; }
        return t;
    }
    static const uint16_t* crc16_table() {
        static uint16_t t[256]; static bool done = false;
        if (!done) { for (int i = 0; i < 256; ++i) { uint16_t c = (uint16_t)(i << 8); for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1)); t[i] = c; } done = true; }
        return t;
    }
    static uint8_t crc8(const uint8_t* p, size_t n) { const uint8_t* t = crc8_table(); uint8_t c = 0; for (size_t i = 0; i < n; ++i) c = t[c ^ p[i]]; return c; }
    static uint16_t crc16(const uint8_t* p, size_t n) { const uint16_t* t = crc16_table(); uint16_t c = 0; for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ t[(c >> 8) ^ p[i]]); return c; }
    void bits_put(uint64_t v, int n) {
        while (n > 0) {
            int take = n > 32 ? 32 : n;
            uint64_t part = (v >> (n - take)) & ((1ull << take) - 1);
            bitacc = (bitacc << take) | part; bitn += take; n -= take;
            while (bitn >= 8) { out.push_back((uint8_t)(bitacc >> (bitn - 8))); bitn -= 8; }
        }
    }
    void bits_flush() { if (bitn) { out.push_back((uint8_t)(bitacc << (8 - bitn))); bitn = 0; } bitacc = 0; }
    void put_utf8(uint32_t v) {
        if (v < 0x80) { bits_put(v, 8); return; }
        int n = v < 0x800 ? 1 : v < 0x10000 ? 2 : v < 0x200000 ? 3 : v < 0x4000000 ? 4 : 5;
        bits_put(((0xFF00 >> (n + 1)) & 0xFF) | (v >> (6 * n)), 8);
        for (int i = n - 1; i >= 0; --i) bits_put(0x80 | ((v >> (6 * i)) & 0x3F), 8);
    }
    // one frame of `n` interleaved samples starting at `buf`
    void encode(const int32_t* buf, uint32_t n, uint32_t ch, uint32_t depth, uint32_t frame_no, uint32_t BS) {
        out.clear(); bitn = 0; bitacc = 0;
        bits_put(0xFFF8, 16);                                          // sync, fixed block size
        bits_put(n == BS ? 0xC : 0x7, 4);                              // 4096, or 16-bit (n-1) at the end of the header
        bits_put(0x0, 4);                                              // sample rate from STREAMINFO
        bits_put(ch - 1, 4);                                           // independent channels
        bits_put(depth == 16 ? 4 : depth == 20 ? 5 : 6, 3); bits_put(0, 1);
        put_utf8(frame_no);
        if (n != BS) bits_put(n - 1, 16);
        bits_flush();
        out.push_back(crc8(out.data(), out.size()));
        res.resize(n);
        for (uint32_t c = 0; c < ch; ++c) {
            auto s = [&](uint32_t i) -> int64_t { return buf[(size_t)i * ch + c]; };
            const uint32_t order = n > 2 ? 2 : 0;
            uint64_t sum = 0;
            bool fits = true;
            for (uint32_t i = order; i < n; ++i) {
                int64_t r = order == 2 ? s(i) - 2 * s(i - 1) + s(i - 2) : s(i);
                if (r > 0x3FFFFFFF || r < -0x3FFFFFFF) fits = false;
                res[i] = (int32_t)r; sum += (uint64_t)(r < 0 ? -r : r);
            }
            if (!fits || order == 0) {                                   // verbatim subframe
                bits_put(0, 1); bits_put(1, 6); bits_put(0, 1);
                for (uint32_t i = 0; i < n; ++i) bits_put((uint64_t)s(i) & ((1ull << depth) - 1), (int)depth);
                continue;
            }
            bits_put(0, 1); bits_put(8 + order, 6); bits_put(0, 1);    // fixed predictor, no wasted bits
            for (uint32_t i = 0; i < order; ++i) bits_put((uint64_t)s(i) & ((1ull << depth) - 1), (int)depth);
            uint32_t k = 0;
            const uint64_t mean = sum / (n - order ? n - order : 1);
            while (k < 30 && (1ull << k) < mean) ++k;
            bits_put(1, 2);                                             // residual coding method 1: 5-bit Rice parameters
            bits_put(0, 4);                                             // partition order 0
            bits_put(k, 5);
            for (uint32_t i = order; i < n; ++i) {
                uint32_t u = res[i] >= 0 ? (uint32_t)res[i] << 1 : (((uint32_t)(-(int64_t)res[i])) << 1) - 1;
                uint32_t q = u >> k;
                while (q >= 32) { bits_put(0, 32); q -= 32; }
                bits_put(1, (int)q + 1);
                if (k) bits_put(u & ((1u << k) - 1), (int)k);
            }
        }
        bits_flush();
        uint16_t c16 = crc16(out.data(), out.size());
        out.push_back((uint8_t)(c16 >> 8)); out.push_back((uint8_t)c16);
    }
};

struct FlacSink : PcmSink {
    FILE* f; uint32_t ch, rate, bits; uint64_t frames = 0; uint32_t frame_no = 0;
    static constexpr uint32_t BS = 4096;
    std::vector<int32_t> buf;      // interleaved pending samples
    uint32_t min_fs = 0xFFFFFF, max_fs = 0;
    std::vector<uint8_t> id3;
    Md5 md5;                       // of the samples as little-endian whole-byte integers, interleaved (FLAC format, STREAMINFO)
    std::vector<FlacFrameEnc> encs;
    bool io_failed = false;
    uint32_t depth() const { return bits == 20 ? 20 : bits; }
    std::string begin() {
        // metadata: STREAMINFO (filled in at close), then the source's tag as VORBIS_COMMENT and PICTURE
        // blocks; the last block carries the 0x80 flag
        std::vector<std::pair<std::string, std::string>> fields;
        std::vector<TagPicture> pics;
        tag_to_vorbis(id3, fields, pics);
        std::vector<std::vector<uint8_t>> blocks;
        if (!fields.empty()) {
            std::vector<uint8_t> b;
            auto le32 = [&](uint32_t v) { uint8_t t[4]; put_le32(t, v); b.insert(b.end(), t, t + 4); };
            const std::string vendor = "dsd2dxd_amd";
            le32((uint32_t)vendor.size()); b.insert(b.end(), vendor.begin(), vendor.end());
            le32((uint32_t)fields.size());
            for (const auto& kv : fields) {
                const std::string e = kv.first + "=" + kv.second;
                le32((uint32_t)e.size()); b.insert(b.end(), e.begin(), e.end());
            }
            b.insert(b.begin(), {4, 0, 0, 0});
            blocks.push_back(std::move(b));
        }
        for (const TagPicture& p : pics) {
            std::vector<uint8_t> b = {6, 0, 0, 0};
            auto be32v = [&](uint32_t v) { uint8_t t[4]; put_be32(t, v); b.insert(b.end(), t, t + 4); };
            be32v(p.type);
            be32v((uint32_t)p.mime.size()); b.insert(b.end(), p.mime.begin(), p.mime.end());
            be32v((uint32_t)p.description.size()); b.insert(b.end(), p.description.begin(), p.description.end());
            be32v(0); be32v(0); be32v(0); be32v(0);                   // width, height, depth, colours: not parsed
            be32v((uint32_t)p.data.size()); b.insert(b.end(), p.data.begin(), p.data.end());
            if (b.size() - 4 < (1u << 24)) blocks.push_back(std::move(b));
        }
        for (auto& b : blocks) { const uint32_t n = (uint32_t)b.size() - 4; b[1] = (uint8_t)(n >> 16); b[2] = (uint8_t)(n >> 8); b[3] = (uint8_t)n; }
        if (!blocks.empty()) blocks.back()[0] |= 0x80;
        uint8_t h[4 + 4 + 34]; memset(h, 0, sizeof(h));
        memcpy(h, "fLaC", 4); h[4] = blocks.empty() ? 0x80 : 0x00; h[7] = 34;   // STREAMINFO
        if (fwrite(h, 1, sizeof(h), f) != sizeof(h)) return "short write";
        for (const auto& b : blocks) if (fwrite(b.data(), 1, b.size(), f) != b.size()) return "short write";
        return "";
    }
    // encodes the pending whole frames (and, at the end, the short last one) in batches, one thread per
    // frame of a batch, and writes them in order
    void drain(bool final) {
        if (encs.empty()) {
            (void)FlacFrameEnc::crc8_table(); (void)FlacFrameEnc::crc16_table();        // built before any thread runs
            unsigned hw = std::thread::hardware_concurrency();
            encs.resize(std::max(1u, std::min(16u, hw ? hw : 1u)));
        }
        const size_t per = (size_t)BS * ch;
        const size_t nfull = buf.size() / per, rem = buf.size() - nfull * per;
        std::vector<std::pair<size_t, uint32_t>> jobs;                   // (first sample, frames) of each FLAC frame
        for (size_t i = 0; i < nfull; ++i) jobs.push_back({i * per, BS});
        if (final && rem) jobs.push_back({nfull * per, (uint32_t)(rem / ch)});
        for (size_t j0 = 0; j0 < jobs.size(); j0 += encs.size()) {
            const size_t nb = std::min(encs.size(), jobs.size() - j0);
            std::vector<std::thread> th;
            for (size_t j = 1; j < nb; ++j)
                th.emplace_back([&, j] { encs[j].encode(buf.data() + jobs[j0 + j].first, jobs[j0 + j].second, ch, depth(), frame_no + (uint32_t)j, BS); });
            encs[0].encode(buf.data() + jobs[j0].first, jobs[j0].second, ch, depth(), frame_no, BS);
            for (auto& t : th) t.join();
            for (size_t j = 0; j < nb; ++j) {
                const std::vector<uint8_t>& o = encs[j].out;
                if (fwrite(o.data(), 1, o.size(), f) != o.size()) io_failed = true;
                if (o.size() < min_fs) min_fs = (uint32_t)o.size();
                if (o.size() > max_fs) max_fs = (uint32_t)o.size();
                frames += jobs[j0 + j].second;
            }
            frame_no += (uint32_t)nb;
        }
        buf.erase(buf.begin(), buf.begin() + (final ? buf.size() : nfull * per));
    }
    std::string write(const uint8_t* p, size_t nbytes) override {
        const size_t sb = bits == 16 ? 2 : 3;
        buf.reserve(buf.size() + nbytes / sb);
        for (size_t i = 0; i + sb <= nbytes; i += sb) {
            int32_t v = sb == 2 ? (int16_t)(p[i] | (p[i + 1] << 8)) : (int32_t)((p[i] | (p[i + 1] << 8) | (p[i + 2] << 16)) << 8) >> 8;
            if (bits == 20) v >>= 4;
            buf.push_back(v);
        }
        if (bits == 20) {                                                // MD5 runs over the 20-bit values as 3-byte integers
            for (size_t i = buf.size() - nbytes / sb; i < buf.size(); ++i) { const int32_t v = buf[i]; const uint8_t le[3] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16)}; md5.update(le, 3); }
        } else md5.update(p, nbytes / sb * sb);                          // 16/24-bit: the little-endian input as it is
        if (buf.size() >= (size_t)BS * ch * 8) drain(false);
        return io_failed ? "short write" : "";
    }
    std::string close() override {
        drain(true);
        uint8_t si[34]; memset(si, 0, sizeof(si));
        put_be16(si, BS); put_be16(si + 2, BS);
        si[4] = min_fs >> 16; si[5] = min_fs >> 8; si[6] = (uint8_t)min_fs; si[7] = max_fs >> 16; si[8] = max_fs >> 8; si[9] = (uint8_t)max_fs;
        const uint64_t v = ((uint64_t)rate << 44) | ((uint64_t)(ch - 1) << 41) | ((uint64_t)(depth() - 1) << 36) | (frames & 0xFFFFFFFFFull);
        for (int i = 0; i < 8; ++i) si[10 + i] = (uint8_t)(v >> (56 - 8 * i));
        md5.finish(si + 18);
        fseek(f, 8, SEEK_SET); fwrite(si, 1, 34, f);
        if (fclose(f) != 0 || io_failed) return "close failed";
        return "";
    }
};

}  // namespace

std::string open_sink(OutputType type, const std::string& path, uint32_t channels, uint32_t rate, uint32_t bit_depth, PcmSink** out,
                      const std::vector<uint8_t>* id3) {
    *out = nullptr;
    if (type == OutputType::Stdout) { *out = new RawSink(stdout); return ""; }
    if (type == OutputType::Flac && bit_depth == 32) return "FLAC cannot hold 32-bit float; choose 16, 20 or 24 bits";
    if (type == OutputType::Flac && rate > 655350) return "FLAC cannot describe this sample rate";
    if (type == OutputType::Flac && channels > 8) return "FLAC holds at most 8 channels";   // 4-bit channel assignment: 8..10 mean stereo decorrelation
    if (type == OutputType::Aiff && bit_depth == 32) return "AIFF cannot hold 32-bit float; use AIFC (C)";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return "cannot create " + path;
    std::string err;
    if (type == OutputType::Wav) { auto* s = new WavSink(); if (id3) s->id3 = *id3; s->f = f; s->ch = channels; s->rate = rate; s->bits = bit_depth; err = s->begin(); *out = s; }
    else if (type == OutputType::Flac) { auto* s = new FlacSink(); if (id3) s->id3 = *id3; s->f = f; s->ch = channels; s->rate = rate; s->bits = bit_depth; err = s->begin(); *out = s; }
    else { auto* s = new AiffSink(); if (id3) s->id3 = *id3; s->f = f; s->ch = channels; s->rate = rate; s->bits = bit_depth; s->aifc = type == OutputType::Aifc; err = s->begin(); *out = s; }
    return err;
}

}  // namespace d2dhost
