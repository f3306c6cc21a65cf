#include "../../../include/rdsd2pcm.hpp"

#include <string.h>
#include <sys/stat.h>

#include <chrono>

#include "../../../include/dsd2dxd_amd.h"
#include "dsd_reader.h"
#include "pcm_sink.h"
#include "id3_tag.h"

namespace rdsd2pcm {

using namespace d2dhost;

DsdFileFormat DsdFileFormat::from(const std::string& path) {
    switch (format_from_path(path)) {
        case d2dhost::DsdFileFormat::Dsf: return {Dsf};
        case d2dhost::DsdFileFormat::Dff: return {Dff};
        case d2dhost::DsdFileFormat::Raw: return {Raw};
        case d2dhost::DsdFileFormat::Stdin: return {Stdin};
        default: return {Unknown};
    }
}

std::vector<std::string> find_dsd_files(const std::vector<std::string>& paths, bool recurse) {
    std::vector<std::string> out;
    std::string err = d2dhost::find_dsd_files(paths, recurse, out);
    if (!err.empty()) throw std::runtime_error(err);
    return out;
}

struct Rdsd2Pcm::Impl {
    d2d_params prm{};
    OutputType output = OutputType::Stdout;
    std::optional<std::string> out_dir;
    bool append_rate = false;
    std::string base_dir;
    std::optional<std::string> in_path;
    DsdInfo info;
    bool level_only = false;
    size_t chunk = 4u << 20;
    std::string out_path;
    double dsp_s = 0.0, audio_s = 0.0;
    std::string tag_warning;
    int artwork_copied = 0;
};

static uint32_t dither_code(DitherType d) { return d == DitherType::TPDF ? 'T' : d == DitherType::Rectangular ? 'R' : d == DitherType::FPD ? 'F' : d == DitherType::NoiseShaped ? 'N' : 'X'; }
static uint32_t filter_code(FilterType f) { return f == FilterType::Equiripple ? 'E' : f == FilterType::XLD ? 'X' : f == FilterType::Dsd2Pcm ? 'D' : 'C'; }

static std::string dirname_of(const std::string& p) { size_t s = p.find_last_of('/'); return s == std::string::npos ? "." : (s == 0 ? "/" : p.substr(0, s)); }
static std::string basename_of(const std::string& p) { size_t s = p.find_last_of('/'); return s == std::string::npos ? p : p.substr(s + 1); }
static std::string stem_of(const std::string& p) { std::string b = basename_of(p); size_t d = b.find_last_of('.'); return d == std::string::npos ? b : b.substr(0, d); }

static void mkdirs(const std::string& dir) {
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') { if (!cur.empty()) mkdir(cur.c_str(), 0777); }
        if (i < dir.size()) cur += dir[i];
    }
}

// "_96K", "_88_2K" (README.md:170-173)
static std::string rate_suffix(uint32_t rate) {
    char b[32];
    if (rate % 1000 == 0) snprintf(b, sizeof(b), "_%uK", rate / 1000);
    else snprintf(b, sizeof(b), "_%u_%uK", rate / 1000, (rate % 1000) / 100);
    return b;
}

static void validate_engine(const d2d_params& p) {
    // a dry creation surfaces every parameter error with the engine's own message; a missing GPU is
    // reported later, when the conversion actually starts
    d2d_engine* e = nullptr;
    int rc = d2d_create(&p, 1, &e);
    if (rc == D2D_OK) { d2d_destroy(e); return; }
    if (rc == D2D_ERR_DEVICE) return;
    throw std::runtime_error(d2d_create_error());
}

Rdsd2Pcm::Rdsd2Pcm(std::unique_ptr<Impl> p) : p_(std::move(p)) {}
Rdsd2Pcm::Rdsd2Pcm(Rdsd2Pcm&&) noexcept = default;
Rdsd2Pcm& Rdsd2Pcm::operator=(Rdsd2Pcm&&) noexcept = default;
Rdsd2Pcm::~Rdsd2Pcm() = default;

Rdsd2Pcm Rdsd2Pcm::create(size_t bit_depth, OutputType output, double level_db, uint32_t output_rate,
                          std::optional<std::string> out_dir, DitherType dither, FmtType fmt, Endianness endian,
                          uint32_t dsd_rate, uint32_t block_size, size_t channels, FilterType filter,
                          bool append_rate, std::string base_dir, std::optional<std::string> in_path) {
    auto im = std::make_unique<Impl>();
    d2d_params& p = im->prm;
    p.struct_size = sizeof(p);
    p.dsd_rate = dsd_rate; p.output_rate = output_rate; p.channels = (uint32_t)channels;
    p.fmt = fmt == FmtType::Planar ? D2D_FMT_PLANAR : D2D_FMT_INTERLEAVED;
    p.endianness = endian == Endianness::MsbFirst ? D2D_MSB_FIRST : D2D_LSB_FIRST;
    p.block_size = block_size; p.filter = filter_code(filter); p.bit_depth = (uint32_t)bit_depth;
    p.dither = dither_code(dither); p.kernel = D2D_KERNEL_AUTO; p.device = 0; p.level_db = level_db; p.seed = 0;
    im->output = output; im->out_dir = std::move(out_dir); im->append_rate = append_rate;
    im->base_dir = std::move(base_dir); im->in_path = std::move(in_path);
    im->info.format = im->in_path ? format_from_path(*im->in_path) : d2dhost::DsdFileFormat::Stdin;
    if (im->in_path && im->info.format == d2dhost::DsdFileFormat::Unknown) im->info.format = d2dhost::DsdFileFormat::Raw;
    im->info.channels = p.channels; im->info.dsd_rate = dsd_rate; im->info.planar = p.fmt == D2D_FMT_PLANAR;
    im->info.msb_first = p.endianness == D2D_MSB_FIRST; im->info.block_size = block_size;
    validate_engine(p);
    return Rdsd2Pcm(std::move(im));
}

Rdsd2Pcm Rdsd2Pcm::from_container(size_t bit_depth, OutputType output, double level_db, uint32_t output_rate,
                                  std::optional<std::string> out_dir, DitherType dither, FilterType filter,
                                  bool append_rate, std::string base_dir, std::string path) {
    DsdInfo info;
    std::string err = probe(path, info);
    if (!err.empty()) throw std::runtime_error(err);
    // the container's own layout replaces the command-line flags (README.md:103-105)
    Rdsd2Pcm r = create(bit_depth, output, level_db, output_rate, std::move(out_dir), dither,
                        info.planar ? FmtType::Planar : FmtType::Interleaved,
                        info.msb_first ? Endianness::MsbFirst : Endianness::LsbFirst, info.dsd_rate, info.block_size,
                        info.channels, filter, append_rate, std::move(base_dir), path);
    r.p_->info = info;
    return r;
}

Rdsd2Pcm Rdsd2Pcm::new_level_check(uint32_t output_rate, std::optional<std::string> path, FmtType fmt, Endianness endian,
                                   size_t channels, uint32_t block_size, uint32_t input_rate) {
    Rdsd2Pcm r = (path && DsdFileFormat::from(*path).is_container())
                     ? from_container(32, OutputType::Stdout, 0.0, output_rate, std::nullopt, DitherType::None,
                                      FilterType::Equiripple, false, ".", *path)
                     : create(32, OutputType::Stdout, 0.0, output_rate, std::nullopt, DitherType::None, fmt, endian,
                              input_rate, block_size, channels, FilterType::Equiripple, false, ".", path);
    r.p_->level_only = true;
    return r;
}

std::string Rdsd2Pcm::file_name() const { return p_->in_path ? basename_of(*p_->in_path) : "stdin"; }
std::string Rdsd2Pcm::output_path() const { return p_->out_path; }
double Rdsd2Pcm::dsp_seconds() const { return p_->dsp_s; }
double Rdsd2Pcm::audio_seconds() const { return p_->audio_s; }
std::string Rdsd2Pcm::warnings() const { return p_->tag_warning; }
void Rdsd2Pcm::set_device(int device) { p_->prm.device = device; }
void Rdsd2Pcm::set_seed(uint64_t seed) { p_->prm.seed = seed; }
void Rdsd2Pcm::set_tap_bits(uint32_t bits) { p_->prm.tap_bits = bits; }
void Rdsd2Pcm::set_chunk_bytes(size_t b) { if (b) p_->chunk = b; }

namespace {
struct RunCtx {
    DsdSource* src; PcmSink* sink; const std::atomic<bool>* cancel; volatile int cancel_int = 0;
    ProgressSender* sender; std::string io_error;
    std::chrono::steady_clock::duration io_time{};
};
long read_cb(void* u, uint8_t* dst, size_t cap) {
    RunCtx* c = (RunCtx*)u;
    auto t0 = std::chrono::steady_clock::now();
    if (c->cancel->load()) c->cancel_int = 1;
    long n = c->src->read(dst, cap);
    c->io_time += std::chrono::steady_clock::now() - t0;
    return n;
}
int write_cb(void* u, const void* pcm, size_t bytes) {
    RunCtx* c = (RunCtx*)u;
    auto t0 = std::chrono::steady_clock::now();
    std::string e = c->sink ? c->sink->write((const uint8_t*)pcm, bytes) : "";
    c->io_time += std::chrono::steady_clock::now() - t0;
    if (!e.empty()) { c->io_error = e; return 1; }
    if (c->cancel->load()) c->cancel_int = 1;
    return 0;
}
void progress_cb(void* u, float pct) {
    RunCtx* c = (RunCtx*)u;
    if (c->sender && *c->sender) (*c->sender)(ProgressUpdate{pct});
}
}  // namespace

static float run(Rdsd2Pcm::Impl& im, const std::atomic<bool>& cancel, ProgressSender& sender, bool levels) {
    DsdSource src;
    std::string err = src.open(im.in_path ? *im.in_path : "-", im.info);
    if (!err.empty()) throw std::runtime_error(err);
    const DsdInfo& info = src.info();
    PcmSink* sink = nullptr;
    im.out_path.clear();
    if (!levels) {
        if (im.output != OutputType::Stdout) {
            std::string dir;
            const std::string in_dir = im.in_path ? dirname_of(*im.in_path) : ".";
            if (im.out_dir) {
                dir = *im.out_dir;
                // keep the input's position below base_dir (src/main.rs:255-273)
                if (im.in_path && !im.base_dir.empty() && in_dir.compare(0, im.base_dir.size(), im.base_dir) == 0 &&
                    in_dir.size() > im.base_dir.size())
                    dir += in_dir.substr(im.base_dir.size());
                mkdirs(dir);
            } else dir = in_dir;
            std::string stem = im.in_path ? stem_of(*im.in_path) : "output";
            if (im.append_rate) stem += rate_suffix(im.prm.output_rate);
            im.out_path = dir + "/" + stem + "." + output_extension((d2dhost::OutputType)(int)im.output);
        }
        // the source's ID3v2 tag travels with the audio (README.md:7); -a also marks the album (README.md:170-173)
        std::vector<uint8_t> tag;
        if (im.output != OutputType::Stdout && im.in_path) {
            std::string twarn;
            err = d2dhost::read_source_tag(*im.in_path, info, tag, twarn);
            if (!err.empty()) throw std::runtime_error(err);
            im.tag_warning = twarn;
            if (!tag.empty() && im.append_rate) d2dhost::append_to_album(tag, d2dhost::album_rate_suffix(im.prm.output_rate));
            // -p: artwork next to the sources follows them into the output tree (README.md:115-119)
            if (im.out_dir) im.artwork_copied = d2dhost::copy_artwork(dirname_of(*im.in_path), dirname_of(im.out_path));
        }
        err = open_sink((d2dhost::OutputType)(int)im.output, im.out_path, im.prm.channels, im.prm.output_rate, im.prm.bit_depth, &sink, &tag);
        if (!err.empty()) throw std::runtime_error(err);
    }
    std::unique_ptr<PcmSink> sink_guard(sink);
    d2d_engine* e = nullptr;
    if (d2d_create(&im.prm, 1, &e) != D2D_OK) throw std::runtime_error(d2d_create_error());
    RunCtx ctx; ctx.src = &src; ctx.sink = sink; ctx.cancel = &cancel; ctx.sender = &sender;
    auto t0 = std::chrono::steady_clock::now();
    int rc = d2d_convert_stream(e, read_cb, &ctx, levels ? nullptr : write_cb, &ctx, &ctx.cancel_int,
                                progress_cb, &ctx, info.bytes_per_channel, im.chunk);
    auto total = std::chrono::steady_clock::now() - t0;
    std::string msg = rc ? d2d_last_error(e) : "";
    float db = 0.f;
    if (!rc && levels) { if (d2d_peak_dbfs(e, 0, &db) != D2D_OK) msg = d2d_last_error(e); }
    const uint64_t bpc = info.bytes_per_channel;
    d2d_destroy(e);
    im.dsp_s = std::chrono::duration<double>(total - ctx.io_time).count();
    im.audio_s = bpc ? (double)bpc * 8.0 / (2822400.0 * im.prm.dsd_rate) : 0.0;
    if (sink) { std::string ce = sink->close(); if (msg.empty() && !ce.empty()) msg = ce; }
    if (!ctx.io_error.empty()) msg = ctx.io_error;
    if (!msg.empty()) throw std::runtime_error(msg);
    return db;
}

void Rdsd2Pcm::do_conversion(const std::atomic<bool>& cancel, ProgressSender sender) { run(*p_, cancel, sender, false); }

float Rdsd2Pcm::check_level(const std::atomic<bool>& cancel, ProgressSender sender) { return run(*p_, cancel, sender, true); }

}  // namespace rdsd2pcm
