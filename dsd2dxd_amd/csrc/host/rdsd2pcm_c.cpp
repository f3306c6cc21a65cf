// rdsd2pcm_c.cpp -- include/rdsd2pcm_c.h over include/rdsd2pcm.hpp: exceptions become codes + a thread-local message.
#include "../../../include/rdsd2pcm_c.h"

#include <atomic>
#include <string>

#include "../../../include/dsd2dxd_amd.h"
#include "../../../include/rdsd2pcm.hpp"

using namespace rdsd2pcm;

struct d2dh_conv {
    Rdsd2Pcm obj;
    std::string file_name, out_path, warnings;
    explicit d2dh_conv(Rdsd2Pcm&& o) : obj(std::move(o)) {}
};

namespace {
thread_local std::string g_err;

int fail(const std::exception& ex) {
    g_err = ex.what();
    if (g_err == "Conversion cancelled") return D2D_ERR_CANCELLED;
    return D2D_ERR_PARAM;
}
DitherType dither_of(uint32_t c) {
    switch (c) { case 'T': case 't': return DitherType::TPDF; case 'R': case 'r': return DitherType::Rectangular;
                 case 'F': case 'f': return DitherType::FPD; case 'X': case 'x': return DitherType::None;
                 case 'N': case 'n': return DitherType::NoiseShaped; }                       // extension
    throw std::runtime_error("Invalid dither type; must be T, R, F, or X");           // src/main.rs:176-180
}
FmtType fmt_of(uint32_t c) {
    switch (c) { case 'I': case 'i': return FmtType::Interleaved; case 'P': case 'p': return FmtType::Planar; }
    throw std::runtime_error("Invalid format; must be I (interleaved) or P (planar)");  // src/main.rs:187-190
}
Endianness endian_of(uint32_t c) { return (c == 'L' || c == 'l') ? Endianness::LsbFirst : Endianness::MsbFirst; }       // :193-197
FilterType filter_of(uint32_t c) {                                                                                       // :199-205
    switch (c) { case 'X': case 'x': return FilterType::XLD; case 'D': case 'd': return FilterType::Dsd2Pcm; case 'C': case 'c': return FilterType::Chebyshev; }
    return FilterType::Equiripple;
}
OutputType output_of(uint32_t c) {                                                                                       // :207-214
    switch (c) { case 'A': case 'a': return OutputType::Aiff; case 'C': case 'c': return OutputType::Aifc;
                 case 'W': case 'w': return OutputType::Wav; case 'F': case 'f': return OutputType::Flac; }
    return OutputType::Stdout;
}
std::optional<std::string> opt(const char* s) { return s ? std::optional<std::string>(s) : std::nullopt; }

template <typename F>
int guarded(F&& f) {
    try { g_err.clear(); f(); return D2D_OK; }
    catch (const std::exception& ex) { return fail(ex); }
}

// the C side polls an int, the C++ side an atomic<bool>: bridge through the progress callback and a
// copy made before and after (the engine itself polls between blocks through its read/write callbacks)
struct CancelBridge {
    const volatile int* src; std::atomic<bool> flag{false};
    explicit CancelBridge(const volatile int* s) : src(s) { poll(); }
    void poll() { if (src && *src) flag.store(true); }
};
}  // namespace

extern "C" {

const char* d2dh_last_error(void) { return g_err.c_str(); }

int d2dh_new(uint32_t bit_depth, uint32_t output, double level_db, uint32_t output_rate, const char* out_dir, uint32_t dither,
             uint32_t fmt, uint32_t endian, uint32_t dsd_rate, uint32_t block_size, uint32_t channels, uint32_t filter,
             int append_rate, const char* base_dir, const char* in_path, d2dh_conv** out) {
    if (!out) { g_err = "null argument"; return D2D_ERR_PARAM; }
    *out = nullptr;
    return guarded([&] {
        *out = new d2dh_conv(Rdsd2Pcm::create(bit_depth, output_of(output), level_db, output_rate, opt(out_dir), dither_of(dither),
                                              fmt_of(fmt), endian_of(endian), dsd_rate, block_size, channels, filter_of(filter),
                                              append_rate != 0, base_dir ? base_dir : ".", opt(in_path)));
    });
}

int d2dh_from_container(uint32_t bit_depth, uint32_t output, double level_db, uint32_t output_rate, const char* out_dir,
                        uint32_t dither, uint32_t filter, int append_rate, const char* base_dir, const char* path, d2dh_conv** out) {
    if (!out || !path) { g_err = "null argument"; return D2D_ERR_PARAM; }
    *out = nullptr;
    return guarded([&] {
        *out = new d2dh_conv(Rdsd2Pcm::from_container(bit_depth, output_of(output), level_db, output_rate, opt(out_dir), dither_of(dither),
                                                      filter_of(filter), append_rate != 0, base_dir ? base_dir : ".", path));
    });
}

int d2dh_new_level_check(uint32_t output_rate, const char* path, uint32_t fmt, uint32_t endian, uint32_t channels,
                         uint32_t block_size, uint32_t input_rate, d2dh_conv** out) {
    if (!out) { g_err = "null argument"; return D2D_ERR_PARAM; }
    *out = nullptr;
    return guarded([&] {
        *out = new d2dh_conv(Rdsd2Pcm::new_level_check(output_rate, opt(path), fmt_of(fmt), endian_of(endian), channels, block_size, input_rate));
    });
}

void d2dh_free(d2dh_conv* c) { delete c; }

int d2dh_do_conversion(d2dh_conv* c, const volatile int* cancel, d2dh_progress_fn progress, void* user) {
    if (!c) { g_err = "null object"; return D2D_ERR_PARAM; }
    CancelBridge br(cancel);
    return guarded([&] {
        c->obj.do_conversion(br.flag, [&](const ProgressUpdate& u) { br.poll(); if (progress) progress(user, u.percent); });
        c->out_path = c->obj.output_path();
        c->warnings = c->obj.warnings();
    });
}

int d2dh_check_level(d2dh_conv* c, const volatile int* cancel, d2dh_progress_fn progress, void* user, float* peak_dbfs) {
    if (!c || !peak_dbfs) { g_err = "null argument"; return D2D_ERR_PARAM; }
    CancelBridge br(cancel);
    return guarded([&] {
        *peak_dbfs = c->obj.check_level(br.flag, [&](const ProgressUpdate& u) { br.poll(); if (progress) progress(user, u.percent); });
    });
}

const char* d2dh_file_name(const d2dh_conv* c) {
    if (!c) return "";
    const_cast<d2dh_conv*>(c)->file_name = c->obj.file_name();
    return c->file_name.c_str();
}
const char* d2dh_output_path(const d2dh_conv* c) { return c ? c->out_path.c_str() : ""; }
const char* d2dh_warnings(const d2dh_conv* c) { return c ? c->warnings.c_str() : ""; }
void d2dh_set_device(d2dh_conv* c, int device) { if (c) c->obj.set_device(device); }
void d2dh_set_seed(d2dh_conv* c, uint64_t seed) { if (c) c->obj.set_seed(seed); }
void d2dh_set_tap_bits(d2dh_conv* c, uint32_t bits) { if (c) c->obj.set_tap_bits(bits); }

int d2dh_find_dsd_files(const char* const* paths, size_t n_paths, int recurse, d2dh_path_fn each, void* user) {
    return guarded([&] {
        std::vector<std::string> in;
        for (size_t i = 0; i < n_paths; ++i) if (paths && paths[i]) in.push_back(paths[i]);
        for (const std::string& p : find_dsd_files(in, recurse != 0)) if (each) each(user, p.c_str());
    });
}

int d2dh_is_container(const char* path) { return path && DsdFileFormat::from(path).is_container() ? 1 : 0; }

}  // extern "C"
