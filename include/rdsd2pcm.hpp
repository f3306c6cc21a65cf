// rdsd2pcm.hpp -- C++ host-side mirror of the rdsd2pcm crate surface that dsd2dxd's CLI uses
// (/root/reference/src/main.rs:27-31 imports; the crate itself is an absent submodule).  Same names,
// argument order and error texts as the call sites, over the C ABI of dsd2dxd_amd.h; the Rust shim in
// INTEGRATION.md is this file transliterated.
//
//   Rdsd2Pcm::create(...)            <- Rdsd2Pcm::new            (src/main.rs:325-342)
//   Rdsd2Pcm::from_container(...)    <- Rdsd2Pcm::from_container (src/main.rs:362-373)
//   Rdsd2Pcm::new_level_check(...)   <- Rdsd2Pcm::new_level_check (src/bin/dsd_levels/main.rs:214-223)
//   do_conversion(cancel, sender)    <- src/main.rs:345,429       throws std::runtime_error (Err(Box<dyn Error>))
//   check_level(cancel, sender)      <- src/bin/dsd_levels/main.rs:252   returns peak dBFS (f32)
//   file_name()                      <- src/main.rs:398
#pragma once
#include <atomic>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace rdsd2pcm {

enum class DitherType { TPDF, Rectangular, FPD, None,              // src/main.rs:172-175
                        NoiseShaped };                            // extension ('N'): see dsd2dxd_amd.h
enum class FmtType { Interleaved, Planar };                       // src/main.rs:185-186
enum class Endianness { LsbFirst, MsbFirst };                     // src/main.rs:194-196
enum class FilterType { Equiripple, XLD, Dsd2Pcm, Chebyshev };    // src/main.rs:200-204
enum class OutputType { Stdout, Aiff, Aifc, Wav, Flac };          // src/main.rs:208-213

struct ProgressUpdate { float percent; };
constexpr float ONE_HUNDRED_PERCENT = 100.0f;                     // src/main.rs:418
using ProgressSender = std::function<void(const ProgressUpdate&)>;

struct DsdFileFormat {
    enum Kind { Dsf, Dff, Raw, Stdin, Unknown } kind;
    static DsdFileFormat from(const std::string& path);            // src/main.rs:361
    bool is_container() const { return kind == Dsf || kind == Dff; }
};

// src/main.rs:275; throws std::runtime_error
std::vector<std::string> find_dsd_files(const std::vector<std::string>& paths, bool recurse);

class Rdsd2Pcm {
   public:
    // dsd_rate: 1, 2, 4 or 8 (the reference converts it with TryFrom<u32>; an invalid value throws here)
    static Rdsd2Pcm create(size_t bit_depth, OutputType output, double level_db, uint32_t output_rate,
                           std::optional<std::string> out_dir, DitherType dither, FmtType fmt, Endianness endian,
                           uint32_t dsd_rate, uint32_t block_size, size_t channels, FilterType filter,
                           bool append_rate, std::string base_dir, std::optional<std::string> in_path);
    static Rdsd2Pcm from_container(size_t bit_depth, OutputType output, double level_db, uint32_t output_rate,
                                   std::optional<std::string> out_dir, DitherType dither, FilterType filter,
                                   bool append_rate, std::string base_dir, std::string path);
    // path: nullopt = stdin (src/bin/dsd_levels/main.rs:214-223 passes Some(path), :273-281 None)
    static Rdsd2Pcm new_level_check(uint32_t output_rate, std::optional<std::string> path, FmtType fmt, Endianness endian,
                                    size_t channels, uint32_t block_size, uint32_t input_rate);
    Rdsd2Pcm(Rdsd2Pcm&&) noexcept;
    Rdsd2Pcm& operator=(Rdsd2Pcm&&) noexcept;
    ~Rdsd2Pcm();

    void do_conversion(const std::atomic<bool>& cancel, ProgressSender sender = nullptr);
    float check_level(const std::atomic<bool>& cancel, ProgressSender sender = nullptr);
    std::string file_name() const;
    std::string output_path() const;          // where do_conversion writes ("" for stdout)
    double dsp_seconds() const;               // time inside the engine during the last run (the "DSP speed" figure)
    double audio_seconds() const;
    std::string warnings() const;             // non-fatal findings of the last run (e.g. a damaged ID3 tag that was not copied)
    // engine knobs a driver may set before the run
    void set_device(int device);
    void set_seed(uint64_t seed);
    void set_tap_bits(uint32_t bits);     // 24 (default) or 32: d2d_params.tap_bits (include/dsd2dxd_amd.h); not a reference option
    void set_chunk_bytes(size_t bytes_per_channel);

    struct Impl;                              // opaque state (public only so the .cpp's helpers can name it)

   private:
    std::unique_ptr<Impl> p_;
    explicit Rdsd2Pcm(std::unique_ptr<Impl> p);
};

}  // namespace rdsd2pcm
