/* dsd2dxd_amd.h -- C ABI of the MI355X-native DSD->PCM decimation engine.
 *
 * This is the drop-in boundary for ONE path of clone206/dsd2dxd: the conversion core that the
 * CLI reaches through the rdsd2pcm crate (the crate itself is an un-vendored submodule; the
 * surface below is reconstructed from its call sites in the reference).  The reference has no
 * FFI of its own for this path, so each entry point names the Rust item a binding would sit
 * behind.  INTEGRATION.md shows the `extern "C"` block and the Rdsd2Pcm shim a maintainer adds.
 *
 *   plain pointers and sizes only; caller-owned buffers; int status returns (0 = ok, <0 = error);
 *   one engine is used by one thread at a time (the reference builds one Rdsd2Pcm per Rayon
 *   worker, src/main.rs:280-300,361-394); different engines are fully independent.
 */
#ifndef DSD2DXD_AMD_H
#define DSD2DXD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2D_ABI_VERSION 5   /* 3: the table blob header names the table variant; 4: d2d_params.tap_bits; 5: d2d_params.debug_flags (was reserved0) */

/* status codes */
enum {
    D2D_OK = 0,
    D2D_ERR_PARAM = -1,        /* invalid parameter value (message says which)                  */
    D2D_ERR_RATE = -2,         /* (dsd rate, output rate) combination not available             */
    D2D_ERR_FILTER = -3,       /* filter type not available for this rate combination           */
    D2D_ERR_DEVICE = -10,      /* no usable HIP device / HIP runtime error                      */
    D2D_ERR_CAPACITY = -20,    /* output buffer too small                                       */
    D2D_ERR_CANCELLED = -30,   /* cancel flag was raised (do_conversion's &AtomicBool)          */
    D2D_ERR_STATE = -40,       /* call sequence error                                           */
    D2D_ERR_IO = -50           /* reader / sink callback failed                                 */
};

/* enum values mirror the reference's CLI characters so a binding can pass them through:
 *   FmtType     src/main.rs:183-191    Endianness src/main.rs:193-197
 *   FilterType  src/main.rs:199-205    DitherType src/main.rs:171-181 */
enum { D2D_FMT_INTERLEAVED = 0, D2D_FMT_PLANAR = 1 };
enum { D2D_LSB_FIRST = 0, D2D_MSB_FIRST = 1 };
enum { D2D_FILTER_EQUIRIPPLE = 'E', D2D_FILTER_XLD = 'X', D2D_FILTER_DSD2PCM = 'D', D2D_FILTER_CHEBYSHEV = 'C' };
enum { D2D_DITHER_TPDF = 'T', D2D_DITHER_RECT = 'R', D2D_DITHER_FPD = 'F', D2D_DITHER_NONE = 'X',
       /* extension (no counterpart in src/main.rs:171-181): TPDF dither inside a second-order error-feedback
        * loop, noise transfer function (1 - z^-1)^2; integer depths, 44.1 kHz-family output rates.  The loop is a
        * recurrence through a rounding; it restarts from zero error at every output index that is a multiple of
        * 8192, so that those segments run side by side in a pass of their own. */
       D2D_DITHER_NOISE_SHAPED = 'N' };
/* which device kernel evaluates the FIR (same numbers either way) */
enum { D2D_KERNEL_AUTO = 0, D2D_KERNEL_LUT = 1, D2D_KERNEL_MFMA = 2 };
/* d2d_params.debug_flags (ABI 5): diagnostic dispatch switches.  0 = production dispatch; the library reads NO environment variable.
 * Each flag sends a conversion down an older kernel or route that produces the SAME bytes -- what the parity tests compare routes with
 * and what A/B measurements time; all are fixed when the engine is created. */
enum {
    D2D_DBG_NO_MX       = 1u << 0,   /* M = 32 / 64 / 128 stay off the fp6 x fp4 kernel (int8 pipelined kernel / older kernels)            */
    D2D_DBG_NO_GAINQ    = 1u << 1,   /* levels other than 0 dB, 20-bit and the float dither go back to the two-group / one-group kernels   */
    D2D_DBG_NO_COOP     = 1u << 2,   /* byte-interleaved input goes through the planar pre-pass instead of the kernels' own staging         */
    D2D_DBG_HOST_STAGED = 1u << 3,   /* host-pointer entry points stage through device buffers even when the GPU can address the memory   */
    D2D_DBG_NO_PIPE     = 1u << 4,   /* stereo conversions stay on the two-group kernel (no software-pipelined kernel)                    */
    D2D_DBG_MFMA_V1     = 1u << 5,   /* the one-group matrix-core kernel of round 1 for every decimation                                   */
    D2D_DBG_NO_INTQ     = 1u << 6,   /* the f64 epilogues instead of the all-integer requantisers                                           */
    D2D_DBG_NS_GENERAL  = 1u << 7,   /* the general noise-shaping kernel for stereo too                                                     */
    /* bits 8..15: waves per block of the matrix-core kernels (0 = their own choice)                                                        */
    D2D_DBG_TAPS32_2PASS = 1u << 16  /* tap_bits = 32: the two scratch passes and the combining pass where the one-pass kernel would serve  */
};

/* The conversion parameters of Rdsd2Pcm::new (src/main.rs:325-342) that concern the hot path.
 * File/sink arguments (output type, out_dir, append_rate, base_dir, in_path) stay on the host
 * side of the binding. */
typedef struct d2d_params {
    uint32_t struct_size;   /* sizeof(d2d_params), for ABI growth                               */
    uint32_t dsd_rate;      /* 1,2,4,8 = DSD64..DSD512     `dsd_rate`     src/main.rs:94-96,334 */
    uint32_t output_rate;   /* Hz                          `output_rate`  src/main.rs:85-92,329 */
    uint32_t channels;      /*                             `channels`     src/main.rs:50-52,336 */
    uint32_t fmt;           /* D2D_FMT_*                   `fmt`          src/main.rs:332       */
    uint32_t endianness;    /* D2D_LSB_FIRST/MSB_FIRST     `endian`       src/main.rs:333       */
    uint32_t block_size;    /* bytes/channel/block, planar `block_size`   src/main.rs:75-78,335 */
    uint32_t filter;        /* D2D_FILTER_*                `filter`       src/main.rs:337       */
    uint32_t bit_depth;     /* 16,20,24 int; 32 float      `bit_depth`    src/main.rs:58-60,326 */
    uint32_t dither;        /* D2D_DITHER_*                `dither`       src/main.rs:331       */
    uint32_t kernel;        /* D2D_KERNEL_*                                                      */
    int32_t  device;        /* HIP device ordinal (Rayon workers can be pinned to GPUs)         */
    double   level_db;      /* volume in dB                `level_db`     src/main.rs:107-110   */
    uint64_t seed;          /* dither seed (counter-based generator)             */
    /* Channel subset (ABI 2; a 64-byte struct without these two means "all channels").  The engine
     * reads the `channels`-wide input but converts only channels [channel_first, channel_first +
     * channel_count) and its PCM frames are `channel_count` wide; filter state, dither and peaks of a
     * channel are the same as in a full conversion, so the ranks of a multi-GPU job can take a few
     * channels each of ONE multichannel stream (config 5) and the host interleaves their
     * frames.  channel_count = 0: all channels. */
    uint32_t channel_first;
    uint32_t channel_count;
    /* Tap grid (ABI 4; a 72-byte struct without these two means 24).  24: the taps are q * 2^-S with 24-bit q -- what every fast
     * kernel is built on.  32: the same designs on the grid q32 * 2^-(S+8) (filters/filter_tables.inc: half32), for callers who
     * need integer output closer to an f64-tap reference (DESIGN.md sections 0 and 4.5): the engine runs the FIR twice -- the 24-bit
     * table, then the small residual table q32 - 256 q -- and combines the two exact integer sums before level, dither and
     * requantisation; about three times the device time.  44.1k-family rates, dither T / R / F / X. */
    uint32_t tap_bits;
    uint32_t debug_flags;   /* D2D_DBG_* (ABI 5; 0 in production)                                    */
} d2d_params;

typedef struct d2d_engine d2d_engine;

/* ---- life cycle -------------------------------------------------------------------------- */

/* Replaces Rdsd2Pcm::new / ::from_container (src/main.rs:325-343,362-374): validates the
 * (filter, dsd rate, output rate, depth, dither) combination, picks the tap tables, builds the
 * device tables.  `n_files` independent files (each `channels` wide) share one engine so that a
 * whole batch goes to the GPU in one launch; n_files = 1 is the per-file object of the reference.
 * On failure *out is NULL and d2d_create_error() (thread-local) has the message the reference
 * stringifies at src/main.rs:343,374,393. */
int d2d_create(const d2d_params* params, uint32_t n_files, d2d_engine** out);
const char* d2d_create_error(void);
void d2d_destroy(d2d_engine* e);
/* Start all files again from the idle history (a fresh Rdsd2Pcm). */
int d2d_reset(d2d_engine* e);
/* Message for the last failing call on this engine (Box<dyn Error> text, src/main.rs:438). */
const char* d2d_last_error(const d2d_engine* e);

/* ---- sizes ------------------------------------------------------------------------------- */

/* Bytes of one interleaved PCM frame: channels * {2,3,3,4} for 16/20/24/32 (20-bit rides in a
 * 3-byte container: build_test_mono.sh:3-8). */
size_t d2d_frame_bytes(const d2d_engine* e);
/* Frames the next translate call on `file` will produce if it is fed `bytes_per_channel`. */
size_t d2d_next_frames(const d2d_engine* e, uint32_t file, size_t bytes_per_channel);

/* ---- the hot path ------------------------------------------------------------------------ */

/* The per-block translate step inside Rdsd2Pcm::do_conversion (src/main.rs:345,429): feed
 * `bytes_per_channel` more bytes per channel of file 0 (channels*bytes_per_channel bytes at `dsd`,
 * laid out as params.fmt/block_size say; README.md:9), get interleaved little-endian PCM frames.
 * FIR history, resampler history, dither counter and peak carry over to the next call.
 * Host pointers; returns when `pcm` is filled.  Buffers the GPU can address (hipHostMalloc / hipHostRegister, 16-byte
 * aligned) are read and written by the kernels in place; any other memory is staged through device buffers. */
int d2d_translate(d2d_engine* e, const uint8_t* dsd, size_t bytes_per_channel,
                  void* pcm, size_t pcm_capacity_bytes, size_t* frames_out);

/* One file of a batch: device-resident input and output for the many-file path. */
typedef struct d2d_file_io {
    const void* dsd;            /* DEVICE pointer, 16-byte aligned                              */
    size_t      bytes_per_channel;
    void*       pcm;            /* DEVICE pointer, 16-byte aligned                              */
    size_t      pcm_capacity_bytes;
    size_t      frames_out;     /* written by the call                                          */
} d2d_file_io;

/* All `n_files` files of the engine advance by one call each, in ONE set of launches on
 * `hip_stream` (a hipStream_t, NULL = the null stream).  Asynchronous: returns once the work is
 * enqueued; io[i].frames_out is known at return.  This is the Rayon par_iter over files
 * (src/main.rs:280-300) turned into a grid dimension.  The engine's state (history, job table,
 * scratch) lives on the device: successive calls on one engine must be ordered on the device, i.e.
 * use one stream or order the streams with events. */
int d2d_translate_batch_device(d2d_engine* e, d2d_file_io* io, uint32_t n_files, void* hip_stream);

/* The same batch with HOST-resident buffers: `dsd` and `pcm` of every d2d_file_io are host pointers.
 * PINNED buffers (hipHostMalloc / hipHostRegister, 16-byte aligned) are addressed by the kernels
 * themselves: one set of launches reads the DSD and writes the frames over the link, so upload,
 * conversion and download overlap by construction and nothing is staged (bench shape: 90 GB/s over
 * the link both ways, `pcie_inclusive`).  Any other memory -- or D2D_DBG_HOST_STAGED, or a file of 2 GiB
 * per channel and more -- takes the pipeline: the batch is cut along time into slices of about
 * `slice_bytes_per_channel` (0 = 4 MiB; whole planar blocks), and upload, conversion and download
 * of consecutive slices run on three streams over double-buffered device staging (77 GB/s with
 * pinned buffers; pageable ones serialise).  Synchronous: returns when every `pcm` is filled;
 * io[i].frames_out is the total.  What the many-file path of a host that holds its files in RAM
 * calls; the reference's analogue is one Rayon worker per file reading,
 * converting and writing block by block (src/main.rs:280-300,345). */
int d2d_translate_batch_host(d2d_engine* e, d2d_file_io* io, uint32_t n_files, size_t slice_bytes_per_channel);

/* Rdsd2Pcm::check_level's result (src/bin/dsd_levels/main.rs:252-262): peak of |sample * gain|
 * seen so far.  Synchronises with the engine's pending work. */
int d2d_peak(d2d_engine* e, uint32_t file, uint32_t channel, double* peak_out);
int d2d_peak_dbfs(d2d_engine* e, uint32_t file, float* dbfs_out);

/* ---- whole-stream driver ------------------------------------------------------------------ */

/* do_conversion(&cancel, progress) for callers that own the I/O (src/main.rs:345,429):
 * `read` fills at most `cap` bytes per channel worth of DSD (returning bytes per channel read,
 * 0 at end of stream, <0 on error), `write` consumes PCM bytes; `cancel` is polled between
 * chunks (the &AtomicBool of src/main.rs:38,429); `progress` receives a percentage and exactly
 * 100.0f last (ONE_HUNDRED_PERCENT, src/main.rs:417-418). `total_bytes_per_channel` may be 0
 * when unknown (stdin): then only the final 100.0f is reported. */
typedef long (*d2d_read_fn)(void* user, uint8_t* dst, size_t cap_bytes_per_channel);
typedef int (*d2d_write_fn)(void* user, const void* pcm, size_t bytes);
typedef void (*d2d_progress_fn)(void* user, float percent);
int d2d_convert_stream(d2d_engine* e, d2d_read_fn read, void* read_user,
                       d2d_write_fn write, void* write_user,
                       const volatile int* cancel, d2d_progress_fn progress, void* progress_user,
                       uint64_t total_bytes_per_channel, size_t chunk_bytes_per_channel);

/* ---- shared tables (multi-GPU) ------------------------------------------------------------ */

/* The device filter tables as one opaque blob, so that rank 0 can broadcast them (RCCL) and
 * the other ranks adopt them instead of rebuilding: size query, export to / import from a
 * DEVICE buffer.  Import checks the blob's header against the engine's own configuration. */
size_t d2d_tables_bytes(const d2d_engine* e);
int d2d_tables_export_device(d2d_engine* e, void* dev_dst, size_t cap, void* hip_stream);
int d2d_tables_import_device(d2d_engine* e, const void* dev_src, size_t bytes, void* hip_stream);

/* ---- introspection ------------------------------------------------------------------------ */

typedef struct d2d_info {
    uint32_t decimation;     /* M of the integer decimator (stage A for the 48k family)        */
    uint32_t ntaps;
    uint32_t scale_bits;     /* taps are q * 2^-scale_bits                                      */
    uint32_t resamp_L, resamp_M, resamp_P;  /* 0 for the 44.1k family                           */
    uint32_t kernel;         /* D2D_KERNEL_LUT or D2D_KERNEL_MFMA actually in use               */
    uint32_t abi_version;
    char     filter_name[32];
} d2d_info;
int d2d_get_info(const d2d_engine* e, d2d_info* out);
/* Name of the device kernel that does the FIR, as rocprofv3 prints it (for bench/profiles). */
const char* d2d_kernel_name(const d2d_engine* e);

/* ---- measurement ---------------------------------------------------------------------------- */

/* When enabled, every translate call brackets its FIR launch with hipEvents on the launch stream;
 * d2d_profile_read() synchronises, returns the summed FIR kernel time and launch count since the
 * last read, and clears them.  (The analogue of the library's own "DSP speed" log line that the
 * reference prints per file -- asset/progress.jpg.) */
int d2d_profile_enable(d2d_engine* e, int on);
int d2d_profile_read(d2d_engine* e, double* fir_ms_total, uint64_t* launches);
/* The same, plus the device time of EVERY kernel of the batch calls (de-interleave, FIR, noise-shaping
 * or stage-B resampler, history carry): a second event pair around the whole call.  The two figures are
 * the engine's "DSP speed" for single- and multi-kernel configurations. */
int d2d_profile_read_all(d2d_engine* e, double* fir_ms_total, double* step_ms_total, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* DSD2DXD_AMD_H */
