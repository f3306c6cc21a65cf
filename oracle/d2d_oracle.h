/* d2d_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, f64) of the DSD->PCM hot path that dsd2dxd's CLI drives through
 * rdsd2pcm (call sites /root/reference/src/main.rs:325-345,361-394,429).  It is the checker for
 * the HIP engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * it.  The product (dsd2dxd_amd/) never links, imports or executes anything in oracle/.
 *
 * PARITY UNPINNED.  The algorithm lives in a third-party dependency that is absent from
 * /root/reference: rdsd2pcm 0.3.0 (path dep Cargo.toml:8, Cargo.lock:543-553, un-vendored
 * submodule .gitmodules:1-3; src/rdsd2pcm/ is empty) on top of dsd-reader 0.2.0 and rand 0.8.5.
 * The reference holds no golden vector, known-answer test or numeric assertion for this path
 * (every script ends in ffplay or /dev/null: run_all_tests.sh:1-12, test_all_44k_mults.sh,
 * test_all_48k_mults.sh), and no Rust toolchain exists here to run it.  What is restated is
 * therefore the DOCUMENTED behaviour (README.md:9,11-12,129-134,143-152,230,236,252,254;
 * src/main.rs:50-110,165-214) plus the published dsd2pcm algorithm the README acknowledges as the
 * code's ancestor (README.md:240): byte FIFO, per-byte lookup tables, symmetric half-stored taps,
 * one output per M input bits, idle pattern 0x69 as initial history.
 */
#ifndef D2D_ORACLE_H
#define D2D_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* Same parameter set as Rdsd2Pcm::new (src/main.rs:325-342), minus the file/sink arguments. */
typedef struct {
    uint32_t dsd_rate;     /* 1,2,4,8 = DSD64..DSD512              (src/main.rs:94-96,334)   */
    uint32_t output_rate;  /* Hz                                      (src/main.rs:85-92,329)   */
    uint32_t channels;     /*                                         (src/main.rs:50-52,336)   */
    uint32_t fmt;          /* 0 Interleaved, 1 Planar                 (src/main.rs:183-191)     */
    uint32_t endianness;   /* 0 LsbFirst, 1 MsbFirst                  (src/main.rs:193-197)     */
    uint32_t block_size;   /* bytes per channel per block (planar)    (src/main.rs:75-78,335)   */
    uint32_t filter;       /* 'E','X','D','C'                         (src/main.rs:199-205)     */
    uint32_t bit_depth;    /* 16,20,24 int; 32 float                  (src/main.rs:58-60)       */
    uint32_t dither;       /* 'T','R','F','X' (src/main.rs:171-181); 'N' = noise-shaped TPDF, an extension   */
    uint32_t fir_mode;     /* 0 = direct bit-by-bit form, 1 = byte-LUT form (dsd2pcm lineage)   */
    double   level_db;     /*                                         (src/main.rs:107-110,328) */
    uint64_t seed;         /* dither seed (the reference's seeding is unknown; see header)      */
} orc_params;

/* returns 0 or a negative code; *err gets a static message */
int  orc_create(const orc_params* p, orc_ctx** out, const char** err);
void orc_destroy(orc_ctx* c);

/* Upper bound on frames produced by a call that feeds `bytes_per_channel` more bytes. */
size_t orc_max_frames(const orc_ctx* c, size_t bytes_per_channel);
/* Bytes per output frame (channels * container bytes). */
size_t orc_frame_bytes(const orc_ctx* c);

/* One block set: `dsd` holds channels*bytes_per_channel bytes in the context's layout.
 * Writes interleaved little-endian PCM frames; state (FIR history, resampler history,
 * dither counter, peak) is carried to the next call. */
int orc_translate(orc_ctx* c, const uint8_t* dsd, size_t bytes_per_channel,
                  void* pcm_out, size_t pcm_capacity_bytes, size_t* frames_out);

/* Same, but also hands back the pre-dither f64 samples (level applied), interleaved; may be NULL. */
int orc_translate_f64(orc_ctx* c, const uint8_t* dsd, size_t bytes_per_channel,
                      void* pcm_out, size_t pcm_capacity_bytes, double* f64_out, size_t* frames_out);

/* The same bytes and state as orc_translate through a streaming organisation (preallocated buffers, integer byte tables
 * in the stream's bit order): the CPU baseline that bench.py times.  Falls back to orc_translate where it does not apply. */
int orc_translate_stream(orc_ctx* c, const uint8_t* dsd, size_t bytes_per_channel,
                         void* pcm_out, size_t pcm_capacity_bytes, size_t* frames_out);

double orc_peak(const orc_ctx* c, uint32_t channel);      /* max |sample*gain| so far */
float  orc_peak_dbfs(const orc_ctx* c);                    /* 20*log10(max over channels) */

/* Introspection used by the tests */
int    orc_filter_info(const orc_ctx* c, int* M, int* ntaps, int* S, int* resamp_L, int* resamp_P);
/* full (both halves) tap i as f64; returns 0.0 outside */
double orc_tap(const orc_ctx* c, int i);
/* Replace the context's taps by `half` (ntaps/2 doubles, 2nd half centre-outward, any values) and rebuild the byte
 * tables from them: with the designs' UNQUANTISED f64 taps (filters/filter_taps_f64.json) the context becomes the
 * f64-tap reference that measures what this build's 24-bit tap grid costs (tests/test_tap_grid.py).  The sums are
 * then ordinary f64 arithmetic in dsd2pcm's order (table i of the newer half + table i of the older half, i ascending). */
int    orc_set_half_taps(orc_ctx* c, const double* half, int n_half);
/* the filter's 32-bit tap grid (q32 * 2^-(S+8)) instead of the 24-bit one: the engine's d2d_params.tap_bits = 32; 44.1k family, T/R/F/X */
int    orc_use_fine_taps(orc_ctx* c);
/* study mode: stage B of the 48k cascade with the f64 coefficients the 2^-28 grid was rounded from (tests/test_tap_grid.py) */
int    orc_use_f64_resamp_coef(orc_ctx* c);
/* study mode: DSD64 / DSD128 -> 96 / 192 / 384 kHz through the two-stage cascade they were defined by before the stages were composed into
 * one polyphase table (call before the first translate); with orc_use_f64_resamp_coef and then orc_set_half_taps: that cascade's f64 design */
int    orc_use_cascade(orc_ctx* c);

/* The dither generator, exposed so tests can pin it. */
uint32_t orc_rng(uint64_t seed, uint32_t channel, uint64_t n);
uint64_t orc_rng_key(uint64_t seed, uint32_t channel);

#ifdef __cplusplus
}
#endif
#endif
