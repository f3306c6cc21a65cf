// d2d_internal.h -- structures shared by the host engine and the device kernels.
// Not part of the C ABI (that is include/dsd2dxd_amd.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace d2d {

constexpr uint32_t DSD64_RATE = 2822400u;
constexpr uint8_t IDLE_BYTE = 0x69;   // dsd2pcm's idle pattern (DC-free), MSB-first-in-time
constexpr int LUT_THREADS = 256;

// One (file, channel) stream for one translate call.  All byte indices are "call-relative":
// 0 = the first byte this call feeds for the channel; negative = carried history.
struct StreamJob {
    const uint8_t* in;      // device: the file's call buffer (all channels) in the layout the FIR kernels read
    const uint8_t* in_raw;  // device: the caller's byte-interleaved buffer when `in` is the engine's planar copy, else null
    const uint8_t* hist;    // device: this channel's `keep` history bytes (raw bit order)
    uint8_t*       hist_next; // device: where the updated history goes (ping-pong buffer)
    void*          out;     // device: the file's interleaved PCM frames for this call
    int32_t*       xs;      // device: 48k cascade only -- xs[i] = stage-A output n0+i as the integer y*2^S
                            // (exact: |sum q s| < 2^31 for every table), xs[-P..-1] = carried
    double*        peak;    // device: this channel's running peak (as non-negative f64)
    uint64_t       L;       // bytes per channel fed by this call
    int64_t        e0;      // one past the newest byte of output n0's window (call-relative)
    uint64_t       n0;      // absolute index of the first FIR output of this call
    uint32_t       nout;    // FIR outputs this call
    uint32_t       ch;      // channel index inside the FILE (input layout, dither key)
    // stage B (48k family)
    uint64_t       m0;      // absolute index of the first resampler output of this call
    uint32_t       nres;    // resampler outputs this call
    uint32_t       rng_key;   // dither key at this call's first output index i0: k32 + hi32(i0)*kstep
    uint32_t       rng_kstep; // added once more where lo32(index) wraps inside this call
    uint32_t       rng_lo0;   // lo32(i0); the index is n (44.1k family) or m (48k family)
    uint32_t       och;       // channel index inside the OUTPUT frame (differs from ch when the engine converts a channel subset)
};

// How to turn an f64 sample into output bytes (a5-a7 of SURVEY 8a).
struct Epilogue {
    double   gain;        // 10^(level/20)
    double   scale;       // gain * 2^(bits-1) for integer depths
    uint64_t seed;
    uint32_t bits;        // 16,20,24,32
    uint32_t dither;      // 'T','R','F','X'
    uint32_t sample_bytes;
    uint32_t channels;
};

struct FirArgs {
    const StreamJob* jobs;
    const void*      tables;   // LUT: f64 nibble tables [ntab][16]; MFMA: int8 B fragments
    uint32_t Wb;               // window bytes = ntaps/8
    uint32_t ntab;             // LUT: number of nibble tables incl. zero padding
    uint32_t pad;              // LUT: zero tables in front
    uint32_t nq;               // LUT: qwords each lane walks
    uint32_t B;                // effective block size (1 = byte interleaved)
    uint32_t keep;             // history bytes per channel
    uint32_t to_scratch;       // 1: write the FIR outputs as integers y*2^S to job.xs (stage A of the 48k cascade)
    uint32_t ksteps;           // MFMA: K steps
    int32_t  scale_bits;       // S of the tap table (h = q * 2^-S)
    uint32_t in_channels;      // channels of the input layout (epi.channels = channels of the output frame; fewer for a channel subset)
    uint64_t sum_abs_q;        // sum |q_j| of the tap table (bounds |y*2^S|)
    uint32_t pipelined;        // two-group MFMA kernels: 0, or the variant of the pipelined kernel whose tap table the engine built (3 dense, 4 sparse,
                               // 5 the fp6 x fp4 kernel of d2d_kernels_mx.hip)
    uint32_t coop;             // 1 (d2d_kernels_mx.hip, scratch flavour): byte-interleaved 4- or 8-channel input, every channel converted: the kernel
                               // de-interleaves inside its staging, a block per (file, tile) with one wave per channel pair; B = 1 then
    uint32_t il2;              // 1 (pipelined frame kernels): byte-interleaved STEREO input (DFF, -f I), both channels converted: the kernel's staging
                               // pulls the channels apart (one v_perm_b32 per channel and eight input bytes); B = 1, no planar copy
    uint32_t mx_exact;         // 1: the table's base-32 digit sums recombine exactly in f32 (d2d_mx.h: mx_exact)
    uint32_t dbg_flags;        // d2d_params.debug_flags (D2D_DBG_*), fixed when the engine was created
    uint32_t taps32;           // 1 (d2d_kernels_mx.hip, stereo frames): the 32-bit tap grid in ONE pass -- `tables` hold the seven-digit fragments of the half32
                               // taps, scale_bits = S + 8, v = sum q32 s is a 64-bit integer, the f64 requantiser finishes (tap_bits = 32, round 4)
    uint32_t mono2;            // 1 (d2d_kernels_mx.hip, frames): a MONO stream served as a planar pair -- jobs 2 f, 2 f + 1 are the two halves of file f's call
                               // (equal lengths and outputs; the second's history is the end of the first, its frames follow the first's)
    Epilogue epi;
};

// the noise-shaping pass ('N' dither): one lane per (8192-output segment, channel) walks its integers in order
struct NoiseShapeArgs {
    const StreamJob* jobs;
    const double* state;       // [nstreams][2]: the last two requantisation errors carried INTO this call
    double*  state_next;       // ... and out of it (ping-pong: reader and writer of a stream's state may be different blocks)
    uint8_t* dump;             // 1 KiB nobody reads: where the stereo kernel's cooperative stores put the pieces that belong to no group
    int32_t  scale_bits;       // S: the scratch holds y * 2^S
    uint32_t nstreams;
    uint32_t max_nout;         // the longest stream's outputs this call (sizes the grid)
    uint32_t cp_bits;          // log2(channels rounded up to a power of two): filled by the launcher
    uint32_t intq;             // 1: |y * 2^S| + a few LSB stay inside int32 (the engine checks the tap table): the all-integer loop may run at unit gain
    // 48k family: the pass runs on stage B's outputs -- index m (job.m0, job.nres), input y as f64 in ys[stream * ys_stride + i] (the
    // resampler's 64-bit integers do not fit the int32 recurrence), peaks already taken by stage B
    const double* ys;
    uint32_t ys_stride;
    uint32_t res;
    uint32_t general;          // 1 (D2D_DBG_NS_GENERAL): the general kernel for stereo frames too
    Epilogue epi;
};

// stage B of the 48k cascade (d2d_kernels_rs.hip): launch_resample2 fills everything but jobs, tables, S and epi
struct Rs2Args {
    const StreamJob* jobs;
    const uint8_t* tables;     // [L/4 blocks][NSTEP][64 lanes][16 bytes] coefficient fragments, then the blocks' row offsets (uint32)
    uint32_t L, Mdn, P, NB, NSTEP;
    int32_t  S, T;             // x = X * 2^-S (stage A's table), g = G * 2^-T (filters/filter_tables.inc)
    uint32_t cw;               // channels a wave converts one after the other: 2 for an even channel count, else 1
    uint32_t nwaves;
    uint32_t off_waves, wave_lds, off_out;   // LDS: start of the per-wave regions, bytes per wave, the wave's output slice inside its region
    uint32_t fast;             // 1: unit gain at 24 or 16 bits, dither T / R / X: the all-integer requantiser (with its guard band)
    uint32_t dkind;            // 0 none, 1 triangular, 2 rectangular
    int32_t  fbits;            // F = S + T - (bits - 1): y in LSB is v * 2^-F
    double*  ys;               // non-null: the samples go as y = (double)v * 2^-(S+T) to ys[stream * ys_stride + i] for the noise-shaping pass, no frames
    uint32_t ys_stride;
    Epilogue epi;
};

// blob header for d2d_tables_export/import
struct TableBlobHeader {
    uint32_t magic;            // 'D2DT'
    uint32_t abi;
    uint32_t kernel;
    uint32_t endianness;
    uint32_t ntaps;
    uint32_t M;
    uint32_t scale_bits;
    uint32_t filter_type;
    uint32_t table_variant;    // layout of the FIR table: 0 LUT / one-group MFMA, 2 two-group (plane 0 unmasked), 3 pipelined (every plane masked),
                               // 4 structured-sparse, 5 fp6 digits, 6 / 7 the composed polyphase tables, 8 fp6 digits of the 32-bit taps -- it depends on
                               // channels, depth, gain and dither, not only on the filter
    uint32_t reserved;
    uint64_t fir_bytes;
    uint64_t resamp_bytes;
};

}  // namespace d2d
