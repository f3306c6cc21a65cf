// d2d_px.h -- geometry of the direct polyphase kernels (d2d_kernels_px.hip), shared by host and device.
//
// DSD64 / DSD128 -> 96 / 192 / 384 kHz: the 48k cascade composed into ONE polyphase filter on the bits (filters/filter_tables.inc: D2D_POLYS)
//
//     y[m] = sum_j c[rho][j] s[q + D - j],   Mp m = Lp q + rho,   c = Q 2^-S (24-bit Q, every phase sums to 2^S)
//
// Five consecutive outputs (a GROUP) take 25 of the 32 rows of one fp6 x fp4 matrix instruction (row = output of the group x base-32 digit
// of the tap); a matrix COLUMN serves G consecutive groups, 5 G outputs being a whole number of the filter's cycles (Lp outputs per Mp
// bits), so that every column of every tile sees the same taps at the same places; a wave-tile is 32 columns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../filters/filter_tables.inc"
#include "d2d_internal.h"

namespace d2d {

constexpr int PX_FRAG_BYTES = 1536;          // a tap fragment: 64 lanes x 16 bytes, then 64 lanes x 8 bytes (32 e2m3 codes per lane)
#ifndef D2D_PX_THREADS
#define D2D_PX_THREADS 512     // waves per block x 64: 512 = two waves per SIMD; 768 = three (at most 168 registers), an A/B build
#endif
constexpr int PX_THREADS = D2D_PX_THREADS;

// first bit (relative to the column's first output) of output o's window: floor(o Mp / Lp)
__host__ __device__ constexpr int px_q(int LP, int MP, int o) { return (int)(((long long)o * MP) / LP); }
// steps (64 stream bits) of a column's window that group g's rows touch: [u0, u1]
__host__ __device__ constexpr int px_u0(int LP, int MP, int g) { return px_q(LP, MP, 5 * g) / 64; }
__host__ __device__ constexpr int px_u1(int LP, int MP, int NP, int g) { return (px_q(LP, MP, 5 * g + 4) + NP - 1) / 64; }
__host__ __device__ constexpr int px_tp(int LP, int MP, int NP, int G) { return px_u1(LP, MP, NP, G - 1) + 1; }
__host__ __device__ constexpr bool px_active(int LP, int MP, int NP, int u, int g) { return u >= px_u0(LP, MP, g) && u <= px_u1(LP, MP, NP, g); }
// index of the matrix instruction (step u, group g) in issue order (step-major) = index of its tap fragment in the table
__host__ __device__ constexpr int px_slot(int LP, int MP, int NP, int G, int u, int g) {
    int k = 0;
    for (int uu = 0; uu <= u; ++uu)
        for (int gg = 0; gg < G; ++gg) {
            if (!px_active(LP, MP, NP, uu, gg)) continue;
            if (uu == u && gg == g) return k;
            ++k;
        }
    return k;
}
// ... and the (step, group) of the k-th matrix instruction
__host__ __device__ constexpr int px_slot_u(int LP, int MP, int NP, int G, int k) {
    int c = 0;
    for (int u = 0;; ++u)
        for (int g = 0; g < G; ++g) {
            if (!px_active(LP, MP, NP, u, g)) continue;
            if (c == k) return u;
            ++c;
        }
}
__host__ __device__ constexpr int px_slot_g(int LP, int MP, int NP, int G, int k) {
    int c = 0;
    for (int u = 0;; ++u)
        for (int g = 0; g < G; ++g) {
            if (!px_active(LP, MP, NP, u, g)) continue;
            if (c == k) return g;
            ++c;
        }
}
__host__ __device__ constexpr int px_nslot(int LP, int MP, int NP, int G) {
    int k = 0;
    for (int g = 0; g < G; ++g) k += px_u1(LP, MP, NP, g) - px_u0(LP, MP, g) + 1;
    return k;
}
__host__ __device__ constexpr int px_sbits(int LP, int MP, int G) { return 5 * G * MP / LP; }          // stream bits between two columns
// 16-byte chunks of a tile's stream: up to 127 bits in front of column 0, 31 column strides, the window, one more dword for the funnel shift
__host__ __device__ constexpr int px_chunks(int LP, int MP, int NP, int G) { return (127 + 31 * px_sbits(LP, MP, G) + 64 * px_tp(LP, MP, NP, G) + 32 + 127) / 128; }
__host__ __device__ constexpr int px_stream_bytes(int LP, int MP, int NP, int G) { return 16 * px_chunks(LP, MP, NP, G) + 16; }

// the direct polyphase kernels' launch arguments
struct PxArgs {
    const StreamJob* jobs;     // n0 / nout: the call's first output index and count; e0: bytes per channel consumed before this call
    const void* tables;        // matrix-core kernel: [px_nslot][PX_FRAG_BYTES] tap fragments; plain kernel: int32 Q[Lp][NP]
    uint32_t Lp, Mp, NP;
    int32_t  D, S;
    uint32_t in_channels;      // channels of the input layout
    uint32_t B, keep;          // effective block size, history bytes per channel
    uint32_t msb;              // 1: the stream's bytes hold their first bit in bit 7
    uint32_t il2;              // 1: byte-interleaved STEREO input, both channels converted: the matrix-core kernel's staging pulls the channels apart (B = 1)
    uint32_t to_scratch;       // 1: the exact integers sum Q s go to job.xs (the noise-shaping pass requantises)
    uint32_t cw;               // channels a wave converts per tile (2, or 1 for a mono file)
    uint32_t ngroups;          // channel groups per file: ceil(channels / cw)
    uint32_t nwaves, off_waves, wave_lds, off_out;
    uint32_t dkind;            // integer requantiser: 0 none, 1 triangular, 2 rectangular
    int32_t  fbits;            // x = v * 2^-fbits LSB at unit gain
    int32_t  qmin_i, qmax_i;
    uint32_t qsh;              // 4: 20-bit samples in a 24-bit container
    Epilogue epi;
};

bool px_supported(const d2d_poly_def& p);                                   // is a matrix-core kernel compiled for this table?
bool px_exact(const d2d_poly_def& p);                                       // do the table's base-32 digit sums recombine exactly in f32?
int px_groups(const d2d_poly_def& p);
std::vector<int8_t> build_px_tables(const d2d_poly_def& p);
hipError_t launch_fir_px(PxArgs& a, const d2d_poly_def& p, uint32_t max_nout, uint32_t nfiles, hipStream_t s);
hipError_t launch_poly_plain(PxArgs& a, const d2d_poly_def& p, uint32_t max_nout, uint32_t nstreams, hipStream_t s);

}  // namespace d2d
