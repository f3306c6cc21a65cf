// d2d_mx.h -- geometry of the fp6 x fp4 matrix-core FIR kernel (d2d_kernels_mx.hip), shared by host and device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "d2d_filters.h"
#include "d2d_internal.h"

namespace d2d {

constexpr int MX_FRAG_BYTES = 1536;          // a tap fragment: 64 lanes x 16 bytes, then 64 lanes x 8 bytes (32 e2m3 codes per lane)

// MB = bytes per output (M / 8), NT = taps, G = groups of PH phases per matrix column.  PH = 6 with the 24-bit taps' five base-32 digits (30 of the
// 32 matrix rows), PH = 4 with the 32-bit taps' seven (28 rows; the one-pass form of tap_bits = 32, round 4)
__host__ __device__ constexpr int mx_cs(int MB, int G, int PH = 6) { return PH * G * MB / 4; }                 // column stride in dwords (PH G outputs)
__host__ __device__ constexpr int mx_dly(int MB, int PH = 6) { return PH * MB / 8; }                          // steps (64 bits) between two groups: PH M / 64
__host__ __device__ constexpr int mx_nf(int MB, int NT, int PH = 6) { return (NT + (PH - 1) * 8 * MB + 24 + 63) / 64; }   // fragments: window of PH phases + up to 3 bytes of misalignment
__host__ __device__ constexpr int mx_nstep(int MB, int NT, int G, int PH = 6) { return mx_nf(MB, NT, PH) + mx_dly(MB, PH) * (G - 1); }
__host__ __device__ constexpr int mx_span_dw(int MB, int NT, int G, int PH = 6) { return 31 * mx_cs(MB, G, PH) + 2 * mx_nstep(MB, NT, G, PH); }
__host__ __device__ constexpr int mx_chunks(int MB, int NT, int G, int PH = 6) { return (mx_span_dw(MB, NT, G, PH) + 3 + 3) / 4; }   // + up to 3 dwords in front
__host__ __device__ constexpr int mx_pf(int MB, int NT, int G, int PH = 6) { return (mx_chunks(MB, NT, G, PH) + 63) / 64; }
#ifndef D2D_MX_NOFLAT
#define D2D_MX_NOFLAT 0     // 1: the padded image for every shape (A/B builds)
#endif
#ifndef D2D_MX_FORCEFLAT
#define D2D_MX_FORCEFLAT 0  // 1: the unpadded image for every shape (A/B builds: bank conflicts on the window reads against fewer registers)
#endif
__host__ __device__ constexpr bool mx_flat(int MB, int G, int PH = 6) { return D2D_MX_FORCEFLAT || (!D2D_MX_NOFLAT && mx_cs(MB, G, PH) % 4 == 2); }        // unpadded LDS image (see the kernel)
__host__ __device__ constexpr int mx_stream_bytes(int MB, int NT, int G, int PH = 6) {
    const int dw = 4 * mx_chunks(MB, NT, G, PH);
    if (mx_flat(MB, G, PH)) return 4 * dw + 16;
    return (((dw + dw / mx_cs(MB, G, PH) + 4) * 4 + 15) & ~15) + 16;   // + a dummy slot for the dwords in front of the window
}

// index of the MFMA (step u, group g) among the MFMAs of a chain, in issue order
__host__ __device__ constexpr int mx_slot(int MB, int NT, int G, int u, int g, int PH = 6) {
    int k = 0;
    for (int uu = 0; uu <= u; ++uu)
        for (int gg = 0; gg < G; ++gg) {
            const int f = uu - mx_dly(MB, PH) * gg;
            if (f < 0 || f >= mx_nf(MB, NT, PH)) continue;
            if (uu == u && gg == g) return k;
            ++k;
        }
    return k;
}

// groups per column: three at M = 32; at M = 64 the tap fragments of three groups do not fit the register file next to two accumulator sets
#ifndef D2D_MX_G4
#define D2D_MX_G4 3
#endif
#ifndef D2D_MX_G8
#define D2D_MX_G8 2
#endif
// M = 128: one group (the fragments of a second one would be held for twelve steps: 84 registers)
__host__ __device__ constexpr int mx_g(int MB) { return MB == 4 ? D2D_MX_G4 : MB == 16 ? 1 : D2D_MX_G8; }

struct Mfma2Args;
bool mx_supported(int MB, int NT);                 // is a kernel compiled for this shape?
bool mx_pairs_supported(int MB, int NT, int npairs);   // ... for `npairs` channel pairs per wave (planar multichannel frames)?
bool mx_gain_supported(int MB, int NT);            // ... and its gain flavours (frames at another level than 0 dB)?
bool mx_exact(const d2d_filter_def& f);            // do the digit sums of this table recombine exactly in f32?
bool mx_wide_supported(int MB, int NT);            // ... the one-pass form of the 32-bit tap grid (seven digits, four phases per group)?
bool mx_wide_exact(const d2d_filter_def& f);       // ... and do the seven digit sums of its half32 taps recombine exactly?
int mx_groups(int MB);
void mx_debug_stamps(unsigned long long out[8]);   // diagnostic (-DD2D_MX_STAMPS=1 builds)
std::vector<int8_t> build_mx_tables(const d2d_filter_def& f, bool msb_first, bool wide = false);   // wide: the 32-bit taps (f.half32)
hipError_t launch_fir_mx(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s);

}  // namespace d2d
