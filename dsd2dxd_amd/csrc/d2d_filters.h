// d2d_filters.h -- which tap table serves which (filter type, DSD rate, output rate), and the
// host-side construction of the device tables from the frozen designs in filters/filter_tables.inc.
//
// The legal combinations follow the reference CLI's documentation:
//   /root/reference/src/main.rs:62-67 (filter availability), :85-92 (rates per DSD rate),
//   README.md:129-134,146-152; test_all_44k_mults.sh / test_all_48k_mults.sh (the matrix exercised).
#pragma once
#include <stdint.h>

#include <string>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../filters/filter_tables.inc"
#include "../../include/dsd2dxd_amd.h"

namespace d2d {

struct FilterChoice {
    const d2d_filter_def* fir = nullptr;     // integer decimator (the only stage for 44.1k multiples)
    const d2d_resamp_def* resamp = nullptr;  // stage B for 48k multiples, else null
    // DSD64 / DSD128 -> 48k multiples: the two stages composed into ONE polyphase filter on the bits (d2d_kernels_px.hip).  `fir` and
    // `resamp` then only say how many frames a call yields (an output exists as soon as the two-stage form would have produced it).
    const d2d_poly_def* poly = nullptr;
};

inline int choose_filters(const d2d_params& p, FilterChoice& out, std::string& err) {
    if (p.dsd_rate != 1 && p.dsd_rate != 2 && p.dsd_rate != 4 && p.dsd_rate != 8) {
        err = "Invalid DSD rate; must be 1, 2, 4 or 8";
        return D2D_ERR_PARAM;
    }
    const uint64_t fs = 2822400ull * p.dsd_rate;
    const uint32_t o = p.output_rate;
    char type = (char)p.filter;
    if (type != 'E' && type != 'X' && type != 'D' && type != 'C') {
        err = "Invalid filter type; must be E, X, D or C";
        return D2D_ERR_FILTER;
    }
    int M = 0;
    const bool fam441 = (o == 88200 || o == 176400 || o == 352800 || o == 705600 || o == 1411200);
    const bool fam48 = (o == 96000 || o == 192000 || o == 384000);
    if (fam441) {
        if (p.dsd_rate == 8 && o != 352800) { err = "DSD512 input: only 352800 output is available"; return D2D_ERR_RATE; }
        if (o == 705600 && p.dsd_rate != 2 && p.dsd_rate != 4) { err = "705600 output needs DSD128 or DSD256 input"; return D2D_ERR_RATE; }
        if (o == 1411200 && p.dsd_rate != 4) { err = "1411200 output needs DSD256 input"; return D2D_ERR_RATE; }
        M = (int)(fs / o);
        if (type == 'X' && !(p.dsd_rate == 1 && o <= 352800)) { err = "XLD filter: DSD64 input and 88200/176400/352800 output only"; return D2D_ERR_FILTER; }
        if (type == 'D' && !(p.dsd_rate == 1 && o == 352800)) { err = "dsd2pcm filter: DSD64 input and 352800 output only"; return D2D_ERR_FILTER; }
        if (type == 'C' && !(p.dsd_rate == 2 && o <= 352800)) { err = "Chebyshev filter: DSD128 input and 88200/176400/352800 output only"; return D2D_ERR_FILTER; }
    } else if (fam48) {
        if (type != 'E') { err = "48 kHz multiples are only available with the equiripple filter"; return D2D_ERR_FILTER; }
        M = 8 * (int)p.dsd_rate;   // stage A always lands on 352.8 kHz
        type = 'A';
        for (int i = 0; i < D2D_NUM_RESAMPLERS; ++i)
            if ((uint32_t)D2D_RESAMPLERS[i].out_rate == o) out.resamp = &D2D_RESAMPLERS[i];
        for (int i = 0; i < D2D_NUM_POLYS; ++i)
            if ((uint32_t)D2D_POLYS[i].out_rate == o && (uint32_t)D2D_POLYS[i].dsd_rate == p.dsd_rate) out.poly = &D2D_POLYS[i];
    } else {
        err = "Invalid output rate";
        return D2D_ERR_RATE;
    }
    for (int i = 0; i < D2D_NUM_FILTERS; ++i)
        if (D2D_FILTERS[i].type == type && D2D_FILTERS[i].M == M) out.fir = &D2D_FILTERS[i];
    if (!out.fir) { err = "no filter table for this combination"; return D2D_ERR_FILTER; }
    return D2D_OK;
}

// full tap j (0..N-1) as the integer q_j (tap = q_j * 2^-S); 2nd half stored centre-outward
inline int32_t tap_q(const d2d_filter_def& f, int j) {
    const int h = f.ntaps / 2;
    return j >= h ? f.half[j - h] : f.half[h - 1 - j];
}

// LUT kernel geometry for decimation byte count MB
struct LutLayout {
    int R, LS, pad, nq, ntab;
};
inline LutLayout lut_layout(int MB, int Wb) {
    LutLayout g;
    g.R = MB >= 8 ? 1 : 8 / MB;
    g.LS = MB >= 8 ? MB : 8;
    g.pad = 2 * (g.R - 1) * MB;
    g.nq = (Wb + (g.R - 1) * MB + 7) / 8;
    g.ntab = g.pad + 16 * g.nq;
    return g;
}

// Nibble tables [ntab][16] of f64.  Table pad+2w serves the HIGH nibble of window byte w, table
// pad+2w+1 its LOW nibble, whatever the stream's bit order: for MSB-first streams the high nibble
// holds the four EARLIER samples (bit 7 first), for LSB-first streams the LATER four (bit 4 first).
inline std::vector<double> build_lut_tables(const d2d_filter_def& f, int MB, bool msb_first) {
    const int Wb = f.ntaps / 8;
    const LutLayout g = lut_layout(MB, Wb);
    std::vector<double> t((size_t)g.ntab * 16, 0.0);
    const double scale = 1.0 / (double)(1ull << f.S);   // exact power of two
    for (int w = 0; w < Wb; ++w)
        for (int nib = 0; nib < 2; ++nib)              // 0 = high nibble of the byte, 1 = low nibble
            for (int x = 0; x < 16; ++x) {
                int64_t acc = 0;
                for (int i = 0; i < 4; ++i) {
                    // bit i of the nibble value x  ->  time position inside the byte
                    int tpos = msb_first ? (nib == 0 ? 3 - i : 7 - i) : (nib == 0 ? 4 + i : i);
                    int64_t q = tap_q(f, 8 * w + tpos);
                    acc += ((x >> i) & 1) ? q : -q;
                }
                t[(size_t)(g.pad + 2 * w + nib) * 16 + x] = (double)acc * scale;
            }
    return t;
}

}  // namespace d2d
