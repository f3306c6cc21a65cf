/* d2d_oracle.c -- TEST INFRASTRUCTURE ONLY (see d2d_oracle.h: "PARITY UNPINNED").
 *
 * Plain-C f64 restatement of the per-block DSD->PCM path behind Rdsd2Pcm::do_conversion
 * (/root/reference/src/main.rs:345,429).  Each function names the reference evidence it follows;
 * where the reference is silent the published dsd2pcm algorithm (README.md:240) is followed and
 * marked [lineage]; anything that is this build's own definition is marked [own].
 *
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared -o liboracle.so d2d_oracle.c -lm
 * (-ffp-contract=off: every f64 operation below is a single IEEE operation, fma() only where
 * written, so the HIP engine can reproduce the same bits.)
 */
#include "d2d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../filters/filter_tables.inc"

#define DSD64_RATE 2822400u
#define IDLE_BYTE 0x69u /* [lineage] dsd2pcm resets its FIFO to 0x69: a DC-free idle pattern */

struct orc_ctx {
    orc_params p;
    /* integer decimator (44.1k family, or stage A of the 48k cascade) */
    const d2d_filter_def* f;
    int M, Mb, N, Wb, S;
    double* taps;     /* N full taps as f64 (exact: q * 2^-S) */
    double* lut;      /* (Wb/2) x 256, half tables, canonical (MSB-first-in-time) bytes */
    /* rational stage B (48k family) */
    const d2d_resamp_def* r; /* NULL for the 44.1k family */
    /* derived */
    uint32_t B;       /* effective block size: 1 for interleaved (README.md:9) */
    double gain;      /* 10^(level/20)  (src/main.rs:107-110) */
    uint32_t C;
    /* state per channel */
    size_t keep;      /* bytes of history kept per channel */
    uint8_t* hist_raw;/* C x keep, raw bytes as fed (bit order as in the stream) */
    uint64_t pos;     /* bytes consumed per channel so far */
    uint64_t nfir;    /* FIR outputs produced so far (per channel) */
    double* xhist;    /* C x P   last P stage-A outputs (48k only) */
    uint64_t nres;    /* stage-B outputs so far */
    double* peak;     /* C */
    double* ns_err;   /* C x 2: the last two requantisation errors of the noise shaper ('N') */
    /* streaming path (orc_translate_stream): integer byte tables in the stream's own bit order, buffers kept between calls */
    int32_t* ilut;    /* Wb x 256 */
    uint8_t* sbuf; size_t sbuf_cap;
    /* DSD64 / DSD128 -> 96 / 192 / 384 kHz: the cascade composed into one polyphase filter on the bits (f and r then only count samples) */
    const d2d_poly_def* poly;
    int cascade_mode; /* orc_use_cascade: the two-stage definition these rates had before (study mode) */
    double* taps_f64; /* orc_set_half_taps on a cascade context: stage A's f64 taps (study mode) */
    int fine;         /* orc_use_fine_taps: the taps are the 32-bit grid's */
    int rs_f64;       /* orc_use_f64_resamp_coef: stage B with the design's f64 coefficients (what the 2^-28 grid costs) */
};

static uint8_t bitrev8(uint8_t v) {
    v = (uint8_t)((v >> 4) | (v << 4));
    v = (uint8_t)(((v & 0xCC) >> 2) | ((v & 0x33) << 2));
    v = (uint8_t)(((v & 0xAA) >> 1) | ((v & 0x55) << 1));
    return v;
}

/* [own] counter-based dither generator.  One 32-bit word per output sample, a pure function of
 * (seed, channel, n): key = splitmix64 finaliser of (seed, channel) -> k32 and an odd step for the
 * high half of n; word = lowbias32(lo32(n) + k32 + hi32(n) * kstep)  (lowbias32: C. Wellons'
 * 2-multiply integer hash).  The reference locks rand 0.8.5 (Cargo.lock:546-552) with unknown
 * seeding inside rdsd2pcm, so its noise sequence is unobservable; bit-exactness is defined against
 * THIS generator, chosen to be cheap on a GPU lane (two 32-bit multiplies). */
static uint64_t splitmix_fin(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint64_t orc_rng_key(uint64_t seed, uint32_t channel) {
    return splitmix_fin((seed ^ ((uint64_t)channel * 0xD1B54A32D192ED03ull)) + 0x9E3779B97F4A7C15ull);
}
uint32_t orc_rng(uint64_t seed, uint32_t channel, uint64_t n) {
    uint64_t k = orc_rng_key(seed, channel);
    uint32_t k32 = (uint32_t)(k >> 32), kstep = (uint32_t)k | 1u;
    uint32_t x = (uint32_t)n + k32 + (uint32_t)(n >> 32) * kstep;
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* Which (filter, dsd rate, output rate) combinations exist: src/main.rs:62-67,85-92;
 * README.md:129-134,146-152; test_all_44k_mults.sh, test_all_48k_mults.sh. */
static int select_filters(const orc_params* p, const d2d_filter_def** f, const d2d_resamp_def** r, const d2d_poly_def** pl,
                          const char** err) {
    *f = NULL; *r = NULL; *pl = NULL;
    if (p->dsd_rate != 1 && p->dsd_rate != 2 && p->dsd_rate != 4 && p->dsd_rate != 8) {
        *err = "Invalid DSD rate; must be 1, 2, 4 or 8"; return -1;
    }
    uint64_t fs = (uint64_t)DSD64_RATE * p->dsd_rate;
    uint32_t o = p->output_rate;
    char type = (char)p->filter;
    int M = 0;
    if (o == 88200 || o == 176400 || o == 352800 || o == 705600 || o == 1411200) {
        if (p->dsd_rate == 8 && o != 352800) { *err = "DSD512 input: only 352800 output is available"; return -2; }
        if (o == 705600 && !(p->dsd_rate == 2 || p->dsd_rate == 4)) { *err = "705600 output needs DSD128 or DSD256 input"; return -2; }
        if (o == 1411200 && p->dsd_rate != 4) { *err = "1411200 output needs DSD256 input"; return -2; }
        M = (int)(fs / o);
        if (type == 'X' && !(p->dsd_rate == 1 && o <= 352800)) { *err = "XLD filter: DSD64 input and 88200/176400/352800 output only"; return -3; }
        if (type == 'D' && !(p->dsd_rate == 1 && o == 352800)) { *err = "dsd2pcm filter: DSD64 input and 352800 output only"; return -3; }
        if (type == 'C' && !(p->dsd_rate == 2 && o <= 352800)) { *err = "Chebyshev filter: DSD128 input and 88200/176400/352800 output only"; return -3; }
    } else if (o == 96000 || o == 192000 || o == 384000) {
        /* 48k multiples: cascaded gentle FIRs (README.md:230).  Stage A -> 352.8 kHz, stage B L/147. */
        if (type != 'E') { *err = "48 kHz multiples are only available with the equiripple filter"; return -3; }
        M = 8 * (int)p->dsd_rate;
        type = 'A';
        for (int i = 0; i < D2D_NUM_RESAMPLERS; ++i)
            if ((uint32_t)D2D_RESAMPLERS[i].out_rate == o) *r = &D2D_RESAMPLERS[i];
        /* [own] DSD64 / DSD128 input: the two stages composed into ONE polyphase filter on the bits (filters/filter_tables.inc: D2D_POLYS) */
        for (int i = 0; i < D2D_NUM_POLYS; ++i)
            if ((uint32_t)D2D_POLYS[i].out_rate == o && (uint32_t)D2D_POLYS[i].dsd_rate == p->dsd_rate) *pl = &D2D_POLYS[i];
    } else { *err = "Invalid output rate"; return -2; }
    if (type != 'E' && type != 'X' && type != 'D' && type != 'C' && type != 'A') { *err = "Invalid filter type"; return -3; }
    for (int i = 0; i < D2D_NUM_FILTERS; ++i)
        if (D2D_FILTERS[i].type == type && D2D_FILTERS[i].M == M) *f = &D2D_FILTERS[i];
    if (!*f) { *err = "no filter table for this combination"; return -3; }
    return 0;
}

int orc_create(const orc_params* p, orc_ctx** out, const char** err) {
    static const char* dummy; if (!err) err = &dummy;
    *out = NULL;
    if (p->channels < 1 || p->channels > 64) { *err = "Invalid channel count"; return -4; }
    if (p->bit_depth != 16 && p->bit_depth != 20 && p->bit_depth != 24 && p->bit_depth != 32) { *err = "Invalid bit depth; must be 16, 20, 24 or 32"; return -5; }
    if (p->dither != 'T' && p->dither != 'R' && p->dither != 'F' && p->dither != 'X' && p->dither != 'N') { *err = "Invalid dither type; must be T, R, F, or X"; return -6; } /* src/main.rs:176-180 */
    if (p->fmt > 1) { *err = "Invalid format; must be I (interleaved) or P (planar)"; return -7; }  /* src/main.rs:187-190 */
    if (p->fmt == 1 && p->block_size == 0) { *err = "Invalid block size"; return -8; }
    const d2d_filter_def* f; const d2d_resamp_def* r; const d2d_poly_def* pl;
    int rc = select_filters(p, &f, &r, &pl, err);
    if (rc) return rc;
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(*c));
    c->p = *p; c->f = f; c->r = r; c->poly = pl;
    c->M = f->M; c->Mb = f->M / 8; c->N = f->ntaps; c->Wb = f->ntaps / 8; c->S = f->S;
    c->C = p->channels;
    c->B = p->fmt == 0 ? 1u : p->block_size;   /* README.md:9: block size 1 for interleaved */
    c->gain = pow(10.0, p->level_db / 20.0);
    /* full taps: 2nd half stored, centre outward (README.md:252) */
    c->taps = (double*)malloc(sizeof(double) * (size_t)c->N);
    double scale = ldexp(1.0, -c->S);
    for (int k = 0; k < c->N / 2; ++k) {
        double v = (double)f->half[k] * scale;
        c->taps[c->N / 2 + k] = v;
        c->taps[c->N / 2 - 1 - k] = v;
    }
    /* [lineage] dsd2pcm ctables: one 256-entry table per 8 taps of the stored half */
    int nt = c->Wb / 2;
    c->lut = (double*)malloc(sizeof(double) * 256 * (size_t)nt);
    for (int i = 0; i < nt; ++i)
        for (int v = 0; v < 256; ++v) {
            double acc = 0.0;
            for (int m = 0; m < 8; ++m)
                acc += (double)(((v >> (7 - m)) & 1) * 2 - 1) * c->taps[c->N / 2 + 8 * i + m];
            c->lut[i * 256 + v] = acc;
        }
    c->keep = (size_t)c->Wb + (size_t)c->Mb;
    if (pl) {   /* the oldest bit an output of the next call can need lies NP - D + M bits before the call's first byte */
        size_t k = (size_t)(pl->NP - pl->D + c->M + 7) / 8 + 2;
        if (k > c->keep) c->keep = k;
    }
    c->hist_raw = (uint8_t*)malloc(c->keep * c->C);
    memset(c->hist_raw, p->endianness ? IDLE_BYTE : bitrev8(IDLE_BYTE), c->keep * c->C);
    if (r) c->xhist = (double*)calloc((size_t)r->P * c->C, sizeof(double));
    c->peak = (double*)calloc(c->C, sizeof(double));
    c->ns_err = (double*)calloc((size_t)c->C * 2, sizeof(double));
    *out = c;
    return 0;
}

void orc_destroy(orc_ctx* c) {
    if (!c) return;
    free(c->taps_f64); free(c->taps); free(c->lut); free(c->hist_raw); free(c->xhist); free(c->peak); free(c->ns_err); free(c->ilut); free(c->sbuf); free(c);
}

size_t orc_frame_bytes(const orc_ctx* c) {
    size_t b = c->p.bit_depth == 16 ? 2 : (c->p.bit_depth == 32 ? 4 : 3); /* 20-bit rides in 3 bytes: build_test_mono.sh:3-8 */
    return b * c->C;
}

static uint64_t fir_outputs_after(const orc_ctx* c, uint64_t pos_bytes) { return pos_bytes / (uint64_t)c->Mb; }
static uint64_t res_outputs_after(const orc_ctx* c, uint64_t nx) {
    if (!c->r) return nx;
    if (nx == 0) return 0;
    return (nx * (uint64_t)c->r->L - 1) / (uint64_t)c->r->Mdn + 1;
}

size_t orc_max_frames(const orc_ctx* c, size_t bytes_per_channel) {
    uint64_t nx = fir_outputs_after(c, c->pos + bytes_per_channel);
    return (size_t)(res_outputs_after(c, nx) - c->nres);
}

/* Channel c's byte j of a call holding L bytes per channel.  Planar = [ch0 blk][ch1 blk]...,
 * interleaved = c0 c1 c0 c1 (block size 1): README.md:9; src/main.rs:54-56,75-78.  [own] a final
 * short block keeps the same shape with the shorter length. */
static size_t layout_addr(uint32_t C, uint32_t B, size_t L, uint32_t ch, size_t j) {
    size_t blk = j / B, off = j % B;
    size_t blen = L - blk * B; if (blen > B) blen = B;
    return blk * (size_t)B * C + (size_t)ch * blen + off;
}

/* bit t (time order) of a raw byte buffer: MSB-first or LSB-first inside each byte
 * (src/main.rs:70-73; DSF is LSB-first, DFF MSB-first -- SURVEY 4.3 measurements) */
static int raw_bit(const uint8_t* raw, size_t t, int msb_first) {
    uint8_t b = raw[t >> 3];
    return msb_first ? (b >> (7 - (t & 7))) & 1 : (b >> (t & 7)) & 1;
}

/* One FIR output, direct form: y = sum_j h[j] * s[e*8 - N + j], s = 2*bit-1.  f64 accumulate
 * (README.md:230,236).  `e` = index one past the newest byte of the window inside `raw`. */
static double fir_direct(const orc_ctx* c, const uint8_t* raw, size_t e) {
    double acc = 0.0;
    size_t t0 = e * 8 - (size_t)c->N;
    for (int j = 0; j < c->N; ++j)
        acc += raw_bit(raw, t0 + (size_t)j, (int)c->p.endianness) ? c->taps[j] : -c->taps[j];
    return acc;
}

/* Same number through the dsd2pcm structure [lineage]: newer half bytes index the half tables
 * directly, older half bytes are bit-reversed and index the same tables (filter symmetry). */
static double fir_lut(const orc_ctx* c, const uint8_t* canon, const uint8_t* canon_rev, size_t e) {
    double acc = 0.0;
    int nt = c->Wb / 2;
    const uint8_t* mid = canon + e - (size_t)nt;
    const uint8_t* midr = canon_rev + e - (size_t)nt;
    for (int i = 0; i < nt; ++i)
        acc += c->lut[i * 256 + mid[i]] + c->lut[i * 256 + midr[-1 - i]];
    return acc;
}

/* Dither + requantise one sample.  `v` = FIR (or cascade) output times the level gain.
 * Depths and containers: src/main.rs:58-60, build_test_*.sh (s16le/s24le/f32le; 20 bit in 24).
 * Dither kinds: src/main.rs:171-181, README.md:11-12,236. */
static void emit_sample(orc_ctx* c, double y, uint32_t ch, uint64_t n, uint8_t* dst) {
    uint32_t bits = c->p.bit_depth;
    uint32_t rnd = orc_rng(c->p.seed, ch, n);
    if (bits == 32) {
        double x = y * c->gain;
        if (c->p.dither == 'F') {
            /* Airwindows "Dither Float" (README.md:236) [lineage: published formula
             *   x += (double(fpd) - 0x7fffffff) * 5.5e-36 * 2^(expon+62),  frexpf(x) -> expon ];
             * [own] fpd comes from the counter generator instead of a running xorshift32, and
             * expon is read from the float's exponent field (0 for zero/denormal). */
            float xf = (float)x; uint32_t fb; memcpy(&fb, &xf, 4);
            int e = (int)((fb >> 23) & 0xFF);
            int expon = e ? e - 126 : 0;
            double t = ((double)rnd - 2147483647.0) * 5.5e-36;
            x = x + ldexp(t, expon + 62);
        } /* [own] T/R/X on float output: plain cast */
        float o = (float)x;
        memcpy(dst, &o, 4);
        return;
    }
    double scale = ldexp(c->gain, (int)bits - 1);  /* exact scaling of the gain */
    double x = y * scale;
    double d = 0.0;
    /* the word's two 16-bit halves are the two uniforms of the triangular pdf; both forms are
     * symmetric about zero, so the dither adds no DC */
    if (c->p.dither == 'T' || c->p.dither == 'N') d = (double)((rnd & 0xFFFFu) + (rnd >> 16) + 1u) * 0x1p-16 - 1.0;   /* triangular, +-1 LSB */
    else if (c->p.dither == 'R') d = (double)(2u * (rnd >> 16) + 1u) * 0x1p-17 - 0.5;          /* rectangular, +-1/2 LSB */
    /* [own] 'F' at an integer depth: no dither */
    /* [own] 'N': TPDF dither inside a second-order error-feedback loop, noise transfer function
     * (1 - z^-1)^2 (an extension: the reference CLI offers T/R/F/X only, src/main.rs:171-181; BASELINE
     * config 3 and the north star ask for a noise-shaped variant).  Per channel, in output order:
     *   w = x - (2*e1 - e2);  r = round(w + d);  e = r - w  (taken before clipping);  e2 = e1; e1 = e.
     * Every operation is one IEEE f64 operation in this order.  The loop restarts from e1 = e2 = 0 at
     * every output index that is a multiple of 8192 (0.093 s at 88.2 kHz): the segments are then
     * independent of each other, which is what lets a GPU run them side by side; the restart adds
     * white noise about 32 dB below one LSB of white noise: the 0.2-4 kHz average of the shaped error rises by 1 dB
     * (27.8 instead of 28.8 dB under plain TPDF at 88.2 kHz, 16 bits) and the bottom of the notch around 1 kHz from -45
     * to -34 dB (tests/test_oracle_kat.py::test_noise_shaped_dither_moves_the_error_out_of_band). */
    double w = x;
    if (c->p.dither == 'N') {
        double* e = c->ns_err + 2 * (size_t)ch;
        if ((n & 8191u) == 0) e[0] = e[1] = 0.0;
        double fb = 2.0 * e[0] - e[1];
        w = x - fb;
    }
    double q = w + d;
    /* [lineage] dsd2pcm main.cpp rounds half away from zero and clips */
    double r = q >= 0.0 ? floor(q + 0.5) : ceil(q - 0.5);
    if (c->p.dither == 'N') {
        double* e = c->ns_err + 2 * (size_t)ch;
        e[1] = e[0];
        e[0] = r - w;
    }
    double lim = ldexp(1.0, (int)bits - 1);
    if (r > lim - 1.0) r = lim - 1.0;
    if (r < -lim) r = -lim;
    int32_t iv = (int32_t)r;
    if (bits == 16) { dst[0] = (uint8_t)iv; dst[1] = (uint8_t)(iv >> 8); return; }
    if (bits == 20) iv = iv * 16;  /* 20 significant bits in the top of a 24-bit container */
    dst[0] = (uint8_t)iv; dst[1] = (uint8_t)(iv >> 8); dst[2] = (uint8_t)(iv >> 16);
}

int orc_translate_f64(orc_ctx* c, const uint8_t* dsd, size_t L, void* pcm_out, size_t cap,
                      double* f64_out, size_t* frames_out) {
    uint32_t C = c->C;
    size_t keep = c->keep;
    uint64_t nfir1 = fir_outputs_after(c, c->pos + L);
    size_t nx = (size_t)(nfir1 - c->nfir);
    uint64_t nres1 = res_outputs_after(c, nfir1);
    size_t nframes = (size_t)(nres1 - c->nres);
    size_t fb = orc_frame_bytes(c), sb = fb / C;
    if (frames_out) *frames_out = 0;
    if (nframes * fb > cap) return -20;
    uint8_t* raw = (uint8_t*)malloc(keep + L + 1);
    uint8_t* canon = (uint8_t*)malloc(keep + L + 1);
    uint8_t* canon_rev = (uint8_t*)malloc(keep + L + 1);
    double* x = (double*)malloc(sizeof(double) * (nx + 1 + (c->r ? (size_t)c->r->P : 0)));
    for (uint32_t ch = 0; ch < C; ++ch) {
        /* a2: select the channel's bytes (planar / interleaved) behind the kept history */
        memcpy(raw, c->hist_raw + (size_t)ch * keep, keep);
        for (size_t j = 0; j < L; ++j) raw[keep + j] = dsd[layout_addr(C, c->B, L, ch, j)];
        for (size_t j = 0; j < keep + L; ++j) {
            uint8_t cb = c->p.endianness ? raw[j] : bitrev8(raw[j]);  /* canonical = MSB first in time */
            canon[j] = cb; canon_rev[j] = bitrev8(cb);
        }
        if (c->poly && !c->cascade_mode) {
            /* a4 [own]: y[m] = sum_j c[rho][j] s[q + D - j],  Mp m = Lp q + rho,  c = Q * 2^-S: the exact integer sum Q s, converted once.
             * An output exists as soon as the two-stage form of these rates would have produced it (nres counts ceil(n_x L / 147)); its
             * newest bit q + D (D < 0) then lies inside the bytes fed so far. */
            const d2d_poly_def* pl = c->poly;
            const double yscale = ldexp(1.0, -pl->S);
            for (size_t o = 0; o < nframes; ++o) {
                const uint64_t m = c->nres + o;
                const uint64_t t = m * (uint64_t)pl->Mp;
                const int64_t q = (int64_t)(t / (uint64_t)pl->Lp); const int rho = (int)(t % (uint64_t)pl->Lp);
                const int32_t* g = pl->q + (size_t)rho * (size_t)pl->NP;
                /* bit b of the stream is bit b - 8 (pos - keep) of raw[] */
                const int64_t b0 = q + pl->D - 8 * ((int64_t)c->pos - (int64_t)keep);
                int64_t isum = 0;
                for (int j = 0; j < pl->NP; ++j) {
                    const int64_t b = b0 - j;
                    const int bit = b >= 0 ? raw_bit(raw, (size_t)b, (int)c->p.endianness) : 0;   /* (never taken: keep covers the window) */
                    isum += bit ? (int64_t)g[j] : -(int64_t)g[j];
                }
                const double acc = (double)isum * yscale;      /* exact: |isum| < 2^31 */
                const double v = acc * c->gain;
                const double a = fabs(v); if (a > c->peak[ch]) c->peak[ch] = a;
                if (f64_out) f64_out[o * C + ch] = v;
                if (pcm_out) emit_sample(c, acc, ch, m, (uint8_t*)pcm_out + o * fb + ch * sb);
            }
            memcpy(c->hist_raw + (size_t)ch * keep, raw + L, keep);
            continue;
        }
        /* raw[keep + j] is stream byte pos + j;  output n ends at stream byte (n+1)*Mb */
        double* xs = x + (c->r ? (size_t)c->r->P : 0);
        for (size_t i = 0; i < nx; ++i) {
            uint64_t n = c->nfir + i;
            size_t e = (size_t)((n + 1) * (uint64_t)c->Mb - c->pos) + keep;
            xs[i] = c->p.fir_mode ? fir_lut(c, canon, canon_rev, e) : fir_direct(c, raw, e);
        }
        if (!c->r) {
            for (size_t i = 0; i < nx; ++i) {
                double v = xs[i] * c->gain;
                double a = fabs(v); if (a > c->peak[ch]) c->peak[ch] = a;
                if (f64_out) f64_out[i * C + ch] = v;
                if (pcm_out) emit_sample(c, xs[i], ch, c->nfir + i, (uint8_t*)pcm_out + i * fb + ch * sb);
            }
        } else {
            /* a4: stage B, polyphase L/147: y[m] = sum_k g[phi][k] * x[i_m - k], t = 147 m, i_m = t div L, phi = t mod L.
             * [own] Both factors are dyadic: x = X * 2^-S with the exact stage-A integers X = sum q s, g = G * 2^-T with the 2^-T grid of
             * filters/filter_tables.inc (every phase sums to 1 exactly), so the sum is the exact integer sum G X (|.| < 2^61); it is
             * converted to f64 once (round to nearest even) and scaled by the power of two.  Any order of summation gives this number:
             * the GPU forms it from int8 limbs on the matrix cores. */
            int P = c->r->P, Lr = c->r->L, Md = c->r->Mdn;
            memcpy(x, c->xhist + (size_t)ch * P, sizeof(double) * (size_t)P); /* x[-P..-1] relative to nfir */
            const double xscale = ldexp(1.0, c->S), yscale = ldexp(1.0, -(c->S + c->r->T));
            for (size_t o = 0; o < nframes; ++o) {
                uint64_t m = c->nres + o;
                uint64_t t = m * (uint64_t)Md;
                uint64_t im = t / (uint64_t)Lr; int phi = (int)(t % (uint64_t)Lr);
                const int32_t* g = c->r->q + (size_t)phi * P;
                /* absolute index im -> xs[im - nfir]; history below */
                const double* xp = xs + (ptrdiff_t)(im - c->nfir);
                int64_t isum = 0;
                if (!c->rs_f64) for (int k = 0; k < P; ++k) {
                    const double xi = xp[-k] * xscale;                    /* x * 2^S is an integer, exactly ... */
                    if (xi != (double)(int64_t)xi) return -21;            /* ... or the restatement is broken: say so instead of truncating */
                    isum += (int64_t)g[k] * (int64_t)xi;
                }
                double acc = (double)isum * yscale;
                if (c->rs_f64) {   /* study mode (orc_use_f64_resamp_coef): the design's own f64 coefficients, summed in f64 in this order */
                    const double* gd = c->r->coef + (size_t)phi * P;
                    acc = 0.0;
                    for (int k = 0; k < P; ++k) acc += gd[k] * xp[-k];
                }
                double v = acc * c->gain;
                double a = fabs(v); if (a > c->peak[ch]) c->peak[ch] = a;
                if (f64_out) f64_out[o * C + ch] = v;
                if (pcm_out) emit_sample(c, acc, ch, m, (uint8_t*)pcm_out + o * fb + ch * sb);
            }
            /* keep the last P stage-A outputs */
            memmove(x, x + nx, sizeof(double) * (size_t)P);
            memcpy(c->xhist + (size_t)ch * P, x, sizeof(double) * (size_t)P);
        }
        memcpy(c->hist_raw + (size_t)ch * keep, raw + L, keep);
    }
    c->pos += L; c->nfir = nfir1; c->nres = nres1;
    if (frames_out) *frames_out = nframes;
    free(raw); free(canon); free(canon_rev); free(x);
    return 0;
}

int orc_translate(orc_ctx* c, const uint8_t* dsd, size_t L, void* pcm_out, size_t cap, size_t* frames_out) {
    return orc_translate_f64(c, dsd, L, pcm_out, cap, NULL, frames_out);
}

/* The same conversion organised the way a tuned CPU converter runs it (bench.py's cpu_baseline leg; checked against
 * orc_translate by tests/test_oracle_kat.py): no per-call allocation, the channel's bytes gathered block by block into a
 * buffer kept between calls, one 256-entry INTEGER table per window byte in the stream's own bit order (no bit reversal;
 * the sums are the exact integers sum q*s, so any order gives the oracle's number), then the identical emit_sample().
 * Integer decimator only (44.1k family, no 'N'); anything else is handed to orc_translate. */
int orc_translate_stream(orc_ctx* c, const uint8_t* dsd, size_t L, void* pcm_out, size_t cap, size_t* frames_out) {
    if (c->r || c->p.dither == 'N' || !pcm_out || c->fine) return orc_translate_f64(c, dsd, L, pcm_out, cap, NULL, frames_out);
    const uint32_t C = c->C;
    const size_t keep = c->keep;
    const int Wb = c->Wb;
    if (!c->ilut) {
        c->ilut = (int32_t*)malloc(sizeof(int32_t) * 256 * (size_t)Wb);
        const int h = c->N / 2;
        for (int k = 0; k < Wb; ++k)
            for (int v = 0; v < 256; ++v) {
                int64_t acc = 0;
                for (int m = 0; m < 8; ++m) {
                    const int j = 8 * k + m;                                  /* tap index = time index inside the window */
                    const int64_t q = j >= h ? c->f->half[j - h] : c->f->half[h - 1 - j];
                    const int bit = c->p.endianness ? (v >> (7 - m)) & 1 : (v >> m) & 1;
                    acc += bit ? q : -q;
                }
                c->ilut[k * 256 + v] = (int32_t)acc;
            }
    }
    uint64_t nfir1 = fir_outputs_after(c, c->pos + L);
    size_t nx = (size_t)(nfir1 - c->nfir);
    size_t fb = orc_frame_bytes(c), sb = fb / C;
    if (frames_out) *frames_out = 0;
    if (nx * fb > cap) return -20;
    if (c->sbuf_cap < keep + L + 8) {
        free(c->sbuf);
        c->sbuf_cap = (keep + L + 8) * 2;
        c->sbuf = (uint8_t*)malloc(c->sbuf_cap);
    }
    uint8_t* raw = c->sbuf;
    const double ys = ldexp(1.0, -c->S);
    for (uint32_t ch = 0; ch < C; ++ch) {
        memcpy(raw, c->hist_raw + (size_t)ch * keep, keep);
        if (c->B == 1) {
            for (size_t j = 0; j < L; ++j) raw[keep + j] = dsd[j * C + ch];
        } else {
            for (size_t j = 0; j < L; j += c->B) {                            /* one memcpy per block of the channel */
                size_t blen = L - j; if (blen > c->B) blen = c->B;
                memcpy(raw + keep + j, dsd + (j / c->B) * (size_t)c->B * C + (size_t)ch * blen, blen);
            }
        }
        double pk = c->peak[ch];
        for (size_t i = 0; i < nx; ++i) {
            const uint64_t n = c->nfir + i;
            const uint8_t* w = raw + (size_t)((n + 1) * (uint64_t)c->Mb - c->pos) + keep - (size_t)Wb;
            int32_t a0 = 0, a1 = 0;
            const int32_t* t = c->ilut;
            int k = 0;
            for (; k + 1 < Wb; k += 2) { a0 += t[k * 256 + w[k]]; a1 += t[(k + 1) * 256 + w[k + 1]]; }
            if (k < Wb) a0 += t[k * 256 + w[k]];
            const double y = (double)(a0 + a1) * ys;                          /* exact: |sum q s| < 2^31 */
            const double a = fabs(y * c->gain); if (a > pk) pk = a;
            emit_sample(c, y, ch, n, (uint8_t*)pcm_out + i * fb + ch * sb);
        }
        c->peak[ch] = pk;
        memcpy(c->hist_raw + (size_t)ch * keep, raw + L, keep);
    }
    c->pos += L; c->nfir = nfir1; c->nres = nfir1;
    if (frames_out) *frames_out = nx;
    return 0;
}

double orc_peak(const orc_ctx* c, uint32_t ch) { return ch < c->C ? c->peak[ch] : 0.0; }

/* check_level -> peak dBFS as f32 (src/bin/dsd_levels/main.rs:252-262) */
float orc_peak_dbfs(const orc_ctx* c) {
    double m = 0.0;
    for (uint32_t i = 0; i < c->C; ++i) if (c->peak[i] > m) m = c->peak[i];
    return (float)(20.0 * log10(m));
}

int orc_filter_info(const orc_ctx* c, int* M, int* ntaps, int* S, int* L, int* P) {
    if (M) *M = c->M; if (ntaps) *ntaps = c->N; if (S) *S = c->S;
    if (L) *L = c->r ? c->r->L : 0; if (P) *P = c->r ? c->r->P : 0;
    return 0;
}

double orc_tap(const orc_ctx* c, int i) { return (i >= 0 && i < c->N) ? c->taps[i] : 0.0; }

static void build_byte_tables(orc_ctx* c) {
    int nt = c->Wb / 2;
    for (int i = 0; i < nt; ++i)
        for (int v = 0; v < 256; ++v) {
            double acc = 0.0;
            for (int m = 0; m < 8; ++m)
                acc += (double)(((v >> (7 - m)) & 1) * 2 - 1) * c->taps[c->N / 2 + 8 * i + m];
            c->lut[i * 256 + v] = acc;
        }
}

/* [own] the optional 32-bit tap grid (filters/filter_tables.inc: half32 = q32, taps q32 * 2^-(S+8); the engine's tap_bits = 32).  Any
 * +-1-weighted sum of these taps is exact in f64 (dyadic, sum |q32| < 2^40), so the f64 forms below ARE the integer definition
 * y = (sum q32 s) * 2^-(S+8); level, dither and requantisation as for the 24-bit tables. */
int orc_use_fine_taps(orc_ctx* c) {
    if (!c) return -1;
    if (c->r || c->p.dither == 'N') return -2;   /* 44.1k family, dither T/R/F/X */
    const double scale = ldexp(1.0, -(c->S + 8));
    for (int k = 0; k < c->N / 2; ++k) {
        const double v = (double)c->f->half32[k] * scale;
        c->taps[c->N / 2 + k] = v;
        c->taps[c->N / 2 - 1 - k] = v;
    }
    build_byte_tables(c);
    c->fine = 1;
    return 0;
}

/* [own] study mode for tests/test_tap_grid.py: stage B of the 48k cascade with the f64 coefficients its integer grid was rounded from */
int orc_use_f64_resamp_coef(orc_ctx* c) {
    if (!c || !c->r) return -1;
    c->rs_f64 = 1;
    return 0;
}

/* [own] study mode: DSD64 / DSD128 -> 96 / 192 / 384 kHz the way they were defined before the stages were composed (stage A to 352.8 kHz, stage B
 * polyphase L/147, both on their integer grids); with orc_use_f64_resamp_coef and orc_set_half_taps on top it is the f64 design of that cascade,
 * which tests/test_tap_grid.py measures the composed tables against. */
int orc_use_cascade(orc_ctx* c) {
    if (!c || !c->r) return -1;
    if (c->pos) return -2;
    c->cascade_mode = 1;
    return 0;
}

int orc_set_half_taps(orc_ctx* c, const double* half, int n_half) {
    if (!c || !half || n_half != c->N / 2) return -1;
    if (c->r && !(c->cascade_mode && c->rs_f64)) return -2;   /* stage B sums the exact stage-A integers: f64 taps only in the all-f64 study mode */
    for (int k = 0; k < n_half; ++k) {
        c->taps[c->N / 2 + k] = half[k];
        c->taps[c->N / 2 - 1 - k] = half[k];
    }
    build_byte_tables(c);
    return 0;
}
