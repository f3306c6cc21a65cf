/* synth.c -- TEST INFRASTRUCTURE ONLY: deterministic synthetic DSD for tests and bench.py.
 *
 * A plain 2nd-order 1-bit delta-sigma modulator driven by a sine or by pink noise, standing in for
 * the reference's fixtures at sizes the fixtures do not reach (1 kHz sine at 0.352 of
 * full scale, pink noise at ~0.098 in-band RMS -- both levels measured from /root/reference/test/).
 * Nothing here comes from the reference; the bit packing convention (MSB- or LSB-first inside a
 * byte) is the one src/main.rs:70-73 documents.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static inline uint64_t sm64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* kind 0: sine(freq, amp, phase) ; kind 1: pink noise with RMS ~ amp ; kind 2: silence (idle tone) */
void synth_dsd(int kind, uint64_t seed, double amp, double freq, double phase, double fs,
               size_t nbytes, int msb_first, uint8_t* out) {
    double i1 = 0.0, i2 = 0.0;
    uint64_t st = seed * 0x2545F4914F6CDD1Dull + 12345;
    double b0 = 0, b1 = 0, b2 = 0;           /* Paul Kellet's economy pink filter state */
    const double w = 2.0 * M_PI * freq / fs;
    double hold = 0.0, sn = 0.0, cs = 1.0;
    const double cw = cos(w), sw = sin(w);
    size_t t = 0;
    for (size_t j = 0; j < nbytes; ++j) {
        unsigned byte = 0;
        for (int k = 0; k < 8; ++k, ++t) {
            double x;
            if (kind == 0) {
                if ((t & 4095) == 0) { sn = sin(w * (double)t + phase); cs = cos(w * (double)t + phase); }
                x = amp * sn;
                double ns = sn * cw + cs * sw; cs = cs * cw - sn * sw; sn = ns;   /* rotate by w */
            }
            else if (kind == 1) {
                if ((t & 31) == 0) {          /* noise generated at fs/32 and held: band-limited enough */
                    double wn = ((double)(sm64(&st) >> 11) * 0x1p-53) * 2.0 - 1.0;
                    b0 = 0.99765 * b0 + wn * 0.0990460;
                    b1 = 0.96300 * b1 + wn * 0.2965164;
                    b2 = 0.57000 * b2 + wn * 1.0526913;
                    hold = (b0 + b1 + b2 + wn * 0.1848) * amp * 0.63;
                }
                x = hold;
            } else x = 0.0;
            if (x > 0.6) x = 0.6; if (x < -0.6) x = -0.6;
            double v = i2 >= 0.0 ? 1.0 : -1.0;
            i1 += x - v;
            i2 += i1 - 2.0 * v;              /* Boser-Wooley style 2nd-order loop, stable for |x| < ~0.7 */
            if (v > 0) byte |= msb_first ? (0x80u >> k) : (1u << k);
        }
        out[j] = (uint8_t)byte;
    }
}
