// d2d_kernels_mfma.hip -- the 1-bit FIR decimator on the int8 matrix cores (gfx950), exact.
//
// The taps are 32-bit integers q (tap = q*2^-S) split into four balanced int8 limbs, the DSD bits
// are 0/1 bytes, products accumulate in int32: v_mfma_i32_32x32x32_i8 computes
//        D[row][4*ph + limb] += sum_k  bit[row][k] * limb_l(q[tau(k) - ph*M])
// where one matrix ROW is a window of the channel's bit stream that serves EIGHT consecutive outputs
// (phases ph = 0..7; the row advances 8*M bits) and the K dimension walks that window 32 bits at a
// time.  Recombining the limbs (D0 + D1<<8 + D2<<16 + D3<<24) gives acc = sum_k q_k b_k exactly, and
// y = (2*acc - 2^S) * 2^-S is the same number the f64 oracle and the LUT kernel produce.
//
// Bit -> int8 expansion costs two VALU ops per operand register: A_v = (W >> p) & 0x01010101 puts
// bit (8b + p) of the stream dword W into byte b; the tap table is laid out for exactly that K
// order (and for the stream's bit order), so no bit reversal or transposition happens at run time.
//
// Replaces: the per-block translate loop inside Rdsd2Pcm::do_conversion
// (/root/reference/src/main.rs:345,429); the crate that holds it is absent from the reference.
#include <hip/hip_runtime.h>

#include "d2d_device.h"
#include <stdio.h>
#include <stdlib.h>

#include "d2d_mfma.h"

namespace d2d {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MFMA_THREADS = 256;
constexpr int MFMA_WAVES = MFMA_THREADS / 64;

struct MfmaArgs {
    FirArgs f;
    double yscale;        // 2^(1-S):            y = fma(acc, yscale, -1)            (exact)
    double c1, c0;        // c1 = yscale*c0:      x = fma(acc, c1, -c0) == round(y*c0), c0 = scale or gain
    uint32_t U;           // dwords of row window per lane half; K steps = 2U
    uint32_t span;        // logical staged bytes per channel (multiple of 16)
    uint32_t pspan;       // physical LDS bytes per channel (one pad dword per row stride)
    uint32_t ls;          // log2(row stride in dwords) = log2(2*MB)
    uint32_t off_waves;   // LDS: start of the per-wave regions (after the shared tap table, if any)
    uint32_t wave_lds;    // LDS bytes per wave
    uint32_t off_out, off_pk;   // inside a wave's region
    // integer-depth epilogue as data: d = fma(term, dmul, dadd), clamp to [qmin, qmax], << qshift
    double dmul, dadd, qmin, qmax;
    uint32_t dsel;        // 1: triangular term, 0: rectangular term
    uint32_t qmul;        // 16 for 20-bit samples in a 24-bit container, else 1
    uint32_t dbg;         // diagnostic ablation mask (env D2D_DBG), 0 in production
    uint32_t nwaves;      // waves per block (4, or fewer when the shared tap table leaves less LDS)
};

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16 bytes of channel c's stream starting at call-relative byte j (j % 16 == 0).
__device__ __forceinline__ u32x4 load_chunk(const StreamJob* jobs, const StreamJob& j0, uint32_t c, int32_t j,
                                            uint32_t C, uint32_t B, uint32_t keep) {
    const uint32_t L = (uint32_t)j0.L;
    if ((B & 15u) == 0 && j >= 0 && (uint32_t)j + 16 <= L) {
        const uint32_t ju = (uint32_t)j;
        const uint32_t blk = (B & (B - 1)) == 0 ? ju >> (31 - __builtin_clz(B)) : ju / B;
        const uint32_t off = ju - blk * B;
        uint32_t blen = L - blk * B;
        if (blen > B) blen = B;
        if ((blen & 15u) == 0) {
            const uint8_t* p = j0.in + (uint64_t)blk * B * C + (uint64_t)c * blen + off;
            return *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(p));
        }
    }
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int b = 0; b < 16; ++b) {
        const uint32_t x = stream_byte(jobs[c], C, B, keep, j + b) << (8 * (b & 3));
        if ((b >> 2) == 0) w[0] |= x; else if ((b >> 2) == 1) w[1] |= x; else if ((b >> 2) == 2) w[2] |= x; else w[3] |= x;
    }
    return u32x4{w[0], w[1], w[2], w[3]};
}

constexpr int MFMA_PF = 3;   // 16-byte chunks per lane fetched one wave-tile ahead

// One wave = one independent worker: it converts wave-tiles of 256 frames x C channels (32 matrix
// rows x 8 phases per channel), staging the packed bits in its own LDS slice, with the next
// wave-tile's bytes already in flight while the current one is multiplied.  No block barrier in the
// loop, so the four waves of a block sit in different phases and the matrix pipe, the VALU (bit
// expansion, dither, requantise) and the memory pipe overlap.
//
// UCT > 0: the tap fragments of all 2*UCT K steps stay in registers for the whole kernel (small
// filters, fully unrolled).  UCT == 0: K steps counted at run time, fragments streamed from a
// block-shared LDS table one pair ahead of the MFMAs that use them.
template <int MB, int UCT>
__global__ __launch_bounds__(UCT > 0 ? 256 : 1024) void d2d_fir_mfma_kernel(MfmaArgs m) {
    const FirArgs& a = m.f;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t C = a.epi.channels, sb = a.epi.sample_bytes, fbytes = sb * C;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* wbase = smem + m.off_waves + wave * m.wave_lds;
    uint32_t* inw = reinterpret_cast<uint32_t*>(wbase);
    uint8_t* outw = wbase + m.off_out;
    double* pkw = reinterpret_cast<double*>(wbase + m.off_pk);
    const StreamJob* jobs = a.jobs + (size_t)blockIdx.y * C;
    const StreamJob j0 = jobs[0];          // in, L, e0, n0, nout are common to a file's channels

    v4i breg[UCT > 0 ? 2 * UCT : 1];
    if constexpr (UCT > 0) {   // tap fragments: L2 -> registers once
        const v4i* s = reinterpret_cast<const v4i*>(a.tables) + lane;
#pragma unroll
        for (int t = 0; t < 2 * UCT; ++t) breg[t] = s[t * 64];
        // retire these loads here and hide their origin from the compiler: otherwise every MFMA in
        // the loop waits on the vector-memory counter, which by then also holds the next
        // wave-tile's prefetch (the counter is in order), and the prefetch stops being one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2 * UCT; ++t) asm volatile("" : "+v"(breg[t]));
    } else {                   // tap fragments: L2 -> LDS once per block
        const uint4* s = reinterpret_cast<const uint4*>(a.tables);
        uint4* d = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < (a.ksteps + 2) * 64; i += blockDim.x) d[i] = s[i];
        __syncthreads();
    }
    for (uint32_t c = 0; c < C; ++c) pkw[c * 64 + lane] = 0.0;
    // per-channel dither keys: global -> this wave's LDS once (inside the loop they would be vector
    // memory loads queued behind the prefetch)
    uint32_t* rngw = reinterpret_cast<uint32_t*>(pkw + C * 64);
    if (lane < C) {
        rngw[lane * 4 + 0] = jobs[lane].rng_key;
        rngw[lane * 4 + 1] = jobs[lane].rng_kstep;
        rngw[lane * 4 + 2] = jobs[lane].rng_lo0;
    }
    wave_sync();

    const uint32_t Urt = UCT > 0 ? (uint32_t)UCT : m.U;
    const uint32_t nwt = (j0.nout + 255u) >> 8;            // wave-tiles in this file
    const uint32_t wstride = gridDim.x * m.nwaves;
    const uint32_t cpc = m.span >> 4;                      // 16-byte chunks per channel
    const uint32_t nch = C * cpc;
    const uint32_t ls = m.ls;
    const uint32_t pdw = m.pspan >> 2;                     // physical dwords per channel
    // the chunks this lane stages are the same for every wave-tile
    uint32_t pf_c[MFMA_PF], pf_q[MFMA_PF];
#pragma unroll
    for (int i = 0; i < MFMA_PF; ++i) {
        const uint32_t ch = lane + 64 * i;
        pf_c[i] = ch / cpc;
        pf_q[i] = ch - pf_c[i] * cpc;
    }
    u32x4 pf[MFMA_PF];
    uint32_t wt = blockIdx.x * m.nwaves + wave;
    auto prefetch = [&](uint32_t w) {
        const int32_t ab = (int32_t)((j0.e0 - (int64_t)a.Wb + (int64_t)w * (256 * MB)) & ~(int64_t)15);
#pragma unroll
        for (int i = 0; i < MFMA_PF; ++i)
            if (lane + 64 * i < nch) pf[i] = load_chunk(jobs, j0, pf_c[i], ab + (int32_t)(pf_q[i] * 16), C, a.B, a.keep);
    };
    auto write_chunk = [&](uint32_t c, uint32_t q, const u32x4& v) {
        uint32_t* base = inw + c * pdw;
        const uint32_t Ld = q * 4;
        base[Ld + (Ld >> ls)] = v.x;
        base[Ld + 1 + ((Ld + 1) >> ls)] = v.y;
        base[Ld + 2 + ((Ld + 2) >> ls)] = v.z;
        base[Ld + 3 + ((Ld + 3) >> ls)] = v.w;
    };
    if (wt < nwt) prefetch(wt);

    const uint32_t K1 = 0x01010101u;
    const uint32_t r = lane & 31, h = lane >> 5;
    for (; wt < nwt; wt += wstride) {
        const int64_t tile_first = j0.e0 - (int64_t)a.Wb + (int64_t)wt * (256 * MB);
        const int64_t abeg = tile_first & ~(int64_t)15;
        const uint32_t d = (uint32_t)(tile_first - abeg);
        const uint32_t sh = d & 3u;
        // staged bytes of this wave-tile: registers -> LDS (row stride padded by one dword)
#pragma unroll
        for (int i = 0; i < MFMA_PF; ++i)
            if (lane + 64 * i < nch && !(m.dbg & 8)) write_chunk(pf_c[i], pf_q[i], pf[i]);
        for (uint32_t ch = lane + 64 * MFMA_PF; ch < nch; ch += 64) {
            const uint32_t c = ch / cpc, q = ch - c * cpc;
            write_chunk(c, q, load_chunk(jobs, j0, c, (int32_t)abeg + (int32_t)(q * 16), C, a.B, a.keep));
        }
        if (wt + wstride < nwt && !(m.dbg & 8)) prefetch(wt + wstride);   // next wave-tile's bytes: in flight during the MFMAs
        wave_sync();

        const uint32_t LB = (d >> 2) + r * (2 * MB) + h * Urt;   // logical dword of this lane's first row word
        for (uint32_t c = 0; c < C; ++c) {
            const uint32_t* rowp = inw + c * pdw;
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (m.dbg & 1) { acc[0] = (int)lane; } else
            if constexpr (UCT > 0) {
                uint32_t rw[UCT + 1];
#pragma unroll
                for (int u = 0; u <= UCT; ++u) rw[u] = rowp[LB + u + ((LB + u) >> ls)];
                __builtin_amdgcn_sched_barrier(0);   // all row words in flight before the first MFMA waits
#pragma unroll
                for (int u = 0; u < UCT; ++u) {
                    const uint32_t W = __builtin_amdgcn_alignbyte(rw[u + 1], rw[u], sh);
                    v4i A0 = {(int)(W & K1), (int)((W >> 1) & K1), (int)((W >> 2) & K1), (int)((W >> 3) & K1)};
                    v4i A1 = {(int)((W >> 4) & K1), (int)((W >> 5) & K1), (int)((W >> 6) & K1), (int)((W >> 7) & K1)};
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(breg[2 * u], A0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(breg[2 * u + 1], A1, acc, 0, 0, 0);
                }
            } else {
                const v4i* bp = reinterpret_cast<const v4i*>(smem) + lane;
                uint32_t cur = rowp[LB + (LB >> ls)], nxt = rowp[LB + 1 + ((LB + 1) >> ls)];
                v4i B0 = bp[0], B1 = bp[64];
                for (uint32_t u = 0; u < Urt; ++u) {
                    // operands of the next pair are fetched before this pair's MFMAs (one extra row
                    // dword and one extra zero fragment pair exist past the end)
                    const uint32_t Ln = LB + u + 2;
                    const uint32_t nn = rowp[Ln + (Ln >> ls)];
                    const v4i B0n = bp[(2 * u + 2) * 64], B1n = bp[(2 * u + 3) * 64];
                    const uint32_t W = __builtin_amdgcn_alignbyte(nxt, cur, sh);
                    v4i A0 = {(int)(W & K1), (int)((W >> 1) & K1), (int)((W >> 2) & K1), (int)((W >> 3) & K1)};
                    v4i A1 = {(int)((W >> 4) & K1), (int)((W >> 5) & K1), (int)((W >> 6) & K1), (int)((W >> 7) & K1)};
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(B0, A0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(B1, A1, acc, 0, 0, 0);
                    cur = nxt; nxt = nn; B0 = B0n; B1 = B1n;
                }
            }
            // The taps are the A operand (matrix row = 4*phase + limb), the bits the B operand (matrix
            // column = stream row), so D[4*ph + limb][r] puts ALL four limbs of four outputs into this
            // lane's own registers: lane (r, h) holds phases ph = h + 2k in acc[4k .. 4k+3].  No
            // transposition; the four samples below are independent straight-line code.
            const StreamJob* jc = jobs + c;
            const uint32_t rkey = rngw[c * 4], rstep = rngw[c * 4 + 1], rlo0 = rngw[c * 4 + 2];
            double pkx = pkw[c * 64 + lane];
            double xv[4];
            uint32_t zv[4], ov[4];
            bool valid[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lo = acc[4 * k] + (acc[4 * k + 1] << 8), hi = acc[4 * k + 2] + (acc[4 * k + 3] << 8);
                const double accd = fma((double)hi, 65536.0, (double)lo);     // exact: sum_k q_k b_k
                ov[k] = 8 * r + h + 2 * k;
                const uint32_t nl = wt * 256 + ov[k];
                valid[k] = nl < j0.nout;
                // x = y*c0 with ONE rounding: acc*c1 - c0 is exactly y*c0 before the fma rounds
                // (to_scratch: c1 = 2^(1-S), c0 = 1, so x = y exactly)
                xv[k] = fma(accd, m.c1, -m.c0);
                const uint32_t nlo = (uint32_t)j0.n0 + nl;
                uint32_t z = nlo + rkey + (nlo < rlo0 ? rstep : 0u);
                z ^= z >> 16; z *= 0x7feb352dU;
                z ^= z >> 15; z *= 0x846ca68bU;
                z ^= z >> 16;
                zv[k] = z;
            }
            if (m.dbg & 2) { if (xv[0] == 1.2345 && zv[1] == 77 && xv[2] == 3.3 && xv[3] == 4.4 && zv[0]+zv[2]+zv[3] == 5) outw[lane] = 1; } else
            if (a.to_scratch) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (valid[k]) as_global(jc->xs)[wt * 256 + ov[k]] = xv[k];
            } else if (a.epi.bits == 32) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pkx = fmax(pkx, valid[k] ? fabs(xv[k]) : 0.0);
                    *reinterpret_cast<float*>(outw + (size_t)(ov[k] * C + c) * 4) = finish_f32(a.epi, xv[k], zv[k]);
                }
            } else {
                int32_t iv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pkx = fmax(pkx, valid[k] ? fabs(xv[k]) : 0.0);
                    // dither term: T = lo16 + hi16 + 1 (x 2^-16, -1), R = 2*hi16 + 1 (x 2^-17, -1/2), none = 0
                    const uint32_t term = m.dsel ? (zv[k] & 0xFFFFu) + (zv[k] >> 16) + 1u : 2u * (zv[k] >> 16) + 1u;
                    const double dd = fma((double)term, m.dmul, m.dadd);
                    const double q = xv[k] + dd;
                    const double rr = fmax(fmin(trunc(q + copysign(0.5, q)), m.qmax), m.qmin);
                    iv[k] = (int32_t)rr * (int32_t)m.qmul;
                }
                if (sb == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) *reinterpret_cast<uint16_t*>(outw + (size_t)(ov[k] * C + c) * 2) = (uint16_t)iv[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint8_t* p = outw + (size_t)(ov[k] * C + c) * 3;
                        p[0] = (uint8_t)iv[k]; p[1] = (uint8_t)(iv[k] >> 8); p[2] = (uint8_t)(iv[k] >> 16);
                    }
                }
            }
            pkw[c * 64 + lane] = pkx;
        }
        if (!a.to_scratch && !(m.dbg & 4)) {
            wave_sync();
            // the wave-tile's interleaved frames: LDS -> HBM, 16 bytes per lane per store
            const uint32_t left = j0.nout - wt * 256;
            const uint32_t nb = (left < 256u ? left : 256u) * fbytes;
            uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * 256 * fbytes;
            const uint32_t nb16 = nb & ~15u;
            for (uint32_t i = lane * 16; i < nb16; i += 64 * 16)
                *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(g + i)) = *reinterpret_cast<const u32x4*>(outw + i);
            for (uint32_t i = nb16 + lane; i < nb; i += 64) as_global(g)[i] = outw[i];
        }
    }
    if (!a.to_scratch) {
        // peak meter: |x| was tracked in the scaled domain; undo the power-of-two part exactly
        const double unscale = a.epi.bits == 32 ? 1.0 : 1.0 / (double)(1u << (a.epi.bits - 1));
        for (uint32_t c = 0; c < C; ++c) {
            double pk = pkw[c * 64 + lane] * unscale;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) pk = fmax(pk, __shfl_xor(pk, o));
            if (lane == 0 && pk > 0.0)
                atomicMax(reinterpret_cast<unsigned long long*>(jobs[c].peak), (unsigned long long)__double_as_longlong(pk));
        }
    }
}

// ---- host side -------------------------------------------------------------------------------

bool mfma_supported(int M, int N) {
    (void)N;
    return M == 8 || M == 16 || M == 32 || M == 64 || M == 128;
}

MfmaLayout mfma_layout(int M, int N) {
    MfmaLayout g;
    g.M = M; g.N = N;
    const int wd = (N + 7 * M + 31) / 32;   // dwords of one row's window
    const int U = (wd + 1) / 2;
    g.ksteps = 2 * U;
    return g;
}

uint32_t mfma_keep_bytes(const MfmaLayout&, int) { return 0; }

static inline int8_t limb_of(int32_t q, int l) {
    // balanced base-256 digits: q = d0 + d1*2^8 + d2*2^16 + d3*2^24, every d in [-128, 127]
    int64_t v = q;
    int8_t dgt = 0;
    for (int i = 0; i <= l; ++i) {
        int64_t dd = ((v + 128) & 255) - 128;
        dgt = (int8_t)dd;
        v = (v - dd) / 256;
    }
    return dgt;
}

std::vector<int8_t> build_mfma_tables(const d2d_filter_def& f, const MfmaLayout& g, bool msb_first) {
    const int U = g.ksteps / 2;
    std::vector<int8_t> t((size_t)(g.ksteps + 2) * 64 * 16, 0);   // +2 zero steps: read-ahead of the LDS variant
    for (int ks = 0; ks < g.ksteps; ++ks)
        for (int l = 0; l < 64; ++l) {
            const int col = l & 31, h = l >> 5, ph = col >> 2, limb = col & 3;
            for (int j = 0; j < 16; ++j) {
                // which bit of the row's window feeds K slot (ks, h, j): see the kernel's A0/A1
                const int wb = 32 * (h * U + (ks >> 1)) + 8 * (j & 3) + 4 * (ks & 1) + (j >> 2);
                const int tau = msb_first ? (wb & ~7) + 7 - (wb & 7) : wb;   // time order inside the window
                const int tap = tau - ph * g.M;
                int8_t v = 0;
                if (tap >= 0 && tap < f.ntaps) v = limb_of(tap_q(f, tap), limb);
                t[((size_t)ks * 64 + l) * 16 + j] = v;
            }
        }
    return t;
}

// (MB, U) pairs whose tap fragments fit in registers: every filter of filters/filter_tables.inc
// with at most 26 K steps.  Anything else takes the run-time-U kernel.
#define D2D_MFMA_REG_VARIANTS(X) \
    X(1, 3) X(1, 4) X(2, 5) X(2, 6) X(2, 7) X(4, 9) X(4, 10) X(4, 12) X(4, 13)

static bool mfma_has_reg_variant(int MB, int U) {
    static const bool off = getenv("D2D_MFMA_NO_REG") != nullptr;   // diagnostic: force the run-time-U kernel
    if (off) return false;
#define X(mb, u) if (MB == mb && U == u) return true;
    D2D_MFMA_REG_VARIANTS(X)
#undef X
    return false;
}

static void mfma_geometry(const FirArgs& a, const MfmaLayout& g, MfmaArgs& m, size_t& smem) {
    const uint32_t C = a.epi.channels;
    const int MB = g.M / 8;
    m.f = a;
    m.yscale = ldexp(1.0, 1 - a.scale_bits);
    m.c0 = a.to_scratch ? 1.0 : (a.epi.bits == 32 ? a.epi.gain : a.epi.scale);
    m.c1 = m.yscale * m.c0;                       // exact: yscale is a power of two
    m.dsel = a.epi.dither == 'T' ? 1u : 0u;
    m.dmul = a.epi.dither == 'T' ? 0x1p-16 : (a.epi.dither == 'R' ? 0x1p-17 : 0.0);
    m.dadd = a.epi.dither == 'T' ? -1.0 : (a.epi.dither == 'R' ? -0.5 : 0.0);
    const double lim = a.epi.bits == 32 ? 1.0 : (double)(1u << (a.epi.bits - 1));
    m.qmax = lim - 1.0; m.qmin = -lim;
    m.qmul = a.epi.bits == 20 ? 16u : 1u;
    { static const char* e = getenv("D2D_DBG"); m.dbg = e ? (uint32_t)atoi(e) : 0u; }
    m.U = (uint32_t)g.ksteps / 2;
    int ls = 0;
    while ((1 << ls) < 2 * MB) ++ls;
    m.ls = (uint32_t)ls;
    // logical staged bytes per channel: 16-byte alignment slack + 31 row strides + one row window
    // (+2 dwords read ahead) + slack for the in-register byte realignment
    m.span = (16u + 31u * 8u * MB + (2 * m.U + 3) * 4u + 16u + 15u) & ~15u;
    const uint32_t ldw = m.span / 4;
    m.pspan = ((ldw + (ldw >> ls) + 2) * 4u + 15u) & ~15u;
    const bool reg = mfma_has_reg_variant(MB, (int)m.U);
    m.off_waves = reg ? 0u : ((uint32_t)g.ksteps + 2u) * 1024u;   // +2: the zero pair the prefetch reads
    m.off_out = C * m.pspan;
    m.off_pk = m.off_out + ((256u * C * a.epi.sample_bytes + 15u) & ~15u);
    m.wave_lds = m.off_pk + C * 64u * 8u + ((C * 16u + 15u) & ~15u);   // peaks + per-channel dither keys
    m.nwaves = MFMA_WAVES;
    if (!reg) { static const char* e = getenv("D2D_MFMA_WAVES"); m.nwaves = e ? (uint32_t)atoi(e) : 16u; }
    while (m.nwaves > 1 && (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds > 160 * 1024) m.nwaves >>= 1;
    smem = (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds;
}

size_t mfma_smem_bytes(const MfmaLayout& g, uint32_t channels, uint32_t sample_bytes, uint32_t* waves_per_block) {
    FirArgs a{};
    a.epi.channels = channels; a.epi.sample_bytes = sample_bytes; a.epi.bits = 24;
    MfmaArgs m{}; size_t smem = 0;
    mfma_geometry(a, g, m, smem);
    if (waves_per_block) *waves_per_block = m.nwaves;
    return smem;
}

template <int MB, int UCT>
static hipError_t launch_mfma_t(const MfmaArgs& m, size_t smem, uint32_t nwt_max, uint32_t nfiles, hipStream_t s) {
    static int blocks_per_cu = 0, ncu = 0;
    static size_t smem_seen = 0;
    if (blocks_per_cu == 0 || smem != smem_seen) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&d2d_fir_mfma_kernel<MB, UCT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        ncu = prop.multiProcessorCount;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, d2d_fir_mfma_kernel<MB, UCT>, (int)(64 * m.nwaves), smem);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb < 1 ? 1 : nb;
        smem_seen = smem;
    }
    // every wave loops over its share of the wave-tiles: launch what is resident at once
    uint32_t gx = (uint32_t)(ncu * blocks_per_cu) / nfiles;
    if (gx < 1) gx = 1;
    const uint32_t need = (nwt_max + m.nwaves - 1) / m.nwaves;
    if (gx > need) gx = need;
    hipLaunchKernelGGL((d2d_fir_mfma_kernel<MB, UCT>), dim3(gx, nfiles), dim3(64 * m.nwaves), smem, s, m);
    return hipGetLastError();
}

hipError_t launch_fir_mfma(const FirArgs& a, const MfmaLayout& g, uint32_t max_nout, uint32_t nstreams, hipStream_t s) {
    if (nstreams == 0 || max_nout == 0) return hipSuccess;
    const uint32_t C = a.epi.channels;
    const uint32_t nfiles = nstreams / C;
    const int MB = g.M / 8;
    MfmaArgs m{};
    size_t smem = 0;
    mfma_geometry(a, g, m, smem);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    const uint32_t nwt = (max_nout + 255u) / 256u;
#define X(mb, u) if (MB == mb && m.U == u && mfma_has_reg_variant(mb, u)) return launch_mfma_t<mb, u>(m, smem, nwt, nfiles, s);
    D2D_MFMA_REG_VARIANTS(X)
#undef X
    switch (MB) {
        case 1: return launch_mfma_t<1, 0>(m, smem, nwt, nfiles, s);
        case 2: return launch_mfma_t<2, 0>(m, smem, nwt, nfiles, s);
        case 4: return launch_mfma_t<4, 0>(m, smem, nwt, nfiles, s);
        case 8: return launch_mfma_t<8, 0>(m, smem, nwt, nfiles, s);
        case 16: return launch_mfma_t<16, 0>(m, smem, nwt, nfiles, s);
        default: return hipErrorInvalidValue;
    }
}

const char* mfma_kernel_name(const MfmaLayout& g) {
    static thread_local char buf[64];
    const int MB = g.M / 8, U = g.ksteps / 2;
    snprintf(buf, sizeof(buf), "d2d_fir_mfma_kernel<%d, %d>", MB, mfma_has_reg_variant(MB, U) ? U : 0);
    return buf;
}

}  // namespace d2d
