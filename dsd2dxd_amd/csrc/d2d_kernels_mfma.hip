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
    double yscale;        // 2^(1-S)
    uint32_t FT;          // frames per block tile (multiple of 256)
    uint32_t U;           // dwords of row window per lane half; K steps = 2U
    uint32_t span;        // staged bytes per channel
    uint32_t off_in, off_d, off_out;   // LDS offsets
};

__device__ __forceinline__ void wave_peak_flush(double pk, double* dst) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pk = fmax(pk, __shfl_xor(pk, o));
    if ((threadIdx.x & 63) == 0 && pk > 0.0)
        atomicMax(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(pk));
}

// UCT > 0: the tap fragments of all 2*UCT K steps stay in registers for the whole block (small
// filters: no LDS traffic in the MFMA loop, fully unrolled).  UCT == 0: K steps counted at run time,
// fragments streamed from LDS one pair ahead of the MFMAs that use them.
template <int MB, int UCT>
__global__ __launch_bounds__(MFMA_THREADS) void d2d_fir_mfma_kernel(MfmaArgs m) {
    const FirArgs& a = m.f;
    extern __shared__ __align__(16) unsigned char smem[];
    uint8_t* btab = smem;
    uint8_t* inb = smem + m.off_in;
    uint8_t* dscr = smem + m.off_d;
    uint8_t* outb = smem + m.off_out;
    const uint32_t C = a.epi.channels, sb = a.epi.sample_bytes, fbytes = sb * C;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const StreamJob* jobs = a.jobs + (size_t)blockIdx.y * C;
    const StreamJob j0 = jobs[0];          // L, e0, n0, nout are common to a file's channels

    v4i breg[UCT > 0 ? 2 * UCT : 1];
    if constexpr (UCT > 0) {   // tap fragments: L2 -> registers once per block
        const v4i* s = reinterpret_cast<const v4i*>(a.tables) + lane;
#pragma unroll
        for (int t = 0; t < 2 * UCT; ++t) breg[t] = s[t * 64];
    } else {                   // tap fragments: L2 -> LDS once per block
        const uint4* s = reinterpret_cast<const uint4*>(a.tables);
        uint4* d = reinterpret_cast<uint4*>(btab);
        for (uint32_t i = tid; i < (a.ksteps + 2) * 64; i += MFMA_THREADS) d[i] = s[i];
    }
    const uint32_t FT = m.FT;
    const uint32_t ntiles = (j0.nout + FT - 1) / FT;
    const uint32_t nwt = C * (FT >> 8);            // wave-tiles (32 rows x 8 phases) per block tile
    const uint32_t nit = (nwt + MFMA_WAVES - 1) / MFMA_WAVES;
    double pk = 0.0;
    uint32_t pk_c = 0xFFFFFFFFu;

    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tile_first = j0.e0 - (int64_t)a.Wb + (int64_t)tile * FT * MB;
        const int64_t abeg = tile_first & ~(int64_t)15;
        const uint32_t d = (uint32_t)(tile_first - abeg);
        __syncthreads();
        for (uint32_t c = 0; c < C; ++c)
            stage_window(inb + c * m.span, jobs[c], C, a.B, a.keep, abeg, m.span, tid, MFMA_THREADS);
        __syncthreads();
        const uint32_t sh = d & 3u;

        for (uint32_t it = 0; it < nit; ++it) {
            const uint32_t wt = it * MFMA_WAVES + wave;
            if (wt < nwt) {
                const uint32_t c = wt % C, sub = wt / C;
                const uint32_t h = lane >> 5;
                {
                    const uint32_t r = sub * 32 + (lane & 31);
                    const uint32_t Urt = UCT > 0 ? (uint32_t)UCT : m.U;
                    const uint32_t* rp = reinterpret_cast<const uint32_t*>(inb + c * m.span + (d & ~3u) + r * (8 * MB)) + h * Urt;
                    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    const uint32_t K1 = 0x01010101u;
                    if constexpr (UCT > 0) {
                        uint32_t rw[UCT + 1];
#pragma unroll
                        for (int u = 0; u <= UCT; ++u) rw[u] = rp[u];
#pragma unroll
                        for (int u = 0; u < UCT; ++u) {
                            const uint32_t W = __builtin_amdgcn_alignbyte(rw[u + 1], rw[u], sh);
                            v4i A0 = {(int)(W & K1), (int)((W >> 1) & K1), (int)((W >> 2) & K1), (int)((W >> 3) & K1)};
                            v4i A1 = {(int)((W >> 4) & K1), (int)((W >> 5) & K1), (int)((W >> 6) & K1), (int)((W >> 7) & K1)};
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, breg[2 * u], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, breg[2 * u + 1], acc, 0, 0, 0);
                        }
                    } else {
                        const v4i* bp = reinterpret_cast<const v4i*>(btab) + lane;
                        uint32_t cur = rp[0], nxt = rp[1];
                        v4i B0 = bp[0], B1 = bp[64];
                        for (uint32_t u = 0; u < m.U; ++u) {
                            // fetch the next pair's operands before this pair's MFMAs (one extra row
                            // dword and one extra zero fragment pair exist past the end)
                            const uint32_t nn = rp[u + 2];
                            const v4i B0n = bp[(2 * u + 2) * 64], B1n = bp[(2 * u + 3) * 64];
                            const uint32_t W = __builtin_amdgcn_alignbyte(nxt, cur, sh);
                            v4i A0 = {(int)(W & K1), (int)((W >> 1) & K1), (int)((W >> 2) & K1), (int)((W >> 3) & K1)};
                            v4i A1 = {(int)((W >> 4) & K1), (int)((W >> 5) & K1), (int)((W >> 6) & K1), (int)((W >> 7) & K1)};
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B1, acc, 0, 0, 0);
                            cur = nxt; nxt = nn; B0 = B0n; B1 = B1n;
                        }
                    }
                    // D[row][col] -> this wave's scratch as [row][col] = [output o = row*8+ph][limb]
                    uint32_t* ds = reinterpret_cast<uint32_t*>(dscr + wave * 4096);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const uint32_t row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                        ds[row * 32 + (lane & 31)] = (uint32_t)acc[reg];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (pk_c != c) {
                    if (pk_c != 0xFFFFFFFFu) wave_peak_flush(pk, jobs[pk_c].peak);
                    pk = 0.0; pk_c = c;
                }
                const StreamJob* jc = jobs + c;
                const uint32_t rkey = jc->rng_key, rstep = jc->rng_kstep, rlo0 = jc->rng_lo0;
                const int4* dsv = reinterpret_cast<const int4*>(dscr + wave * 4096);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t o = lane + 64 * k;
                    const int4 Lm = dsv[o];
                    const int lo = Lm.x + (Lm.y << 8), hi = Lm.z + (Lm.w << 8);
                    const double accd = fma((double)hi, 65536.0, (double)lo);   // exact
                    const double y = fma(accd, m.yscale, -1.0);                  // exact: (2*acc - 2^S) * 2^-S
                    const uint32_t fl = sub * 256 + o;
                    const uint32_t nl = tile * FT + fl;
                    if (nl < j0.nout) {
                        if (a.to_scratch) {
                            jc->xs[nl] = y;
                        } else {
                            const uint32_t nlo = (uint32_t)(j0.n0 + nl);
                            uint32_t x = nlo + rkey + (nlo < rlo0 ? rstep : 0u);
                            x ^= x >> 16; x *= 0x7feb352dU;
                            x ^= x >> 15; x *= 0x846ca68bU;
                            x ^= x >> 16;
                            uint8_t* p = outb + (size_t)(fl * C + c) * sb;
                            if (a.epi.bits == 32) {
                                *reinterpret_cast<float*>(p) = quantise_f32(a.epi, y, x);
                            } else {
                                const int32_t iv = quantise_int(a.epi, y, x);
                                if (a.epi.bits == 16) {
                                    *reinterpret_cast<uint16_t*>(p) = (uint16_t)iv;
                                } else {
                                    p[0] = (uint8_t)iv; p[1] = (uint8_t)(iv >> 8); p[2] = (uint8_t)(iv >> 16);
                                }
                            }
                            pk = fmax(pk, fabs(y * a.epi.gain));
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (!a.to_scratch) {
            // interleaved frames of this tile: LDS -> HBM in 16-byte pieces
            const uint32_t left = j0.nout - tile * FT;
            const uint32_t nb = (left < FT ? left : FT) * fbytes;
            uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)tile * FT * fbytes;
            const uint32_t nb16 = nb & ~15u;
            for (uint32_t i = tid * 16; i < nb16; i += MFMA_THREADS * 16)
                *reinterpret_cast<uint4*>(g + i) = *reinterpret_cast<const uint4*>(outb + i);
            for (uint32_t i = nb16 + tid; i < nb; i += MFMA_THREADS) g[i] = outb[i];
        }
    }
    if (!a.to_scratch && pk_c != 0xFFFFFFFFu) wave_peak_flush(pk, jobs[pk_c].peak);
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

static uint32_t mfma_ft(uint32_t C) { return C == 1 ? 1024u : (C == 2 ? 512u : 256u); }

template <int MB, int UCT>
static hipError_t launch_mfma_t(const MfmaArgs& m, size_t smem, dim3 grid, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&d2d_fir_mfma_kernel<MB, UCT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((d2d_fir_mfma_kernel<MB, UCT>), grid, dim3(MFMA_THREADS), smem, s, m);
    return hipGetLastError();
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
    m.FT = mfma_ft(C);
    m.U = (uint32_t)g.ksteps / 2;
    // staged bytes per channel: alignment slack + rows + one row window (+2 dwords read ahead)
    m.span = (16u + (m.FT / 8 - 1) * 8u * MB + (2 * m.U + 3) * 4u + 16u + 15u) & ~15u;
    const bool reg = mfma_has_reg_variant(MB, (int)m.U);
    m.off_in = reg ? 0u : ((uint32_t)g.ksteps + 2u) * 1024u;   // +2: the zero pair the prefetch reads
    m.off_d = m.off_in + C * m.span;
    m.off_out = m.off_d + MFMA_WAVES * 4096u;
    smem = (size_t)m.off_out + ((m.FT * C * a.epi.sample_bytes + 15u) & ~15u);
}

size_t mfma_smem_bytes(const MfmaLayout& g, uint32_t channels, uint32_t sample_bytes) {
    FirArgs a{};
    a.epi.channels = channels; a.epi.sample_bytes = sample_bytes;
    MfmaArgs m{}; size_t smem = 0;
    mfma_geometry(a, g, m, smem);
    return smem;
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
    const uint32_t ntiles = (max_nout + m.FT - 1) / m.FT;
    uint32_t gx = ntiles;
    const uint32_t cap = (2048 + nfiles - 1) / nfiles;    // ~8 blocks per CU in flight, tiles looped inside
    if (gx > cap) gx = cap;
    dim3 grid(gx, nfiles);
#define X(mb, u) if (MB == mb && m.U == u) return launch_mfma_t<mb, u>(m, smem, grid, s);
    D2D_MFMA_REG_VARIANTS(X)
#undef X
    switch (MB) {
        case 1: return launch_mfma_t<1, 0>(m, smem, grid, s);
        case 2: return launch_mfma_t<2, 0>(m, smem, grid, s);
        case 4: return launch_mfma_t<4, 0>(m, smem, grid, s);
        case 8: return launch_mfma_t<8, 0>(m, smem, grid, s);
        case 16: return launch_mfma_t<16, 0>(m, smem, grid, s);
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
