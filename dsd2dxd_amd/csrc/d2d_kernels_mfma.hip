// d2d_kernels_mfma.hip -- the 1-bit FIR decimator on the int8 matrix cores (gfx950), exact.
//
// Arithmetic.  The taps are 24-bit integers q (tap = q*2^-S).  A DSD bit is fed to the MFMA where
// it already sits inside its stream byte: operand register = W & (0x01010101 << p) holds bit p of
// four stream bytes as the int8 value 2^p (p = 7: -128), ONE VALU op per register.  The tap table
// compensates: the K slot that reads bit position p holds q*2^(7-p) (negated for p = 7) split into
// four balanced int8 limbs, so every product is 2^7 * q * bit and the int32 accumulators hold
//        D[4*ph + limb][row] = sum_k  limb_l(...) * 2^p * bit      ->  sum_limbs = 128 * sum_k q_k b_k
// exactly.  y = (2*acc - 2^S) * 2^-S is the number the f64 oracle and the LUT kernel produce.
//
// Geometry.  One matrix column = one "row window" of a channel's bit stream that serves EIGHT
// consecutive outputs (phases ph = 0..7, the window advances 8*M bits per row); the K dimension walks
// the window 32 bits at a time; the 32 matrix rows are 8 phases x 4 limbs, so all four limbs of an
// output land in ONE lane's accumulator registers and are recombined without any data movement.
//
// Schedule.  One wave = one independent worker converting wave-tiles of 256 frames of its block's
// channels (all of a mono/stereo file, one channel pair of a multichannel one): it stages the
// packed bytes of those channels in its own LDS slice (next wave-tile's bytes already in flight),
// runs TWO channels' MFMA chains together (each tap fragment read from the block-shared LDS table
// feeds two MFMAs, the row words of both channels come in one ds_read_b64), then dithers,
// requantises and packs in registers, and stores whole interleaved frames with 16-byte stores.
// No block barrier inside the loop: the waves of a block drift apart, so matrix pipe, VALU and
// memory pipe overlap.
//
// Replaces: the per-block translate loop inside Rdsd2Pcm::do_conversion
// (/root/reference/src/main.rs:345,429); the crate that holds it is absent from the reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "d2d_device.h"
#include "d2d_launch.h"
#include "d2d_mfma.h"

namespace d2d {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int MFMA_MAX_THREADS = 768;   // 12 waves = 3 per SIMD, what 168 VGPRs allow
constexpr int MFMA_PF = 3;   // 16-byte chunks per lane fetched one wave-tile ahead

// Diagnostics (phase ablations, in-kernel cycle stamps, start staggering, the MFMA-phase token) exist
// only in a build with -DD2D_DIAG=1 (make DIAG=1); the production kernel carries none of their state.
#ifndef D2D_DIAG
#define D2D_DIAG 0
#endif

// diagnostic build only (D2D_DBG bit 4): wave-cycles per phase, summed over all waves
__device__ unsigned long long d2d_stamp_acc[8];

struct MfmaArgs {
    FirArgs f;
    double c1, c0;        // x = fma(acc128, c1, -c0) == round(y*c0): c1 = 2^(1-S-7)*c0, c0 = scale | gain | 1
    // integer-depth epilogue as data: d = fma(term, dmul, dadd), clamp to [qmin_i, qmax_i], << qsh
    double dmul, dadd;
    uint32_t dsel;        // 1: triangular term, 0: rectangular term
    uint32_t dkind;       // 0: no dither, 1: triangular, 2: rectangular (chooses the register epilogue's instantiation)
    uint32_t qsh;         // 4 for 20-bit samples in a 24-bit container, else 0
    int32_t qmin_i, qmax_i;
    uint32_t wide;        // 1: limb sums may exceed 2^23, recombine in f64
    uint32_t U;           // dwords of row window per lane half; K steps = 2U
    uint32_t span;        // logical staged bytes per channel (multiple of 16)
    uint32_t ppair;       // physical LDS bytes per channel PAIR (dword-interleaved, padded rows)
    uint32_t ls;          // log2(row stride in dwords) = log2(2*MB)
    uint32_t off_waves;   // LDS: start of the per-wave regions (after the shared tap table)
    uint32_t wave_lds;    // LDS bytes per wave
    uint32_t off_out, off_pk;   // inside a wave's region
    uint32_t nwaves;      // waves per block
    uint32_t ngroups;     // channel groups per file: 1 for mono/stereo, else one block column per channel PAIR
    uint32_t dbg;         // diagnostic ablation mask (env D2D_DBG), 0 in production
    uint32_t stagger;     // start offset between wave slots, in units of 1024 cycles
};

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16 bytes of channel c's stream starting at call-relative byte j (j % 16 == 0).
// (jobs[c] is the job of file channel cbase + c of a file with C channels)
__device__ __forceinline__ u32x4 load_chunk(const StreamJob* jobs, const StreamJob& j0, uint32_t c, uint32_t cbase, int32_t j,
                                            uint32_t C, uint32_t B, uint32_t keep) {
    const uint32_t L = (uint32_t)j0.L;
    if ((B & 15u) == 0 && j >= 0 && (uint32_t)j + 16 <= L) {
        const uint32_t ju = (uint32_t)j;
        const uint32_t blk = (B & (B - 1)) == 0 ? ju >> (31 - __builtin_clz(B)) : ju / B;
        const uint32_t off = ju - blk * B;
        uint32_t blen = L - blk * B;
        if (blen > B) blen = B;
        if ((blen & 15u) == 0) {
            const uint8_t* p = j0.in + (uint64_t)blk * B * C + (uint64_t)(cbase + c) * blen + off;
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

template <int MB>
__global__ __launch_bounds__(MFMA_MAX_THREADS, 3) void d2d_fir_mfma_kernel(MfmaArgs m) {
    const FirArgs& a = m.f;
    const uint32_t dbg = D2D_DIAG ? m.dbg : 0u;
    extern __shared__ __align__(16) unsigned char smem[];
    // A block serves one channel group of one file: all channels for mono/stereo, one channel PAIR
    // otherwise (many channels would not leave LDS for more than a few waves, and staging them all
    // at once overruns the register prefetch).  Ct = channels of the file (input layout), Cs = channels
    // the engine converts = width of the output frame (fewer than Ct for a channel subset), C =
    // channels of this group, cbase = its first channel among the Cs; jobs[c].ch is the channel's
    // index in the FILE.
    const uint32_t Ct = a.in_channels, Cs = a.epi.channels, sb = a.epi.sample_bytes;
    uint32_t fidx, grp_;                                         // (XCD-aware: the channel pairs of a file write into the same frames, d2d_device.h)
    row_to_file_group(blockIdx.y, gridDim.y, m.ngroups, gridDim.x, fidx, grp_);
    const uint32_t cbase = grp_ * 2u;
    const uint32_t C = m.ngroups == 1 ? Cs : (Cs - cbase < 2u ? Cs - cbase : 2u);
    const uint32_t fbytes = sb * C;                        // frame bytes inside the wave's LDS out-slice
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* wbase = smem + m.off_waves + wave * m.wave_lds;
    uint8_t* outw = wbase + m.off_out;
    double* pkw = reinterpret_cast<double*>(wbase + m.off_pk);
    uint32_t* rngw = reinterpret_cast<uint32_t*>(pkw + C * 64);
    const StreamJob* jobs = a.jobs + (size_t)fidx * Cs + cbase;   // jobs[c]: converted channel cbase + c
    const StreamJob j0 = jobs[0];          // in, L, e0, n0, nout are common to a file's channels

    const int64_t first0 = j0.e0 - (int64_t)a.Wb;          // first byte of output 0's window
    const uint32_t d = (uint32_t)(first0 & 15);
    {   // tap fragments: L2 -> LDS once per block (six extra zero K steps for the read-ahead).  The
        // table comes in four variants, one per byte misalignment of the window inside its first
        // LDS dword: the row words are used as they lie, the taps are shifted instead.
        const uint4* s = reinterpret_cast<const uint4*>(a.tables) + (size_t)(d & 3u) * ((a.ksteps + 6) * 64);
        uint4* dl = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < (a.ksteps + 6) * 64; i += blockDim.x) dl[i] = s[i];
    }
#if D2D_DIAG
    uint32_t* tokens = reinterpret_cast<uint32_t*>(smem + (a.ksteps + 6) * 1024);
    if (tid < 16) tokens[tid] = 0;
#endif
    for (uint32_t c = 0; c < C; ++c) pkw[c * 64 + lane] = 0.0;
    if (lane < C) {   // per-channel dither keys: global -> this wave's LDS once
        rngw[lane * 4 + 0] = jobs[lane].rng_key;
        rngw[lane * 4 + 1] = jobs[lane].rng_kstep;
        rngw[lane * 4 + 2] = jobs[lane].rng_lo0;
    }
    __syncthreads();

    const uint32_t U = m.U;
    constexpr uint32_t ls = MB == 1 ? 1 : MB == 2 ? 2 : MB == 4 ? 3 : MB == 8 ? 4 : 5;   // log2(row stride in dwords)
    const uint32_t nwt = (j0.nout + 255u) >> 8;            // wave-tiles in this file
    const uint32_t wstride = gridDim.x * m.nwaves;
    const uint32_t cpc = m.span >> 4;                      // 16-byte chunks per channel
    const uint32_t nch = C * cpc;
    const uint32_t npairs = (C + 1) >> 1;
    // The staging geometry does not change from wave-tile to wave-tile (a wave-tile advances the
    // stream by 256*MB bytes, a multiple of 16): chunk -> (channel, LDS address) once.
    uint32_t pf_c[MFMA_PF], pf_q[MFMA_PF];
    auto lds_word_addr = [&](uint32_t c, uint32_t Ld) -> uint32_t {   // byte offset of staged dword Ld of channel c
        return (c >> 1) * m.ppair + (2u * (Ld + (Ld >> ls)) + (c & 1u)) * 4u;
    };
    // dword k of a 16-byte chunk lands 8k bytes behind the chunk's first dword (the other channel of
    // the pair sits in between), plus one 8-byte row pad after every second dword when the row stride
    // is two dwords (MB = 1); with wider rows a chunk never straddles a pad.
    auto chunk_dword_off = [](int k) -> uint32_t { return 8u * (uint32_t)k + (ls == 1 ? 8u * (uint32_t)(k >> 1) : 0u); };
    uint32_t pf_a[MFMA_PF];                                // LDS byte offset of the chunk's first dword
#pragma unroll
    for (int i = 0; i < MFMA_PF; ++i) {
        const uint32_t ch = lane + 64 * i;
        pf_c[i] = ch / cpc;
        pf_q[i] = ch - pf_c[i] * cpc;
        pf_a[i] = lds_word_addr(pf_c[i], pf_q[i] * 4);
    }
    u32x4 pf[MFMA_PF];
    uint32_t wt = blockIdx.x * m.nwaves + wave;
    auto tile_abeg = [&](uint32_t w) -> int32_t { return (int32_t)((first0 + (int64_t)w * (256 * MB)) & ~(int64_t)15); };
    // planar layout with power-of-two blocks of >= 16 bytes: a staged range that lies inside the
    // call's FULL blocks needs no per-chunk checks (wave-uniform test, a handful of VALU per chunk)
    const uint32_t Bsz = a.B, Lcall = (uint32_t)j0.L;
    const bool pow2B = Bsz >= 16 && (Bsz & (Bsz - 1)) == 0;
    const uint32_t bshift = pow2B ? 31 - __builtin_clz(Bsz) : 0;
    const uint32_t full_bytes = pow2B ? (Lcall >> bshift) << bshift : 0;   // bytes per channel in full blocks
    // inside that path a staged range shorter than one block crosses at most one block boundary:
    // address = (uniform start of the first block group) + (0 or one group) + channel offset + offset
    // inside the block -- five 32-bit VALU per chunk and a scalar base for the load
    const bool short_span = pow2B && m.span <= Bsz && ((uint64_t)Bsz * Ct * 2u) < (1ull << 32);
    uint32_t pf_cs[MFMA_PF];                               // (cbase + channel) << bshift
#pragma unroll
    for (int i = 0; i < MFMA_PF; ++i) pf_cs[i] = (j0.ch + pf_c[i]) << bshift;   // the group's channels are consecutive in the file
    const uint32_t group_bytes = Bsz * Ct;
    auto prefetch = [&](uint32_t w) {
        const int32_t ab = tile_abeg(w);
        if (short_span && ab >= 0 && (uint32_t)ab + m.span <= full_bytes) {
            const uint8_t* gbase = j0.in + ((uint64_t)(((uint32_t)ab >> bshift) * Ct) << bshift);   // uniform
            const uint32_t r0 = (uint32_t)ab & (Bsz - 1);
#pragma unroll
            for (int i = 0; i < MFMA_PF; ++i)
                if (lane + 64 * i < nch) {
                    const uint32_t rr = r0 + pf_q[i] * 16;                          // < 2 * Bsz
                    const uint32_t voff = ((rr >> bshift) ? group_bytes : 0u) + pf_cs[i] + (rr & (Bsz - 1));
                    pf[i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(gbase) + voff);
                }
        } else if (pow2B && ab >= 0 && (uint32_t)ab + m.span <= full_bytes) {
#pragma unroll
            for (int i = 0; i < MFMA_PF; ++i)
                if (lane + 64 * i < nch) {
                    const uint32_t j = (uint32_t)ab + pf_q[i] * 16;
                    const uint64_t off = (uint64_t)((j >> bshift) * Ct) << bshift;   // start of the block group
                    const uint8_t* p = j0.in + off + (((j0.ch + pf_c[i]) << bshift) + (j & (Bsz - 1)));
                    pf[i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(p));
                }
        } else {
#pragma unroll
            for (int i = 0; i < MFMA_PF; ++i)
                if (lane + 64 * i < nch) pf[i] = load_chunk(jobs, j0, pf_c[i], j0.ch, ab + (int32_t)(pf_q[i] * 16), Ct, a.B, a.keep);
        }
    };
    if (wt < nwt) prefetch(wt);
#if D2D_DIAG
    {   // Waves that run the same program fall into lockstep (all in the MFMA loop together, then all in
        // the VALU epilogue together) and the two pipes never overlap.  A one-off start offset per wave
        // slot keeps them apart: the work per wave-tile is identical, so the offset persists.
        const uint32_t slot = ((wave >> 2) + 2u * (blockIdx.x & 1u) + (dbg >> 8)) & 3u;
        for (uint32_t i = 0; i < slot * m.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif

#if D2D_DIAG
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last = 0;
    auto stamp = [&](int slot) {
        if (dbg & 16) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            if (slot >= 0) st_sum[slot] += t - st_last;
            st_last = t;
        }
    };
#else
    auto stamp = [](int) {};
#endif
    const uint32_t r = lane & 31, h = lane >> 5;
    // this lane's row words: logical dword X0 + u of row r, pair-interleaved and padded in LDS
    const uint32_t X0 = (d >> 2) + h * U;
    const uint32_t rbase = 8u * ((2u * MB + 1u) * r);
    const v4i* bp = reinterpret_cast<const v4i*>(smem) + lane;
    const uint32_t K1 = 0x01010101u;

    stamp(-1);
    for (; wt < nwt; wt += wstride) {
        // staged bytes of this wave-tile: registers -> LDS
#pragma unroll
        for (int i = 0; i < MFMA_PF; ++i)
            if (lane + 64 * i < nch) {
                uint8_t* dst = wbase + pf_a[i];
                *reinterpret_cast<uint32_t*>(dst + chunk_dword_off(0)) = pf[i].x;
                *reinterpret_cast<uint32_t*>(dst + chunk_dword_off(1)) = pf[i].y;
                *reinterpret_cast<uint32_t*>(dst + chunk_dword_off(2)) = pf[i].z;
                *reinterpret_cast<uint32_t*>(dst + chunk_dword_off(3)) = pf[i].w;
            }
        for (uint32_t ch = lane + 64 * MFMA_PF; ch < nch; ch += 64) {   // many channels / long windows
            const uint32_t c = ch / cpc, q = ch - c * cpc;
            const u32x4 v = load_chunk(jobs, j0, c, j0.ch, tile_abeg(wt) + (int32_t)(q * 16), Ct, a.B, a.keep);
            *reinterpret_cast<uint32_t*>(wbase + lds_word_addr(c, q * 4 + 0)) = v.x;
            *reinterpret_cast<uint32_t*>(wbase + lds_word_addr(c, q * 4 + 1)) = v.y;
            *reinterpret_cast<uint32_t*>(wbase + lds_word_addr(c, q * 4 + 2)) = v.z;
            *reinterpret_cast<uint32_t*>(wbase + lds_word_addr(c, q * 4 + 3)) = v.w;
        }
        stamp(0);
        if (wt + wstride < nwt) prefetch(wt + wstride);   // next wave-tile's bytes: in flight during the MFMAs
        wave_sync();
        stamp(1);
        const bool full = wt * 256u + 256u <= j0.nout;

        bool stored_from_regs = false;
        for (uint32_t pr = 0; pr < npairs; ++pr) {
            const uint32_t c0 = 2 * pr, c1 = c0 + 1;
            const bool two = c1 < C;
            const uint8_t* prow = wbase + pr * m.ppair + rbase;
            v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            v16i acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            auto row_word = [&](uint32_t t) -> u32x2 {      // both channels' dword t of this lane's row
                return *reinterpret_cast<const u32x2*>(prow + 8u * (t + (t >> ls)));
            };
            auto kpair = [&](const u32x2& w, const v4i& B0, const v4i& B1, auto two_tag, auto first_tag) {
                constexpr bool TWO = decltype(two_tag)::value;
                constexpr bool FIRST = decltype(first_tag)::value;   // start from the inline constant 0: no zero-fill
                const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const uint32_t W0 = w.x, W1 = w.y;
                const v4i A00 = {(int)(W0 & K1), (int)(W0 & (K1 << 1)), (int)(W0 & (K1 << 2)), (int)(W0 & (K1 << 3))};
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B0, A00, FIRST ? zero : acc0, 0, 0, 0);
                if constexpr (TWO) {
                    const v4i A10 = {(int)(W1 & K1), (int)(W1 & (K1 << 1)), (int)(W1 & (K1 << 2)), (int)(W1 & (K1 << 3))};
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B0, A10, FIRST ? zero : acc1, 0, 0, 0);
                }
                const v4i A01 = {(int)(W0 & (K1 << 4)), (int)(W0 & (K1 << 5)), (int)(W0 & (K1 << 6)), (int)(W0 & (K1 << 7))};
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B1, A01, acc0, 0, 0, 0);
                if constexpr (TWO) {
                    const v4i A11 = {(int)(W1 & (K1 << 4)), (int)(W1 & (K1 << 5)), (int)(W1 & (K1 << 6)), (int)(W1 & (K1 << 7))};
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(B1, A11, acc1, 0, 0, 0);
                }
            };
            auto chain = [&](auto two_tag) {
                // Two operand sets in ping-pong: set P serves K pair u, set Q serves u+1; as soon as
                // a set's MFMAs are issued its registers are reloaded for two pairs later, so every
                // LDS read has one whole K pair (four MFMAs) of cover and no register is copied.
                // Zero fragments and spare row words exist past the end for the read-ahead.
                u32x2 wP = row_word(X0), wQ = row_word(X0 + 1);
                v4i P0 = bp[0], P1 = bp[64], Q0 = bp[2 * 64], Q1 = bp[3 * 64];
                const v4i* bq = bp;
                // the first K pair starts both accumulators from the inline constant 0 (U >= 1 always)
                kpair(wP, P0, P1, two_tag, std::true_type{});
                if (U < 2) return;
                wP = row_word(X0 + 2); P0 = bq[4 * 64]; P1 = bq[5 * 64];
                kpair(wQ, Q0, Q1, two_tag, std::false_type{});
                wQ = row_word(X0 + 3); Q0 = bq[6 * 64]; Q1 = bq[7 * 64];
                bq += 4 * 64;
                uint32_t u = 2;
                for (; u + 2 <= U; u += 2) {
                    kpair(wP, P0, P1, two_tag, std::false_type{});
                    wP = row_word(X0 + u + 2); P0 = bq[4 * 64]; P1 = bq[5 * 64];
                    kpair(wQ, Q0, Q1, two_tag, std::false_type{});
                    wQ = row_word(X0 + u + 3); Q0 = bq[6 * 64]; Q1 = bq[7 * 64];
                    bq += 4 * 64;
                }
                if (u < U) kpair(wP, P0, P1, two_tag, std::false_type{});
            };
            // Waves that share a SIMD (w and w+4 of a block) take turns in the MFMA phase: while one
            // multiplies, the other runs its VALU/memory phases, so the two pipes overlap instead of
            // all waves queueing on the matrix pipe together and then on the VALU together.
#if D2D_DIAG
            const bool use_token = (dbg & 64) != 0;
            if (use_token) {
                if (lane == 0) { while (atomicCAS(&tokens[wave & 3u], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(2); }
                __builtin_amdgcn_wave_barrier();
            }
#endif
            if (dbg & 1) { acc0[0] = (int)lane; acc1[0] = (int)r; }
            else if (two) chain(std::true_type{}); else chain(std::false_type{});
#if D2D_DIAG
            if (use_token) {
                asm volatile("" :: "v"(acc0[15]), "v"(acc1[15]));
                if (lane == 0) atomicExch(&tokens[wave & 3u], 0u);
            }
#endif
            if (dbg & 16) { asm volatile("" :: "v"(acc0[15]), "v"(acc1[15])); }
            stamp(2);

            // ---- epilogue: lane (r, h) owns phases ph = h + 2k of row r for both channels ----
            // (the flags that are fixed per launch -- wide recombination, stage-A scratch output -- arrive as
            // tags and are dispatched once per tile, so that the four samples of a lane share a basic block)
            auto finish = [&](const v16i& acc, uint32_t c, auto full_tag, auto wide_tag, auto scratch_tag, auto kind_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                constexpr bool WIDE = decltype(wide_tag)::value;
                constexpr bool SCRATCH = decltype(scratch_tag)::value;
                constexpr int KIND = decltype(kind_tag)::value;          // 0 no noise word needed, 1 triangular, 2 rectangular, 3 float FPD
                if constexpr (!WIDE) {
                    // Two all-integer forms (no f64 instruction).  128*sum q b = lo + 65536*hi with lo a multiple of 128, so
                    // v = sum q s = (lo >> 6) + (hi << 10) - 2^S in int32 (wraps are harmless, |v| < 2^31).
                    //  * the stage-A scratch wants exactly v;
                    //  * float output at 0 dB: x = v * 2^-S, and rounding v to f32 then scaling by the power of two is the same
                    //    rounding as (float)x; |x| is monotonic in |v|, so the peak is tracked on v.
                    const bool f32_int = KIND == 0 && a.epi.bits == 32 && m.c0 == 1.0 && !SCRATCH;
                    if (SCRATCH || f32_int) {
                        int32_t v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int32_t lo = acc[4 * k] + (acc[4 * k + 1] << 8);
                            const uint32_t hi = (uint32_t)acc[4 * k + 2] + ((uint32_t)acc[4 * k + 3] << 8);
                            v[k] = (int32_t)((uint32_t)(lo >> 6) + (hi << 10) - (1u << a.scale_bits));
                        }
                        const uint32_t nl0 = wt * 256u + (8 * r + 4 * h);
                        if constexpr (SCRATCH) {
                            typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
                            if (FULL || nl0 + 3 < j0.nout) {
                                *reinterpret_cast<D2D_GLOBAL i32x4*>(as_global(jobs[c].xs + nl0)) = i32x4{v[0], v[1], v[2], v[3]};
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (nl0 + k < j0.nout) as_global(jobs[c].xs)[nl0 + k] = v[k];
                            }
                        } else {
                            const float sc = __builtin_ldexpf(1.0f, -a.scale_bits);
                            uint32_t vm = 0;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const bool ok = FULL || (nl0 + k < j0.nout);
                                const uint32_t av = (uint32_t)(v[k] < 0 ? -v[k] : v[k]);
                                vm = max(vm, ok ? av : 0u);
                                *reinterpret_cast<float*>(outw + (size_t)((8 * r + 4 * h + k) * C + c) * 4) = (float)v[k] * sc;
                            }
                            pkw[c * 64 + lane] = fmax(pkw[c * 64 + lane], ldexp((double)vm, -a.scale_bits));
                        }
                        return;
                    }
                }
                double xv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double accd;
                    if constexpr (WIDE) {
                        accd = fma((double)acc[4 * k + 3], 16777216.0,
                                   fma((double)acc[4 * k + 2], 65536.0, fma((double)acc[4 * k + 1], 256.0, (double)acc[4 * k])));
                    } else {
                        const int lo = acc[4 * k] + (acc[4 * k + 1] << 8), hi = acc[4 * k + 2] + (acc[4 * k + 3] << 8);
                        accd = fma((double)hi, 65536.0, (double)lo);          // exact: 128 * sum_k q_k b_k
                    }
                    // x = y*c0 with ONE rounding: acc*c1 - c0 is exactly y*c0 before the fma rounds
                    xv[k] = fma(accd, m.c1, -m.c0);
                }
                if constexpr (SCRATCH) {
                    // stage A of the 48k cascade (or the input of the noise-shaping pass): the exact integers
                    // y * 2^S; a lane's four frames are consecutive, one 16-byte store
                    const uint32_t nl0 = wt * 256u + (8 * r + 4 * h);
                    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
                    if (FULL || nl0 + 3 < j0.nout) {
                        *reinterpret_cast<D2D_GLOBAL i32x4*>(as_global(jobs[c].xs + nl0)) = i32x4{(int32_t)xv[0], (int32_t)xv[1], (int32_t)xv[2], (int32_t)xv[3]};
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (nl0 + k < j0.nout) as_global(jobs[c].xs)[nl0 + k] = (int32_t)xv[k];
                    }
                    return;
                }
                const uint32_t rkey = rngw[c * 4], rstep = rngw[c * 4 + 1], rlo0 = rngw[c * 4 + 2];
                double pkx = pkw[c * 64 + lane];
                uint32_t zv[4] = {0, 0, 0, 0};
                if constexpr (KIND != 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t nlo = (uint32_t)j0.n0 + wt * 256u + (8 * r + 4 * h + k);
                        uint32_t z = nlo + rkey + (nlo < rlo0 ? rstep : 0u);
                        z ^= z >> 16; z *= 0x7feb352dU;
                        z ^= z >> 15; z *= 0x846ca68bU;
                        z ^= z >> 16;
                        zv[k] = z;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = FULL || (wt * 256u + (8 * r + 4 * h + k) < j0.nout);
                    // plain v_max_f64 with the |.| source modifier (fmax() would canonicalise both operands first)
                    const double cand = ok ? xv[k] : 0.0;
                    asm("v_max_f64 %0, %1, |%2|" : "=v"(pkx) : "v"(pkx), "v"(cand));
                }
                pkw[c * 64 + lane] = pkx;
                if (a.epi.bits == 32) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        *reinterpret_cast<float*>(outw + (size_t)((8 * r + 4 * h + k) * C + c) * 4) =
                            KIND == 3 ? finish_f32(a.epi, xv[k], zv[k]) : (float)xv[k];        // finish_f32 without FPD is the plain cast
                } else {
                    int32_t iv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // dither term: T = lo16 + hi16 + 1 (x 2^-16, -1), R = 2*hi16 + 1 (x 2^-17, -1/2), none = 0
                        double q = xv[k] + 0.0;                             // "none": what quantise_int() does (a -0 becomes +0)
                        if constexpr (KIND == 1 || KIND == 2) {
                            const uint32_t term = KIND == 1 ? (zv[k] & 0xFFFFu) + (zv[k] >> 16) + 1u : 2u * (zv[k] >> 16) + 1u;
                            q = xv[k] + fma((double)term, m.dmul, m.dadd);
                        }
                        // round half away from zero; v_cvt_i32_f64 saturates, the clip is an integer med3
                        int32_t ri;
                        const double t = q + copysign(0.5, q);
                        asm("v_cvt_i32_f64 %0, %1" : "=v"(ri) : "v"(t));
                        iv[k] = min(max(ri, m.qmin_i), m.qmax_i) << m.qsh;
                    }
                    if (sb == 2) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            *reinterpret_cast<uint16_t*>(outw + (size_t)((8 * r + 4 * h + k) * C + c) * 2) = (uint16_t)iv[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint8_t* p = outw + (size_t)((8 * r + 4 * h + k) * C + c) * 3;
                            p[0] = (uint8_t)iv[k]; p[1] = (uint8_t)(iv[k] >> 8); p[2] = (uint8_t)(iv[k] >> 16);
                        }
                    }
                }
            };
            auto finish_tile = [&](auto full_tag) {
                auto both = [&](auto wide_tag, auto scratch_tag, auto kind_tag) {
                    finish(acc0, c0, full_tag, wide_tag, scratch_tag, kind_tag);
                    if (two) finish(acc1, c1, full_tag, wide_tag, scratch_tag, kind_tag);
                };
                auto kinds = [&](auto wide_tag) {
                    using K0 = std::integral_constant<int, 0>;
                    if (a.to_scratch) { both(wide_tag, std::true_type{}, K0{}); return; }
                    const uint32_t kind = a.epi.bits == 32 ? (a.epi.dither == 'F' ? 3u : 0u) : m.dkind;
                    if (kind == 1) both(wide_tag, std::false_type{}, std::integral_constant<int, 1>{});
                    else if (kind == 2) both(wide_tag, std::false_type{}, std::integral_constant<int, 2>{});
                    else if (kind == 3) both(wide_tag, std::false_type{}, std::integral_constant<int, 3>{});
                    else both(wide_tag, std::false_type{}, K0{});
                };
                if (m.wide) kinds(std::true_type{}); else kinds(std::false_type{});
            };
            const bool reg_store = full && two && Cs == 2 && sb == 3 && m.qsh == 0 && !m.wide && !a.to_scratch;
            bool stored_from_regs_now = false;
            if (dbg & 2) { if (acc0[0] == 0x12345 && acc1[5] == 77 && acc0[9] + acc1[13] + acc0[15] + acc1[2] == 99) outw[lane] = 1; }
            else if (reg_store) {
                // Stereo 24-bit, whole tile: no LDS round trip.  Lane (r, h) owns frames 4h .. 4h+3 of
                // row r for both channels = 24 contiguous output bytes; a few byte permutes pack them
                // and the wave stores 1536 contiguous bytes.
                // (frame by frame, both channels together: two independent sample pipelines in flight
                // and only the six packed output words stay live)
                // The dither kind is fixed per launch: the body is instantiated per kind and chosen ONCE per
                // tile, so the eight sample pipelines sit in one basic block (a per-sample branch on a
                // uniform flag would cut the block there and stop the scheduler from interleaving them).
                auto body = [&](auto kind_tag) {
                    constexpr int KIND = decltype(kind_tag)::value;       // 0 none, 1 triangular, 2 rectangular
                    const uint32_t key0 = rngw[c0 * 4], st0 = rngw[c0 * 4 + 1], lo00 = rngw[c0 * 4 + 2];
                    const uint32_t key1 = rngw[c1 * 4], st1 = rngw[c1 * 4 + 1], lo01 = rngw[c1 * 4 + 2];
                    double pk0 = pkw[c0 * 64 + lane], pk1 = pkw[c1 * 64 + lane];
                    const int32_t qmax_v = m.qmax_i;
                    auto one = [&](const v16i& acc, int k, uint32_t key, uint32_t stp, uint32_t lo0, double& pk) -> uint32_t {
                        // (narrow recombination: the wide case does not take this path)
                        const double accd = fma((double)(acc[4 * k + 2] + (acc[4 * k + 3] << 8)), 65536.0, (double)(acc[4 * k] + (acc[4 * k + 1] << 8)));
                        const double x = fma(accd, m.c1, -m.c0);
                        asm("v_max_f64 %0, %1, |%2|" : "=v"(pk) : "v"(pk), "v"(x));
                        double q = x;
                        if constexpr (KIND != 0) {
                            const uint32_t nlo = (uint32_t)j0.n0 + wt * 256u + (8 * r + 4 * h + k);
                            uint32_t z = nlo + key + (nlo < lo0 ? stp : 0u);
                            z ^= z >> 16; z *= 0x7feb352dU;
                            z ^= z >> 15; z *= 0x846ca68bU;
                            z ^= z >> 16;
                            const uint32_t term = KIND == 1 ? (z & 0xFFFFu) + (z >> 16) + 1u : 2u * (z >> 16) + 1u;
                            q = x + fma((double)term, m.dmul, m.dadd);
                        } else {
                            q = x + 0.0;                                    // what quantise_int() does for "none" (a -0 becomes +0)
                        }
                        // round half away from zero: the conversion itself truncates toward zero (and saturates),
                        // the clip is one integer med3 (one bound has to sit in a VGPR: one SGPR per VOP3 on gfx9)
                        int32_t ri, o;
                        const double t = q + copysign(0.5, q);
                        asm("v_cvt_i32_f64 %0, %1" : "=v"(ri) : "v"(t));
                        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(o) : "v"(ri), "s"(m.qmin_i), "v"(qmax_v));
                        return (uint32_t)o;                                   // (qsh == 0 on this path)
                    };
                    // frames k, k+1 -> 12 bytes: [L0 L1 L2 R0 | R1 R2 L0' L1' | L2' R0' R1' R2']
                    uint32_t w0, w1, w2, w3, w4, w5;
                    {
                        const uint32_t La = one(acc0, 0, key0, st0, lo00, pk0), Ra = one(acc1, 0, key1, st1, lo01, pk1);
                        const uint32_t Lb = one(acc0, 1, key0, st0, lo00, pk0), Rb = one(acc1, 1, key1, st1, lo01, pk1);
                        w0 = __builtin_amdgcn_perm(Ra, La, 0x04020100u);
                        w1 = __builtin_amdgcn_perm(Lb, Ra, 0x05040201u);
                        w2 = __builtin_amdgcn_perm(Rb, Lb, 0x06050402u);
                    }
                    {
                        const uint32_t La = one(acc0, 2, key0, st0, lo00, pk0), Ra = one(acc1, 2, key1, st1, lo01, pk1);
                        const uint32_t Lb = one(acc0, 3, key0, st0, lo00, pk0), Rb = one(acc1, 3, key1, st1, lo01, pk1);
                        w3 = __builtin_amdgcn_perm(Ra, La, 0x04020100u);
                        w4 = __builtin_amdgcn_perm(Lb, Ra, 0x05040201u);
                        w5 = __builtin_amdgcn_perm(Rb, Lb, 0x06050402u);
                    }
                    pkw[c0 * 64 + lane] = pk0; pkw[c1 * 64 + lane] = pk1;
                    const u32x4 o4 = {w0, w1, w2, w3};
                    const u32x2 o2 = {w4, w5};
                    uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * 1536 + 48u * r + 24u * h;
                    *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(g)) = o4;
                    *reinterpret_cast<D2D_GLOBAL u32x2*>(as_global(g + 16)) = o2;
                };
                if (m.dkind == 1) body(std::integral_constant<int, 1>{});
                else if (m.dkind == 2) body(std::integral_constant<int, 2>{});
                else body(std::integral_constant<int, 0>{});
            }
            else if (full && two && Cs == 2 && a.epi.bits == 32 && !m.wide && !a.to_scratch && a.epi.dither != 'F' && m.c0 == 1.0) {
                // Stereo float at 0 dB, whole tile: no LDS round trip either.  Lane (r, h) owns frames 4h .. 4h+3 of row r for
                // both channels = 32 contiguous output bytes, two 16-byte stores; x = v * 2^-S with v = sum q s an int32, and
                // rounding v to f32 then scaling by the power of two is the rounding of (float)x.
                const float sc = __builtin_ldexpf(1.0f, -a.scale_bits);
                uint32_t vm0 = 0, vm1 = 0;
                float L[4], R[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t lo0 = acc0[4 * k] + (acc0[4 * k + 1] << 8), lo1 = acc1[4 * k] + (acc1[4 * k + 1] << 8);
                    const uint32_t hi0 = (uint32_t)acc0[4 * k + 2] + ((uint32_t)acc0[4 * k + 3] << 8), hi1 = (uint32_t)acc1[4 * k + 2] + ((uint32_t)acc1[4 * k + 3] << 8);
                    const int32_t v0 = (int32_t)((uint32_t)(lo0 >> 6) + (hi0 << 10) - (1u << a.scale_bits));
                    const int32_t v1 = (int32_t)((uint32_t)(lo1 >> 6) + (hi1 << 10) - (1u << a.scale_bits));
                    vm0 = max(vm0, (uint32_t)(v0 < 0 ? -v0 : v0)); vm1 = max(vm1, (uint32_t)(v1 < 0 ? -v1 : v1));
                    L[k] = (float)v0 * sc; R[k] = (float)v1 * sc;
                }
                pkw[c0 * 64 + lane] = fmax(pkw[c0 * 64 + lane], ldexp((double)vm0, -a.scale_bits));
                pkw[c1 * 64 + lane] = fmax(pkw[c1 * 64 + lane], ldexp((double)vm1, -a.scale_bits));
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * 2048 + 64u * r + 32u * h;
                *reinterpret_cast<D2D_GLOBAL f32x4*>(as_global(g)) = f32x4{L[0], R[0], L[1], R[1]};
                *reinterpret_cast<D2D_GLOBAL f32x4*>(as_global(g + 16)) = f32x4{L[2], R[2], L[3], R[3]};
                stored_from_regs_now = true;
            }
            else if (full) finish_tile(std::true_type{});
            else finish_tile(std::false_type{});
            stored_from_regs = reg_store || stored_from_regs_now;
        }
        stamp(3);
        if (!a.to_scratch && !(dbg & 4) && !stored_from_regs) {
            wave_sync();
            // the wave-tile's interleaved frames: LDS -> HBM, 16 bytes per lane per store
            const uint32_t left = j0.nout - wt * 256;
            const uint32_t nfr = left < 256u ? left : 256u;
            if (m.ngroups == 1) {
                const uint32_t nb = nfr * fbytes;
                uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * 256 * fbytes;
                const uint32_t nb16 = nb & ~15u;
                for (uint32_t i = lane * 16; i < nb16; i += 64 * 16)
                    *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(g + i)) = *reinterpret_cast<const u32x4*>(outw + i);
                for (uint32_t i = nb16 + lane; i < nb; i += 64) as_global(g)[i] = outw[i];
            } else {
                // this group's `fbytes` bytes of every frame sit sb*cbase bytes into the file's frame
                const uint32_t gstride = sb * Cs;
                uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * 256 * gstride + sb * cbase;
                for (uint32_t fr = lane; fr < nfr; fr += 64)
                    for (uint32_t b = 0; b < fbytes; ++b) as_global(g)[(size_t)fr * gstride + b] = outw[fr * fbytes + b];
            }
        }
        stamp(4);
    }
#if D2D_DIAG
    if ((dbg & 16) && lane == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(&d2d_stamp_acc[i], st_sum[i]);
#endif
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
    const int wd = (N + 7 * M + 24 + 31) / 32;   // dwords of one row's window (+ up to 3 bytes of misalignment)
    const int U = (wd + 1) / 2;
    g.ksteps = 2 * U;
    return g;
}

static inline int8_t limb_of(int64_t v, int l) {
    // balanced base-256 digits: v = d0 + d1*2^8 + d2*2^16 + d3*2^24, every d in [-128, 127]
    int8_t dgt = 0;
    for (int i = 0; i <= l; ++i) {
        int64_t dd = ((v + 128) & 255) - 128;
        dgt = (int8_t)dd;
        v = (v - dd) / 256;
    }
    return dgt;
}

// Tap fragments [4 byte shifts][ksteps + 6][64 lanes][16 bytes].  Lane l supplies matrix row (l & 31) = 4*phase + limb
// for the K slots (l >> 5)*16 + j; slot (ks, h, j) reads bit `wb` of the row window (see the
// kernel's A00/A01), which sits at bit position p = wb & 7 of its stream byte and therefore arrives
// as 2^p (p = 7: -128): the table holds q * 2^(7-p), negated for p = 7.
std::vector<int8_t> build_mfma_tables(const d2d_filter_def& f, const MfmaLayout& g, bool msb_first) {
    const int U = g.ksteps / 2;
    const size_t per = (size_t)(g.ksteps + 6) * 64 * 16;           // +6 zero steps: the kernel's read-ahead
    std::vector<int8_t> t(4 * per, 0);
    for (int sh = 0; sh < 4; ++sh)                                  // window starts `sh` bytes into its first dword
        for (int ks = 0; ks < g.ksteps; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 31, h = l >> 5, limb = row & 3;
                // D row i lands in lane half (i >> 2) & 1, register group i >> 3: give that slot output
                // phase 4*half + group, so lane (r, half) owns the four CONSECUTIVE outputs 8r + 4*half + k
                const int ph = 4 * ((row >> 2) & 1) + (row >> 3);
                for (int j = 0; j < 16; ++j) {
                    const int p = 4 * (ks & 1) + (j >> 2);                        // register v = j>>2 of step ks
                    const int wb = 32 * (h * U + (ks >> 1)) + 8 * (j & 3) + p;     // bit of the LDS row words
                    const int tau = (msb_first ? (wb & ~7) + 7 - (wb & 7) : wb) - 8 * sh;   // its time index in the window
                    const int tap = tau - ph * g.M;
                    int8_t v = 0;
                    if (tau >= 0 && tap >= 0 && tap < f.ntaps) {
                        int64_t q = tap_q(f, tap);
                        q = p == 7 ? -q : q * (int64_t)(1 << (7 - p));
                        v = limb_of(q, limb);
                    }
                    t[sh * per + ((size_t)ks * 64 + l) * 16 + j] = v;
                }
            }
    return t;
}

static void mfma_geometry(const FirArgs& a, const MfmaLayout& g, MfmaArgs& m, size_t& smem) {
    // channels per block: all of a mono/stereo file, one pair of a multichannel one
    m.ngroups = a.epi.channels <= 2 ? 1u : (a.epi.channels + 1u) / 2u;
    const uint32_t C = a.epi.channels <= 2 ? a.epi.channels : 2u;
    const int MB = g.M / 8;
    m.f = a;
    m.c0 = a.to_scratch ? ldexp(1.0, a.scale_bits) : (a.epi.bits == 32 ? a.epi.gain : a.epi.scale);   // scratch: the integer y*2^S
    m.c1 = ldexp(m.c0, 1 - a.scale_bits - 7);     // exact: a power-of-two multiple of c0
    m.dsel = a.epi.dither == 'T' ? 1u : 0u;
    m.dkind = a.epi.dither == 'T' ? 1u : (a.epi.dither == 'R' ? 2u : 0u);
    m.dmul = a.epi.dither == 'T' ? 0x1p-16 : (a.epi.dither == 'R' ? 0x1p-17 : 0.0);
    m.dadd = a.epi.dither == 'T' ? -1.0 : (a.epi.dither == 'R' ? -0.5 : 0.0);
    m.qsh = a.epi.bits == 20 ? 4u : 0u;
    m.qmin_i = a.epi.bits == 32 ? 0 : -(1 << (a.epi.bits - 1)); m.qmax_i = a.epi.bits == 32 ? 0 : (1 << (a.epi.bits - 1)) - 1;
    m.U = (uint32_t)g.ksteps / 2;
    // |limb sum| <= (bytes of row window) * 255 * 128; below 2^23 the pairs recombine in int32
    m.wide = (uint64_t)g.ksteps * 4u * 255u * 128u >= (1u << 23) ? 1u : 0u;
    int ls = 0;
    while ((1 << ls) < 2 * MB) ++ls;
    m.ls = (uint32_t)ls;
    // logical staged bytes per channel: 16-byte alignment slack + 31 row strides + one row window
    // (+3 dwords read ahead) + slack for the in-register byte realignment
    m.span = (16u + 31u * 8u * MB + (2 * m.U + 5) * 4u + 16u + 15u) & ~15u;
    const uint32_t ldw = m.span / 4;
    m.ppair = ((2u * (ldw + (ldw >> ls) + 2u)) * 4u + 15u) & ~15u;
    m.off_waves = ((uint32_t)g.ksteps + 6u) * 1024u + 64u;   // + MFMA-phase tokens
    m.off_out = ((C + 1) / 2) * m.ppair;
    m.off_pk = m.off_out + ((256u * C * a.epi.sample_bytes + 15u) & ~15u);
    m.wave_lds = m.off_pk + C * 64u * 8u + ((C * 16u + 15u) & ~15u);   // peaks + per-channel dither keys
#if D2D_DIAG
    { static const char* e = getenv("D2D_DBG"); m.dbg = e ? (uint32_t)atoi(e) : 0u; }          // (make DIAG=1 builds only: never the shipped library)
    { static const char* e = getenv("D2D_STAGGER"); m.stagger = e ? (uint32_t)atoi(e) : 0u; }
#endif
    const uint32_t wdbg = (a.dbg_flags >> 8) & 0xFFu;      // diagnostic override (d2d_params.debug_flags bits 8..15)
    m.nwaves = wdbg ? wdbg : 12u;
    if (m.nwaves < 1 || m.nwaves > 12) m.nwaves = 12;
    // largest block that fits the CU's LDS, keeping the waves evenly spread over the four SIMDs
    while (m.nwaves > 1 && (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds > 160 * 1024)
        m.nwaves = m.nwaves > 8 ? 8 : m.nwaves > 4 ? 4 : m.nwaves >> 1;
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

template <int MB>
static hipError_t launch_mfma_t(const MfmaArgs& m, size_t smem, uint32_t nwt_max, uint32_t nfiles, hipStream_t s) {
    static KernelPrep prep;
    int dev = 0;
    hipError_t e = prep.max_dynamic_lds(reinterpret_cast<const void*>(&d2d_fir_mfma_kernel<MB>), 160 * 1024, &dev);
    if (e != hipSuccess) return e;
    int blocks_per_cu, ncu;
    {
        std::lock_guard<std::mutex> g(prep.mu);
        if (prep.blocks_per_cu[dev] == 0 || smem != prep.smem_seen[dev] || m.nwaves != prep.nwaves_seen[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            int nb = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, d2d_fir_mfma_kernel<MB>, (int)(64 * m.nwaves), smem);
            if (e != hipSuccess) return e;
            prep.ncu[dev] = prop.multiProcessorCount;
            prep.blocks_per_cu[dev] = nb < 1 ? 1 : nb;
            prep.smem_seen[dev] = smem; prep.nwaves_seen[dev] = m.nwaves;
        }
        blocks_per_cu = prep.blocks_per_cu[dev]; ncu = prep.ncu[dev];
    }
    // every wave loops over its share of the wave-tiles: launch what is resident at once
    uint32_t gx = (uint32_t)(ncu * blocks_per_cu) / nfiles;
    if (gx < 1) gx = 1;
    const uint32_t need = (nwt_max + m.nwaves - 1) / m.nwaves;
    if (gx > need) gx = need;
    hipLaunchKernelGGL((d2d_fir_mfma_kernel<MB>), dim3(gx, nfiles), dim3(64 * m.nwaves), smem, s, m);
    d2d_last_launched_kernel = launched_name<MB>("d2d_fir_mfma_kernel");
    return hipGetLastError();
}

hipError_t launch_fir_mfma(const FirArgs& a, const MfmaLayout& g, uint32_t max_nout, uint32_t nstreams, hipStream_t s) {
    if (nstreams == 0 || max_nout == 0) return hipSuccess;
    const uint32_t C = a.epi.channels;
    const int MB = g.M / 8;
    MfmaArgs m{};
    size_t smem = 0;
    mfma_geometry(a, g, m, smem);
    const uint32_t nfiles = (nstreams / C) * m.ngroups;       // grid rows: one per (file, channel group)
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    const uint32_t nwt = (max_nout + 255u) / 256u;
    switch (MB) {
        case 1: return launch_mfma_t<1>(m, smem, nwt, nfiles, s);
        case 2: return launch_mfma_t<2>(m, smem, nwt, nfiles, s);
        case 4: return launch_mfma_t<4>(m, smem, nwt, nfiles, s);
        case 8: return launch_mfma_t<8>(m, smem, nwt, nfiles, s);
        case 16: return launch_mfma_t<16>(m, smem, nwt, nfiles, s);
        default: return hipErrorInvalidValue;
    }
}

void mfma_debug_stamps(unsigned long long out[8]) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(d2d_stamp_acc), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d2d_stamp_acc), z, sizeof(z));
}

const char* mfma_kernel_name(const MfmaLayout& g) {
    switch (g.M / 8) {
        case 1: return "d2d_fir_mfma_kernel<1>";
        case 2: return "d2d_fir_mfma_kernel<2>";
        case 4: return "d2d_fir_mfma_kernel<4>";
        case 8: return "d2d_fir_mfma_kernel<8>";
        default: return "d2d_fir_mfma_kernel<16>";
    }
}

}  // namespace d2d
