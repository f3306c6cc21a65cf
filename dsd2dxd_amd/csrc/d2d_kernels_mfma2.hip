// d2d_kernels_mfma2.hip -- the int8 matrix-core FIR decimator, second generation (gfx950), exact.
//
// Same arithmetic as d2d_kernels_mfma.hip (taps are 24-bit integers q, tap = q*2^-S; a stream dword W
// becomes operand registers W & (0x01010101 << p), the tap table holds q*2^(7-p) in four balanced int8
// limbs, the int32 accumulators hold 128 * sum q_k b_k exactly), different geometry:
//
//   * One matrix column = one "row window" that serves SIXTEEN consecutive outputs: two groups of eight
//     phases.  The expanded stream operand of a K step (eight v_and per stream dword) is shared by the
//     two groups' MFMAs, and the two groups read the SAME tap fragments, group 1 MB pair-steps later
//     (its window starts 8*M bits further on).  Every stream bit is expanded 2.0x instead of 3.1x.
//   * Lane half h takes the dwords 2u+h of the row window, so a K "pair step" u covers 64 contiguous
//     bits and the all-zero tap blocks in front of group 1 / behind group 0 are never issued:
//     2 * 2 * NPG MFMAs per 512 outputs, the same count as the one-group kernel.
//   * A wave converts a tile of 512 outputs of ONE channel per chain (the channels of a stereo file one
//     after the other); the chain is fully unrolled: every LDS address is the lane's base register plus
//     an immediate, the loop carries no address arithmetic.
//   * The tile's bytes are fetched as aligned 16-byte chunks and written to LDS shifted so that the
//     window of matrix column r starts at staged dword 4*MB*r exactly (one pad dword per row stride keeps
//     the 32 lanes' reads on distinct banks); only the byte misalignment (0..3) is left to the tap table,
//     which comes in four pre-shifted variants.
//
// Replaces: the per-block translate loop inside Rdsd2Pcm::do_conversion
// (/root/reference/src/main.rs:345,429); the crate that holds it is absent from the reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "d2d_mfma2_dev.h"
#include "d2d_mx.h"

namespace d2d {

#if D2D_DIAG
__device__ unsigned long long d2d_m2_stamps[8];
#endif

#ifndef D2D_M2_THREADS
#define D2D_M2_THREADS 768
#endif
// EPI: 0 = any format through the wave's LDS out-slice, 1 = stereo 24-bit packed in registers, 2 = the exact integers y*2^S to the
// stage-A scratch (48k cascade, noise-shaping pass)
template <int MB, int NPG, int CH, int EPI>
__global__ __launch_bounds__(D2D_M2_THREADS) void d2d_fir_mfma2_kernel(Mfma2Args m) {
    using G = M2Geom<MB>;
    constexpr int RS = G::RS, LSH = G::LSH;
    constexpr int TP = NPG + MB;                                    // pair steps of one chain
    constexpr int NCHK = m2_chunks(MB, NPG);
    constexpr int PF = m2_pf(MB, NPG);
    const FirArgs& a = m.f;
    const uint32_t dbg = D2D_DIAG ? m.dbg : 0u;
    extern __shared__ __align__(16) unsigned char smem[];
    // A block serves one channel group of one file: all channels for mono/stereo, one channel PAIR
    // otherwise.  Ct = channels of the file (input layout), Cs = channels the engine converts = width of
    // the output frame, C = channels of this group (== CH except for the odd last channel of a
    // multichannel file, whose block runs the CH = 2 code with the second chain skipped).
    const uint32_t Ct = a.in_channels, Cs = a.epi.channels, sb = a.epi.sample_bytes;
    uint32_t fidx, grp_;                                         // (XCD-aware: the channel pairs of a file write into the same frames, d2d_device.h)
    row_to_file_group(blockIdx.y, gridDim.y, m.ngroups, gridDim.x, fidx, grp_);
    const uint32_t cbase = grp_ * 2u;
    const uint32_t C = m.ngroups == 1 ? Cs : (Cs - cbase < 2u ? Cs - cbase : 2u);
    const uint32_t fbytes = sb * C;                        // frame bytes inside the wave's LDS out-slice
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* wbase = smem + m.off_waves + wave * m.wave_lds;
    uint8_t* outw = wbase + m.off_out;
    const StreamJob* jobs = a.jobs + (size_t)fidx * Cs + cbase;   // jobs[c]: converted channel cbase + c
    const StreamJob j0 = jobs[0];          // in, L, e0, n0, nout are common to a file's channels

    const int64_t first0 = j0.e0 - (int64_t)a.Wb;          // first byte of output 0's window
    const uint32_t sh = (uint32_t)(first0 & 3);            // its misalignment inside the staged dword
    {   // tap fragments: L2 -> LDS once per block; the variant for this byte misalignment
        const uint4* s = reinterpret_cast<const uint4*>(a.tables) + (size_t)sh * (2 * NPG * 64);
        uint4* dl = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < 2 * NPG * 64; i += blockDim.x) dl[i] = s[i];
    }
    __syncthreads();

    const uint32_t nwt = (j0.nout + (M2_TILE - 1)) / M2_TILE;      // wave-tiles in this file
    const uint32_t wstride = gridDim.x * m.nwaves;
    const uint32_t r = lane & 31, h = lane >> 5;

    // ---- staging geometry (does not change from tile to tile: a tile advances the stream by 512*MB bytes) ----
    // Global loads are 16-byte aligned chunks of the channel's stream; the first window dword of the tile
    // is X0 dwords into chunk 0.  LDS dword L (L = 0: that first window dword) lives at L + (L >> LSH):
    // a chunk's dwords k < X0 (they belong to the aligned group before) go to wlo + 4k, the others to
    // whi + 4k; neither run crosses a pad.  The dwords in front of the window (chunk 0, k < X0) land in
    // a dummy slot behind the buffer.
    const uint32_t X0 = (uint32_t)(first0 >> 2) & 3u;
    constexpr uint32_t DUMMY = (uint32_t)m2_stream_bytes(MB, NPG) - 16u;
    uint32_t wlo[PF], whi[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const uint32_t q = lane + 64u * i;
        const uint32_t Lh = 4u * q;                                  // L of dword k = X0
        whi[i] = 4u * (Lh + (Lh >> LSH)) - 4u * X0;
        const uint32_t Ll = 4u * q - X0;                             // L of dword k = 0 (q > 0)
        wlo[i] = q == 0 ? DUMMY : 4u * (Ll + (Ll >> LSH));
    }
    const uint32_t lane16 = lane * 16u;
    const uint32_t Bsz = a.B, Lcall = (uint32_t)j0.L;
    const bool pow2B = Bsz >= 16 && (Bsz & (Bsz - 1)) == 0;
    const uint32_t bshift = pow2B ? 31 - __builtin_clz(Bsz) : 0;
    const uint32_t full_bytes = pow2B ? (Lcall >> bshift) << bshift : 0;   // bytes per channel in full blocks
    const uint32_t jump = (Ct - 1u) * Bsz;                                 // from a block's end to the channel's next block
    const bool fast_layout = pow2B && (uint64_t)full_bytes * Ct < (1ull << 32) && jump < (1u << 24);
    auto tile_ab16 = [&](uint32_t w) -> int32_t { return (int32_t)((first0 + (int64_t)w * (M2_TILE * MB)) & ~(int64_t)15); };

    u32x4 pf[PF];
    auto issue_loads = [&](uint32_t w, uint32_t c) {
        const int32_t ab = tile_ab16(w);
        const uint32_t chf = jobs[c].ch;                                    // the channel's index in the file
        if (fast_layout && ab >= 0 && (uint32_t)ab + 16u * NCHK <= full_bytes) {
            // planar power-of-two blocks, the whole staged range inside the call's full blocks:
            // uniform start of the first block + offset inside it + one `jump` per block boundary crossed
            const uint32_t blk0 = (uint32_t)ab >> bshift, r0 = (uint32_t)ab & (Bsz - 1);
            const uint8_t* base = j0.in + ((uint64_t)(blk0 * Ct + chf) << bshift);
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (lane + 64u * i < (uint32_t)NCHK) {
                    const uint32_t off = r0 + 1024u * i + lane16;
                    const uint32_t o = __umul24(off >> bshift, jump) + off;
                    pf[i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(base) + o);
                }
        } else {
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (lane + 64u * i < (uint32_t)NCHK) pf[i] = gather_chunk(jobs + c, Ct, a.B, a.keep, ab + (int32_t)(lane16 + 1024u * i));
        }
    };
    auto write_lds_x = [&](auto xc) {
        constexpr int X = decltype(xc)::value;
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (lane + 64u * i < (uint32_t)NCHK) {
                const uint32_t v[4] = {pf[i].x, pf[i].y, pf[i].z, pf[i].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<uint32_t*>(wbase + (k < X ? wlo[i] : whi[i]) + 4 * k) = v[k];
            }
    };
    auto write_lds = [&]() {
        if (X0 == 0) write_lds_x(std::integral_constant<int, 0>{});
        else if (X0 == 1) write_lds_x(std::integral_constant<int, 1>{});
        else if (X0 == 2) write_lds_x(std::integral_constant<int, 2>{});
        else write_lds_x(std::integral_constant<int, 3>{});
    };

    // this lane's row window: staged dwords RS*r + 2u + h, u = 0 .. TP-1, padded by one dword per RS
    const uint8_t* rb = wbase + 4u * ((RS + 1) * r + h);
    const v4i* tp = reinterpret_cast<const v4i*>(smem) + lane;      // fragment f: tp[64 * f]
    const uint32_t K1 = 0x01010101u;
#ifndef D2D_M2_NO_VMASK
    // the eight plane masks parked in VGPRs: a VOP2 with only VGPR operands is the cheapest encoding to issue
    uint32_t km[8];
#pragma unroll
    for (int p_ = 0; p_ < 8; ++p_) { km[p_] = K1 << p_; asm volatile("" : "+v"(km[p_])); }
#endif

    // dither keys of the block's channels (uniform)
    uint32_t rkey[CH], rstep[CH], rlo0[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t cc = (uint32_t)c < C ? c : 0;
        rkey[c] = jobs[cc].rng_key; rstep[c] = jobs[cc].rng_kstep; rlo0[c] = jobs[cc].rng_lo0;
    }
    double pk[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) pk[c] = 0.0;

    uint32_t wt = blockIdx.x * m.nwaves + wave;
    if (wt < nwt) issue_loads(wt, 0);
#if D2D_DIAG
    if (dbg & 8) {      // static priority by the wave's slot on its SIMD (waves w, w+4, w+8, w+12 share one)
        const uint32_t slot = wave >> 2;
        if (slot == 1) __builtin_amdgcn_s_setprio(1); else if (slot == 2) __builtin_amdgcn_s_setprio(2); else if (slot == 3) __builtin_amdgcn_s_setprio(3);
    }
    if (dbg & 32) {     // one-off start offset per slot
        const uint32_t slot = wave >> 2;
        for (uint32_t i = 0; i < slot * (dbg >> 8); ++i) __builtin_amdgcn_s_sleep(32);
    }
#endif

    static_assert(EPI != 1 || CH == 2, "the register-packed epilogue is the stereo one");
#if D2D_DIAG
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last = 0;
    auto stamp = [&](int slot) {
        if (dbg & 256) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (slot >= 0) st_sum[slot] += t - st_last;
            st_last = t;
        }
    };
    stamp(-1);
#else
    auto stamp = [](int) {};
#endif
    // constants of the all-integer epilogue, parked in VGPRs (opaque to the compiler so that they stay there)
    int32_t kF = m.fbits;
    uint32_t kSh = 16u - (uint32_t)m.fbits, kSh17 = 17u - (uint32_t)m.fbits, kMask = (1u << m.fbits) - 1u;
    uint32_t kC1 = 0x7feb352dU, kC2 = 0x846ca68bU, kTm = (uint32_t)-32767;
    int32_t kBiasH = (int32_t)(1u << (a.scale_bits - m.fbits));
    // |x| <= qmax - 2 LSB keeps x + d inside the range whatever the dither: v0 within 2^S -+ (qmax - 2) * 2^F
    int32_t kHiSafe = (int32_t)((1u << a.scale_bits) + (((uint32_t)m.qmax_i - 2u) << m.fbits)), kLoSafe = (int32_t)((1u << a.scale_bits) - (((uint32_t)m.qmax_i - 2u) << m.fbits));
    asm volatile("" : "+v"(kF), "+v"(kSh), "+v"(kSh17), "+v"(kMask), "+v"(kC1), "+v"(kC2), "+v"(kTm), "+v"(kBiasH), "+v"(kHiSafe), "+v"(kLoSafe));
    const uint32_t lane_fr = 16u * r + 4u * h;              // the lane's first frame inside a tile
    int32_t vmn[CH], vmx[CH];                               // running extremes of v0 = v + 2^S on the fast path
#pragma unroll
    for (int c = 0; c < CH; ++c) { vmn[c] = 1 << a.scale_bits; vmx[c] = 1 << a.scale_bits; }
    // EPI 1: a full tile's packed frames
    u32x4 pend4[2]; u32x2 pend2[2];
    auto flush_pending = [&](uint32_t pend_wt) {
        uint8_t* gout = reinterpret_cast<uint8_t*>(j0.out) + (size_t)pend_wt * (M2_TILE * 6) + 96u * r + 24u * h;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (dbg & 64) { asm volatile("" :: "v"(pend4[g]), "v"(pend2[g])); continue; }
            *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(gout + 48 * g)) = pend4[g];
            *reinterpret_cast<D2D_GLOBAL u32x2*>(as_global(gout + 48 * g + 16)) = pend2[g];
        }
    };
    for (; wt < nwt; wt += wstride) {
        const bool full = wt * (uint32_t)M2_TILE + (uint32_t)M2_TILE <= j0.nout;
        uint32_t held[8];                                   // EPI 1: channel 0's eight samples
        static_for<0, CH>([&](auto cc_) {
            constexpr int c = decltype(cc_)::value;
            if ((uint32_t)c >= C) return;
            wave_sync2();                                   // the previous chain's reads are done (same wave: in order)
            if (!(dbg & 4)) {
                write_lds();
                // next unit's bytes: in flight during this chain
                if ((uint32_t)c + 1 < C) issue_loads(wt, c + 1);
                else if (wt + wstride < nwt) issue_loads(wt + wstride, 0);
            }
            wave_sync2();
            stamp(0);

            // ---- the chain: TP pair steps, two groups of eight phases ----
#if D2D_DIAG
            if (dbg & 16) __builtin_amdgcn_s_setprio(3);
#endif
            v16i acc[2];
            if (dbg & 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc[0][i] = (int)(lane * 128u * (i & 1)); acc[1][i] = (int)(r * 256u * (i & 1)); }
            } else {
                const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                static_for<0, TP>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const uint32_t W = *reinterpret_cast<const uint32_t*>(rb + 4 * (2 * u + ((2 * u) >> LSH)));
#ifndef D2D_M2_NO_VMASK
                    // plane 0 goes in UNMASKED where the limb sums allow it: as int8 a stream byte is the sum of its eight
                    // masked planes, so the table's other planes carry (their tap - the plane-0 tap) and the products still
                    // add up to 2^7 * q * bit exactly -- one v_and in eight less
                    const v4i lo = {(int)(m2_unmask0(NPG) ? W : (W & km[0])), (int)(W & km[1]), (int)(W & km[2]), (int)(W & km[3])};
                    const v4i hi = {(int)(W & km[4]), (int)(W & km[5]), (int)(W & km[6]), (int)(W & km[7])};
#else
                    const v4i lo = {(int)(W & K1), (int)(W & (K1 << 1)), (int)(W & (K1 << 2)), (int)(W & (K1 << 3))};
                    const v4i hi = {(int)(W & (K1 << 4)), (int)(W & (K1 << 5)), (int)(W & (K1 << 6)), (int)(W & (K1 << 7))};
#endif
                    static_for<0, 2>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        constexpr int pp = u - MB * g;
                        if constexpr (pp >= 0 && pp < NPG) {
                            const v4i F0 = tp[64 * (2 * pp)], F1 = tp[64 * (2 * pp + 1)];
                            if constexpr (pp == 0) acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F0, lo, zero, 0, 0, 0);
                            else acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F0, lo, acc[g], 0, 0, 0);
                            acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F1, hi, acc[g], 0, 0, 0);
                        }
                    });
                });
            }

#if D2D_DIAG
            if (dbg & 16) { asm volatile("" :: "v"(acc[0][15]), "v"(acc[1][15])); __builtin_amdgcn_s_setprio(0); }
#endif
#if D2D_DIAG
            if (dbg & 256) { asm volatile("" :: "v"(acc[0][15]), "v"(acc[1][15])); }
#endif
            stamp(1);
            // ---- epilogue: lane (r, h) owns outputs 16r + 8g + 4h + k of the tile, k = 0..3 ----
            const uint32_t nl_base = wt * (uint32_t)M2_TILE + 16u * r + 4u * h;      // + 8g + k
            auto recombine = [&](const v16i& A, int k, auto wide_tag) -> double {
                constexpr bool WIDE = decltype(wide_tag)::value;
                double accd;
                if constexpr (WIDE) {
                    accd = fma((double)A[4 * k + 3], 16777216.0,
                               fma((double)A[4 * k + 2], 65536.0, fma((double)A[4 * k + 1], 256.0, (double)A[4 * k])));
                } else {
                    const int lo = A[4 * k] + (A[4 * k + 1] << 8), hi = A[4 * k + 2] + (A[4 * k + 3] << 8);
                    accd = fma((double)hi, 65536.0, (double)lo);          // exact: 128 * sum_k q_k b_k
                }
                return fma(accd, m.c1, -m.c0);                            // x = y*c0 with ONE rounding
            };
            auto noise = [&](uint32_t nl) -> uint32_t {
                const uint32_t nlo = (uint32_t)j0.n0 + nl;
                uint32_t z = nlo + rkey[c] + (nlo < rlo0[c] ? rstep[c] : 0u);
                z ^= z >> 16; z *= 0x7feb352dU;
                z ^= z >> 15; z *= 0x846ca68bU;
                z ^= z >> 16;
                return z;
            };
            // one integer-depth sample: dither, round half away from zero, clip
            auto quant = [&](double x, uint32_t nl, auto kind_tag) -> int32_t {
                constexpr int KIND = decltype(kind_tag)::value;
                double q;
                if constexpr (KIND == 1 || KIND == 2) {
                    const uint32_t z = noise(nl);
                    const uint32_t term = KIND == 1 ? (z & 0xFFFFu) + (z >> 16) + 1u : 2u * (z >> 16) + 1u;
                    q = x + fma((double)term, m.dmul, m.dadd);
                } else {
                    q = x + 0.0;                                          // what quantise_int() does for "none" (a -0 becomes +0)
                }
                int32_t ri, o;
                const double t = q + copysign(0.5, q);
                asm("v_cvt_i32_f64 %0, %1" : "=v"(ri) : "v"(t));         // truncates toward zero, saturates
                const int32_t qmax_v = m.qmax_i;
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(o) : "v"(ri), "s"(m.qmin_i), "v"(qmax_v));
                return o;
            };
            // all-integer variant for unit gain (level 0 dB): x = v * 2^-F LSB with v = sum q s an integer, the
            // dither term an integer number of 2^-16 LSB -- the same real numbers, no f64 instruction.
            //   z = v*2^(16-F) + dterm   (units of 2^-16 LSB, dterm = t - 65535 (T) | 2*hi16 - 65535 ... (R))
            //   r = round half away from zero of z / 65536
            auto quant_int = [&](const v16i& A, int k, uint32_t nl, auto kind_tag, uint32_t& vabs) -> int32_t {
                constexpr int KIND = decltype(kind_tag)::value;
                // 128*sum q b = t0 + 65536*t1 with t0 a multiple of 128; v = 2*sum q b - 2^S (wraps are harmless: |v| < 2^31)
                const int32_t t0 = A[4 * k] + (A[4 * k + 1] << 8);
                const uint32_t t1 = (uint32_t)A[4 * k + 2] + ((uint32_t)A[4 * k + 3] << 8);
                const int32_t v = (int32_t)((uint32_t)(t0 >> 6) + (t1 << 10) - (1u << a.scale_bits));
                vabs = (uint32_t)(v < 0 ? -v : v);
                const int F = m.fbits;                                    // 0 < F <= 16 on this path
                const int32_t vh = v >> F;                                // floor(x)
                const uint32_t vl = (uint32_t)v & ((1u << F) - 1u);       // its fraction, F bits
                int32_t rr;
                if constexpr (KIND == 2) {
                    // units of 2^-17 LSB: (2*hi16 + 1)*2^-17 - 1/2
                    const uint32_t z = noise(nl);
                    const int32_t w = (int32_t)(vl << (17 - F)) + (int32_t)(2u * (z >> 16) + 1u) - 65536;
                    const int32_t neg = (vh + (w >> 17)) >> 31;           // -1 when x + d < 0
                    rr = vh + ((w + 65536 + neg) >> 17);
                } else {
                    // units of 2^-16 LSB: (lo16 + hi16 + 1)*2^-16 - 1, or nothing
                    int32_t w = (int32_t)(vl << (16 - F));
                    if constexpr (KIND == 1) {
                        const uint32_t z = noise(nl);
                        w += (int32_t)((z & 0xFFFFu) + (z >> 16)) - 65535;
                    }
                    const int32_t neg = (vh + (w >> 16)) >> 31;
                    rr = vh + ((w + 32768 + neg) >> 16);
                }
                int32_t o;
                const int32_t qmax_v = m.qmax_i;
                asm("v_med3_i32 %0, %1, %2, %3" : "=v"(o) : "v"(rr), "s"(m.qmin_i), "v"(qmax_v));
                return o;
            };


            if (dbg & 2) {
                if (acc[0][0] == 0x12345 && acc[1][5] == 77 && acc[0][9] + acc[1][13] + acc[0][15] + acc[1][2] == 99) outw[lane] = 1;
            } else if constexpr (EPI == 1) {
                // Stereo 24-bit: no LDS round trip.  Per group the lane owns 4 consecutive frames of both
                // channels = 24 contiguous output bytes.
                auto body = [&](auto kind_tag, auto intq_tag) {
                    constexpr bool INTQ = decltype(intq_tag)::value;
                    uint32_t cur[8];
                    if constexpr (INTQ) {
                        constexpr int KIND = decltype(kind_tag)::value;
                        // Fast form: whole tile, no clip possible (running extremes of v inside the safe range), the
                        // dither counter does not wrap inside the tile, no exact rounding tie in any lane.  Anything
                        // else recomputes the tile with the general per-sample code below.  All constants sit in
                        // VGPRs: a VOP2 with only VGPR operands issues at twice the rate of a VOP3 or of one that
                        // reads an SGPR (tools/ubench/issue_rate.hip).
                        const uint32_t first = (uint32_t)j0.n0 + wt * (uint32_t)M2_TILE;      // lo32 of the tile's first output index
                        bool slow = !full || first > 0xFFFFFFFFu - (uint32_t)M2_TILE;
                        if (!slow) {
                            const uint32_t key_eff = rkey[c] + (first < rlo0[c] ? rstep[c] : 0u);
                            const uint32_t zb = first + key_eff + lane_fr;                  // + 8g + k
                            uint32_t anytie = 0;
#pragma unroll
                            for (int g = 0; g < 2; ++g)
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const v16i& A = acc[g];
                                    // v0 = 2*sum q b = (A0 >> 6) + 4*A1 + 2^10*A2 + 2^18*A3  (A0 is a multiple of 128)
                                    // (negative taps can take it below zero: signed, |v0 - 2^S| <= sum|q| keeps it inside int32)
                                    uint32_t u0 = (uint32_t)(A[4 * k] >> 6);
                                    u0 = ((uint32_t)A[4 * k + 1] << 2) + u0;
                                    u0 = ((uint32_t)A[4 * k + 2] << 10) + u0;
                                    u0 = ((uint32_t)A[4 * k + 3] << 18) + u0;
                                    const int32_t v0 = (int32_t)u0;
                                    vmn[c] = min(vmn[c], v0); vmx[c] = max(vmx[c], v0);
                                    const uint32_t vhb = (uint32_t)(v0 >> kF);             // floor(x) + 2^(S-F)
                                    const uint32_t vl = u0 & kMask;                        // fraction of x, F bits
                                    int32_t cc;
                                    if constexpr (KIND == 0) {
                                        const int32_t w = (int32_t)((vl << kSh) + 32768u);
                                        anytie |= (uint32_t)((w & 0xFFFF) == 0);
                                        cc = w >> 16;
                                    } else {
                                        uint32_t z = zb + (uint32_t)(8 * g + k);
                                        if (!(dbg & 128)) {
                                        z ^= z >> 16; z *= kC1;
                                        z ^= z >> 15; z *= kC2;
                                        z ^= z >> 16;
                                        }
                                        if constexpr (KIND == 1) {
                                            // units of 2^-16 LSB: x + (lo16 + hi16 + 1)*2^-16 - 1 + 1/2
                                            const int32_t w = (int32_t)((vl << kSh) + __builtin_amdgcn_sad_u16(z, 0u, kTm));
                                            anytie |= (uint32_t)((w & 0xFFFF) == 0);
                                            cc = w >> 16;
                                        } else {
                                            // units of 2^-17 LSB: x + (2*hi16 + 1)*2^-17 - 1/2 + 1/2: odd, never a tie
                                            const int32_t w = (int32_t)((vl << kSh17) + ((z >> 15) | 1u));
                                            cc = w >> 17;
                                        }
                                    }
                                    cur[4 * g + k] = (uint32_t)((int32_t)(vhb + (uint32_t)cc) - kBiasH);
                                }
                            // clip or tie anywhere in the wave: redo the tile the careful way
                            slow = __builtin_amdgcn_ballot_w64(anytie != 0 || vmx[c] > kHiSafe || vmn[c] < kLoSafe) != 0;
                        }
                        if (slow) {
                            uint32_t vmax = 0;
#pragma unroll
                            for (int g = 0; g < 2; ++g)
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    uint32_t va;
                                    cur[4 * g + k] = (uint32_t)quant_int(acc[g], k, nl_base + 8u * g + k, kind_tag, va);
                                    vmax = max(vmax, full || nl_base + 8u * g + k < j0.nout ? va : 0u);
                                }
                            pk[c] = fmax(pk[c], ldexp((double)vmax, -m.fbits));   // |x| = |v| * 2^-F exactly
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 2; ++g)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const double x = recombine(acc[g], k, std::false_type{});
                                const double cand = full || nl_base + 8u * g + k < j0.nout ? x : 0.0;
                                asm("v_max_f64 %0, %1, |%2|" : "=v"(pk[c]) : "v"(pk[c]), "v"(cand));
                                cur[4 * g + k] = (uint32_t)quant(x, nl_base + 8u * g + k, kind_tag);
                            }
                    }
                    if constexpr (c == 0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) held[i] = cur[i];
                    } else {
                        if (full) {
#pragma unroll
                            for (int g = 0; g < 2; ++g) {
                                // frames k, k+1 -> 12 bytes: [L0 L1 L2 R0 | R1 R2 L0' L1' | L2' R0' R1' R2']
                                const uint32_t La = held[4 * g], Ra = cur[4 * g], Lb = held[4 * g + 1], Rb = cur[4 * g + 1];
                                const uint32_t Lc = held[4 * g + 2], Rc = cur[4 * g + 2], Ld = held[4 * g + 3], Rd = cur[4 * g + 3];
                                pend4[g] = u32x4{__builtin_amdgcn_perm(Ra, La, 0x04020100u), __builtin_amdgcn_perm(Lb, Ra, 0x05040201u),
                                                 __builtin_amdgcn_perm(Rb, Lb, 0x06050402u), __builtin_amdgcn_perm(Rc, Lc, 0x04020100u)};
                                pend2[g] = u32x2{__builtin_amdgcn_perm(Ld, Rc, 0x05040201u), __builtin_amdgcn_perm(Rd, Ld, 0x06050402u)};
                            }
                            flush_pending(wt);
                        } else {
                            // the file's last, partial tile: frame by frame, three 2-byte stores each
                            uint8_t* gout = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * (M2_TILE * 6) + 96u * r + 24u * h;
#pragma unroll 1
                            for (int i = 0; i < 8; ++i) {
                                const uint32_t g = (uint32_t)i >> 2, k = (uint32_t)i & 3u;
                                if (nl_base + 8u * g + k < j0.nout) {
                                    uint32_t L = 0, R = 0;
#pragma unroll
                                    for (int q = 0; q < 8; ++q) { L = i == q ? held[q] : L; R = i == q ? cur[q] : R; }
                                    D2D_GLOBAL uint16_t* p16 = reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(gout + 48u * g + 6u * k));
                                    p16[0] = (uint16_t)L; p16[1] = (uint16_t)(((L >> 16) & 0xFFu) | (R << 8)); p16[2] = (uint16_t)(R >> 8);
                                }
                            }
                        }
                    }
                };
                auto by_kind = [&](auto intq_tag) {
                    if (m.dkind == 1) body(std::integral_constant<int, 1>{}, intq_tag);
                    else if (m.dkind == 2) body(std::integral_constant<int, 2>{}, intq_tag);
                    else body(std::integral_constant<int, 0>{}, intq_tag);
                };
                if (m.intq) by_kind(std::true_type{}); else by_kind(std::false_type{});
            } else {
                // EPI 0: through the wave's LDS out-slice; EPI 2: straight to the stage-A scratch
                auto finish = [&](auto full_tag, auto wide_tag, auto kind_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    constexpr int KIND = decltype(kind_tag)::value;       // 0 none, 1 triangular, 2 rectangular, 3 float FPD
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const uint32_t fl0 = 16u * r + 8u * g + 4u * h;                  // frame inside the tile
                        const uint32_t nl0 = wt * (uint32_t)M2_TILE + fl0;
                        if constexpr (EPI == 2 && !decltype(wide_tag)::value) {
                            // the scratch wants the exact integer y * 2^S = sum q s itself: (A0 >> 6) + 4*A1 + 2^10*A2 + 2^18*A3 - 2^S in
                            // int32 (A0 is a multiple of 128; wraps are harmless, |sum q s| < 2^31) -- no f64 instruction
                            int32_t iv[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const v16i& A = acc[g];
                                uint32_t u0 = (uint32_t)(A[4 * k] >> 6);
                                u0 = ((uint32_t)A[4 * k + 1] << 2) + u0;
                                u0 = ((uint32_t)A[4 * k + 2] << 10) + u0;
                                u0 = ((uint32_t)A[4 * k + 3] << 18) + u0;
                                iv[k] = (int32_t)(u0 - (1u << a.scale_bits));
                            }
                            if (FULL || nl0 + 3 < j0.nout) {
                                *reinterpret_cast<D2D_GLOBAL i32x4*>(as_global(jobs[c].xs + nl0)) = i32x4{iv[0], iv[1], iv[2], iv[3]};
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (nl0 + k < j0.nout) as_global(jobs[c].xs)[nl0 + k] = iv[k];
                            }
                            continue;
                        }
                        double xv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) xv[k] = recombine(acc[g], k, wide_tag);
                        if constexpr (EPI == 2) {
                            // stage A of the 48k cascade / input of the noise-shaping pass: the exact integers y*2^S
                            if (FULL || nl0 + 3 < j0.nout) {
                                *reinterpret_cast<D2D_GLOBAL i32x4*>(as_global(jobs[c].xs + nl0)) =
                                    i32x4{(int32_t)xv[0], (int32_t)xv[1], (int32_t)xv[2], (int32_t)xv[3]};
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (nl0 + k < j0.nout) as_global(jobs[c].xs)[nl0 + k] = (int32_t)xv[k];
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const bool ok = FULL || (nl0 + k < j0.nout);
                                const double cand = ok ? xv[k] : 0.0;
                                asm("v_max_f64 %0, %1, |%2|" : "=v"(pk[c]) : "v"(pk[c]), "v"(cand));
                            }
                            if (a.epi.bits == 32) {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    *reinterpret_cast<float*>(outw + (size_t)((fl0 + k) * C + c) * 4) =
                                        KIND == 3 ? finish_f32(a.epi, xv[k], noise(nl0 + k)) : (float)xv[k];
                            } else {
                                int32_t iv[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    if constexpr (KIND == 1) iv[k] = quant(xv[k], nl0 + k, std::integral_constant<int, 1>{}) << m.qsh;
                                    else if constexpr (KIND == 2) iv[k] = quant(xv[k], nl0 + k, std::integral_constant<int, 2>{}) << m.qsh;
                                    else iv[k] = quant(xv[k], nl0 + k, std::integral_constant<int, 0>{}) << m.qsh;
                                }
                                if (sb == 2) {
#pragma unroll
                                    for (int k = 0; k < 4; ++k)
                                        *reinterpret_cast<uint16_t*>(outw + (size_t)((fl0 + k) * C + c) * 2) = (uint16_t)iv[k];
                                } else {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        uint8_t* p = outw + (size_t)((fl0 + k) * C + c) * 3;
                                        p[0] = (uint8_t)iv[k]; p[1] = (uint8_t)(iv[k] >> 8); p[2] = (uint8_t)(iv[k] >> 16);
                                    }
                                }
                            }
                        }
                    }
                };
                auto kinds = [&](auto full_tag, auto wide_tag) {
                    using K0 = std::integral_constant<int, 0>;
                    if constexpr (EPI == 2) { finish(full_tag, wide_tag, K0{}); return; }
                    const uint32_t kind = a.epi.bits == 32 ? (a.epi.dither == 'F' ? 3u : 0u) : m.dkind;
                    if (kind == 1) finish(full_tag, wide_tag, std::integral_constant<int, 1>{});
                    else if (kind == 2) finish(full_tag, wide_tag, std::integral_constant<int, 2>{});
                    else if (kind == 3) finish(full_tag, wide_tag, std::integral_constant<int, 3>{});
                    else finish(full_tag, wide_tag, K0{});
                };
                auto wides = [&](auto full_tag) { if (m.wide) kinds(full_tag, std::true_type{}); else kinds(full_tag, std::false_type{}); };
                if (full) wides(std::true_type{}); else wides(std::false_type{});
            }
            stamp(2);
        });
        if (EPI == 0 && !(dbg & 2)) {
            wave_sync2();
            // the wave-tile's interleaved frames: LDS -> HBM, 16 bytes per lane per store
            const uint32_t left = j0.nout - wt * (uint32_t)M2_TILE;
            const uint32_t nfr = left < (uint32_t)M2_TILE ? left : (uint32_t)M2_TILE;
            if (m.ngroups == 1) {
                const uint32_t nb = nfr * fbytes;
                uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * M2_TILE * fbytes;
                const uint32_t nb16 = nb & ~15u;
                for (uint32_t i = lane * 16; i < nb16; i += 64 * 16)
                    *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(g + i)) = *reinterpret_cast<const u32x4*>(outw + i);
                for (uint32_t i = nb16 + lane; i < nb; i += 64) as_global(g)[i] = outw[i];
            } else {
                // this group's `fbytes` bytes of every frame sit sb*cbase bytes into the file's frame
                const uint32_t gstride = sb * Cs;
                uint8_t* g = reinterpret_cast<uint8_t*>(j0.out) + (size_t)wt * M2_TILE * gstride + sb * cbase;
                for (uint32_t fr = lane; fr < nfr; fr += 64)
                    for (uint32_t b = 0; b < fbytes; ++b) as_global(g)[(size_t)fr * gstride + b] = outw[fr * fbytes + b];
            }
        }
    }
#if D2D_DIAG
    if ((dbg & 256) && lane == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(&d2d_m2_stamps[i], st_sum[i]);
#endif
    if (EPI != 2) {
        // peak meter: |x| was tracked in the scaled domain; undo the power-of-two part exactly
        const double unscale = a.epi.bits == 32 ? 1.0 : 1.0 / (double)(1u << (a.epi.bits - 1));
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if ((uint32_t)c >= C) continue;
            double p = pk[c];
            if (EPI == 1 && m.intq) {
                const int32_t b = 1 << a.scale_bits;
                const int32_t dev = max(vmx[c] - b, b - vmn[c]);
                p = fmax(p, ldexp((double)dev, -m.fbits));
            }
            p *= unscale;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) p = fmax(p, __shfl_xor(p, o));
            if (lane == 0 && p > 0.0)
                atomicMax(reinterpret_cast<unsigned long long*>(jobs[c].peak), (unsigned long long)__double_as_longlong(p));
        }
    }
}

// ---- host side -------------------------------------------------------------------------------

int mfma2_pairs(int M, int N) { return (N + 7 * M + 24 + 63) / 64; }

static inline int8_t limb_of2(int64_t v, int l) {
    // balanced base-256 digits: v = d0 + d1*2^8 + d2*2^16 + d3*2^24, every d in [-128, 127]
    int8_t dgt = 0;
    for (int i = 0; i <= l; ++i) {
        int64_t dd = ((v + 128) & 255) - 128;
        dgt = (int8_t)dd;
        v = (v - dd) / 256;
    }
    return dgt;
}

// Tap fragments [4 byte shifts][2*NPG][64 lanes][16 bytes].  Fragment 2*pp + n serves pair step pp of a
// group's window, bit positions 4n .. 4n+3 of every byte.  Lane l supplies matrix row (l & 31) =
// 4*slot + limb for the K slots of lane half hh = l >> 5, i.e. the staged dword 2*pp + hh of the window;
// slot j of the lane = byte (j & 3), plane (j >> 2) -> bit position p = 4n + (j >> 2) of that byte,
// which arrives as 2^p (p = 7: -128): the table holds q * 2^(7-p), negated for p = 7.
std::vector<int8_t> build_mfma2_tables(const d2d_filter_def& f, bool msb_first, bool unmask0_wanted) {
    const int NPG = mfma2_pairs(f.M, f.ntaps);
    const bool unmask0 = unmask0_wanted && m2_unmask0(NPG);
    const size_t per = (size_t)(2 * NPG) * 64 * 16;
    std::vector<int8_t> t(4 * per, 0);
    for (int sh = 0; sh < 4; ++sh)                                  // window starts `sh` bytes into its first dword
        for (int fr = 0; fr < 2 * NPG; ++fr)
            for (int l = 0; l < 64; ++l) {
                const int pp = fr >> 1, n = fr & 1;
                const int row = l & 31, hh = l >> 5, limb = row & 3;
                // D row i lands in lane half (i >> 2) & 1, register group i >> 3: give that slot output
                // phase 4*half + group, so lane (r, half) owns the four CONSECUTIVE outputs 4*half + k of a group
                const int ph = 4 * ((row >> 2) & 1) + (row >> 3);
                for (int j = 0; j < 16; ++j) {
                    const int p = 4 * n + (j >> 2);
                    const int wb = 32 * (2 * pp + hh) + 8 * (j & 3) + p;                     // bit of the staged window
                    const int tau = (msb_first ? (wb & ~7) + 7 - (wb & 7) : wb) - 8 * sh;   // its time index in the window
                    const int tap = tau - ph * f.M;
                    auto entry = [&](int pp_) -> int64_t {                                   // q * 2^(7-p) (p = 7: -q) of bit position pp_ of this byte
                        const int wb_ = 32 * (2 * pp + hh) + 8 * (j & 3) + pp_;
                        const int tau_ = (msb_first ? (wb_ & ~7) + 7 - (wb_ & 7) : wb_) - 8 * sh;
                        const int tap_ = tau_ - ph * f.M;
                        if (tau_ < 0 || tap_ < 0 || tap_ >= f.ntaps) return 0;
                        const int64_t q = tap_q(f, tap_);
                        return pp_ == 7 ? -q : q * (int64_t)(1 << (7 - pp_));
                    };
                    (void)tau; (void)tap;
                    int64_t T = entry(p);
                    if (unmask0 && p != 0) T -= entry(0);                                 // plane 0 arrives unmasked (see the kernel)
                    t[sh * per + ((size_t)fr * 64 + l) * 16 + j] = limb_of2(T, limb);
                }
            }
    return t;
}

// (MB, NPG) pairs with a compiled kernel
#ifdef D2D_M2_DEV
#define D2D_M2_SHAPES(X) X(4, 13)
#else
// M = 32 and 64 only: with 1 or 2 bytes per output a 512-output tile holds so little stream that the per-tile work
// (staging, waits, the epilogue) outweighs the shorter chain and the one-group kernel is faster (measured: DSD64 -> 352.8 kHz
// float 451 against 514 Gsamples/s, the M = 8 stage A of the 48k cascade 4.7 against 3.8 ms)
#define D2D_M2_SHAPES(X) X(4, 10) X(4, 12) X(4, 13) X(8, 19) X(8, 24) X(8, 25)
#endif

bool mfma2_supported(int M, int N) {
    const int MB = M / 8, NPG = mfma2_pairs(M, N);
#define X(mb, npg) if (MB == mb && NPG == npg) return true;
    D2D_M2_SHAPES(X)
#undef X
    return false;
}

// the epilogue flavour of a launch: 2 = the integers for the stage-A scratch, 1 = stereo 24-bit packed in registers, 0 = anything via LDS
static int mfma2_epilogue(const FirArgs& a, const Mfma2Args& m) {
    if (a.to_scratch) return 2;
    return a.epi.channels == 2 && a.epi.sample_bytes == 3 && m.qsh == 0 && !m.wide ? 1 : 0;
}

// the pipelined kernel serves the register-packed stereo flavour with the all-integer requantiser; its accumulators start
// from -2^(S-18) in the limb-3 rows
static bool mfma3_eligible(const FirArgs& a, const Mfma2Args& m, int MB, int NPG, int NT) {
    // stereo, 24-bit packed or 16-bit frames, the all-integer requantiser (unit gain)
    // the exact integers for the stage-A scratch: every channel pair of an even channel count
    if (a.to_scratch) return a.epi.channels >= 2 && a.epi.channels % 2 == 0 && !m.wide && a.scale_bits >= 18 && a.scale_bits <= 30 &&
                             a.sum_abs_q != 0 && a.sum_abs_q < (1ull << 31) && mfma3_scr_supported(MB, NPG);
    const bool frames_ok = !a.to_scratch && a.epi.channels == 2 && (a.epi.sample_bytes == 3 || a.epi.sample_bytes == 2) && m.qsh == 0 && !m.wide;
    const bool shape_ok = a.scale_bits >= 18 && a.scale_bits <= 30 && mfma3_supported(MB, NPG, NT);
    // ... or 32-bit float at 0 dB without the float dither: the sample is (float)v * 2^-S
    const bool noint = (a.dbg_flags & D2D_DBG_NO_INTQ) != 0;
    const bool float_ok = !noint && !a.to_scratch && a.epi.channels == 2 && a.epi.bits == 32 && a.epi.dither != 'F' && a.epi.gain == 1.0 && !m.wide &&
                          a.sum_abs_q != 0 && a.sum_abs_q < (1ull << 31);
    // (M = 8 float frames too since the pipelined kernel stages that shape's frames through LDS: 4.18 against 4.50 ms on the one-group kernel)
    if (float_ok && shape_ok) return true;
    if (m.gainq && shape_ok && MB < 4) return true;            // another level in dB, M = 8 and 16: the f64 requantiser inside the pipelined epilogue
    return frames_ok && m.intq && shape_ok;
}

static void mfma2_geometry(const FirArgs& a, int MB, int NPG, Mfma2Args& m, size_t& smem);

// Planar frames of an even channel count above two on the fp6 kernel: a wave converts every pair of a tile and stores whole frames
// (d2d_kernels_mx.hip, NPR).  The pairs per wave, or 1.  (Byte-interleaved multichannel input reaches the kernel as the engine's planar copy.)
static uint32_t mx_pairs(const FirArgs& a, int MB, int N) {
    const uint32_t C = a.epi.channels;
    if (a.to_scratch || a.mono2 || a.il2 || a.coop || C < 4 || C % 2) return 1u;
    if (a.B < 16 || (a.B & (a.B - 1)) != 0) return 1u;
    return mx_pairs_supported(MB, N, (int)(C / 2u)) ? C / 2u : 1u;
}

// the same conversions as mfma3_eligible, shape apart (the caller checks mx_supported and the engine mx_exact)
static bool mx_eligible(const FirArgs& a, const Mfma2Args& m, int MB, int N) {
    const bool range_ok = a.scale_bits >= 20 && a.scale_bits <= 30 && a.sum_abs_q != 0 && a.sum_abs_q < (1ull << 31) && a.mx_exact;
    if (a.to_scratch) return a.epi.channels >= 2 && a.epi.channels % 2 == 0 && range_ok;
    const bool noint = (a.dbg_flags & D2D_DBG_NO_INTQ) != 0;
    const bool stereo = (a.epi.channels == 2 || mx_pairs(a, MB, N) > 1u) && m.qsh == 0;       // (or whole frames of several pairs)
    const bool float_ok = !noint && stereo && a.epi.bits == 32 && a.epi.dither != 'F' && a.epi.gain == 1.0;
    const bool frames_ok = stereo && (a.epi.sample_bytes == 3 || a.epi.sample_bytes == 2) && m.intq;
    return range_ok && (float_ok || frames_ok);
}

int mfma2_pipelined(const FirArgs& a, int M, int N) {
    if (a.taps32) return 5;                                   // (the engine has checked mx_wide_supported and mx_wide_exact)
    if (a.dbg_flags & D2D_DBG_NO_PIPE) return 0;
    const int MB = M / 8, NPG = mfma2_pairs(M, N);
    const bool nomx = (a.dbg_flags & D2D_DBG_NO_MX) != 0;
    Mfma2Args m{}; size_t smem = 0;
    mfma2_geometry(a, MB, NPG, m, smem);
    // M = 128 (DSD256 -> 88.2 kHz, DSD512 -> 176.4 kHz): only the fp6 kernel has the LDS for that tap table; its conditions are its own
    // (S = 30: no biased accumulators, and the int8 kernels' limb-sum bound `wide` does not apply)
    if (MB == 16) {
        const bool noint16 = (a.dbg_flags & D2D_DBG_NO_INTQ) != 0;
        const bool range_ok = a.scale_bits >= 20 && a.scale_bits <= 30 && a.sum_abs_q != 0 && a.sum_abs_q < (1ull << 31) && a.mx_exact;
        if (nomx || !mx_supported(MB, N) || !range_ok || noint16) return 0;
        if (a.to_scratch) return a.epi.channels >= 2 && a.epi.channels % 2 == 0 ? 5 : 0;
        const bool depth_ok = a.epi.bits == 32 ? true : ((a.epi.bits == 24 || a.epi.bits == 20 || a.epi.bits == 16) && m.fbits > 0 && m.fbits <= 16 && a.epi.dither != 'F');
        const bool mp = mx_pairs(a, MB, N) > 1u;
        if ((a.epi.channels != 2 && !mp) || !depth_ok || a.epi.dither == 'N') return 0;
        if (a.epi.gain == 1.0 && m.qsh == 0 && !(a.epi.bits == 32 && a.epi.dither == 'F')) return 5;
        if (mp) return 0;                                          // (several pairs per wave: unit gain only)
        return mx_gain_supported(MB, N) && !(a.dbg_flags & D2D_DBG_NO_GAINQ) ? 5 : 0;
    }
    // the fp6 x fp4 kernel (d2d_kernels_mx.hip) serves what the pipelined int8 kernel serves at M = 32 and 64: 5
    if (!nomx && mx_supported(MB, N) && mx_eligible(a, m, MB, N)) return 5;
    // ... and stereo frames at another level than 0 dB (its gain flavours)
    if (!nomx && m.gainq && mx_gain_supported(MB, N) && a.mx_exact && a.scale_bits >= 20 && a.scale_bits <= 30 && (a.epi.bits == 32 || a.epi.sample_bytes == 2 || a.epi.sample_bytes == 3)) return 5;
    if (!mfma2_supported(M, N) && !mfma3_supported(MB, NPG, N) && !(a.to_scratch && mfma3_scr_supported(MB, NPG))) return 0;
    if (!mfma3_eligible(a, m, MB, NPG, N)) return 0;
    return 3;
}

static void mfma2_geometry(const FirArgs& a, int MB, int NPG, Mfma2Args& m, size_t& smem) {
    m.ngroups = a.epi.channels <= 2 ? 1u : (a.epi.channels + 1u) / 2u;
    m.npairs = 1u;
    const uint32_t C = a.epi.channels <= 2 ? a.epi.channels : 2u;
    m.f = a;
    m.c0 = a.to_scratch ? ldexp(1.0, a.scale_bits) : (a.epi.bits == 32 ? a.epi.gain : a.epi.scale);   // scratch: the integer y*2^S
    m.c1 = ldexp(m.c0, 1 - a.scale_bits - 7);     // exact: a power-of-two multiple of c0
    m.dkind = a.epi.dither == 'T' ? 1u : (a.epi.dither == 'R' ? 2u : 0u);
    m.dmul = a.epi.dither == 'T' ? 0x1p-16 : (a.epi.dither == 'R' ? 0x1p-17 : 0.0);
    m.dadd = a.epi.dither == 'T' ? -1.0 : (a.epi.dither == 'R' ? -0.5 : 0.0);
    m.qsh = a.epi.bits == 20 ? 4u : 0u;
    m.qmin_i = a.epi.bits == 32 ? 0 : -(1 << (a.epi.bits - 1)); m.qmax_i = a.epi.bits == 32 ? 0 : (1 << (a.epi.bits - 1)) - 1;
    // |limb sum| <= (bytes of a group's window) * 255 * 128; below 2^23 the pairs recombine in int32
    m.wide = m2_unmask0(NPG) ? 0u : ((uint64_t)NPG * 8u * 255u * 128u >= (1u << 23) ? 1u : 0u);
    m.fbits = a.scale_bits - ((int)a.epi.bits - 1);
    const bool noint = (a.dbg_flags & D2D_DBG_NO_INTQ) != 0;
    // (the fast form carries v0 = v + 2^S in an int32: 2^S + sum|q| has to stay below 2^31)
    m.intq = (!noint && !a.to_scratch && a.epi.bits != 32 && a.epi.gain == 1.0 && !m.wide && m.fbits > 0 && m.fbits <= 16 &&
              a.sum_abs_q != 0 && (1ull << a.scale_bits) + a.sum_abs_q < (1ull << 31)) ? 1u : 0u;
    // stereo 16/24-bit (dither T, R, none) or float (no float dither) frames at another level than 0 dB, and 20-bit frames and the float dither (the CLI's default for -b 32) at any level
    // (the all-integer requantiser has no 20-in-24 form; the f64 one shifts its result)
    m.gainq = (!noint && !(a.dbg_flags & D2D_DBG_NO_GAINQ) && !a.to_scratch && a.epi.channels == 2 && (a.epi.gain != 1.0 || m.qsh != 0 || (a.epi.bits == 32 && a.epi.dither == 'F')) && !m.wide &&
               (a.epi.dither != 'F' || a.epi.bits == 32) && a.epi.dither != 'N' && (a.epi.bits == 32 || (m.fbits > 0 && m.fbits <= 16)) && a.sum_abs_q != 0 && a.sum_abs_q < (1ull << 31)) ? 1u : 0u;
    m.off_waves = (uint32_t)(2 * NPG) * 1024u;
    m.off_out = (uint32_t)m2_stream_bytes(MB, NPG);
    // (only the LDS-staged epilogue needs the output slice)
    const bool lds_out = mfma2_epilogue(a, m) == 0;
    m.wave_lds = m.off_out + (lds_out ? (((uint32_t)M2_TILE * C * a.epi.sample_bytes + 15u) & ~15u) : 0u);
#if D2D_DIAG
    { static const char* e = getenv("D2D_DBG"); m.dbg = e ? (uint32_t)atoi(e) : 0u; }     // (make DIAG=1 builds only: never the shipped library)
#endif
    const uint32_t wdbg = (a.dbg_flags >> 8) & 0xFFu;      // diagnostic override (d2d_params.debug_flags bits 8..15)
    m.nwaves = wdbg ? wdbg : 12u;
    if (m.nwaves < 1 || m.nwaves > 12) m.nwaves = 12;
    // largest block that fits the CU's LDS, keeping the waves evenly spread over the four SIMDs
    while (m.nwaves > 1 && (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds > 160 * 1024)
        m.nwaves = m.nwaves > 8 ? 8 : m.nwaves > 4 ? 4 : m.nwaves >> 1;
    smem = (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds;
}

size_t mfma2_smem_bytes(int M, int N, uint32_t channels, uint32_t sample_bytes, uint32_t* waves_per_block) {
    FirArgs a{};
    a.epi.channels = channels; a.epi.sample_bytes = sample_bytes; a.epi.bits = 24;
    Mfma2Args m{}; size_t smem = 0;
    mfma2_geometry(a, M / 8, mfma2_pairs(M, N), m, smem);
    if (waves_per_block) *waves_per_block = m.nwaves;
    return smem;
}

template <int MB, int NPG, int CH, int EPI>
static hipError_t launch_mfma2_t(Mfma2Args& m, size_t smem, uint32_t nwt_max, uint32_t nrows, hipStream_t s) {
    static KernelPrep prep;
    int dev = 0;
    const void* fn = reinterpret_cast<const void*>(&d2d_fir_mfma2_kernel<MB, NPG, CH, EPI>);
    hipError_t e = prep.max_dynamic_lds(fn, 160 * 1024, &dev);
    if (e != hipSuccess) return e;
    int blocks_per_cu, ncu;
    {
        std::lock_guard<std::mutex> g(prep.mu);
        // the cache is keyed on what the caller ASKED for (LDS bytes, waves); what the register file then admitted is kept next to it
        if (prep.blocks_per_cu[dev] == 0 || smem != prep.smem_seen[dev] || m.nwaves != prep.nwaves_seen[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            const size_t smem_req = smem; const uint32_t nwaves_req = m.nwaves;
            // the register file may admit fewer waves than LDS does: shrink the block until one fits
            int nb = 0;
            for (;;) {
                e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, d2d_fir_mfma2_kernel<MB, NPG, CH, EPI>, (int)(64 * m.nwaves), smem);
                if (e != hipSuccess) return e;
                if (nb >= 1 || m.nwaves <= 4) break;
                m.nwaves -= 4;
                smem = (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds;
            }
            prep.ncu[dev] = prop.multiProcessorCount;
            prep.blocks_per_cu[dev] = nb < 1 ? 1 : nb;
            prep.smem_seen[dev] = smem_req; prep.nwaves_seen[dev] = nwaves_req;
            prep.smem_used[dev] = smem; prep.nwaves_used[dev] = m.nwaves;
        }
        blocks_per_cu = prep.blocks_per_cu[dev]; ncu = prep.ncu[dev];
        m.nwaves = prep.nwaves_used[dev]; smem = prep.smem_used[dev];
    }
    // every wave loops over its share of the wave-tiles: launch what is resident at once
    uint32_t gx = (uint32_t)(ncu * blocks_per_cu) / nrows;
    if (gx < 1) gx = 1;
    const uint32_t need = (nwt_max + m.nwaves - 1) / m.nwaves;
    if (gx > need) gx = need;
    hipLaunchKernelGGL((d2d_fir_mfma2_kernel<MB, NPG, CH, EPI>), dim3(gx, nrows), dim3(64 * m.nwaves), smem, s, m);
    d2d_last_launched_kernel = launched_name<MB, NPG, CH, EPI>("d2d_fir_mfma2_kernel");
    return hipGetLastError();
}

hipError_t launch_fir_mfma2(const FirArgs& a, int M, int N, uint32_t max_nout, uint32_t nstreams, hipStream_t s) {
    if (nstreams == 0 || max_nout == 0) return hipSuccess;
    const uint32_t C = a.epi.channels;
    const int MB = M / 8, NPG = mfma2_pairs(M, N);
    Mfma2Args m{};
    size_t smem = 0;
    mfma2_geometry(a, MB, NPG, m, smem);
    const uint32_t nrows = (nstreams / C) * m.ngroups;       // grid rows: one per (file, channel group)
    if (a.pipelined == 5) {                                  // the fp6 kernel has its own LDS layout (and M = 128 no two-group one at all)
        m.npairs = mx_pairs(a, MB, N);
        if (m.npairs > 1u) return launch_fir_mx(m, MB, N, max_nout, nstreams / C, s);     // one block row per file
        if (MB == 16) m.gainq = (!a.to_scratch && (a.epi.gain != 1.0 || m.qsh != 0 || (a.epi.bits == 32 && a.epi.dither == 'F'))) ? 1u : 0u;
        return launch_fir_mx(m, MB, N, max_nout, nrows, s);
    }
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    const uint32_t nwt = (max_nout + (M2_TILE - 1)) / M2_TILE;
    // the epilogue flavour: the integers for the stage-A scratch, stereo 24-bit packed in registers, or anything via LDS
    const int epi = mfma2_epilogue(a, m);
    // stereo 24-bit at 0 dB: the software-pipelined kernel (d2d_kernels_mfma3.hip), same results; the engine chose it (and its
    // table variant) when it was created
    if (a.pipelined == 5) return launch_fir_mx(m, MB, N, max_nout, nrows, s);
    if (a.pipelined) return launch_fir_mfma3(m, (int)a.pipelined, MB, NPG, N, nwt, nrows, s);
#define X(mb, npg)                                                                                  \
    if (MB == mb && NPG == npg) {                                                                   \
        if (C == 1) return epi == 2 ? launch_mfma2_t<mb, npg, 1, 2>(m, smem, nwt, nrows, s) : launch_mfma2_t<mb, npg, 1, 0>(m, smem, nwt, nrows, s); \
        return epi == 2 ? launch_mfma2_t<mb, npg, 2, 2>(m, smem, nwt, nrows, s)                      \
             : epi == 1 ? launch_mfma2_t<mb, npg, 2, 1>(m, smem, nwt, nrows, s) : launch_mfma2_t<mb, npg, 2, 0>(m, smem, nwt, nrows, s); \
    }
    D2D_M2_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

#if D2D_DIAG
void mfma2_debug_stamps(unsigned long long out[8]) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(d2d_m2_stamps), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d2d_m2_stamps), z, sizeof(z));
}
#else
void mfma2_debug_stamps(unsigned long long out[8]) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif

}  // namespace d2d
