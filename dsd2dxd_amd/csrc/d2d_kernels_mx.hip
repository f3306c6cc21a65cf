// d2d_kernels_mx.hip -- the FIR decimator on the fp6 x fp4 matrix-core instruction (gfx950), exact, software-pipelined.
//
// Same arithmetic contract, staging, pipelining and epilogue as d2d_kernels_mfma3.hip (stereo at 0 dB: 24-bit, 16-bit, float frames,
// and the exact integers for the stage-A / noise-shaper scratch); what changes is the instruction that carries the dot product and,
// with it, the geometry.  v_mfma_i32_32x32x32_i8 spends 32 cycles on 32 stream bits per column; v_mfma_scale_f32_32x32x64_f8f6f4
// with an fp6 (e2m3) A operand and an fp4 (e2m1) B operand spends the same 32 cycles on 64 (tools/ubench/mfma_shapes.hip), and one
// v_and turns a stream dword into EIGHT operand slots instead of four:
//
//   * B operand = the bit stream.  A nibble that holds one stream bit is an e2m1 number: 0b0001 = 0.5, 0b0010 = 1.0.  A stream dword W
//     becomes the lane's four operand registers W & 0x11111111, W & 0x22222222, (W >> 2) & 0x11111111, (W >> 2) & 0x22222222 (five
//     vector instructions for 32 bits; the int8 form needs eight): K slot 8p + n of the lane is bit 4n + p of its dword.
//   * A operand = the taps.  2q (q the 24-bit tap) is written in five balanced base-32 digits d in [-16, 15]; the slot that meets a
//     0.5-valued bit holds d/4, the slot that meets a 1.0-valued bit d/8 -- both exact in e2m3 (multiples of 1/8 up to 2, of 1/4 up to
//     4) -- and the B scale of the instruction is 2^3, so every product is the integer d * bit and the f32 accumulators hold the exact
//     digit sums (|sum| <= 16 * 752 << 2^24).  Matrix row = (phase, digit): 6 phases x 5 digits = 30 of the 32 rows; a lane half owns
//     the three phases 3h .. 3h+2 of every group with all five digits of a sample in its own registers.
//   * v = sum q s = S0 + 32 S1 + 2^10 S2 + 2^15 (S3 + 32 S4), accumulators started from -2^S in the digit-4 rows: two f32 fma
//     (|.| < 2^24: exact), one more for the high part, two conversions, one shift-add.
//   * One matrix column serves 6 G consecutive outputs (G groups of six phases); group g reads the tap fragments of group 0
//     6 M / 64 steps later, a step being 64 stream bits (lane half h takes dword 2u + h).  M = 32, E filter: 12 fragments, 15 steps and
//     24 MFMAs per 384 outputs = 32 per 512 (the int8 form: 52), 100 operand-expansion instructions per 512 outputs (136).
//
// Frames leave through a per-wave LDS slice (a lane owns runs of three samples, not of four): samples in as dwords, out as groups of
// four frames, packed and stored as in the int8 kernel.
//
// Replaces: the per-block translate loop inside Rdsd2Pcm::do_conversion
// (/root/reference/src/main.rs:345,429); the crate that holds it is absent from the reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "d2d_mfma2_dev.h"
#include "d2d_mx.h"

namespace d2d {

#ifndef D2D_MX_THREADS
#define D2D_MX_THREADS 512     // waves per block x 64: 512 = two waves per SIMD (256 registers each); 768 = three (168), an A/B build
#endif
#ifndef D2D_MX_ABL
#define D2D_MX_ABL 0
#endif

#ifndef D2D_MX_STAMPS
#define D2D_MX_STAMPS 0
#endif
#if D2D_MX_STAMPS
// per-wave s_memtime ticks (-DD2D_MX_STAMPS=1, tools/ab_mx.sh): [0] min, [1] max, [2] sum, [3] count of the waves' lifetimes; sums over all waves of
// [4] staging (LDS writes, next prefetch, stores), [5] the two regions (chain + epilogue), [6] what follows a region; [7] sum of s_memrealtime
__device__ unsigned long long d2d_mx_stamps[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
#endif

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int32_t i32x3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ int32_t mx_lshl_add(int32_t x, uint32_t sh, int32_t y) {
    int32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(sh), "v"(y));
    return d;
}
__device__ __forceinline__ uint32_t mx_min3_u16(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t d;
    asm("v_min3_u16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}
__device__ __forceinline__ int32_t mx_min3(int32_t x, int32_t y, int32_t z) {
    int32_t d;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}
__device__ __forceinline__ int32_t mx_max3(int32_t x, int32_t y, int32_t z) {
    int32_t d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
    return d;
}

// v_perm_b32 selector of the dword whose first byte is byte o of sample s0 (SBY bytes per sample, packed): the rest of s0 (second source), then s0 + 1 (first source)
__host__ __device__ constexpr uint32_t mx_pack_sel(int SBY, int o) {
    uint32_t sel = 0;
    for (int j = 0; j < 4; ++j) { const int t = o + j; sel |= (uint32_t)(t < SBY ? t : 4 + (t - SBY)) << (8 * j); }
    return sel;
}

// KIND: 0 no dither, 1 triangular, 2 rectangular (unit gain, all-integer requantiser); 4, 5, 6: the same dithers at any level in dB (the
// f64 requantiser of the definition inside the pipelined epilogue, no careful path).  Stereo; SBY = bytes per sample: 3 (24-bit packed frames), 2 (16-bit),
// 4 (32-bit float, KIND 0 only) or 0 (the exact integers y * 2^S to the scratch lines of a channel pair).
// NPR > 1 (planar multichannel frames, unit gain): a wave converts ALL the NPR channel pairs of a tile, one after the other through the same
// two stream buffers and accumulator sets -- the pipelined loop's trips are (tile, pair), the pair unrolled -- into a slice
// [2 NPR channels][TILE], and the tile's WHOLE frames leave together (a pair storing its own 6 bytes of every 18-byte frame left each line
// to three partial writes: 14.0 ms against the 4.7 ms of the same samples as stereo, profiles/r04_experiments.txt item 10).
// ND = 7 (tap_bits = 32 in ONE pass, round 4): the 32-bit taps in seven base-32 digits, four phases per group (28 of the 32 matrix rows; a lane half owns two
// phases), v = sum q32 s as a 64-bit integer from three f32 parts, requantised by the f64 flavour's epilogue (KIND 4-7 only).
template <int MB, int NT, int G, int KIND, int SBY, int NPR = 1, int ND = 5>
__global__ __launch_bounds__(D2D_MX_THREADS) void d2d_fir_mx_kernel(Mfma2Args m) {
    constexpr bool WIDE = ND == 7;
    static_assert(ND == 5 || ND == 7, "five digits (24-bit taps) or seven (32-bit taps)");
    constexpr int PH = WIDE ? 4 : 6, PHH = PH / 2;                  // phases (outputs) per group; per lane half
    using vint = std::conditional_t<WIDE, int64_t, int32_t>;        // v = sum q s
    constexpr int CS = mx_cs(MB, G, PH), DLY = mx_dly(MB, PH), NF = mx_nf(MB, NT, PH), TP = mx_nstep(MB, NT, G, PH);
    constexpr int OC = PH * G, TILE = 32 * OC, NS = PHH * G;        // outputs per column / per tile; samples per lane and channel
    constexpr int NCHK = mx_chunks(MB, NT, G, PH), PF = mx_pf(MB, NT, G, PH);
    constexpr uint32_t SB = (uint32_t)mx_stream_bytes(MB, NT, G, PH);
    // FLAT: the column stride is 2 mod 4 dwords, so the 32 lanes of a half already read 16 different banks from an unpadded image: the
    // chunks go to LDS as they come, one 16-byte write each, at their own 16-byte slots
    constexpr bool FLAT = mx_flat(MB, G, PH);
    constexpr int NCH = 2 * NPR;                                    // channels a wave converts
    constexpr uint32_t FB = (uint32_t)NCH * (SBY ? SBY : 1);        // bytes per frame
    static_assert(NPR == 1 || (SBY != 0 && KIND < 4), "several pairs per wave: frames at unit gain");
    constexpr uint32_t TBL16 = (uint32_t)NF * (MX_FRAG_BYTES / 16); // 16-byte units of one table variant
    constexpr bool SCR = SBY == 0;
    constexpr int DK = KIND & 3;                                    // the dither kind
    constexpr bool GN = KIND >= 4;                                  // any level: x = fl(v * (scale * 2^-S)), q = x + d, round half away, clip -- in f64
    static_assert(!GN || !SCR, "the scratch holds integers");
    static_assert(!WIDE || (GN && NPR == 1 && MB < 16), "32-bit taps: the f64 requantiser, stereo");
    constexpr uint32_t dbg = D2D_MX_ABL;                  // compile-time ablation mask: 1 no chain, 2 no epilogue, 4 no staging, 8 never slow, 64 no stores
    const FirArgs& a = m.f;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t Ct = a.in_channels;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // COOP (scratch flavour, byte-interleaved 4- or 8-channel input, a.coop): a block = one tile of ALL the file's channel pairs, wave p
    // converts pair p, and the waves de-interleave the tile's bytes together (below); otherwise a block row = a file or one of its pairs
    const bool coop = SCR && a.coop;
    const uint32_t fidx = coop ? blockIdx.y : (SCR ? blockIdx.y / m.ngroups : blockIdx.y);
    const uint32_t cbase = coop ? 2u * wave : (SCR ? (blockIdx.y - fidx * m.ngroups) * 2u : 0u);
    uint8_t* wbase = smem + m.off_waves + wave * m.wave_lds;       // [channel 0 stream buffer | channel 1 stream buffer | output slice]
    const StreamJob* jobs = a.jobs + (size_t)fidx * (SCR ? a.epi.channels : (uint32_t)NCH) + cbase;
    const StreamJob j0 = jobs[0];          // in, L, e0, n0, nout are common to a file's channels

    const int64_t first0 = j0.e0 - (int64_t)a.Wb;          // first byte of output 0's window
    const uint32_t sh = (uint32_t)(first0 & 3);            // its misalignment inside the staged dword
    {   // tap fragments: L2 -> LDS once per block; the variant for this byte misalignment
        const uint4* s = reinterpret_cast<const uint4*>(a.tables) + (size_t)sh * TBL16;
        uint4* dl = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < TBL16; i += blockDim.x) dl[i] = s[i];
    }
    __syncthreads();

    const uint32_t nwt = (j0.nout + (TILE - 1)) / TILE;            // wave-tiles in this file
    const uint32_t wstride = coop ? gridDim.x : gridDim.x * m.nwaves;
    const uint32_t r = lane & 31, h = lane >> 5;

    // ---- staging geometry: window dword L of a tile sits at LDS dword L + L / CS (one pad dword per column stride: CS is even, so the
    // 32 lanes of a half read distinct banks); a chunk's four dwords each carry their own address (CS need not be a multiple of 4).
    // FLAT: chunk q at byte 16 q, window dword L at LDS dword L + X0 ----
    const uint32_t X0 = (uint32_t)(first0 >> 2) & 3u;
    constexpr uint32_t DUMMY = SB - 16u;
    // IL (a.il2: byte-interleaved stereo -- DFF files, the CLI's default -f I -- both channels converted): the tile's
    // frames come as they lie in memory, 2 NCHK pieces of 16 bytes = eight frames each, in two halves of PF pieces per lane; one
    // v_perm_b32 per channel and dword pair pulls a channel's bytes (run_loop below).  Piece g holds a channel's bytes 8 g .. 8 g + 7.
    const bool il = NPR == 1 && a.il2 != 0 && !coop;          // (the scratch flavour too: there it is what the fixed-order loop is used for)
    auto pad_addr = [&](int32_t L, uint32_t k) -> uint32_t { return L < 0 ? DUMMY + 4u * k : 4u * ((uint32_t)L + (uint32_t)L / (uint32_t)CS); };
    uint32_t wad[FLAT ? 1 : PF][4];
    if constexpr (!FLAT) {
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // IL: entry 2 H + kk = dword kk of the piece the lane holds in slot i of half H
                const int32_t L = il ? (int32_t)(2u * (lane + 64u * ((uint32_t)PF * (k >> 1) + i))) - (int32_t)X0 + (k & 1)
                                     : (int32_t)(4u * (lane + 64u * i)) - (int32_t)X0 + k;
                wad[i][k] = pad_addr(L, (uint32_t)k);
            }
    }
    // MONO2 (a.mono2): a MONO stream served as a planar pair -- "channel" 0 = the first half of the call's bytes, "channel" 1 = the second half
    // (its history: the end of the first half), each with its own frames: the two halves of the call are converted side by side by the
    // stereo machinery and leave as two mono streams.  Its one "block" is the half call: as a power of two past every offset (2^31) the
    // block arithmetic below degenerates to base + offset.
    const bool mono2 = a.mono2 != 0;
    const uint32_t Bsz = mono2 ? 0x80000000u : a.B, Lcall = (uint32_t)j0.L;
    const bool pow2B = Bsz >= 16 && (Bsz & (Bsz - 1)) == 0;
    const uint32_t bshift = pow2B ? 31 - __builtin_clz(Bsz) : 0;
    const uint32_t full_bytes = il || mono2 ? Lcall : pow2B ? (Lcall >> bshift) << bshift : 0;
    const uint32_t jump = (Ct - 1u) * Bsz;
    const bool fast_layout = mono2 || (pow2B && (uint64_t)full_bytes * Ct < (1ull << 32) && jump < (1u << 24));
    auto tile_ab16 = [&](uint32_t w) -> int32_t { return (int32_t)((first0 + (int64_t)w * (TILE * MB)) & ~(int64_t)15); };

    uint32_t lofs[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) { const uint32_t q = lane + 64u * i; lofs[i] = 16u * (q < (uint32_t)NCHK ? q : (uint32_t)NCHK - 1u); }
    uint64_t chan_off[NCH];                             // where a channel's bytes start inside a block group (MONO2: inside the call)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t chf = (uint32_t)__builtin_amdgcn_readfirstlane((int)jobs[c].ch);
        chan_off[c] = mono2 ? (c ? (uint64_t)Lcall : 0ull) : (uint64_t)chf << bshift;
    }
    // one prefetch register set: a channel's bytes are requested one chain ahead (about two microseconds)
    u32x4 pf[PF];
    auto issue_loads = [&](uint32_t w, auto cc, auto af) {     // cc: the CHANNEL (of the wave's 2 NPR) whose bytes are requested
        constexpr int c = decltype(cc)::value;
        constexpr bool AF = decltype(af)::value;
        const int32_t ab = tile_ab16(w);
        if (AF || (fast_layout && ab >= 0 && (uint32_t)ab + 16u * NCHK <= full_bytes)) {
            const uint32_t blk0 = (uint32_t)ab >> bshift, r0 = (uint32_t)ab & (Bsz - 1);
            const uint8_t* base = j0.in + ((uint64_t)(blk0 * Ct) << bshift) + chan_off[c];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const uint32_t off = r0 + lofs[i];
                const uint32_t o = __umul24(off >> bshift, jump) + off;
                pf[i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(base) + o);
            }
        } else {
            if constexpr (!AF) {
#pragma unroll
                for (int i = 0; i < PF; ++i) pf[i] = gather_chunk(jobs + c, Ct, a.B, a.keep, ab + (int32_t)lofs[i]);
            }
        }
    };
    auto write_lds_t = [&](auto cc, auto calc) {      // calc: the addresses are worked out here (IL: wad holds the other set)
        constexpr int c = decltype(cc)::value;
        uint8_t* buf = wbase + c * SB;
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (lane + 64u * i < (uint32_t)NCHK) {
                if constexpr (FLAT) *reinterpret_cast<u32x4*>(buf + 16u * lane + 1024u * i) = pf[i];
                else {
                    const uint32_t v[4] = {pf[i].x, pf[i].y, pf[i].z, pf[i].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (decltype(calc)::value) *reinterpret_cast<uint32_t*>(buf + pad_addr((int32_t)(4u * (lane + 64u * i)) - (int32_t)X0 + k, (uint32_t)k)) = v[k];
                        else *reinterpret_cast<uint32_t*>(buf + wad[i][k]) = v[k];
                    }
                }
            }
    };
    auto write_lds = [&](auto cc) { write_lds_t(cc, std::false_type{}); };
    // IL staging: half H of the joint tile = pieces lane + 64 (PF H + i)
    auto il_issue = [&](uint32_t w, auto hc) {
        constexpr uint32_t H = decltype(hc)::value;
        const uint8_t* src = j0.in + 2u * (size_t)(uint32_t)tile_ab16(w);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            uint32_t g = lane + 64u * ((uint32_t)PF * H + (uint32_t)i);
            g = g < 2u * (uint32_t)NCHK ? g : 2u * (uint32_t)NCHK - 1u;           // (slots past the last piece re-read it; their writes are masked)
            pf[i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(src) + 16u * g);
        }
    };
    auto il_put = [&](uint32_t c, auto hc, int i, uint32_t x, uint32_t y) {       // eight bytes of channel c from slot i of half H
        constexpr uint32_t H = decltype(hc)::value;
        const uint32_t g = lane + 64u * ((uint32_t)PF * H + (uint32_t)i);
        uint8_t* buf = wbase + c * SB;
        if (g < 2u * (uint32_t)NCHK) {
            if constexpr (FLAT) { u32x2 v; v.x = x; v.y = y; *reinterpret_cast<u32x2*>(buf + 8u * g) = v; }
            else { *reinterpret_cast<uint32_t*>(buf + wad[i][2u * H]) = x; *reinterpret_cast<uint32_t*>(buf + wad[i][2u * H + 1u]) = y; }
        }
    };

    // ---- COOP: the tile's bytes of all channels come as they lie in memory (frame after frame), each wave fetching a share of the 16-byte
    // pieces; a piece holds 16 / Ct frames, from which one v_perm_b32 per channel pair pulls the pair's bytes -- two 2-byte (Ct = 8) or
    // 4-byte (Ct = 4) runs that go straight into the pair's two stream buffers, whichever wave owns them.  Two block barriers per tile.
    constexpr int PFI = SCR ? (NCHK + 31) / 32 : 1;           // pieces per lane: NCHK * Ct pieces over Ct / 2 waves of 64 lanes
    u32x4 pfi[PFI];
    uint32_t cad[PFI];                                         // where a piece's frames start inside a stream buffer
    const uint32_t nw = m.nwaves;
    if constexpr (SCR) {
        if (coop) {
            const uint32_t ts_sh = Ct == 8 ? 1u : 2u;          // log2(frames per piece)
#pragma unroll
            for (int i = 0; i < PFI; ++i) {
                const uint32_t g = lane + 64u * (wave + nw * (uint32_t)i);
                const uint32_t tau = g << ts_sh;                // first frame of the piece = byte of a channel's stream, relative to the tile's first chunk
                const int32_t Lw = (int32_t)(tau >> 2) - (int32_t)X0;
                if constexpr (FLAT) cad[i] = tau;                 // the unpadded image: a channel's bytes as they come
                else cad[i] = Lw < 0 ? DUMMY + (tau & 3u) : 4u * ((uint32_t)Lw + (uint32_t)Lw / (uint32_t)CS) + (tau & 3u);
            }
        }
    }
    auto coop_fast = [&](uint32_t w) -> bool {                 // the tile's bytes all lie inside this call's data (block-uniform)
        const int32_t ab = tile_ab16(w);
        return ab >= 0 && (uint32_t)ab + 16u * NCHK <= Lcall;
    };
    auto coop_issue = [&](uint32_t w) {
        const uint8_t* src = j0.in + (size_t)(uint32_t)tile_ab16(w) * Ct;
#pragma unroll
        for (int i = 0; i < PFI; ++i) {
            uint32_t g = lane + 64u * (wave + nw * (uint32_t)i);
            g = g < (uint32_t)NCHK * Ct ? g : (uint32_t)NCHK * Ct - 1u;      // (lanes past the last piece re-read it; their writes are masked)
            pfi[i] = *reinterpret_cast<D2D_GLOBAL const u32x4*>(as_global(src) + 16u * g);
        }
    };
    auto coop_write = [&]() {
        uint8_t* b0 = smem + m.off_waves;
#pragma unroll
        for (int i = 0; i < PFI; ++i) {
            const uint32_t g = lane + 64u * (wave + nw * (uint32_t)i);
            if (g < (uint32_t)NCHK * Ct) {
                const uint32_t d[4] = {pfi[i].x, pfi[i].y, pfi[i].z, pfi[i].w};
                if (Ct == 8) {
                    // dwords 0, 1 = frame 0 (channels 0-3, 4-7), dwords 2, 3 = frame 1: pair p = channels 2p, 2p+1
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const uint32_t v = __builtin_amdgcn_perm(d[2 + (p >> 1)], d[p >> 1], (p & 1) ? 0x07030602u : 0x05010400u);   // [c.t0 c.t1 c'.t0 c'.t1]
                        uint8_t* pb = b0 + (uint32_t)p * m.wave_lds + cad[i];
                        *reinterpret_cast<uint16_t*>(pb) = (uint16_t)v;
                        *reinterpret_cast<uint16_t*>(pb + SB) = (uint16_t)(v >> 16);
                    }
                } else {
                    // Ct = 4: dword k = frame k (channels 0-3)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const uint32_t sel = p ? 0x07030602u : 0x05010400u;
                        const uint32_t x = __builtin_amdgcn_perm(d[1], d[0], sel), y = __builtin_amdgcn_perm(d[3], d[2], sel);   // [c.t0 c.t1 c'.t0 c'.t1], [c.t2 c.t3 c'.t2 c'.t3]
                        uint8_t* pb = b0 + (uint32_t)p * m.wave_lds + cad[i];
                        *reinterpret_cast<uint32_t*>(pb) = __builtin_amdgcn_perm(y, x, 0x05040100u);
                        *reinterpret_cast<uint32_t*>(pb + SB) = __builtin_amdgcn_perm(y, x, 0x07060302u);
                    }
                }
            }
        }
    };

    // tap fragment f: 16 bytes per lane at f * 1536 + 16 lane, 8 more at f * 1536 + 1024 + 8 lane
    const uint8_t* tp16 = smem + 16u * lane;
    const uint8_t* tp8 = smem + 1024u + 8u * lane;
    uint32_t kmA = 0x11111111u, kmB = 0x22222222u;
    asm volatile("" : "+v"(kmA), "+v"(kmB));
    int scA = 0x7f7f7f7f, scB = (int)0x82828282u;          // e8m0 scales: A x 1, B x 8 (every product becomes an integer)
    asm volatile("" : "+v"(scA), "+v"(scB));
    // accumulators start from -2^S: the digit-4 rows (weight 2^20) of every sample
    // (EB, the dithered integer depths: the accumulators start from zero instead -- sixteen registers less -- and the -2^S rides in the
    // three-operand add that applies the dither; the extremes are then kept on v + 2^S)
    // (M = 128: S = 30 and sum |q| > 2^30, v + 2^S does not fit an int32: the accumulators start from -2^S there)
    constexpr bool EB = MB < 16 && (((KIND == 1 || KIND == 2) && (SBY == 2 || SBY == 3)) || GN);
    v16f cinit;
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = (!EB && i < 15 && (i % 5) == 4) ? -(float)(1 << (a.scale_bits - 20)) : 0.0f;
    if constexpr (!EB) asm volatile("" : "+v"(cinit));
    const vint kBias = EB ? ((vint)1 << a.scale_bits) : (vint)0;      // (WIDE: scale_bits = S + 8, a 64-bit bias)

    // One chain: TP steps of 64 stream bits; group g runs its NF MFMAs from step DLY g on, with the fragments group 0 read
    // DLY g steps earlier; LDS reads are issued AHEAD steps before their use; `hook(k)` is whatever else the wave does behind its k-th MFMA.
    auto chain = [&](uint32_t c, v16f (&acc)[G], auto&& hook) {
        const uint8_t* rbc = wbase + c * SB + (FLAT ? 4u * (CS * r + h + X0) : 4u * ((CS + 1) * r + h));
        uint32_t W[TP];
        v4i F4[NF]; u32x2 F2[NF];
        auto rdW = [&](auto uc) { constexpr int u = decltype(uc)::value; W[u] = *reinterpret_cast<const uint32_t*>(rbc + 4 * (2 * u + (FLAT ? 0 : (2 * u) / CS))); };
        auto rdF = [&](auto fc) {
            constexpr int f = decltype(fc)::value;
            F4[f] = *reinterpret_cast<const v4i*>(tp16 + MX_FRAG_BYTES * f);
            F2[f] = *reinterpret_cast<const u32x2*>(tp8 + MX_FRAG_BYTES * f);
        };
#ifndef D2D_MX_AHEAD
#define D2D_MX_AHEAD 2
#endif
        constexpr int AHEAD = D2D_MX_AHEAD;
        static_for<0, AHEAD>([&](auto uc) { rdW(uc); rdF(uc); });
        static_for<0, TP>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u + AHEAD < TP) rdW(std::integral_constant<int, u + AHEAD>{});
            if constexpr (u + AHEAD < NF) rdF(std::integral_constant<int, u + AHEAD>{});
            const uint32_t w = W[u], w2 = w >> 2;
            const v8i Bv = {(int)(w & kmA), (int)(w & kmB), (int)(w2 & kmA), (int)(w2 & kmB), 0, 0, 0, 0};
            static_for<0, G>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int f = u - DLY * g;
                if constexpr (f >= 0 && f < NF) {
                    const v8i Av = {F4[f].x, F4[f].y, F4[f].z, F4[f].w, (int)F2[f].x, (int)F2[f].y, 0, 0};
                    if constexpr (f == 0) acc[g] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Av, Bv, cinit, 2, 4, 0, scA, 0, scB);
                    else acc[g] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Av, Bv, acc[g], 2, 4, 0, scA, 0, scB);
                    // whatever else the wave does rides BEHIND an MFMA: an in-order wave that issues two MFMAs back to back sits out the
                    // first one's 32 cycles in the matrix pipe
                    hook(std::integral_constant<int, mx_slot(MB, NT, G, u, g, PH)>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    };
    auto no_hook = [](auto) {};
    auto pin = [&](v16f (&acc)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+v"(acc[g]));
    };
    // ... and every accumulator set stays live AS A WHOLE until its last sample has been taken from it.  Register 15 of a set belongs to no
    // output row, so it is dead the moment the chain ends -- and the allocator then hands it to the next temporary, which may be the result of one
    // of the inline-asm instructions of the epilogue (v_lshl_add_u32, v_min3 ...): the compiler's hazard recogniser does not look inside inline
    // asm, no wait states are inserted, and the chain's last MFMA, still in flight, lands its zero row on top of the value (round 4: sample 0 of
    // every lane of a mono pair's second half came out as -2^S -> the negative rail, after a change that moved the allocation).
    // (the in/out form of `pin`: with an input-only operand of sixteen registers the kernel silently fails to instantiate -- no diagnostic, no code object)
    auto hold_acc = [&](v16f (&acc)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+v"(acc[g]));
    };

    uint8_t* const mono_out[2] = {reinterpret_cast<uint8_t*>(jobs[0].out), reinterpret_cast<uint8_t*>(jobs[1].out)};      // (MONO2: each half's own frames)
    // dither keys of the two channels (uniform)
    uint32_t rkey[NCH], rstep[NCH], rlo0[NCH];
    vint vdev[NCH];                                      // running max |v| of a channel, fast path and careful path alike (one register per channel)
#pragma unroll
    for (int c = 0; c < NCH; ++c) { rkey[c] = jobs[c].rng_key; rstep[c] = jobs[c].rng_kstep; rlo0[c] = jobs[c].rng_lo0; vdev[c] = 0; }

    // constants of the fast epilogue, parked in VGPRs
    const int F_ = SBY == 4 ? 1 : m.fbits;                  // 0 < F <= 16 (integer depths)
    float kFs = ldexpf(1.0f, -a.scale_bits);                // float output: y = v * 2^-S
    asm volatile("" : "+v"(kFs));
    uint32_t kF = (uint32_t)F_, kSh = 16u - (uint32_t)F_, kShR = 32u - (uint32_t)F_;
    uint32_t kC1 = 0x7feb352dU, kC2 = 0x846ca68bU, kTm = (uint32_t)-32767;
    uint32_t k15 = MB == 16 ? 10u : 15u;
    float k32 = 32.0f, k1024 = 1024.0f;
    int32_t kHalf = 1 << (F_ - 1);
    asm volatile("" : "+v"(kF), "+v"(kSh), "+v"(kShR), "+v"(kC1), "+v"(kC2), "+v"(kTm), "+v"(k15), "+v"(k32), "+v"(k1024), "+v"(kHalf));
    const int32_t kSafe = (int32_t)(((uint32_t)m.qmax_i - 2u) << F_);
    vint kNegBias = -kBias;
    asm volatile("" : "+v"(kNegBias));
    const uint32_t lane_fr = (uint32_t)OC * r + (uint32_t)PHH * h;     // the lane's first sample inside a tile; sample i = PHH g + q sits at lane_fr + PH g + q

    // v = sum q s of sample q of a group's accumulators: digits S0..S4 = registers 5q .. 5q+4 (exact integers)
    auto recombine = [&](const v16f& A, int q) -> vint {
        if constexpr (WIDE) {
            // seven digits: three f32 parts (each below 2^24: mx_wide_exact), v = lo + 2^15 mid + 2^25 hi in 64 bits
            const float lo = __builtin_fmaf(A[7 * q + 2], k1024, __builtin_fmaf(A[7 * q + 1], k32, A[7 * q]));
            const float mid = __builtin_fmaf(A[7 * q + 4], k32, A[7 * q + 3]);
            const float hi = __builtin_fmaf(A[7 * q + 6], k32, A[7 * q + 5]);
            return (int64_t)(int32_t)lo + ((int64_t)(int32_t)mid << 15) + ((int64_t)(int32_t)hi << 25);
        } else if constexpr (MB == 16) {
            // 2192 taps: S0 + 32 S1 + 1024 S2 can pass 2^24; split after two digits instead (mx_exact checks this form for M = 128)
            const float lo = __builtin_fmaf(A[5 * q + 1], k32, A[5 * q]);
            const float hi = __builtin_fmaf(A[5 * q + 4], k1024, __builtin_fmaf(A[5 * q + 3], k32, A[5 * q + 2]));
            return mx_lshl_add((int32_t)hi, k15, (int32_t)lo);             // (k15 holds 10 here)
        } else {
            const float lo = __builtin_fmaf(A[5 * q + 2], k1024, __builtin_fmaf(A[5 * q + 1], k32, A[5 * q]));
            const float hi = __builtin_fmaf(A[5 * q + 4], k32, A[5 * q + 3]);
            return mx_lshl_add((int32_t)hi, k15, (int32_t)lo);
        }
    };
    // (the channel is a compile-time constant everywhere: a lambda left out of line would otherwise index the per-channel arrays at run time,
    // which sends them to scratch memory)
    auto noise = [&](auto cc, uint32_t nl) -> uint32_t {
        constexpr uint32_t c = decltype(cc)::value;
        const uint32_t nlo = (uint32_t)j0.n0 + nl;
        uint32_t z = nlo + rkey[c] + (nlo < rlo0[c] ? rstep[c] : 0u);
        z ^= z >> 16; z *= 0x7feb352dU;
        z ^= z >> 15; z *= 0x846ca68bU;
        z ^= z >> 16;
        return z;
    };
    // GN: x = fl(v * kCg) is the oracle's y * scale (y = v * 2^-S exactly; the float flavour: y * gain); then d2d_device.h: finish_int
    // with the hash word's dither term t (triangular: lo16 + hi16 + 1, rectangular: 2 hi16 + 1)
    double kCg = ldexp(a.epi.bits == 32 ? a.epi.gain : a.epi.scale, -a.scale_bits);
    double kLim = a.epi.bits == 32 ? 1.0 : (double)(1u << (a.epi.bits - 1));
    if constexpr (GN) asm volatile("" : "+v"(kCg), "+v"(kLim));
    auto quant_gain = [&](vint v, uint32_t t) -> int32_t {
        const double x = (double)v * kCg;
        if constexpr (SBY == 4) {
            if constexpr (DK == 3) {
                // Airwindows "Dither Float" as d2d_device.h: quantise_f32 states it (t = the raw hash word)
                const uint32_t fb = __float_as_uint((float)x);
                const int e = (int)((fb >> 23) & 0xFFu);
                const int expon = e ? e - 126 : 0;
                const double tt = ((double)t - 2147483647.0) * 5.5e-36;
                return __float_as_int((float)(x + ldexp(tt, expon + 62)));
            }
            return __float_as_int((float)x);
        }
        double q = x;
        if constexpr (DK == 1) q = x + fma((double)t, 0x1p-16, -1.0);
        else if constexpr (DK == 2) q = x + fma((double)t, 0x1p-17, -0.5);
        const double rq = fmax(fmin(trunc(q + copysign(0.5, q)), kLim - 1.0), -kLim);
        return (int32_t)rq << m.qsh;                                   // (20-bit samples ride in 24 bits as r << 4)
    };
    // the general per-sample requantiser (any tile): x = v * 2^-F LSB, dither in 2^-16 (2^-17) LSB, round half away, clip
    auto quant_slow = [&](vint v, auto cc, uint32_t nl) -> int32_t {
        if constexpr (GN) {
            uint32_t t = 0;
            if constexpr (DK != 0) { const uint32_t z = noise(cc, nl); t = DK == 1 ? (z & 0xFFFFu) + (z >> 16) + 1u : DK == 2 ? 2u * (z >> 16) + 1u : z; }
            return quant_gain(v, t);
        }
        const int F = m.fbits;
        const int32_t vh = v >> F;
        const uint32_t vl = (uint32_t)v & ((1u << F) - 1u);
        int32_t rr;
        if constexpr (KIND == 2) {
            const uint32_t z = noise(cc, nl);
            const int32_t w = (int32_t)(vl << (17 - F)) + (int32_t)(2u * (z >> 16) + 1u) - 65536;
            const int32_t neg = (vh + (w >> 17)) >> 31;
            rr = vh + ((w + 65536 + neg) >> 17);
        } else {
            int32_t w = (int32_t)(vl << (16 - F));
            if constexpr (KIND == 1) {
                const uint32_t z = noise(cc, nl);
                w += (int32_t)((z & 0xFFFFu) + (z >> 16)) - 65535;
            }
            const int32_t neg = (vh + (w >> 16)) >> 31;
            rr = vh + ((w + 32768 + neg) >> 16);
        }
        return min(max(rr, m.qmin_i), m.qmax_i);
    };

    int32_t* ob = reinterpret_cast<int32_t*>(wbase + m.off_out);       // the wave's output slice [channel][TILE] (dwords)
    // ---- the fast epilogue of one (tile, channel), cut into jobs that ride on the steps of a chain ----
    struct Fast {
        uint32_t zb;            // hash input of the lane's first sample
        uint32_t T;             // the sample in work: its dither term ...
        vint v;                 // ... and its v = sum q s
        int32_t res[SCR ? NS : 1];
        int32_t* slot;          // the lane's first sample of this channel in the wave's output slice
        vint vprev; uint32_t wprev;
        vint tmn, tmx; uint32_t tie;
    };
    auto fast_begin = [&](Fast& f, uint32_t tile, auto cc) {
        constexpr uint32_t c = decltype(cc)::value;
        const uint32_t first = (uint32_t)j0.n0 + tile * (uint32_t)TILE;
        const uint32_t key_eff = rkey[c] + (first < rlo0[c] ? rstep[c] : 0u);
        f.zb = first + key_eff + lane_fr;
        f.tmn = kBias; f.tmx = kBias; f.tie = 0xFFFFu;
        f.slot = ob + c * TILE + lane_fr;
    };
    constexpr int JPS = DK == 0 ? 2 : 3;                    // jobs per sample: [hash,] recombine, finish
    constexpr int NJ = JPS * NS;                            // jobs per epilogue
    constexpr int NSLOT = NF * G;                           // MFMAs of a chain
    auto fast_job = [&](Fast& f, const v16f (&o)[G], auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int i = j / JPS;                          // sample 0..NS-1: group i / PHH, q = i % PHH
        constexpr int t = j % JPS + (DK == 0 ? 1 : 0);      // 0 hash, 1 recombine, 2 finish
        if constexpr (t == 0) {
            uint32_t z = f.zb + (uint32_t)(PH * (i / PHH) + (i % PHH));
            z ^= z >> 16; z *= kC1;
            z ^= z >> 15; z *= kC2;
            z ^= z >> 16;
            if constexpr (GN) f.T = DK == 1 ? __builtin_amdgcn_sad_u16(z, 0u, 1u) : DK == 2 ? ((z >> 15) | 1u) : z;      // lo16 + hi16 + 1; 2 hi16 + 1; the float dither's word
            else if constexpr (KIND == 1) f.T = __builtin_amdgcn_sad_u16(z, 0u, kTm);   // lo16 + hi16 - 32767, units of 2^-16 LSB
            else f.T = z >> kShR;                                                         // (2*hi16 + 1) >> (17 - F)
            asm volatile("" : "+v"(f.T));
        } else if constexpr (t == 1) {
            f.v = recombine(o[i / PHH], i % PHH);
            asm volatile("" : "+v"(f.v));
        } else {
            const vint v = f.v;
            int32_t s;
            if constexpr (GN) {
                s = 0;
            } else if constexpr (KIND == 1) {
                if constexpr (EB) s = v + ((int32_t)f.T >> kSh) + kNegBias; else s = v + ((int32_t)f.T >> kSh);
                const uint32_t w = (uint32_t)mx_lshl_add(v, kSh, (int32_t)f.T);            // low 16 bits zero: an exact tie
                if constexpr (i & 1) f.tie = mx_min3_u16(f.tie, f.wprev, w);
                else if constexpr (i == NS - 1) f.tie = mx_min3_u16(f.tie, w, w);
                else f.wprev = w;
            } else if constexpr (KIND == 2) {
                if constexpr (EB) s = v + (int32_t)f.T + kNegBias; else s = v + (int32_t)f.T;
            } else if constexpr (SBY == 4 || SCR) {
                s = 0;
            } else {
                s = v + kHalf + (v >> 31);                                                 // round half away from zero
            }
            int32_t rv;
            if constexpr (GN) rv = quant_gain(v + kNegBias, DK != 0 ? f.T : 0u);
            else if constexpr (SBY == 4) rv = __float_as_int((float)v * kFs);
            else if constexpr (SCR) rv = v;
            else rv = s >> kF;
            // the sample goes straight into the wave's output slice (the tile that sat there left before this region began); the scratch
            // flavour keeps it in a register for its store after the region
            if constexpr (SCR) { f.res[i] = rv; asm volatile("" : "+v"(f.res[i])); }
            else f.slot[PH * (i / PHH) + (i % PHH)] = rv;
            if constexpr (WIDE) { f.tmn = v < f.tmn ? v : f.tmn; f.tmx = v > f.tmx ? v : f.tmx; }
            else if constexpr (!SCR) {
                if constexpr (i & 1) { f.tmn = mx_min3(f.tmn, f.vprev, v); f.tmx = mx_max3(f.tmx, f.vprev, v); }
                else if constexpr (i == NS - 1) { f.tmn = min(f.tmn, v); f.tmx = max(f.tmx, v); }
                else f.vprev = v;
            }
        }
    };
    // job j rides behind MFMA (j * NSLOT) / NJ of the chain
    auto fast_hook = [&](Fast& f, const v16f (&o)[G], auto kc) {
        constexpr int k = decltype(kc)::value;
        static_for<0, NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr ((j * NSLOT) / NJ == k) fast_job(f, o, jc);
        });
    };
    auto fast_failed = [&](const Fast& f, uint32_t tile) -> bool {
        const uint32_t first = (uint32_t)j0.n0 + tile * (uint32_t)TILE;
        const bool full = tile * (uint32_t)TILE + (uint32_t)TILE <= j0.nout;
        if (SCR || (dbg & 8)) return false;
        if (!full || first > 0xFFFFFFFFu - (uint32_t)TILE) return true;
        if constexpr (SBY == 4 || GN) return false;          // float: nothing clips, nothing ties; any level: the f64 requantiser is the definition
        const bool bad = (KIND == 1 && (f.tie & 0xFFFFu) == 0) || f.tmx > kSafe + kBias || f.tmn < kBias - kSafe;
        return __builtin_amdgcn_ballot_w64(bad) != 0;
    };
    // the careful way, sample by sample, from a chain's accumulators
    auto redo_acc = [&](v16f (&t)[G], uint32_t tile, auto cc, int32_t (&out)[NS]) {
        constexpr uint32_t c = decltype(cc)::value;
        const bool full = tile * (uint32_t)TILE + (uint32_t)TILE <= j0.nout;
        const uint32_t nl_base = tile * (uint32_t)TILE + lane_fr;
        vint lo = kBias, hi = kBias;                         // (of the samples that exist, on v + kBias like the fast path's)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const uint32_t nl = nl_base + (uint32_t)PH * (i / PHH) + (i % PHH);
            const vint vb = recombine(t[i / PHH], i % PHH), v = vb - kBias;
            if constexpr (SBY == 4 && !GN) out[i] = __float_as_int((float)v * kFs); else out[i] = quant_slow(v, cc, nl);
            const bool live = full || nl < j0.nout;
            lo = live && vb < lo ? vb : lo;
            hi = live && vb > hi ? vb : hi;
        }
        if constexpr (WIDE) { const vint d = hi - kBias > kBias - lo ? hi - kBias : kBias - lo; vdev[c] = d > vdev[c] ? d : vdev[c]; }
        else vdev[c] = mx_max3(vdev[c], hi - kBias, kBias - lo);
        hold_acc(t);
    };
    // ... after the channel's chain run again (its stream bytes are still in that channel's buffer): the tiles at a call's edges
    auto redo = [&](uint32_t cbuf, uint32_t tile, auto cc, int32_t (&out)[NS]) {
        v16f t[G];
        chain(cbuf, t, no_hook);
        pin(t);                                     // the chain ends here, its results are complete before the first one is read
        __builtin_amdgcn_s_sleep(1);
        redo_acc(t, tile, cc, out);
    };
    auto tile_full = [&](uint32_t tile) -> bool { return tile * (uint32_t)TILE + (uint32_t)TILE <= j0.nout; };
    // a channel's samples of the tile in flight -> the wave's output slice [channel][TILE] (dwords): a lane owns runs of three
    auto put_samples = [&](uint32_t c, const int32_t (&v)[NS]) {
        int32_t* d = ob + c * TILE + lane_fr;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < PHH; ++k) d[PH * g + k] = v[PHH * g + k];
    };
    // the tile's frames out of the slice: a lane takes groups of four consecutive frames (24 / 16 / 32 contiguous bytes)
    constexpr int NQ = TILE / 4, QPASS = (NQ + 63) / 64;
    // (always inline: out of line, the closure's captures -- the job, the slice pointer, the lane -- would live in scratch memory for the whole kernel)
    auto store_tile = [&](uint32_t tile, bool known_full = false) __attribute__((always_inline)) {
        const bool full = known_full || tile_full(tile);
        uint8_t* gout = reinterpret_cast<uint8_t*>(j0.out) + (size_t)tile * (TILE * FB);
        const uint32_t nl0 = tile * (uint32_t)TILE;
        if constexpr (SBY != 0) {
            if (mono2) {
                // two mono streams: a lane takes four consecutive samples of a half (12 / 8 / 16 contiguous bytes; the second half's frames start at
                // any byte of the caller's buffer)
                typedef uint32_t u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
                typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
                typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    uint8_t* gc = mono_out[c] + (size_t)nl0 * SBY;
#pragma unroll
                    for (int p = 0; p < QPASS; ++p) {
                        const uint32_t Q = lane + 64u * p;
                        if ((NQ % 64) != 0 && p == QPASS - 1 && Q >= (uint32_t)NQ) continue;
                        const i32x4 v = *reinterpret_cast<const i32x4*>(ob + c * TILE + 4 * Q);
                        const uint32_t s0 = (uint32_t)v.x, s1 = (uint32_t)v.y, s2 = (uint32_t)v.z, s3 = (uint32_t)v.w;
                        uint8_t* gq = gc + 4u * SBY * Q;
                        if (full) {
                            if (dbg & 64) { asm volatile("" :: "v"(v)); continue; }
                            if constexpr (SBY == 3) *reinterpret_cast<D2D_GLOBAL u32x3_a1*>(as_global(gq)) = u32x3_a1{(s0 & 0x00FFFFFFu) | (s1 << 24), ((s1 >> 8) & 0xFFFFu) | (s2 << 16), ((s2 >> 16) & 0xFFu) | (s3 << 8)};
                            else if constexpr (SBY == 2) *reinterpret_cast<D2D_GLOBAL u32x2_a1*>(as_global(gq)) = u32x2_a1{(s0 & 0xFFFFu) | (s1 << 16), (s2 & 0xFFFFu) | (s3 << 16)};
                            else *reinterpret_cast<D2D_GLOBAL u32x4_a1*>(as_global(gq)) = u32x4_a1{s0, s1, s2, s3};
                        } else {
                            const uint32_t sv[4] = {s0, s1, s2, s3};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (nl0 + 4u * Q + k < j0.nout) {
                                    D2D_GLOBAL uint8_t* pb = as_global(gq + SBY * k);
#pragma unroll
                                    for (int b = 0; b < SBY; ++b) pb[b] = (uint8_t)(sv[k] >> (8 * b));
                                }
                        }
                    }
                }
                return;
            }
        }
        if constexpr (NPR > 1) {
            // frames of 2 NPR channels: a lane takes four consecutive frames = 4 FB contiguous bytes, two frames at a time (fewer live registers);
            // every dword of them is one v_perm_b32 of two neighbouring samples (sample NCH k + c = frame k, channel c)
            typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
            constexpr int NH = NPR * (int)SBY;                   // dwords of two frames
#pragma unroll
            for (int p = 0; p < QPASS; ++p) {
                const uint32_t Q = lane + 64u * p;
                if ((NQ % 64) != 0 && p == QPASS - 1 && Q >= (uint32_t)NQ) continue;
                static_for<0, 2>([&](auto hc) {
                    constexpr int h = decltype(hc)::value;
                    uint32_t S[2 * NCH];
                    static_for<0, NCH>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        const u32x2 v = *reinterpret_cast<const u32x2*>(ob + c * TILE + 4 * Q + 2 * h);
                        S[c] = v.x; S[NCH + c] = v.y;
                    });
                    uint8_t* gh = gout + 4u * FB * Q + 2u * FB * h;
                    if (full) {
                        if (dbg & 64) {
#pragma unroll
                            for (int i = 0; i < 2 * NCH; ++i) asm volatile("" :: "v"(S[i]));
                        } else {
                            uint32_t D[NH];
                            static_for<0, NH>([&](auto dc) {
                                constexpr int d = decltype(dc)::value;
                                if constexpr (SBY == 4) D[d] = S[d];
                                else {
                                    constexpr int b0 = 4 * d, s0 = b0 / (int)SBY, o = b0 % (int)SBY;
                                    constexpr int s1 = s0 + 1 < 2 * NCH ? s0 + 1 : s0;
                                    D[d] = __builtin_amdgcn_perm(S[s1], S[s0], mx_pack_sel((int)SBY, o));
                                }
                            });
                            static_for<0, NH / 4>([&](auto ic) {
                                constexpr int i = decltype(ic)::value;
                                *reinterpret_cast<D2D_GLOBAL u32x4_a4*>(as_global(gh + 16 * i)) = u32x4_a4{D[4 * i], D[4 * i + 1], D[4 * i + 2], D[4 * i + 3]};
                            });
                            constexpr int R0 = NH / 4 * 4;
                            if constexpr (NH - R0 >= 2) *reinterpret_cast<D2D_GLOBAL u32x2_a4*>(as_global(gh + 4 * R0)) = u32x2_a4{D[R0], D[R0 + 1]};
                            if constexpr ((NH - R0) % 2 == 1) *reinterpret_cast<D2D_GLOBAL uint32_t*>(as_global(gh + 4 * (NH - 1))) = D[NH - 1];
                        }
                    } else {
                        // the file's last, partial tile: byte by byte
                        static_for<0, 2 * NCH>([&](auto ic) {
                            constexpr int i = decltype(ic)::value, k = i / NCH, c = i % NCH;
                            if (nl0 + 4u * Q + 2u * h + k < j0.nout) {
                                D2D_GLOBAL uint8_t* pb = as_global(gh + FB * k + c * (int)SBY);
#pragma unroll
                                for (int b = 0; b < (int)SBY; ++b) pb[b] = (uint8_t)(S[i] >> (8 * b));
                            }
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < QPASS; ++p) {
            const uint32_t Q = lane + 64u * p;
            if ((NQ % 64) != 0 && p == QPASS - 1 && Q >= (uint32_t)NQ) continue;
            const i32x4 Lq = *reinterpret_cast<const i32x4*>(ob + 4 * Q), Rq = *reinterpret_cast<const i32x4*>(ob + TILE + 4 * Q);
            const uint32_t La = Lq.x, Lb = Lq.y, Lc = Lq.z, Ld = Lq.w, Ra = Rq.x, Rb = Rq.y, Rc = Rq.z, Rd = Rq.w;
            uint8_t* gq = gout + 4u * FB * Q;
            if (full) {
                if (dbg & 64) { asm volatile("" :: "v"(Lq), "v"(Rq)); continue; }
                if constexpr (SBY == 3) {
                    // frames k, k+1 -> 12 bytes: [L0 L1 L2 R0 | R1 R2 L0' L1' | L2' R0' R1' R2']
                    const u32x4 p4 = {__builtin_amdgcn_perm(Ra, La, 0x04020100u), __builtin_amdgcn_perm(Lb, Ra, 0x05040201u),
                                      __builtin_amdgcn_perm(Rb, Lb, 0x06050402u), __builtin_amdgcn_perm(Rc, Lc, 0x04020100u)};
                    const u32x2 p2 = {__builtin_amdgcn_perm(Ld, Rc, 0x05040201u), __builtin_amdgcn_perm(Rd, Ld, 0x06050402u)};
                    *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(gq)) = p4;
                    *reinterpret_cast<D2D_GLOBAL u32x2*>(as_global(gq + 16)) = p2;
                } else if constexpr (SBY == 4) {
                    *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(gq)) = u32x4{La, Ra, Lb, Rb};
                    *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(gq + 16)) = u32x4{Lc, Rc, Ld, Rd};
                } else {
                    *reinterpret_cast<D2D_GLOBAL u32x4*>(as_global(gq)) =
                        u32x4{__builtin_amdgcn_perm(Ra, La, 0x05040100u), __builtin_amdgcn_perm(Rb, Lb, 0x05040100u),
                              __builtin_amdgcn_perm(Rc, Lc, 0x05040100u), __builtin_amdgcn_perm(Rd, Ld, 0x05040100u)};
                }
            } else {
                // the file's last, partial tile: frame by frame (24-bit: three 2-byte stores each)
                const uint32_t Ls[4] = {La, Lb, Lc, Ld}, Rs[4] = {Ra, Rb, Rc, Rd};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (nl0 + 4u * Q + k < j0.nout) {
                        const uint32_t Lv = Ls[k], Rv = Rs[k];
                        D2D_GLOBAL uint16_t* p16 = reinterpret_cast<D2D_GLOBAL uint16_t*>(as_global(gq + FB * k));
                        if constexpr (SBY == 3) { p16[0] = (uint16_t)Lv; p16[1] = (uint16_t)(((Lv >> 16) & 0xFFu) | (Rv << 8)); p16[2] = (uint16_t)(Rv >> 8); }
                        else if constexpr (SBY == 4) { p16[0] = (uint16_t)Lv; p16[1] = (uint16_t)(Lv >> 16); p16[2] = (uint16_t)Rv; p16[3] = (uint16_t)(Rv >> 16); }
                        else { p16[0] = (uint16_t)Lv; p16[1] = (uint16_t)Rv; }
                    }
                }
            }
        }
    };
    // SCR: the lane's runs of three consecutive integers of channel c go straight to that channel's scratch line
    auto store_scr = [&](uint32_t tile, uint32_t c, const int32_t (&v)[NS]) {
        D2D_GLOBAL int32_t* xs = as_global(jobs[c].xs) + (size_t)tile * TILE + lane_fr;
        const uint32_t nl = tile * (uint32_t)TILE + lane_fr;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (nl + (uint32_t)PH * g + (uint32_t)PHH - 1u < j0.nout) {
#pragma unroll
                for (int k = 0; k < PHH; ++k) xs[PH * g + k] = v[PHH * g + k];
            } else
#pragma unroll
                for (int k = 0; k < PHH; ++k) if (nl + (uint32_t)PH * g + k < j0.nout) xs[PH * g + k] = v[PHH * g + k];
        }
    };
    // SCR, the pipelined loop: both channels' integers of a tile leave the slice as 16-byte rows along their scratch lines (a lane storing its
    // own runs of three left every line to several partial writes: stage A wrote twice the bytes of its scratch, profiles/r03_summary_c5.txt)
    auto store_scr_tile = [&](uint32_t tile) {
        const uint32_t nl0 = tile * (uint32_t)TILE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            D2D_GLOBAL int32_t* xs = as_global(jobs[c].xs) + (size_t)nl0;
#pragma unroll
            for (int p = 0; p < QPASS; ++p) {
                const uint32_t Q = lane + 64u * p;
                if ((NQ % 64) != 0 && p == QPASS - 1 && Q >= (uint32_t)NQ) continue;
                const i32x4 v = *reinterpret_cast<const i32x4*>(ob + c * TILE + 4 * Q);
                if (nl0 + 4u * Q + 3u < j0.nout) *reinterpret_cast<D2D_GLOBAL i32x4*>(xs + 4u * Q) = v;
                else {
                    const int32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (nl0 + 4u * Q + k < j0.nout) xs[4u * Q + k] = e[k];
                }
            }
        }
    };
    auto merge_extremes = [&](const Fast& f, auto cc) { constexpr uint32_t c = decltype(cc)::value; if constexpr (WIDE) { const vint d = f.tmx - kBias > kBias - f.tmn ? f.tmx - kBias : kBias - f.tmn; vdev[c] = d > vdev[c] ? d : vdev[c]; } else vdev[c] = mx_max3(vdev[c], f.tmx - kBias, kBias - f.tmn); };

    const uint32_t wv = coop ? blockIdx.x : blockIdx.x * m.nwaves + wave;       // this wave's (COOP: this block's) index among the file's tile workers
#if D2D_MX_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), rt_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_sum[3] = {0, 0, 0}, st_last = t_start;
    auto stamp = [&](int slot) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_sum[slot] += t - st_last; st_last = t; };
#else
    auto stamp = [](int) {};
#endif
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // The pipelined loop over the tiles t_begin + wv + k * wstride < t_end:
    //   region A (tile t):  chain of channel 0  ||  requantise channel 1 of tile t-1; its frames leave after the next prefetch is out
    //   region B (tile t):  chain of channel 1  ||  requantise channel 0 of tile t
    //   IL (byte-interleaved stereo; every tile of the range inside the call): a piece carries both channels, so channel 0's bytes of the
    //   NEXT tile go to its buffer while channel 1's chain runs, and channel 1's wait in `keep` (two registers per piece) until their
    //   buffer is free one region later; each piece is fetched once, a whole chain ahead:
    //   A start: [pf = 2nd half of tile t]  ch0 part -> buf0, ch1 part -> buf1, keep (1st half, ch1) -> buf1;  request 1st half of tile t+1
    //   B start: [pf = 1st half of tile t+1]  ch0 part -> buf0, ch1 part -> keep;                              request 2nd half of tile t+1
    auto run_loop = [&](uint32_t t_begin, uint32_t t_end, auto af, auto ilc) {
        constexpr bool AF = decltype(af)::value;
        constexpr bool IL = decltype(ilc)::value;
        static_assert(!IL || AF, "the interleaved staging has no gather path");
        [[maybe_unused]] uint32_t keep[IL ? 2 * PF : 1];
        [[maybe_unused]] auto il_first = [&]() {           // pf = a tile's first half: channel 0's bytes to its buffer, channel 1's kept
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                il_put(0u, C0{}, i, __builtin_amdgcn_perm(pf[i].y, pf[i].x, 0x06040200u), __builtin_amdgcn_perm(pf[i].w, pf[i].z, 0x06040200u));
                keep[2 * i] = __builtin_amdgcn_perm(pf[i].y, pf[i].x, 0x07050301u);
                keep[2 * i + 1] = __builtin_amdgcn_perm(pf[i].w, pf[i].z, 0x07050301u);
            }
        };
        [[maybe_unused]] auto il_second = [&]() {          // pf = the second half: both channels' bytes, and the kept ones, to their buffers
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                il_put(0u, C1{}, i, __builtin_amdgcn_perm(pf[i].y, pf[i].x, 0x06040200u), __builtin_amdgcn_perm(pf[i].w, pf[i].z, 0x06040200u));
                il_put(1u, C1{}, i, __builtin_amdgcn_perm(pf[i].y, pf[i].x, 0x07050301u), __builtin_amdgcn_perm(pf[i].w, pf[i].z, 0x07050301u));
                il_put(1u, C0{}, i, keep[2 * i], keep[2 * i + 1]);
            }
        };
        uint32_t wt = t_begin + wv;
        if (wt < t_end) {
            if (coop) { if (coop_fast(wt)) coop_issue(wt); }
            else if constexpr (IL) { il_issue(wt, C0{}); wave_sync2(); il_first(); il_issue(wt, C1{}); }
            else issue_loads(wt, C0{}, af);
            // AF: every trip issues the same loads and stores in the same order (the first trip stores whatever the slice holds to its
            // own tile, rewritten one trip later; the last trip re-requests its own tile): the compiler can then count its waits
            if (AF && !SCR && !(dbg & 64)) store_tile(wt, true);
        }
        v16f accA[G], accB[G];                                  // channel 0's / channel 1's accumulators
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) accB[g][i] = 0.0f;
        bool have_prev = false;
        uint32_t pw = wt;                                       // the tile whose channel 1 still waits for its epilogue
        for (; wt < t_end; wt += wstride) {
            const bool more = wt + wstride < t_end;
            const uint32_t nxt = more ? wt + wstride : wt;
            // a trip = (tile wt, channel pair pp): its channels 2 pp (buffer 0, accA) and 2 pp + 1 (buffer 1, accB); the epilogue that still
            // waits when it starts is channel 1 of the trip before -- the last pair of the tile before for pp = 0, whose frames are then complete
            static_for<0, NPR>([&](auto ppc) {
            constexpr int pp = decltype(ppc)::value;
            using CH0 = std::integral_constant<int, 2 * pp>;
            using CH1 = std::integral_constant<int, 2 * pp + 1>;
            using CHN = std::integral_constant<int, pp + 1 < NPR ? 2 * pp + 2 : 0>;      // the next trip's first channel
            constexpr uint32_t chp = pp == 0 ? (uint32_t)NCH - 1u : 2u * pp - 1u;
            using CHP = std::integral_constant<uint32_t, chp>;
            using CHA = std::integral_constant<uint32_t, 2u * pp>;
            const uint32_t tp = pp == 0 ? pw : wt;
            const bool prev = pp > 0 || have_prev;
            // ---- region A ----
            stamp(2);
            if (coop) {
                // (the tile index is the block's: every wave takes the same branches and meets the same barriers)
                __syncthreads();                                  // every wave is done with the tile before
                if (coop_fast(wt)) coop_write();
                else {                                            // a tile at the call's edges: every wave gathers its own two channels byte by byte
                    issue_loads(wt, C0{}, std::false_type{}); write_lds(C0{});
                    issue_loads(wt, C1{}, std::false_type{}); write_lds(C1{});
                }
                if (more && coop_fast(nxt)) coop_issue(nxt);
                __syncthreads();
            } else if constexpr (IL) {
                wave_sync2();
                il_second();
                il_issue(nxt, C0{});
                wave_sync2();
            } else {
                wave_sync2();
                if (!(dbg & 4)) {
                    write_lds(C0{});
                    issue_loads(wt, CH1{}, af);
                }
                wave_sync2();
            }
            stamp(0);
            {
                Fast f;
                fast_begin(f, tp, CHP{});
                if (dbg & 2) chain(0u, accA, no_hook);
                else if (dbg & 1) { static_for<0, NJ>([&](auto jc) { fast_job(f, accB, jc); }); for (int g = 0; g < G; ++g) accA[g] = cinit + (float)lane; }
                else chain(0u, accA, [&](auto uc) { fast_hook(f, accB, uc); });
                pin(accA);                      // the chain ends HERE (or the compiler sinks its MFMAs into the blocks that use them, behind the epilogue)
                stamp(1);
                if (prev) {
                    if constexpr (SCR) put_samples(chp, f.res);
                    else {
                        if (!(dbg & 3) && fast_failed(f, tp)) { int32_t o[NS]; redo_acc(accB, tp, CHP{}, o); put_samples(chp, o); } else merge_extremes(f, CHP{});
                    }
                }
                // (the scratch flavour's region tails run no inline-asm instruction, and its next epilogue starts a whole staging phase later; holding
                // the sets there cost <8, 688, 2, 0, 0> thirty spilled registers and config 5 a fifth of its stage A)
                if constexpr (!SCR) hold_acc(accB);
            }
            // ---- region B ----
            stamp(2);
            wave_sync2();
            if constexpr (IL) {
                il_first();
                il_issue(nxt, C1{});
            } else if (!(dbg & 4) && !coop) {
                write_lds(C1{});
                if constexpr (pp + 1 < NPR) issue_loads(wt, CHN{}, af);
                else if (AF || more) issue_loads(nxt, C0{}, af);
            }
            if constexpr (pp == 0) {
                if constexpr (AF) { if (!SCR) store_tile(pw, true); }
                else if (have_prev && !SCR) store_tile(pw);
                if constexpr (SCR) { if (have_prev) { wave_sync2(); store_scr_tile(pw); } }
            }
            wave_sync2();
            stamp(0);
            {
                Fast f;
                fast_begin(f, wt, CHA{});
                if (dbg & 2) chain(1u, accB, no_hook);
                else if (dbg & 1) { static_for<0, NJ>([&](auto jc) { fast_job(f, accA, jc); }); for (int g = 0; g < G; ++g) accB[g] = cinit - (float)lane; }
                else chain(1u, accB, [&](auto uc) { fast_hook(f, accA, uc); });
                pin(accB);
                stamp(1);
                if constexpr (SCR) put_samples(0, f.res);
                else {
                    if (!(dbg & 3) && fast_failed(f, wt)) { int32_t o[NS]; redo_acc(accA, wt, CHA{}, o); put_samples(CHA::value, o); } else merge_extremes(f, CHA{});
                }
                if constexpr (!SCR) hold_acc(accA);
            }
            });
            have_prev = true; pw = wt;
        }
        if (have_prev) {
            // drain: the last channel of the wave's last tile
            constexpr uint32_t chl = (uint32_t)NCH - 1u;
            using CHL = std::integral_constant<uint32_t, chl>;
            Fast f;
            fast_begin(f, pw, CHL{});
            static_for<0, NJ>([&](auto jc) { fast_job(f, accB, jc); });
            if constexpr (SCR) { hold_acc(accB); put_samples(1, f.res); wave_sync2(); store_scr_tile(pw); wave_sync2(); }
            else {
                if (fast_failed(f, pw)) { int32_t o[NS]; redo_acc(accB, pw, CHL{}, o); put_samples(chl, o); } else merge_extremes(f, CHL{});
                hold_acc(accB);
                wave_sync2();
                store_tile(pw);
                wave_sync2();
            }
        }
    };
    // One tile the careful way, start to finish (call edges: the window reaches into the carried history or past the call's full
    // blocks, so its bytes are gathered one by one).
    auto slow_tile = [&](uint32_t t) {
        if constexpr (NPR > 1) {
            static_for<0, NPR>([&](auto ppc) {
                constexpr int pp = decltype(ppc)::value;
                wave_sync2();
                issue_loads(t, std::integral_constant<int, 2 * pp>{}, std::false_type{}); write_lds(C0{});
                issue_loads(t, std::integral_constant<int, 2 * pp + 1>{}, std::false_type{}); write_lds(C1{});
                wave_sync2();
                {
                    int32_t o0[NS];
                    redo(0u, t, std::integral_constant<uint32_t, 2u * pp>{}, o0);
                    put_samples(2u * pp, o0);
                }
                {
                    int32_t o1[NS];
                    redo(1u, t, std::integral_constant<uint32_t, 2u * pp + 1u>{}, o1);
                    put_samples(2u * pp + 1u, o1);
                }
            });
            wave_sync2();
            store_tile(t);
            return;
        }
        wave_sync2();
        issue_loads(t, C0{}, std::false_type{});
        if (il) write_lds_t(C0{}, std::true_type{}); else write_lds(C0{});
        issue_loads(t, C1{}, std::false_type{});
        if (il) write_lds_t(C1{}, std::true_type{}); else write_lds(C1{});
        wave_sync2();
        // (one channel at a time, its samples put away before the other channel's chain starts: shorter live ranges)
        if constexpr (SCR) {
            // the exact integers need no careful path: the chain, then every job of the epilogue at once
            static_for<0, 2>([&](auto cc) {
                constexpr uint32_t c = decltype(cc)::value;
                v16f acc[G];
                chain(c, acc, no_hook);
                pin(acc);
                Fast f;
                fast_begin(f, t, std::integral_constant<uint32_t, c>{});
                static_for<0, NJ>([&](auto jc) { fast_job(f, acc, jc); });
                hold_acc(acc);
                store_scr(t, c, f.res);
            });
        } else {
            {
                int32_t o0[NS];
                redo(0u, t, std::integral_constant<uint32_t, 0u>{}, o0);
                put_samples(0, o0);
            }
            {
                int32_t o1[NS];
                redo(1u, t, std::integral_constant<uint32_t, 1u>{}, o1);
                put_samples(1, o1);
            }
            wave_sync2();
            store_tile(t);
        }
    };
#ifndef D2D_MX_SCR_AF
#define D2D_MX_SCR_AF 1           // the planar scratch flavour walks the call's inner tiles in the fixed-order loop too (round 4: +1.4 .. 3 %; 0: the general loop)
#endif
    if (SCR ? (il || (D2D_MX_SCR_AF && fast_layout && !coop)) : (fast_layout || il)) {
        // the tiles [t_lo, t_hi) lie inside the call's full blocks: the loop without the gather path; the few around them one by one
        const int64_t T = (int64_t)TILE * MB;
        auto is_fast = [&](uint32_t w) { const int32_t ab = tile_ab16(w); return ab >= 0 && (uint32_t)ab + 16u * NCHK <= full_bytes; };
        uint32_t t_lo = first0 >= 0 ? 0u : (uint32_t)((-first0 + T - 1) / T);
        if (t_lo > nwt) t_lo = nwt;
        uint32_t t_hi = t_lo;
        {
            const int64_t room = (int64_t)full_bytes - 16 * NCHK - first0;
            if (room >= 0) { const int64_t e = room / T + 1; t_hi = e > (int64_t)nwt ? nwt : (uint32_t)e; if (t_hi < t_lo) t_hi = t_lo; }
            while (t_hi > t_lo && !is_fast(t_hi - 1)) --t_hi;
            while (t_hi < nwt && t_hi >= t_lo && is_fast(t_hi) && (t_hi > t_lo || is_fast(t_lo))) ++t_hi;
        }
        { const uint32_t nfull = j0.nout / (uint32_t)TILE; if (t_hi > nfull) t_hi = nfull > t_lo ? nfull : t_lo; }     // whole tiles only
        if (NPR == 1 && il) { if constexpr (NPR == 1) run_loop(t_lo, t_hi, std::true_type{}, std::true_type{}); }       // (several pairs per wave: planar input only)
        else if constexpr (!SCR || D2D_MX_SCR_AF) run_loop(t_lo, t_hi, std::true_type{}, std::false_type{});
        const uint32_t n_edge = t_lo + (nwt - t_hi);
        for (uint32_t i = wv; i < n_edge; i += wstride) slow_tile(i < t_lo ? i : t_hi + (i - t_lo));
    } else {
        run_loop(0u, nwt, std::false_type{}, std::false_type{});
    }

#if D2D_MX_STAMPS
    if (lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        atomicMin(&d2d_mx_stamps[0], dt); atomicMax(&d2d_mx_stamps[1], dt); atomicAdd(&d2d_mx_stamps[2], dt); atomicAdd(&d2d_mx_stamps[3], 1ull);
        for (int i = 0; i < 3; ++i) atomicAdd(&d2d_mx_stamps[4 + i], st_sum[i]);
        atomicAdd(&d2d_mx_stamps[7], __builtin_amdgcn_s_memrealtime() - rt_start);      // constant 100 MHz: sum[2] / sum[7] = core clock / 100 MHz
    }
#endif
    if constexpr (SCR) return;                              // (stage B / the noise shaper keep the peaks)
    // peak meter: |x| in LSB; undo the power-of-two part exactly
    const double unscale = 1.0 / (double)(1u << (a.epi.bits - 1));   // (float: fbits = S - 31, so dev * 2^-fbits * 2^-31 = dev * 2^-S)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const vint dev = vdev[c];                                          // |x| = |v| * 2^-F exactly
        double p = ldexp((double)dev, -m.fbits) * unscale;
        if constexpr (GN) p = p * a.epi.gain;                          // |y| is exact: one rounding, as the oracle's |y * gain| of the largest sample
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p = fmax(p, __shfl_xor(p, o));
        if (lane == 0 && p > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(jobs[c].peak), (unsigned long long)__double_as_longlong(p));
    }
}

// ---- host side -------------------------------------------------------------------------------
// (MB, taps) of the filters this kernel serves: X_M32, C_M32, E_M32, A_M32, A_M64, C_M64, E_M64, E_M128.  The file is compiled in four parts
// (Makefile: D2D_MX_PART), each with the kernels of some shapes; part 0 also holds the table builder and the dispatcher.
#ifndef D2D_MX_PART
#define D2D_MX_PART 0
#endif
// one shape per object (Makefile: -DD2D_MX_PART=0..7), so that a clean build spreads over the cores; part 0 also holds the table
// builder and the dispatcher
#define D2D_MX_SHAPES_0(X) X(4, 560)
#ifdef D2D_MX_DEV
#define D2D_MX_SHAPES_1(X)
#define D2D_MX_SHAPES_2(X)
#define D2D_MX_SHAPES_3(X)
#define D2D_MX_SHAPES_4(X)
#define D2D_MX_SHAPES_5(X)
#define D2D_MX_SHAPES_6(X)
#define D2D_MX_SHAPES_7(X)
#else
#define D2D_MX_SHAPES_1(X) X(4, 352)
#define D2D_MX_SHAPES_2(X) X(4, 384)
#define D2D_MX_SHAPES_3(X) X(4, 512)
#define D2D_MX_SHAPES_4(X) X(8, 688)
#define D2D_MX_SHAPES_5(X) X(8, 1024)
#define D2D_MX_SHAPES_6(X) X(8, 1104)
#define D2D_MX_SHAPES_7(X) X(16, 2192)
#endif
#define D2D_MX_SHAPES(X) D2D_MX_SHAPES_0(X) D2D_MX_SHAPES_1(X) D2D_MX_SHAPES_2(X) D2D_MX_SHAPES_3(X) D2D_MX_SHAPES_4(X) D2D_MX_SHAPES_5(X) D2D_MX_SHAPES_6(X) D2D_MX_SHAPES_7(X)
// the f64 flavours (KIND + 4: other levels, 20-bit, the float dither) of the shapes that serve frames (not the cascade's A filters), one
// object each too (Makefile: -DD2D_MX_GPART=0..5, D2D_MX_PART=99)
#define D2D_MX_GSHAPES_0(X) X(4, 560)
#ifdef D2D_MX_DEV
#define D2D_MX_GSHAPES_1(X)
#define D2D_MX_GSHAPES_2(X)
#define D2D_MX_GSHAPES_3(X)
#define D2D_MX_GSHAPES_4(X)
#define D2D_MX_GSHAPES_5(X)
#else
#define D2D_MX_GSHAPES_1(X) X(4, 384)
#define D2D_MX_GSHAPES_2(X) X(4, 512)
#define D2D_MX_GSHAPES_3(X) X(8, 1024)
#define D2D_MX_GSHAPES_4(X) X(8, 1104)
#define D2D_MX_GSHAPES_5(X) X(16, 2192)
#endif
#define D2D_MX_GSHAPES(X) D2D_MX_GSHAPES_0(X) D2D_MX_GSHAPES_1(X) D2D_MX_GSHAPES_2(X) D2D_MX_GSHAPES_3(X) D2D_MX_GSHAPES_4(X) D2D_MX_GSHAPES_5(X)
// several channel pairs per wave (planar multichannel frames: NPR = 2 quad, 3 a 5.1 stream, 4 a 7.1 / eight-channel stream), one object per (shape, pairs)
// (Makefile: -DD2D_MX_MPART=n, D2D_MX_PART=99): X(object, MB, taps, pairs)
#ifdef D2D_MX_DEV
#define D2D_MX_MLIST(X) X(0, 4, 560, 3)
#else
#define D2D_MX_MLIST(X) X(0, 4, 560, 3) X(1, 8, 1104, 3) X(2, 16, 2192, 3) X(3, 4, 560, 2) X(4, 4, 560, 4) X(5, 8, 1104, 2) X(6, 8, 1104, 4)
#endif
template <int N> struct MxPairsPart;
#define X(n, mb, nt, npr) template <> struct MxPairsPart<n> { static constexpr int MBv = mb, NTv = nt, NPRv = npr; }; \
                          hipError_t launch_fir_mx_mp##n(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s);
D2D_MX_MLIST(X)
#undef X
// the one-pass form of the 32-bit tap grid (seven digits, four phases per group; stereo frames through the f64 requantiser): one object (Makefile: -DD2D_MX_WPART=0, D2D_MX_PART=99)
#define D2D_MX_WSHAPES_0(X) X(4, 560)
#ifdef D2D_MX_DEV
#define D2D_MX_WSHAPES_1(X)
#else
#define D2D_MX_WSHAPES_1(X) X(8, 1104)
#endif
#define D2D_MX_WSHAPES(X) D2D_MX_WSHAPES_0(X) D2D_MX_WSHAPES_1(X)
hipError_t launch_fir_mx_wide0(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s);
hipError_t launch_fir_mx_wide1(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s);
#define D2D_MX_DECL(n) hipError_t launch_fir_mx_part##n(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s); \
                       hipError_t launch_fir_mx_gain##n(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s);
D2D_MX_DECL(0) D2D_MX_DECL(1) D2D_MX_DECL(2) D2D_MX_DECL(3) D2D_MX_DECL(4) D2D_MX_DECL(5) D2D_MX_DECL(6) D2D_MX_DECL(7)

#if D2D_MX_PART == 0
bool mx_supported(int MB, int NT) {
#define X(mb, nt) if (MB == mb && NT == nt) return true;
    D2D_MX_SHAPES(X)
#undef X
    return false;
}
bool mx_pairs_supported(int MB, int NT, int npairs) {
#define X(n, mb, nt, npr) if (MB == mb && NT == nt && npairs == npr) return true;
    D2D_MX_MLIST(X)
#undef X
    return false;
}
bool mx_wide_supported(int MB, int NT) {
#define X(mb, nt) if (MB == mb && NT == nt) return true;
    D2D_MX_WSHAPES(X)
#undef X
    return false;
}
bool mx_gain_supported(int MB, int NT) {
#define X(mb, nt) if (MB == mb && NT == nt) return true;
    D2D_MX_GSHAPES(X)
#undef X
    return false;
}

// e2m3 code of x (a multiple of 1/8 up to 2, of 1/4 up to 4, of 1/2 up to 7.5)
static uint32_t e2m3_code(double x) {
    const uint32_t s = x < 0 ? 32u : 0u;
    const double ax = fabs(x);
    for (uint32_t c = 0; c < 32; ++c) {
        const uint32_t e = c >> 3, mm = c & 7;
        const double v = e ? (1.0 + mm / 8.0) * (double)(1 << (e - 1)) : mm * 0.125;
        if (v == ax) return ax == 0 ? 0u : (s | c);
    }
    fprintf(stderr, "d2d: %g is not an e2m3 number\n", x);
    abort();
}
// balanced base-32 digit l of v: v = sum d_l 32^l, every d in [-16, 15]
static int digit32(int64_t v, int l) {
    int dd = 0;
    for (int i = 0; i <= l; ++i) {
        dd = (int)(((v + 16) & 31) - 16);
        v = (v - dd) / 32;
    }
    return dd;
}

// The recombination v = lo + 2^15 hi with lo = S0 + 32 S1 + 2^10 S2 and hi = S3 + 32 S4 is done in f32: exact while every value that can
// occur stays below 2^24.  A digit sum over ANY subset of the window's bits is bounded by the sum of the digits' magnitudes.
bool mx_exact(const d2d_filter_def& f) {
    int64_t sa[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < f.ntaps; ++k)
        for (int l = 0; l < 5; ++l) { const int d = digit32(2 * (int64_t)tap_q(f, k), l); sa[l] += d < 0 ? -d : d; }
    // the kernel's two f32 parts: digits 0-2 | 3-4 (M = 128: 0-1 | 2-4), the -2^S start value in digit 4
    const bool s23 = f.M == 128;
    const int64_t lo = s23 ? sa[0] + 32 * sa[1] : sa[0] + 32 * sa[1] + 1024 * sa[2];
    const int64_t hi = s23 ? sa[2] + 32 * sa[3] + 1024 * (sa[4] + ((int64_t)1 << (f.S - 20))) : sa[3] + 32 * (sa[4] + ((int64_t)1 << (f.S - 20)));
    // 2 q has to fit five digits: |2q| <= 16 * (32^5 - 1) / 31
    for (int k = 0; k < f.ntaps; ++k) { const int64_t q2 = 2 * (int64_t)tap_q(f, k); if (q2 > 16236247 || q2 < -17318416) return false; }
    return f.S >= 20 && f.S <= 30 && lo < (1 << 24) && hi < (1 << 24);
}

// Tap fragments: [4 byte shifts][NF fragments][64 lanes x 16 bytes | 64 lanes x 8 bytes].  Fragment f multiplies the stream dwords
// 2f (lane half 0) and 2f + 1 (half 1) of a column's window.  A lane l = matrix row l & 31, K half l >> 5; its element j (a 6-bit
// e2m3 code at bits [6j, 6j+6) of the lane's 192) meets B register p = j >> 3, nibble n = j & 7 = bit 4n + p of the dword, which
// arrives as 0.5 (p even) or 1.0 (p odd).  D row i lands in lane half (i >> 2) & 1, register 4 (i >> 3) + (i & 3) = 5 q + digit:
// phase 3 half + q (wide: register 7 q + digit, phase 2 half + q).
// full 32-bit tap j (0..N-1) of the 2^-(S+8) grid (filters/filter_tables.inc: half32), stored like the 24-bit halves
static inline int64_t tap_q32(const d2d_filter_def& f, int j) {
    const int h = f.ntaps / 2;
    return j >= h ? f.half32[j - h] : f.half32[h - 1 - j];
}
// the one-pass form of the 32-bit grid: 2 q32 in seven balanced base-32 digits, v = lo + 2^15 mid + 2^25 hi with lo = S0 + 32 S1 + 2^10 S2,
// mid = S3 + 32 S4, hi = S5 + 32 S6, each formed in f32 (accumulators from zero; the -2^(S+8) is subtracted in 64 bits)
bool mx_wide_exact(const d2d_filter_def& f) {
    if (!f.half32) return false;
    int64_t sa[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < f.ntaps; ++k) {
        const int64_t q2 = 2 * tap_q32(f, k);
        int64_t back = 0, w = 1;
        for (int l = 0; l < 7; ++l) { const int d = digit32(q2, l); sa[l] += d < 0 ? -d : d; back += d * w; w *= 32; }
        if (back != q2) return false;                                  // 2 q32 does not fit seven digits
    }
    const int64_t lo = sa[0] + 32 * sa[1] + 1024 * sa[2], mid = sa[3] + 32 * sa[4], hi = sa[5] + 32 * sa[6];
    return f.S + 8 >= 28 && f.S + 8 <= 40 && lo < (1 << 24) && mid < (1 << 24) && hi < (1 << 24);
}

std::vector<int8_t> build_mx_tables(const d2d_filter_def& f, bool msb_first, bool wide) {
    const int M = f.M, N = f.ntaps, MB = M / 8;
    const int PH = wide ? 4 : 6, PHH = PH / 2, ND = wide ? 7 : 5;
    const int NF = mx_nf(MB, N, PH);
    const size_t per = (size_t)NF * MX_FRAG_BYTES;
    std::vector<int8_t> t(4 * per, 0);
    for (int sh = 0; sh < 4; ++sh)
        for (int fr = 0; fr < NF; ++fr)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 31, kh = l >> 5;
                const int half = (row >> 2) & 1, rho = 4 * (row >> 3) + (row & 3);
                uint32_t regs[6] = {0, 0, 0, 0, 0, 0};
                if (rho < PHH * ND) {
                    const int ph = PHH * half + rho / ND, dg = rho % ND;
                    for (int j = 0; j < 32; ++j) {
                        const int p = j >> 3, n = j & 7;
                        const int wb = 32 * (2 * fr + kh) + 4 * n + p;                             // bit of the staged window
                        const int tau = (msb_first ? (wb & ~7) + 7 - (wb & 7) : wb) - 8 * sh;     // its time index in the window
                        const int tap = tau - ph * M;
                        if (tau < 0 || tap < 0 || tap >= N) continue;
                        const int d = digit32(wide ? 2 * tap_q32(f, tap) : 2 * (int64_t)tap_q(f, tap), dg);
                        const uint32_t code = e2m3_code((p & 1) ? d * 0.125 : d * 0.25);
                        for (int b = 0; b < 6; ++b) if ((code >> b) & 1) regs[(6 * j + b) >> 5] |= 1u << ((6 * j + b) & 31);
                    }
                }
                int8_t* fb = &t[sh * per + (size_t)fr * MX_FRAG_BYTES];
                memcpy(fb + (size_t)l * 16, regs, 16);
                memcpy(fb + 1024 + (size_t)l * 8, regs + 4, 8);
            }
    return t;
}

#endif   // part 0

template <int MB, int NT, int G, int KIND, int SBY, int NPR = 1, int ND = 5>
static hipError_t launch_mx_t(Mfma2Args& m, uint32_t max_nout, uint32_t nrows, hipStream_t s) {
    static KernelPrep prep;
    int dev = 0;
    constexpr int PH = ND == 7 ? 4 : 6;
    const void* fn = reinterpret_cast<const void*>(&d2d_fir_mx_kernel<MB, NT, G, KIND, SBY, NPR, ND>);
    hipError_t e = prep.max_dynamic_lds(fn, 160 * 1024, &dev);
    if (e != hipSuccess) return e;
    constexpr uint32_t TILE = 32u * (uint32_t)PH * G;
    // LDS: the shared tap table, then per wave two stream buffers and the output slice; eight waves per block = two per SIMD
    m.off_waves = (uint32_t)mx_nf(MB, NT, PH) * MX_FRAG_BYTES;
    m.off_out = 2u * (uint32_t)mx_stream_bytes(MB, NT, G, PH);
    m.wave_lds = m.off_out + 2u * (uint32_t)NPR * TILE * 4u; // the slice: a row of TILE dwords per channel (the scratch flavour too: its integers leave as rows of the slice)
    const uint32_t wdbg = (m.f.dbg_flags >> 8) & 0xFFu;   // diagnostic override (d2d_params.debug_flags bits 8..15)
    uint32_t nwaves = wdbg ? wdbg : (uint32_t)(D2D_MX_THREADS / 64);
    if (nwaves < 1 || nwaves > D2D_MX_THREADS / 64) nwaves = D2D_MX_THREADS / 64;
    while (nwaves > 1 && (size_t)m.off_waves + (size_t)nwaves * m.wave_lds > 160 * 1024) { if (NPR > 1) --nwaves; else nwaves >>= 1; }
    const bool coop = SBY == 0 && m.f.coop;                 // a block = all channel pairs of a file on one tile: one wave per pair, one grid row per file
    if (coop) { nwaves = m.f.epi.channels / 2u; nrows /= m.ngroups; }
    m.nwaves = nwaves;
    const size_t smem = (size_t)m.off_waves + (size_t)m.nwaves * m.wave_lds;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    int blocks_per_cu, ncu;
    {
        std::lock_guard<std::mutex> g(prep.mu);
        if (prep.blocks_per_cu[dev] == 0 || smem != prep.smem_seen[dev] || m.nwaves != prep.nwaves_seen[dev]) {
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
            int nb = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, d2d_fir_mx_kernel<MB, NT, G, KIND, SBY, NPR, ND>, (int)(64 * m.nwaves), smem);
            if (e != hipSuccess) return e;
            prep.ncu[dev] = prop.multiProcessorCount;
            prep.blocks_per_cu[dev] = nb < 1 ? 1 : nb;
            prep.smem_seen[dev] = smem; prep.nwaves_seen[dev] = m.nwaves;
        }
        blocks_per_cu = prep.blocks_per_cu[dev]; ncu = prep.ncu[dev];
    }
    // every wave loops over its share of the wave-tiles: launch what is resident at once
    const uint32_t nwt_max = (max_nout + TILE - 1) / TILE;
    uint32_t gx = (uint32_t)(ncu * blocks_per_cu) / nrows;
    if (gx < 1) gx = 1;
    const uint32_t need = coop ? nwt_max : (nwt_max + m.nwaves - 1) / m.nwaves;
    if (gx > need) gx = need;
    hipLaunchKernelGGL((d2d_fir_mx_kernel<MB, NT, G, KIND, SBY, NPR, ND>), dim3(gx, nrows), dim3(64 * m.nwaves), smem, s, m);
    d2d_last_launched_kernel = launched_name<MB, NT, G, KIND, SBY, NPR, ND>("d2d_fir_mx_kernel");     // (all seven arguments: the way rocprofv3 prints the instantiation)
    return hipGetLastError();
}

#define D2D_MX_LAUNCH(mb, nt)                                                                                      \
    if (MB == mb && NT == nt) {                                                                                    \
        constexpr int G = mx_g(mb);                                                                                \
        if (m.f.to_scratch) return launch_mx_t<mb, nt, G, 0, 0>(m, max_nout, nrows, s);                             \
        if (m.f.epi.sample_bytes == 4) return launch_mx_t<mb, nt, G, 0, 4>(m, max_nout, nrows, s);                  \
        if (m.f.epi.sample_bytes == 2) {                                                                           \
            if (m.dkind == 1) return launch_mx_t<mb, nt, G, 1, 2>(m, max_nout, nrows, s);                           \
            if (m.dkind == 2) return launch_mx_t<mb, nt, G, 2, 2>(m, max_nout, nrows, s);                           \
            return launch_mx_t<mb, nt, G, 0, 2>(m, max_nout, nrows, s);                                             \
        }                                                                                                          \
        if (m.dkind == 1) return launch_mx_t<mb, nt, G, 1, 3>(m, max_nout, nrows, s);                               \
        if (m.dkind == 2) return launch_mx_t<mb, nt, G, 2, 3>(m, max_nout, nrows, s);                               \
        return launch_mx_t<mb, nt, G, 0, 3>(m, max_nout, nrows, s);                                                 \
    }
#define D2D_MX_PART_FN(n, shapes)                                                                                  \
    hipError_t launch_fir_mx_part##n(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s) { \
        shapes(D2D_MX_LAUNCH)                                                                                      \
        return hipErrorInvalidValue;                                                                               \
    }
#ifdef D2D_MX_WPART
// (wide: the groups per column of the 24-bit form -- twelve outputs at M = 32, eight at M = 64)
#define D2D_MX_WLAUNCH(mb, nt)                                                                                     \
    if (MB == mb && NT == nt) {                                                                                    \
        constexpr int G = mx_g(mb);                                                                                \
        if (m.f.epi.sample_bytes == 4) {                                                                           \
            if (m.f.epi.dither == 'F') return launch_mx_t<mb, nt, G, 7, 4, 1, 7>(m, max_nout, nrows, s);            \
            return launch_mx_t<mb, nt, G, 4, 4, 1, 7>(m, max_nout, nrows, s);                                       \
        }                                                                                                          \
        if (m.f.epi.sample_bytes == 2) {                                                                           \
            if (m.dkind == 1) return launch_mx_t<mb, nt, G, 5, 2, 1, 7>(m, max_nout, nrows, s);                     \
            if (m.dkind == 2) return launch_mx_t<mb, nt, G, 6, 2, 1, 7>(m, max_nout, nrows, s);                     \
            return launch_mx_t<mb, nt, G, 4, 2, 1, 7>(m, max_nout, nrows, s);                                       \
        }                                                                                                          \
        if (m.dkind == 1) return launch_mx_t<mb, nt, G, 5, 3, 1, 7>(m, max_nout, nrows, s);                         \
        if (m.dkind == 2) return launch_mx_t<mb, nt, G, 6, 3, 1, 7>(m, max_nout, nrows, s);                         \
        return launch_mx_t<mb, nt, G, 4, 3, 1, 7>(m, max_nout, nrows, s);                                           \
    }
#if D2D_MX_WPART == 0
hipError_t launch_fir_mx_wide0(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s) {
    D2D_MX_WSHAPES_0(D2D_MX_WLAUNCH)
    return hipErrorInvalidValue;
}
#else
hipError_t launch_fir_mx_wide1(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s) {
    D2D_MX_WSHAPES_1(D2D_MX_WLAUNCH)
    return hipErrorInvalidValue;
}
#endif
#elif defined(D2D_MX_MPART)
#define D2D_MX_CAT2(a, b) a##b
#define D2D_MX_CAT(a, b) D2D_MX_CAT2(a, b)
hipError_t D2D_MX_CAT(launch_fir_mx_mp, D2D_MX_MPART)(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s) {
    using P = MxPairsPart<D2D_MX_MPART>;
    constexpr int mb = P::MBv, nt = P::NTv, npr = P::NPRv, G = mx_g(mb);
    if (MB != mb || NT != nt || (int)m.npairs != npr) return hipErrorInvalidValue;
    if (m.f.epi.sample_bytes == 4) return launch_mx_t<mb, nt, G, 0, 4, npr>(m, max_nout, nrows, s);
    if (m.f.epi.sample_bytes == 2) {
        if (m.dkind == 1) return launch_mx_t<mb, nt, G, 1, 2, npr>(m, max_nout, nrows, s);
        if (m.dkind == 2) return launch_mx_t<mb, nt, G, 2, 2, npr>(m, max_nout, nrows, s);
        return launch_mx_t<mb, nt, G, 0, 2, npr>(m, max_nout, nrows, s);
    }
    if (m.dkind == 1) return launch_mx_t<mb, nt, G, 1, 3, npr>(m, max_nout, nrows, s);
    if (m.dkind == 2) return launch_mx_t<mb, nt, G, 2, 3, npr>(m, max_nout, nrows, s);
    return launch_mx_t<mb, nt, G, 0, 3, npr>(m, max_nout, nrows, s);
}
#elif defined(D2D_MX_GPART)
#define D2D_MX_GLAUNCH(mb, nt)                                                                                     \
    if (MB == mb && NT == nt) {                                                                                    \
        constexpr int G = mx_g(mb);                                                                                \
        if (m.f.epi.sample_bytes == 4) {                                                                           \
            if (m.f.epi.dither == 'F') return launch_mx_t<mb, nt, G, 7, 4>(m, max_nout, nrows, s);                  \
            return launch_mx_t<mb, nt, G, 4, 4>(m, max_nout, nrows, s);                                             \
        }                                                                                                          \
        if (m.f.epi.sample_bytes == 2) {                                                                           \
            if (m.dkind == 1) return launch_mx_t<mb, nt, G, 5, 2>(m, max_nout, nrows, s);                           \
            if (m.dkind == 2) return launch_mx_t<mb, nt, G, 6, 2>(m, max_nout, nrows, s);                           \
            return launch_mx_t<mb, nt, G, 4, 2>(m, max_nout, nrows, s);                                             \
        }                                                                                                          \
        if (m.dkind == 1) return launch_mx_t<mb, nt, G, 5, 3>(m, max_nout, nrows, s);                               \
        if (m.dkind == 2) return launch_mx_t<mb, nt, G, 6, 3>(m, max_nout, nrows, s);                               \
        return launch_mx_t<mb, nt, G, 4, 3>(m, max_nout, nrows, s);                                                 \
    }
#define D2D_MX_GPART_FN(n, shapes)                                                                                 \
    hipError_t launch_fir_mx_gain##n(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s) { \
        shapes(D2D_MX_GLAUNCH)                                                                                     \
        return hipErrorInvalidValue;                                                                               \
    }
#if D2D_MX_GPART == 0
D2D_MX_GPART_FN(0, D2D_MX_GSHAPES_0)
#elif D2D_MX_GPART == 1
D2D_MX_GPART_FN(1, D2D_MX_GSHAPES_1)
#elif D2D_MX_GPART == 2
D2D_MX_GPART_FN(2, D2D_MX_GSHAPES_2)
#elif D2D_MX_GPART == 3
D2D_MX_GPART_FN(3, D2D_MX_GSHAPES_3)
#elif D2D_MX_GPART == 4
D2D_MX_GPART_FN(4, D2D_MX_GSHAPES_4)
#else
D2D_MX_GPART_FN(5, D2D_MX_GSHAPES_5)
#endif
#elif D2D_MX_PART == 0
D2D_MX_PART_FN(0, D2D_MX_SHAPES_0)
hipError_t launch_fir_mx(Mfma2Args& m, int MB, int NT, uint32_t max_nout, uint32_t nrows, hipStream_t s) {
#define D2D_MX_ROUTE(list, fn) { auto route = [&]() -> int { list(X) return 0; }; if (route()) return fn(m, MB, NT, max_nout, nrows, s); }
#define X(mb, nt) if (MB == mb && NT == nt) return 1;
    if (m.f.taps32) {
        D2D_MX_ROUTE(D2D_MX_WSHAPES_0, launch_fir_mx_wide0)
#ifndef D2D_MX_DEV
        D2D_MX_ROUTE(D2D_MX_WSHAPES_1, launch_fir_mx_wide1)
#endif
        return hipErrorInvalidValue;
    }
    if (m.npairs > 1) {
#define Y(n, mb, nt, npr) if (MB == mb && NT == nt && (int)m.npairs == npr) return launch_fir_mx_mp##n(m, MB, NT, max_nout, nrows, s);
        D2D_MX_MLIST(Y)
#undef Y
        return hipErrorInvalidValue;
    }
    if (m.gainq && !m.f.to_scratch) {
        D2D_MX_ROUTE(D2D_MX_GSHAPES_0, launch_fir_mx_gain0)
#ifndef D2D_MX_DEV
        D2D_MX_ROUTE(D2D_MX_GSHAPES_1, launch_fir_mx_gain1) D2D_MX_ROUTE(D2D_MX_GSHAPES_2, launch_fir_mx_gain2) D2D_MX_ROUTE(D2D_MX_GSHAPES_3, launch_fir_mx_gain3)
        D2D_MX_ROUTE(D2D_MX_GSHAPES_4, launch_fir_mx_gain4) D2D_MX_ROUTE(D2D_MX_GSHAPES_5, launch_fir_mx_gain5)
#endif
        return hipErrorInvalidValue;
    }
    D2D_MX_ROUTE(D2D_MX_SHAPES_0, launch_fir_mx_part0)
#ifndef D2D_MX_DEV
    D2D_MX_ROUTE(D2D_MX_SHAPES_1, launch_fir_mx_part1) D2D_MX_ROUTE(D2D_MX_SHAPES_2, launch_fir_mx_part2) D2D_MX_ROUTE(D2D_MX_SHAPES_3, launch_fir_mx_part3)
    D2D_MX_ROUTE(D2D_MX_SHAPES_4, launch_fir_mx_part4) D2D_MX_ROUTE(D2D_MX_SHAPES_5, launch_fir_mx_part5) D2D_MX_ROUTE(D2D_MX_SHAPES_6, launch_fir_mx_part6)
    D2D_MX_ROUTE(D2D_MX_SHAPES_7, launch_fir_mx_part7)
#endif
#undef X
#undef D2D_MX_ROUTE
    return hipErrorInvalidValue;
}
int mx_groups(int MB) { return mx_g(MB); }
#if D2D_MX_STAMPS
void mx_debug_stamps(unsigned long long out[8]) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(d2d_mx_stamps), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(d2d_mx_stamps), z, sizeof(z));
}
#else
void mx_debug_stamps(unsigned long long out[8]) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif
#elif D2D_MX_PART == 1
D2D_MX_PART_FN(1, D2D_MX_SHAPES_1)
#elif D2D_MX_PART == 2
D2D_MX_PART_FN(2, D2D_MX_SHAPES_2)
#elif D2D_MX_PART == 3
D2D_MX_PART_FN(3, D2D_MX_SHAPES_3)
#elif D2D_MX_PART == 4
D2D_MX_PART_FN(4, D2D_MX_SHAPES_4)
#elif D2D_MX_PART == 5
D2D_MX_PART_FN(5, D2D_MX_SHAPES_5)
#elif D2D_MX_PART == 6
D2D_MX_PART_FN(6, D2D_MX_SHAPES_6)
#else
D2D_MX_PART_FN(7, D2D_MX_SHAPES_7)
#endif

}  // namespace d2d
