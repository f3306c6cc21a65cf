// d2d_engine.cpp -- host side of the engine and the C ABI of include/dsd2dxd_amd.h.
//
// One Engine is what the reference calls an Rdsd2Pcm (one per file, /root/reference/src/main.rs:
// 325-342,361-394), widened to `n_files` files so that the Rayon par_iter over files
// (src/main.rs:280-300) becomes a grid dimension of one launch.  The engine owns, in HBM:
//   * the filter tables (nibble LUTs or int8 MFMA fragments; stage-B coefficients)
//   * per (file, channel): `keep` history bytes (ping-pong), the running peak, and for the 48k
//     cascade an f64 scratch line [P carried | stage-A outputs of this call]
// No CPU fallback exists: without a HIP device d2d_create fails with D2D_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dsd2dxd_amd.h"
#include "d2d_filters.h"
#include "d2d_internal.h"
#include "d2d_launch.h"
#include "d2d_mfma.h"
#include "d2d_mx.h"
#include "d2d_px.h"

using namespace d2d;

namespace {

thread_local std::string g_create_error;

struct FileState {
    uint64_t pos = 0;    // bytes per channel consumed
    uint64_t nfir = 0;   // FIR outputs produced
    uint64_t nres = 0;   // stage-B outputs produced (48k family)
};

constexpr int JOB_SLOTS = 8;

}  // namespace

struct d2d_engine {
    d2d_params p{};
    uint32_t n_files = 1;
    FilterChoice fc;
    int M = 0, Mb = 0, N = 0, Wb = 0, S = 0;
    uint32_t B = 1, keep = 0, nstreams = 0;
    uint32_t Cin = 0;          // channels of the input layout
    uint32_t C = 0, c0 = 0;    // channels converted (streams per file, width of the output frame) and the first of them
    uint32_t kernel = D2D_KERNEL_LUT;
    LutLayout lut{};
    MfmaLayout mfma{};
    bool mfma_v2 = false;      // the two-group matrix-core kernel (d2d_kernels_mfma2.hip) serves this shape
    int mfma_pipe = 0;         // ... through its software-pipelined variant (d2d_kernels_mfma3.hip; stereo 24-bit at 0 dB): 3 dense chain, 4 sparse chain
    std::string kname;
    std::string launched;      // the FIR kernel the last call really enqueued (d2d_last_launched_kernel)
    Epilogue epi{};
    std::string err;
    std::vector<FileState> files;

    // device
    void* d_fir_tables = nullptr; size_t fir_table_bytes = 0;
    double* d_resamp = nullptr;   size_t resamp_bytes = 0;
    uint8_t* d_hist[2] = {nullptr, nullptr}; int hist_cur = 0;
    double* d_peak = nullptr;
    int32_t* d_scratch = nullptr; size_t scratch_stride = 0;  // stage-A integers per stream (multiple of 4)
    bool noise_shape = false;             // 'N' dither: the FIR writes integers, a sequential pass requantises
    double* d_ns[2] = {nullptr, nullptr}; int ns_cur = 0;   // its state: two errors per stream, ping-pong between calls
    uint8_t* d_ns_dump = nullptr;                            // NoiseShapeArgs::dump
    double* d_ys = nullptr; size_t ys_stride = 0;            // 'N' on the 48k family: stage B's outputs as f64, one line per stream
    uint32_t xs_hist = 0;                 // samples carried in front of each scratch line (P of the resampler, else 0)
    StreamJob* d_jobs = nullptr;
    StreamJob* h_jobs = nullptr;          // pinned, JOB_SLOTS x nstreams
    hipEvent_t job_ev[JOB_SLOTS]{}; bool job_ev_used[JOB_SLOTS]{}; int job_slot = 0;
    hipStream_t own_stream = nullptr;     // used by the host-pointer entry points
    // d2d_translate_batch_host: upload / convert / download streams, their events, double-buffered staging
    hipStream_t hb_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t hb_ev[6] = {};
    uint8_t* hb_in[2] = {nullptr, nullptr}; uint8_t* hb_out[2] = {nullptr, nullptr};
    size_t hb_in_stride = 0, hb_out_stride = 0;
    hipStream_t last_stream = nullptr;
    // measurement
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool; size_t prof_used = 0;     // around the FIR launch
    std::vector<std::pair<hipEvent_t, hipEvent_t>> step_pool; size_t step_used = 0;     // around every kernel of a batch call
    // staging for d2d_translate (host pointers)
    uint8_t* d_in = nullptr; size_t d_in_cap = 0;
    // planar copies of byte-interleaved inputs (one slice per file), see d2d_deinterleave_kernel
    uint8_t* d_planar = nullptr; size_t planar_stride = 0;
    bool deinterleave = false;
    bool coop = false;                    // byte-interleaved 4/8-channel input de-interleaved inside the fp6 kernel's staging (FirArgs::coop)
    // tap_bits = 32: the FIR runs twice into the scratch (the 24-bit table, then the residual table q32 - 256 q) and d2d_fine_combine_kernel
    // finishes v = 256 v_hi + v_lo; the second half of the scratch, of the job table and `lo_*` belong to the second pass
    bool fine = false;
    bool taps32 = false;                  // tap_bits = 32 in ONE pass (round 4): stereo at M = 32 on the fp6 kernel's seven-digit flavour -- one table, no scratch, no combining pass
    std::vector<int32_t> lo_half;
    d2d_filter_def lo_def{};
    void* d_fir_tables_lo = nullptr;
    int mfma_pipe_lo = 0;
    // DSD64 / DSD128 -> 48k multiples: one polyphase pass over the bits (d2d_kernels_px.hip); fc.fir / fc.resamp only count frames then
    const d2d_poly_def* poly = nullptr;
    bool poly_plain = false;              // ... through the bit-by-bit kernel (D2D_KERNEL_LUT engines)
    bool cascade() const { return fc.resamp && !poly; }      // the two-kernel 48k path (DSD256 / DSD512 input)
    // MONO2 (round 4): a mono stream on the fp6 pipelined kernel as a planar PAIR -- the two halves of a call converted side by side (FirArgs::mono2);
    // calls it does not fit (odd sizes, very short ones) take the engine's ordinary mono kernel: same bytes either way
    bool mono2_ok = false;
    int mono2_pipe = 0;                   // 5: the fp6 kernel (M = 32, 64, 128), 3: the int8 pipelined kernel (M = 8, 16)
    void* d_fir_tables_m2 = nullptr;
    bool il2 = false;                     // byte-interleaved stereo input de-interleaved inside the pipelined frame kernels' staging (FirArgs::il2)
    uint8_t* d_out = nullptr; size_t d_out_cap = 0;

    int fail(int code, const std::string& m) { err = m; return code; }
    int hip_fail(hipError_t e, const char* what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return D2D_ERR_DEVICE;
    }
};

#define HIPCHK(e_, call)                                                  \
    do {                                                                  \
        hipError_t _r = (call);                                           \
        if (_r != hipSuccess) return (e_)->hip_fail(_r, #call);           \
    } while (0)

static uint64_t res_outputs_after(const d2d_engine* e, uint64_t nx) {
    if (!e->fc.resamp) return nx;
    if (nx == 0) return 0;
    return (nx * (uint64_t)e->fc.resamp->L - 1) / (uint64_t)e->fc.resamp->Mdn + 1;
}

// key of the counter-based dither generator for one channel (DESIGN.md "dither")
static uint64_t rng_key64(uint64_t seed, uint32_t channel) {
    uint64_t z = (seed ^ ((uint64_t)channel * 0xD1B54A32D192ED03ull)) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static size_t sample_bytes_of(uint32_t bits) { return bits == 16 ? 2 : (bits == 32 ? 4 : 3); }

static int validate(const d2d_params& p, std::string& err) {
    if (p.channels < 1 || p.channels > 64) { err = "Invalid channel count"; return D2D_ERR_PARAM; }
    if (p.bit_depth != 16 && p.bit_depth != 20 && p.bit_depth != 24 && p.bit_depth != 32) {
        err = "Invalid bit depth; must be 16, 20, 24 or 32"; return D2D_ERR_PARAM;
    }
    if (p.dither != 'T' && p.dither != 'R' && p.dither != 'F' && p.dither != 'X' && p.dither != 'N') {
        err = "Invalid dither type; must be T, R, F, or X"; return D2D_ERR_PARAM;   // src/main.rs:176-180
    }
    if (p.fmt != D2D_FMT_INTERLEAVED && p.fmt != D2D_FMT_PLANAR) {
        err = "Invalid format; must be I (interleaved) or P (planar)"; return D2D_ERR_PARAM;  // src/main.rs:187-190
    }
    if (p.endianness != D2D_LSB_FIRST && p.endianness != D2D_MSB_FIRST) { err = "Invalid endianness"; return D2D_ERR_PARAM; }
    if (p.fmt == D2D_FMT_PLANAR && p.block_size == 0) { err = "Invalid block size"; return D2D_ERR_PARAM; }
    if (p.kernel > D2D_KERNEL_MFMA) { err = "Invalid kernel selector"; return D2D_ERR_PARAM; }
    if (!isfinite(p.level_db)) { err = "Invalid level"; return D2D_ERR_PARAM; }
    if (p.channel_first >= p.channels || p.channel_count > p.channels - p.channel_first) { err = "Invalid channel subset"; return D2D_ERR_PARAM; }
    return D2D_OK;
}

static void free_device(d2d_engine* e) {
    if (e->d_fir_tables) hipFree(e->d_fir_tables);
    if (e->d_resamp) hipFree(e->d_resamp);
    if (e->d_hist[0]) hipFree(e->d_hist[0]);
    if (e->d_hist[1]) hipFree(e->d_hist[1]);
    if (e->d_peak) hipFree(e->d_peak);
    if (e->d_scratch) hipFree(e->d_scratch);
    if (e->d_fir_tables_lo) hipFree(e->d_fir_tables_lo);
    if (e->d_fir_tables_m2) hipFree(e->d_fir_tables_m2);
    for (int i = 0; i < 2; ++i) if (e->d_ns[i]) hipFree(e->d_ns[i]);
    if (e->d_ns_dump) hipFree(e->d_ns_dump);
    if (e->d_ys) hipFree(e->d_ys);
    if (e->d_jobs) hipFree(e->d_jobs);
    if (e->h_jobs) hipHostFree(e->h_jobs);
    if (e->d_in) hipFree(e->d_in);
    if (e->d_planar) hipFree(e->d_planar);
    if (e->d_out) hipFree(e->d_out);
    for (int i = 0; i < JOB_SLOTS; ++i)
        if (e->job_ev[i]) hipEventDestroy(e->job_ev[i]);
    if (e->own_stream) hipStreamDestroy(e->own_stream);
    for (int i = 0; i < 3; ++i) if (e->hb_stream[i]) hipStreamDestroy(e->hb_stream[i]);
    for (int i = 0; i < 6; ++i) if (e->hb_ev[i]) hipEventDestroy(e->hb_ev[i]);
    for (int b = 0; b < 2; ++b) { if (e->hb_in[b]) hipFree(e->hb_in[b]); if (e->hb_out[b]) hipFree(e->hb_out[b]); }
    for (auto& pr : e->prof_pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto& pr : e->step_pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
}

static int reset_state(d2d_engine* e) {
    HIPCHK(e, hipSetDevice(e->p.device));
    const size_t hbytes = (size_t)e->nstreams * e->keep;
    const uint8_t idle = e->p.endianness == D2D_MSB_FIRST ? IDLE_BYTE : (uint8_t)0x96;  // 0x69 bit-reversed
    HIPCHK(e, hipMemset(e->d_hist[0], idle, hbytes));
    HIPCHK(e, hipMemset(e->d_hist[1], idle, hbytes));
    HIPCHK(e, hipMemset(e->d_peak, 0, sizeof(double) * e->nstreams));
    if (e->d_scratch) HIPCHK(e, hipMemset(e->d_scratch, 0, sizeof(int32_t) * e->scratch_stride * e->nstreams * (e->fine ? 2u : 1u)));
    for (int i = 0; i < 2; ++i) if (e->d_ns[i]) HIPCHK(e, hipMemset(e->d_ns[i], 0, sizeof(double) * 2 * e->nstreams));
    for (auto& f : e->files) f = FileState{};
    e->hist_cur = 0;
    return D2D_OK;
}

// the part of a FIR launch's arguments that is fixed when the engine is created
static void fir_args_static(const d2d_engine* e, FirArgs& a, bool lo_pass = false) {
    const d2d_filter_def& fd = lo_pass ? e->lo_def : *e->fc.fir;
    a.tables = lo_pass ? e->d_fir_tables_lo : e->d_fir_tables;
    a.Wb = (uint32_t)e->Wb;
    a.ntab = (uint32_t)e->lut.ntab; a.pad = (uint32_t)e->lut.pad; a.nq = (uint32_t)e->lut.nq;
    a.B = e->B; a.keep = e->keep;
    a.to_scratch = (e->cascade() || e->noise_shape || e->fine) ? 1u : 0u;
    a.ksteps = (uint32_t)e->mfma.ksteps;
    a.scale_bits = e->S + (e->taps32 ? 8 : 0);
    a.taps32 = e->taps32 ? 1u : 0u;
    a.in_channels = e->Cin;
    uint64_t sa = 0;
    for (int j = 0; j < e->N; ++j) { const int64_t q = tap_q(fd, j); sa += (uint64_t)(q < 0 ? -q : q); }
    a.sum_abs_q = sa;
    a.epi = e->epi;
    a.pipelined = (uint32_t)(lo_pass ? e->mfma_pipe_lo : e->mfma_pipe);
    a.mx_exact = mx_exact(fd) ? 1u : 0u;
    a.coop = e->coop ? 1u : 0u;
    a.il2 = e->il2 ? 1u : 0u;
    a.dbg_flags = e->p.debug_flags;
}

namespace d2d { thread_local const char* d2d_last_launched_kernel = nullptr; }

extern "C" {

const char* d2d_create_error(void) { return g_create_error.c_str(); }

int d2d_create(const d2d_params* params, uint32_t n_files, d2d_engine** out) {
    if (out) *out = nullptr;
    if (!params || !out) { g_create_error = "null argument"; return D2D_ERR_PARAM; }
    constexpr size_t legacy_size = offsetof(d2d_params, channel_first);           // ABI 1: no channel subset
    constexpr size_t abi3_size = offsetof(d2d_params, tap_bits);                  // ABI 2, 3: no tap grid (ABI 4 and 5 have today's size: reserved0 became debug_flags)
    if (params->struct_size != sizeof(d2d_params) && params->struct_size != legacy_size && params->struct_size != abi3_size) {
        g_create_error = "d2d_params.struct_size mismatch"; return D2D_ERR_PARAM;
    }
    if (n_files < 1 || n_files > 65535) { g_create_error = "Invalid file count"; return D2D_ERR_PARAM; }
    d2d_engine* e = new d2d_engine();
    memset(&e->p, 0, sizeof(e->p));
    memcpy(&e->p, params, params->struct_size);
    e->p.struct_size = sizeof(d2d_params);
    e->n_files = n_files;
    int rc = validate(e->p, g_create_error);
    if (rc == D2D_OK) rc = choose_filters(e->p, e->fc, g_create_error);
    if (rc != D2D_OK) { delete e; return rc; }
    const d2d_filter_def& f = *e->fc.fir;
    e->M = f.M; e->Mb = f.M / 8; e->N = f.ntaps; e->Wb = f.ntaps / 8; e->S = f.S;
    e->Cin = e->p.channels;
    e->C = e->p.channel_count ? e->p.channel_count : e->p.channels - e->p.channel_first;
    e->c0 = e->p.channel_first;
    e->B = e->p.fmt == D2D_FMT_INTERLEAVED ? 1u : e->p.block_size;   // README.md:9
    if (e->B == 1) {   // byte interleaved: mono is already planar; otherwise a planar copy is made per call
        e->deinterleave = e->Cin > 1;
        e->B = 4096;
    }
    e->nstreams = n_files * e->C;
    e->files.resize(n_files);
    e->epi.gain = pow(10.0, e->p.level_db / 20.0);
    e->epi.scale = e->p.bit_depth == 32 ? e->epi.gain : ldexp(e->epi.gain, (int)e->p.bit_depth - 1);
    e->epi.seed = e->p.seed;
    e->epi.bits = e->p.bit_depth;
    e->epi.dither = e->p.dither;
    if (e->p.dither == 'N') {
        if (e->p.bit_depth == 32) e->epi.dither = 'X';                     // float output: nothing to shape
        else e->noise_shape = true;
    }
    if (e->p.tap_bits != 0 && e->p.tap_bits != 24 && e->p.tap_bits != 32) { g_create_error = "Invalid tap grid; must be 24 or 32 bits"; delete e; return D2D_ERR_PARAM; }
    if (e->p.tap_bits == 32) {
        if (e->fc.resamp || e->noise_shape) { g_create_error = "32-bit taps serve the 44.1k-family rates with dither T, R, F or X"; delete e; return D2D_ERR_PARAM; }
        e->fine = true;
        e->lo_half.resize((size_t)f.ntaps / 2);
        for (int k = 0; k < f.ntaps / 2; ++k) e->lo_half[(size_t)k] = (int32_t)((int64_t)f.half32[k] - ((int64_t)f.half[k] << 8));
        e->lo_def = f; e->lo_def.half = e->lo_half.data(); e->lo_def.half32 = nullptr;
    }
    e->epi.sample_bytes = (uint32_t)sample_bytes_of(e->p.bit_depth);
    e->epi.channels = e->C;
    e->poly = e->fc.poly;
    e->lut = lut_layout(e->Mb, e->Wb);
    e->mfma = mfma_layout(e->M, e->N);
    uint32_t mfma_waves = 0;
    bool mfma_ok = mfma_supported(e->M, e->N) &&
                   mfma_smem_bytes(e->mfma, e->C, e->epi.sample_bytes, &mfma_waves) <= 160 * 1024;
    {   // the two-group kernel wherever its shape is compiled and four waves fit in LDS (D2D_MFMA_V1=1: the older one)
        const bool v1 = (e->p.debug_flags & D2D_DBG_MFMA_V1) != 0;
        uint32_t w2 = 0;
        if (!v1 && mfma2_supported(e->M, e->N) &&
            mfma2_smem_bytes(e->M, e->N, e->C, e->epi.sample_bytes, &w2) <= 160 * 1024 && w2 >= 4) {
            e->mfma_v2 = true; mfma_ok = true; mfma_waves = w2;
        }
        // M = 8 and 16: the two-group geometry only through the pipelined kernel (stereo 16/24-bit/float frames at 0 dB); every other
        // format of those rates stays on the one-group kernel
        if (!v1 && !e->mfma_v2 && (e->M < 32 || e->M == 128) && e->p.kernel != D2D_KERNEL_LUT) {
            FirArgs a{}; fir_args_static(e, a);
            if (mfma2_pipelined(a, e->M, e->N)) { e->mfma_v2 = true; mfma_ok = true; mfma_waves = 8; }
        }
    }
    // AUTO: the matrix-core kernel whenever a full 4-wave block fits in LDS (it works per channel pair,
    // so only an extremely long window can fail this; then the LUT kernel)
    e->kernel = e->p.kernel == D2D_KERNEL_AUTO ? (mfma_ok && mfma_waves >= 4 ? D2D_KERNEL_MFMA : D2D_KERNEL_LUT) : e->p.kernel;
    if (e->poly) {
        // the matrix-core form wherever a kernel is compiled for the table and its digit sums are exact in f32 (all six shipped tables)
        const bool px_ok = px_supported(*e->poly) && px_exact(*e->poly);
        e->kernel = e->p.kernel == D2D_KERNEL_AUTO ? (px_ok ? D2D_KERNEL_MFMA : D2D_KERNEL_LUT) : e->p.kernel;
        mfma_ok = px_ok;
        e->poly_plain = e->kernel == D2D_KERNEL_LUT;
        e->mfma_v2 = false;
    }
    if (e->kernel == D2D_KERNEL_MFMA && !mfma_ok) {
        g_create_error = "MFMA kernel does not support this configuration (decimation or LDS budget)"; delete e; return D2D_ERR_PARAM;
    }
    e->keep = (uint32_t)(e->Wb + e->Mb);
    // (direct polyphase: the oldest bit an output of the next call can need lies NP - D + M bits before the call's first byte)
    if (e->poly) e->keep = std::max<uint32_t>(e->keep, (uint32_t)((e->poly->NP - e->poly->D + e->M + 7) / 8 + 2));
    e->keep = (e->keep + 15u) & ~15u;

    // ---- device side: fail loudly when there is no GPU ----
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available (") + (he != hipSuccess ? hipGetErrorString(he) : "device count 0") +
                         "); this engine has no CPU path";
        delete e; return D2D_ERR_DEVICE;
    }
    if (e->p.device < 0 || e->p.device >= ndev) { g_create_error = "Invalid device ordinal"; delete e; return D2D_ERR_DEVICE; }
    auto bail = [&](hipError_t r, const char* what) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(r);
        free_device(e); delete e; return D2D_ERR_DEVICE;
    };
#define CK(call) do { hipError_t _r = (call); if (_r != hipSuccess) return bail(_r, #call); } while (0)
    CK(hipSetDevice(e->p.device));
    CK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    const bool msb = e->p.endianness == D2D_MSB_FIRST;
    if (e->poly) {
        if (e->poly_plain) {
            e->fir_table_bytes = (size_t)e->poly->Lp * e->poly->NP * sizeof(int32_t);
            CK(hipMalloc(&e->d_fir_tables, e->fir_table_bytes));
            CK(hipMemcpy(e->d_fir_tables, e->poly->q, e->fir_table_bytes, hipMemcpyHostToDevice));
        } else {
            // byte-interleaved stereo (DFF files, the CLI's default -f I): de-interleaved inside the kernel's staging, no planar copy (D2D_NO_COOP=1: the pre-pass)
            if (e->deinterleave && e->Cin == 2 && e->C == 2 && !(e->p.debug_flags & D2D_DBG_NO_COOP)) { e->il2 = true; e->deinterleave = false; e->B = 1; }
            const std::vector<int8_t> t = build_px_tables(*e->poly);
            e->fir_table_bytes = t.size();
            CK(hipMalloc(&e->d_fir_tables, e->fir_table_bytes));
            CK(hipMemcpy(e->d_fir_tables, t.data(), e->fir_table_bytes, hipMemcpyHostToDevice));
        }
    } else if (e->kernel == D2D_KERNEL_LUT) {
        std::vector<double> t = build_lut_tables(f, e->Mb, msb);
        e->fir_table_bytes = t.size() * sizeof(double);
        CK(hipMalloc(&e->d_fir_tables, e->fir_table_bytes));
        CK(hipMemcpy(e->d_fir_tables, t.data(), e->fir_table_bytes, hipMemcpyHostToDevice));
    } else {
        // 32-bit taps in ONE pass (round 4) where the fp6 kernel's seven-digit flavour is compiled for the table and its digit sums are exact: stereo frames,
        // any depth, dither and level (D2D_DBG_TAPS32_2PASS: the two scratch passes and the combining pass, which serve everything else)
        if (e->fine && e->mfma_v2 && e->C == 2 && e->Cin == 2 && mx_wide_supported(e->M / 8, e->N) && mx_wide_exact(f) &&
            !(e->p.debug_flags & (D2D_DBG_TAPS32_2PASS | D2D_DBG_NO_MX | D2D_DBG_NO_PIPE | D2D_DBG_MFMA_V1 | D2D_DBG_NO_GAINQ))) {
            e->fine = false; e->taps32 = true;
        }
        if (e->mfma_v2) { FirArgs a{}; fir_args_static(e, a); e->mfma_pipe = mfma2_pipelined(a, e->M, e->N); }
        // byte-interleaved 4- or 8-channel input into the scratch (48k family, noise shaping) through the fp6 kernel: no planar copy, the
        // kernel's staging de-interleaves (D2D_DBG_NO_COOP: the pre-pass)
        {
            const bool nocoop = (e->p.debug_flags & D2D_DBG_NO_COOP) != 0;
            if (e->deinterleave && e->mfma_pipe == 5 && (e->fc.resamp || e->noise_shape) && !e->fine && e->C == e->Cin && (e->Cin == 8 || e->Cin == 4) &&
                !nocoop) {
                e->coop = true; e->deinterleave = false; e->B = 1;
            }
            // byte-interleaved stereo (DFF files, the CLI's default -f I) into frames through a pipelined kernel (fp6: M = 32, 64; int8: M = 8, 16):
            // the same, inside one wave
            // (the scratch flavours too: stereo DFF input into the 48k cascade and the noise shaper; not the two passes of 32-bit taps)
            if (e->deinterleave && ((e->mfma_pipe == 5 || (e->mfma_pipe == 3 && e->M < 64)) && !e->fine) && e->Cin == 2 && e->C == 2 &&
                !nocoop) {
                e->il2 = true; e->deinterleave = false; e->B = 1;
            }
        }
        std::vector<int8_t> t = e->mfma_pipe == 5 ? build_mx_tables(f, msb, e->taps32)
                              : e->mfma_v2 ? build_mfma2_tables(f, msb, !e->mfma_pipe) : build_mfma_tables(f, e->mfma, msb);
        e->fir_table_bytes = t.size();
        CK(hipMalloc(&e->d_fir_tables, e->fir_table_bytes));
        CK(hipMemcpy(e->d_fir_tables, t.data(), e->fir_table_bytes, hipMemcpyHostToDevice));
        if (e->fine) {
            // the residual table goes through the same builders; which pipelined kernel serves it is decided on ITS digits
            if (e->mfma_v2) { FirArgs a{}; fir_args_static(e, a, true); e->mfma_pipe_lo = mfma2_pipelined(a, e->M, e->N); }
            const d2d_filter_def& fl = e->lo_def;
            std::vector<int8_t> tl = e->mfma_pipe_lo == 5 ? build_mx_tables(fl, msb)
                                   : e->mfma_v2 ? build_mfma2_tables(fl, msb, !e->mfma_pipe_lo) : build_mfma_tables(fl, e->mfma, msb);
            CK(hipMalloc(&e->d_fir_tables_lo, tl.size()));
            CK(hipMemcpy(e->d_fir_tables_lo, tl.data(), tl.size(), hipMemcpyHostToDevice));
        }
    }
    if (!e->poly && e->kernel == D2D_KERNEL_MFMA && e->Cin == 1 && e->C == 1 && !e->fine && !e->noise_shape && !e->fc.resamp &&
        !(e->p.debug_flags & (D2D_DBG_NO_PIPE | D2D_DBG_MFMA_V1 | D2D_DBG_NO_MX))) {
        // would the stereo conversion of this format run a pipelined kernel?  Then so can a mono stream, two halves of a call at a time
        FirArgs a2{}; fir_args_static(e, a2);
        a2.epi.channels = 2; a2.in_channels = 2;
        const int p2 = mfma2_pipelined(a2, e->M, e->N);
        if (p2 == 5 || p2 == 3) {
            const std::vector<int8_t> t2 = p2 == 5 ? build_mx_tables(f, msb) : build_mfma2_tables(f, msb, false);
            e->mono2_pipe = p2;
            CK(hipMalloc(&e->d_fir_tables_m2, t2.size()));
            CK(hipMemcpy(e->d_fir_tables_m2, t2.data(), t2.size(), hipMemcpyHostToDevice));
            e->mono2_ok = true;
        }
    }
    if (e->fine) {
        if (e->kernel == D2D_KERNEL_LUT) {
            std::vector<double> tl = build_lut_tables(e->lo_def, e->Mb, msb);
            CK(hipMalloc(&e->d_fir_tables_lo, tl.size() * sizeof(double)));
            CK(hipMemcpy(e->d_fir_tables_lo, tl.data(), tl.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        e->scratch_stride = 4096;
        CK(hipMalloc((void**)&e->d_scratch, sizeof(int32_t) * e->scratch_stride * e->nstreams * 2));
    }
    if (e->cascade()) {
        const std::vector<int8_t> rt = build_resamp2_table(*e->fc.resamp);
        e->resamp_bytes = rt.size();
        CK(hipMalloc((void**)&e->d_resamp, e->resamp_bytes));
        CK(hipMemcpy(e->d_resamp, rt.data(), e->resamp_bytes, hipMemcpyHostToDevice));
        e->xs_hist = (uint32_t)e->fc.resamp->P;
        e->scratch_stride = (size_t)e->fc.resamp->P + 4096;
        CK(hipMalloc((void**)&e->d_scratch, sizeof(int32_t) * e->scratch_stride * e->nstreams));
    }
    if (e->noise_shape) {
        if (!e->cascade()) {
            e->scratch_stride = 4096;
            CK(hipMalloc((void**)&e->d_scratch, sizeof(int32_t) * e->scratch_stride * e->nstreams));
        }
        for (int i = 0; i < 2; ++i) CK(hipMalloc((void**)&e->d_ns[i], sizeof(double) * 2 * e->nstreams));
        CK(hipMalloc((void**)&e->d_ns_dump, 1024));
    }
    const size_t hbytes = (size_t)e->nstreams * e->keep;
    CK(hipMalloc((void**)&e->d_hist[0], hbytes));
    CK(hipMalloc((void**)&e->d_hist[1], hbytes));
    CK(hipMalloc((void**)&e->d_peak, sizeof(double) * e->nstreams));
    const size_t njobs = (size_t)e->nstreams * (e->fine ? 2u : e->mono2_ok ? 3u : 1u);      // (32-bit taps: the second pass's jobs behind the first's; MONO2: the half-call jobs)
    CK(hipMalloc((void**)&e->d_jobs, sizeof(StreamJob) * njobs));
    CK(hipHostMalloc((void**)&e->h_jobs, sizeof(StreamJob) * njobs * JOB_SLOTS, hipHostMallocDefault));
    for (int i = 0; i < JOB_SLOTS; ++i) CK(hipEventCreateWithFlags(&e->job_ev[i], hipEventDisableTiming));
#undef CK
    rc = reset_state(e);
    if (rc != D2D_OK) { g_create_error = e->err; free_device(e); delete e; return rc; }
    *out = e;
    return D2D_OK;
}

void d2d_destroy(d2d_engine* e) {
    if (!e) return;
    hipSetDevice(e->p.device);
    hipDeviceSynchronize();
    free_device(e);
    delete e;
}

int d2d_reset(d2d_engine* e) {
    if (!e) return D2D_ERR_PARAM;
    hipSetDevice(e->p.device);
    hipDeviceSynchronize();
    return reset_state(e);
}

const char* d2d_last_error(const d2d_engine* e) { return e ? e->err.c_str() : "null engine"; }

size_t d2d_frame_bytes(const d2d_engine* e) { return e ? (size_t)e->epi.sample_bytes * e->C : 0; }

size_t d2d_next_frames(const d2d_engine* e, uint32_t file, size_t L) {
    if (!e || file >= e->n_files) return 0;
    const FileState& f = e->files[file];
    uint64_t nfir1 = (f.pos + L) / (uint64_t)e->Mb;
    return e->fc.resamp ? (size_t)(res_outputs_after(e, nfir1) - f.nres) : (size_t)(nfir1 - f.nfir);
}

static int grow_scratch(d2d_engine* e, size_t need_stride, hipStream_t s) {
    if (need_stride <= e->scratch_stride) return D2D_OK;
    size_t ns = (std::max(need_stride, e->scratch_stride * 2) + 3) & ~(size_t)3;
    int32_t* nb = nullptr;
    HIPCHK(e, hipMalloc((void**)&nb, sizeof(int32_t) * ns * e->nstreams * (e->fine ? 2u : 1u)));
    const size_t P = (size_t)e->xs_hist;
    if (P) HIPCHK(e, hipMemcpy2DAsync(nb, ns * sizeof(int32_t), e->d_scratch, e->scratch_stride * sizeof(int32_t),
                                      P * sizeof(int32_t), e->nstreams, hipMemcpyDeviceToDevice, s));
    HIPCHK(e, hipStreamSynchronize(s));
    HIPCHK(e, hipFree(e->d_scratch));
    e->d_scratch = nb; e->scratch_stride = ns;
    return D2D_OK;
}

int d2d_translate_batch_device(d2d_engine* e, d2d_file_io* io, uint32_t n_files, void* hip_stream) {
    if (!e) return D2D_ERR_PARAM;
    if (!io || n_files != e->n_files) return e->fail(D2D_ERR_PARAM, "file count does not match the engine");
    hipStream_t s = (hipStream_t)hip_stream;
    HIPCHK(e, hipSetDevice(e->p.device));
    const size_t fb = d2d_frame_bytes(e);
    const uint32_t C = e->C;
    // plan
    uint32_t max_nx = 0, max_frames = 0;
    std::vector<uint64_t> nfir1(n_files), nres1(n_files);
    for (uint32_t f = 0; f < n_files; ++f) {
        const FileState& st = e->files[f];
        const size_t L = io[f].bytes_per_channel;
        if (L >= (1ull << 31)) return e->fail(D2D_ERR_PARAM, "bytes_per_channel must be below 2 GiB per call");
        if (L && (!io[f].dsd || ((uintptr_t)io[f].dsd & 15))) return e->fail(D2D_ERR_PARAM, "dsd device pointer must be non-null and 16-byte aligned");
        nfir1[f] = (st.pos + L) / (uint64_t)e->Mb;
        nres1[f] = res_outputs_after(e, nfir1[f]);
        const uint64_t nx = nfir1[f] - st.nfir;
        const uint64_t frames = e->fc.resamp ? nres1[f] - st.nres : nx;
        if (frames * fb > io[f].pcm_capacity_bytes) return e->fail(D2D_ERR_CAPACITY, "pcm buffer too small");
        if (frames && (!io[f].pcm || ((uintptr_t)io[f].pcm & 15))) return e->fail(D2D_ERR_PARAM, "pcm device pointer must be non-null and 16-byte aligned");
        max_nx = std::max<uint32_t>(max_nx, (uint32_t)nx);
        max_frames = std::max<uint32_t>(max_frames, (uint32_t)frames);
        io[f].frames_out = (size_t)frames;
    }
    if (e->cascade() || e->noise_shape || e->fine) {
        int rc = grow_scratch(e, (size_t)e->xs_hist + (e->poly ? max_frames : max_nx), s);
        if (rc) return rc;
    }
    if (e->noise_shape && e->cascade() && (size_t)max_frames + 8 > e->ys_stride) {
        HIPCHK(e, hipStreamSynchronize(s));
        if (e->d_ys) HIPCHK(e, hipFree(e->d_ys));
        e->d_ys = nullptr; e->ys_stride = 0;
        const size_t ns = ((size_t)max_frames + 8 + 1023) & ~(size_t)1023;
        HIPCHK(e, hipMalloc((void**)&e->d_ys, sizeof(double) * ns * e->nstreams));
        e->ys_stride = ns;
    }
    uint32_t max_L = 0;
    for (uint32_t f = 0; f < n_files; ++f) max_L = std::max<uint32_t>(max_L, (uint32_t)io[f].bytes_per_channel);
    if (e->deinterleave) {
        const size_t need = (((size_t)max_L * e->Cin) + 4095) & ~(size_t)4095;
        if (need > e->planar_stride) {
            HIPCHK(e, hipStreamSynchronize(s));
            if (e->d_planar) HIPCHK(e, hipFree(e->d_planar));
            e->d_planar = nullptr; e->planar_stride = 0;
            HIPCHK(e, hipMalloc((void**)&e->d_planar, need * n_files));
            e->planar_stride = need;
        }
    }
    // job table: pinned slot -> device
    const int slot = e->job_slot;
    e->job_slot = (slot + 1) % JOB_SLOTS;
    if (e->job_ev_used[slot]) HIPCHK(e, hipEventSynchronize(e->job_ev[slot]));
    const size_t njobs = (size_t)e->nstreams * (e->fine ? 2u : e->mono2_ok ? 3u : 1u);
    StreamJob* hj = e->h_jobs + (size_t)slot * njobs;
    const int cur = e->hist_cur;
    for (uint32_t f = 0; f < n_files; ++f) {
        const FileState& st = e->files[f];
        for (uint32_t c = 0; c < C; ++c) {
            const uint32_t sidx = f * C + c;
            StreamJob& j = hj[sidx];
            j.in = e->deinterleave ? e->d_planar + (size_t)f * e->planar_stride : (const uint8_t*)io[f].dsd;
            j.in_raw = e->deinterleave ? (const uint8_t*)io[f].dsd : nullptr;
            j.hist = e->d_hist[cur] + (size_t)sidx * e->keep;
            j.hist_next = e->d_hist[cur ^ 1] + (size_t)sidx * e->keep;
            j.out = io[f].pcm;
            j.xs = e->d_scratch ? e->d_scratch + (size_t)sidx * e->scratch_stride + e->xs_hist : nullptr;
            j.peak = e->d_peak + sidx;
            j.L = io[f].bytes_per_channel;
            j.e0 = (int64_t)((st.nfir + 1) * (uint64_t)e->Mb) - (int64_t)st.pos;
            j.n0 = st.nfir;
            j.nout = (uint32_t)(nfir1[f] - st.nfir);
            if (e->poly) { j.e0 = (int64_t)st.pos; j.n0 = st.nres; j.nout = (uint32_t)(nres1[f] - st.nres); }   // (PxArgs::jobs)
            j.ch = e->c0 + c;
            j.och = c;
            j.m0 = st.nres;
            j.nres = e->fc.resamp ? (uint32_t)(nres1[f] - st.nres) : 0;
            const uint64_t i0 = e->fc.resamp ? st.nres : st.nfir;     // index the dither counter runs on
            const uint64_t k = rng_key64(e->p.seed, e->c0 + c);
            j.rng_kstep = (uint32_t)k | 1u;
            j.rng_key = (uint32_t)(k >> 32) + (uint32_t)(i0 >> 32) * j.rng_kstep;
            j.rng_lo0 = (uint32_t)i0;
        }
    }
    // MONO2: every file's call splits into two equal halves of whole outputs and whole 16-byte chunks, long enough to hold the second half's history
    bool mono2 = e->mono2_ok && max_nx > 0;
    for (uint32_t f = 0; f < n_files && mono2; ++f) {
        const StreamJob& j = hj[f];
        const uint64_t half = j.L / 2;
        mono2 = (j.L % 2 == 0) && (half % 16 == 0) && (half % (uint64_t)e->Mb == 0) && half >= e->keep && (j.nout % 2 == 0) &&
                (uint64_t)(uint32_t)j.n0 + j.nout <= 0xFFFFFFFFull;
    }
    if (mono2)
        for (uint32_t f = 0; f < n_files; ++f) {
            StreamJob ja = hj[f];
            const uint64_t half = ja.L / 2;
            ja.L = half; ja.nout /= 2; ja.ch = 0; ja.och = 0;
            StreamJob jb = ja;
            jb.ch = 1;
            jb.hist = ja.in + half - e->keep;                           // the end of the first half
            jb.out = (uint8_t*)ja.out + (size_t)ja.nout * fb;
            jb.rng_key = ja.rng_key + ja.nout;                          // (the kernel hashes (first half's index + key): the second half's indices lie nout further on)
            hj[e->nstreams + 2 * f] = ja; hj[e->nstreams + 2 * f + 1] = jb;
        }
    if (e->fine)
        for (uint32_t i = 0; i < e->nstreams; ++i) { hj[e->nstreams + i] = hj[i]; hj[e->nstreams + i].xs = hj[i].xs + (size_t)e->nstreams * e->scratch_stride; }
    HIPCHK(e, hipMemcpyAsync(e->d_jobs, hj, sizeof(StreamJob) * njobs, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipEventRecord(e->job_ev[slot], s));
    e->job_ev_used[slot] = true;

    std::pair<hipEvent_t, hipEvent_t>* ps = nullptr;
    if (e->profiling && max_nx) {
        if (e->step_used == e->step_pool.size()) {
            std::pair<hipEvent_t, hipEvent_t> n{};
            HIPCHK(e, hipEventCreate(&n.first));
            HIPCHK(e, hipEventCreate(&n.second));
            e->step_pool.push_back(n);
        }
        ps = &e->step_pool[e->step_used++];
        HIPCHK(e, hipEventRecord(ps->first, s));
    }
    if (e->deinterleave) HIPCHK(e, launch_deinterleave(e->d_jobs, n_files, e->Cin, C, max_L, s));

    FirArgs a{};
    a.jobs = e->d_jobs;
    fir_args_static(e, a);
    std::pair<hipEvent_t, hipEvent_t>* pe = nullptr;
    if (e->profiling && max_nx) {
        if (e->prof_used == e->prof_pool.size()) {
            std::pair<hipEvent_t, hipEvent_t> n{};
            HIPCHK(e, hipEventCreate(&n.first));
            HIPCHK(e, hipEventCreate(&n.second));
            e->prof_pool.push_back(n);
        }
        pe = &e->prof_pool[e->prof_used++];
        HIPCHK(e, hipEventRecord(pe->first, s));
    }
    if (e->poly) {
        PxArgs px{};
        px.jobs = e->d_jobs; px.tables = e->d_fir_tables;
        px.in_channels = e->Cin; px.B = e->B; px.keep = e->keep; px.msb = e->p.endianness == D2D_MSB_FIRST ? 1u : 0u;
        px.to_scratch = e->noise_shape ? 1u : 0u;
        px.il2 = e->il2 ? 1u : 0u;
        px.epi = e->epi;
        if (e->poly_plain) HIPCHK(e, launch_poly_plain(px, *e->poly, max_frames, e->nstreams, s));
        else HIPCHK(e, launch_fir_px(px, *e->poly, max_frames, n_files, s));
    } else if (mono2) {
        FirArgs a2 = a;
        a2.jobs = e->d_jobs + e->nstreams; a2.tables = e->d_fir_tables_m2;
        a2.epi.channels = 2; a2.in_channels = 2; a2.B = 0x40000000u;     // (one "block" per half: the gather path's layout rule then reads half c at c * half)
        a2.pipelined = (uint32_t)e->mono2_pipe; a2.mono2 = 1;
        HIPCHK(e, launch_fir_mfma2(a2, e->M, e->N, max_nx / 2, 2 * n_files, s));
    } else if (max_nx) {
        if (e->kernel == D2D_KERNEL_LUT) {
            const uint32_t per_tile = lut_outputs_per_tile(e->Mb);
            HIPCHK(e, launch_fir_lut(a, e->Mb, (max_nx + per_tile - 1) / per_tile, e->nstreams, s));
        } else {
            if (e->mfma_v2) HIPCHK(e, launch_fir_mfma2(a, e->M, e->N, max_nx, e->nstreams, s));
            else HIPCHK(e, launch_fir_mfma(a, e->mfma, max_nx, e->nstreams, s));
        }
    }
    if ((e->poly ? max_frames : max_nx) && d2d_last_launched_kernel) e->launched = d2d_last_launched_kernel;
    if (e->fine && max_nx) {
        // second pass: the residual taps, into the second half of the scratch
        FirArgs al{};
        al.jobs = e->d_jobs + e->nstreams;
        fir_args_static(e, al, true);
        if (e->kernel == D2D_KERNEL_LUT) {
            const uint32_t per_tile = lut_outputs_per_tile(e->Mb);
            HIPCHK(e, launch_fir_lut(al, e->Mb, (max_nx + per_tile - 1) / per_tile, e->nstreams, s));
        } else {
            if (e->mfma_v2) HIPCHK(e, launch_fir_mfma2(al, e->M, e->N, max_nx, e->nstreams, s));
            else HIPCHK(e, launch_fir_mfma(al, e->mfma, max_nx, e->nstreams, s));
        }
    }
    if (pe) HIPCHK(e, hipEventRecord(pe->second, s));
    if (e->fine && max_nx) {
        // the matrix-core kernels write 2 sum(q b) - 2^S, which is sum(q s) only for a table that sums to 2^S: the residual table sums to 0
        const int64_t lo_bias = e->kernel == D2D_KERNEL_LUT ? 0 : ((int64_t)1 << e->S);
        HIPCHK(e, launch_fine_combine(e->d_jobs, e->nstreams, max_nx, (size_t)e->nstreams * e->scratch_stride, lo_bias, e->S + 8, e->epi, s));
    }
    if (e->cascade()) {
        Rs2Args r{};
        r.jobs = e->d_jobs; r.tables = reinterpret_cast<const uint8_t*>(e->d_resamp);
        r.S = e->S; r.epi = e->epi;
        if (e->noise_shape) { r.ys = e->d_ys; r.ys_stride = (uint32_t)e->ys_stride; }
        HIPCHK(e, launch_resample2(r, *e->fc.resamp, max_frames, n_files, s));
        HIPCHK(e, launch_xhist(e->d_jobs, e->nstreams, (uint32_t)e->fc.resamp->P, s));
    }
    if (e->noise_shape) {
        NoiseShapeArgs ns{};
        ns.jobs = e->d_jobs; ns.state = e->d_ns[e->ns_cur]; ns.state_next = e->d_ns[e->ns_cur ^ 1]; ns.dump = e->d_ns_dump;
        ns.scale_bits = e->poly ? e->poly->S : e->S; ns.nstreams = e->nstreams; ns.max_nout = e->fc.resamp ? max_frames : max_nx; ns.epi = e->epi;
        if (e->cascade()) { ns.ys = e->d_ys; ns.ys_stride = (uint32_t)e->ys_stride; ns.res = 1; }
        {
            const bool noint = (e->p.debug_flags & D2D_DBG_NO_INTQ) != 0;
            uint64_t sa = 0;
            if (e->poly) {
                for (int ph = 0; ph < e->poly->Lp; ++ph) {
                    uint64_t sp = 0;
                    for (int j = 0; j < e->poly->NP; ++j) { const int64_t q = e->poly->q[(size_t)ph * e->poly->NP + j]; sp += (uint64_t)(q < 0 ? -q : q); }
                    sa = std::max(sa, sp);
                }
            } else { FirArgs fa{}; fir_args_static(e, fa); sa = fa.sum_abs_q; }
            ns.intq = (!noint && sa + (1ull << 24) < (1ull << 31)) ? 1u : 0u;
            ns.general = (e->p.debug_flags & D2D_DBG_NS_GENERAL) ? 1u : 0u;
        }
        // a stream whose call ends exactly on a segment boundary, or feeds nothing, writes no state: start the next buffer from the current one
        HIPCHK(e, hipMemcpyAsync(e->d_ns[e->ns_cur ^ 1], e->d_ns[e->ns_cur], sizeof(double) * 2 * e->nstreams, hipMemcpyDeviceToDevice, s));
        HIPCHK(e, launch_noise_shape(ns, s));
        e->ns_cur ^= 1;
    }
    HIPCHK(e, launch_history(e->d_jobs, e->nstreams, e->Cin, e->B, e->keep, s));
    if (ps) HIPCHK(e, hipEventRecord(ps->second, s));
    e->hist_cur = cur ^ 1;
    for (uint32_t f = 0; f < n_files; ++f) {
        FileState& st = e->files[f];
        st.pos += io[f].bytes_per_channel;
        st.nfir = nfir1[f];
        st.nres = nres1[f];
    }
    e->last_stream = s;
    return D2D_OK;
}

static int ensure_cap(d2d_engine* e, uint8_t** buf, size_t* cap, size_t need) {
    if (need <= *cap) return D2D_OK;
    size_t n = std::max(need, *cap * 2);
    n = (n + 4095) & ~(size_t)4095;
    HIPCHK(e, hipStreamSynchronize(e->own_stream));
    if (*buf) HIPCHK(e, hipFree(*buf));
    *buf = nullptr; *cap = 0;
    HIPCHK(e, hipMalloc((void**)buf, n));
    *cap = n;
    return D2D_OK;
}

// Host memory the GPU can address itself (hipHostMalloc, hipHostRegister; device memory passes too): the kernels then read the DSD
// and write the frames over the link with no staging copy at all -- one pass in which upload, conversion and download overlap by
// construction.  Measured on the bench batch (tools/zero_copy_probe.py): 52.9 ms per step = 89.6 GB/s over the link (reads alone
// 54.7 GB/s, writes alone 44.4), against 61.7 ms through the sliced three-stream pipeline below.  D2D_HOST_STAGED=1 keeps the pipeline.
static bool device_view(const void* p, void** dev) {
    if (!p) return false;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if ((at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) && at.devicePointer &&
        ((uintptr_t)at.devicePointer & 15) == 0) {               // (the batch entry point wants 16-byte aligned buffers)
        *dev = at.devicePointer;
        return true;
    }
    return false;
}
static bool host_staged_forced(const d2d_engine* e) { return (e->p.debug_flags & D2D_DBG_HOST_STAGED) != 0; }

int d2d_translate(d2d_engine* e, const uint8_t* dsd, size_t L, void* pcm, size_t cap, size_t* frames_out) {
    if (!e) return D2D_ERR_PARAM;
    if (frames_out) *frames_out = 0;
    if (e->n_files != 1) return e->fail(D2D_ERR_STATE, "d2d_translate needs a single-file engine");
    if (L && !dsd) return e->fail(D2D_ERR_PARAM, "null dsd pointer");
    HIPCHK(e, hipSetDevice(e->p.device));
    const size_t frames = d2d_next_frames(e, 0, L);
    const size_t out_bytes = frames * d2d_frame_bytes(e);
    if (out_bytes > cap) return e->fail(D2D_ERR_CAPACITY, "pcm buffer too small");
    if (out_bytes && !pcm) return e->fail(D2D_ERR_PARAM, "null pcm pointer");
    const size_t in_bytes = L * e->Cin;
    hipStream_t s = e->own_stream;
    void *vin = nullptr, *vout = nullptr;
    if (in_bytes && out_bytes && !host_staged_forced(e) && device_view(dsd, &vin) && device_view(pcm, &vout)) {
        d2d_file_io io{};
        io.dsd = vin; io.bytes_per_channel = L; io.pcm = vout; io.pcm_capacity_bytes = cap;
        int rc = d2d_translate_batch_device(e, &io, 1, s);
        if (rc) return rc;
        HIPCHK(e, hipStreamSynchronize(s));
        if (frames_out) *frames_out = io.frames_out;
        return D2D_OK;
    }
    int rc = ensure_cap(e, &e->d_in, &e->d_in_cap, std::max<size_t>(in_bytes, 16));
    if (rc) return rc;
    rc = ensure_cap(e, &e->d_out, &e->d_out_cap, std::max<size_t>(out_bytes, 16));
    if (rc) return rc;
    if (in_bytes) HIPCHK(e, hipMemcpyAsync(e->d_in, dsd, in_bytes, hipMemcpyHostToDevice, s));
    d2d_file_io io{};
    io.dsd = e->d_in; io.bytes_per_channel = L; io.pcm = e->d_out; io.pcm_capacity_bytes = e->d_out_cap;
    rc = d2d_translate_batch_device(e, &io, 1, s);
    if (rc) return rc;
    if (out_bytes) HIPCHK(e, hipMemcpyAsync(pcm, e->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    if (frames_out) *frames_out = io.frames_out;
    return D2D_OK;
}

int d2d_translate_batch_host(d2d_engine* e, d2d_file_io* io, uint32_t n_files, size_t slice) {
    if (!e) return D2D_ERR_PARAM;
    if (!io || n_files != e->n_files) return e->fail(D2D_ERR_PARAM, "file count does not match the engine");
    HIPCHK(e, hipSetDevice(e->p.device));
    const uint32_t C = e->Cin;                       // the uploads move whole input frames
    const size_t fb = d2d_frame_bytes(e);
    if (slice == 0) slice = 4u << 20;
    // whole planar blocks per slice, for ANY block size (-s takes any value, src/main.rs:75-78): every slice then
    // starts on a block-group boundary, which is what the kernels' addressing of a call assumes.  (No rounding to 16
    // bytes: the staging buffers are 256-byte aligned per file whatever the slice length.)
    if (e->B > 1) slice = std::max<size_t>(e->B, slice / e->B * e->B);
    size_t max_L = 0;
    for (uint32_t f = 0; f < n_files; ++f) {
        if (io[f].bytes_per_channel && !io[f].dsd) return e->fail(D2D_ERR_PARAM, "null dsd pointer");
        max_L = std::max(max_L, io[f].bytes_per_channel);
        io[f].frames_out = 0;
    }
    if (max_L == 0) return D2D_OK;
    if (!host_staged_forced(e)) {
        std::vector<d2d_file_io> vio(io, io + n_files);
        // one call for the whole batch: every file below the per-call limit, and the cascade's / noise shaper's / 32-bit taps' scratch for all
        // of it at once (4 B per stage-A sample, 8 more per output where the two combine; two int32 halves with 32-bit taps) within half of
        // the free device memory -- otherwise the sliced pipeline below, which needs one slice of scratch
        bool direct = max_L < (1ull << 31);
        if (direct && (e->cascade() || e->noise_shape || e->fine)) {
            size_t free_b = 0, total_b = 0;
            HIPCHK(e, hipMemGetInfo(&free_b, &total_b));
            const double per_stream = (double)max_L / (double)e->Mb * (e->cascade() && e->noise_shape ? 12.0 : e->fine ? 8.0 : 4.0);
            direct = per_stream * (double)e->nstreams < 0.5 * (double)free_b;
        }
        for (uint32_t f = 0; f < n_files && direct; ++f) {
            if (!io[f].bytes_per_channel) continue;
            void *vin = nullptr, *vout = nullptr;
            direct = device_view(io[f].dsd, &vin) && device_view(io[f].pcm, &vout);
            vio[f].dsd = vin; vio[f].pcm = vout;
        }
        if (direct) {
            hipStream_t s = e->own_stream;
            int rc = d2d_translate_batch_device(e, vio.data(), n_files, s);
            if (rc) { hipStreamSynchronize(s); return rc; }
            HIPCHK(e, hipStreamSynchronize(s));
            for (uint32_t f = 0; f < n_files; ++f) io[f].frames_out = vio[f].frames_out;
            return D2D_OK;
        }
    }
    // capacity of one slice's output: the engine never emits more than ceil(bytes*8/M)+1 frames per call
    const double ratio = e->fc.resamp ? (double)e->fc.resamp->L / (double)e->fc.resamp->Mdn / (double)e->M : 1.0 / (double)e->M;
    const size_t in_stride = (slice * C + 255) & ~(size_t)255;
    const size_t out_stride = (((size_t)((double)slice * 8.0 * ratio) + 4) * fb + 255) & ~(size_t)255;
    if (!e->hb_in[0] || e->hb_in_stride < in_stride || e->hb_out_stride < out_stride) {
        HIPCHK(e, hipDeviceSynchronize());
        for (int b = 0; b < 2; ++b) {
            if (e->hb_in[b]) hipFree(e->hb_in[b]);
            if (e->hb_out[b]) hipFree(e->hb_out[b]);
            e->hb_in[b] = e->hb_out[b] = nullptr;
            HIPCHK(e, hipMalloc((void**)&e->hb_in[b], in_stride * n_files));
            HIPCHK(e, hipMalloc((void**)&e->hb_out[b], out_stride * n_files));
        }
        e->hb_in_stride = in_stride; e->hb_out_stride = out_stride;
    }
    if (!e->hb_stream[0]) {
        for (int i = 0; i < 3; ++i) HIPCHK(e, hipStreamCreateWithFlags(&e->hb_stream[i], hipStreamNonBlocking));
        for (int i = 0; i < 6; ++i) HIPCHK(e, hipEventCreateWithFlags(&e->hb_ev[i], hipEventDisableTiming));
    }
    hipStream_t s_in = e->hb_stream[0], s_c = e->hb_stream[1], s_out = e->hb_stream[2];
    hipEvent_t* in_done = e->hb_ev; hipEvent_t* comp_done = e->hb_ev + 2; hipEvent_t* out_done = e->hb_ev + 4;
    std::vector<size_t> done(n_files, 0);
    std::vector<d2d_file_io> dio(n_files);
    // (equal slices: ramping them up and down -- slice/8, /4, /2, full ... -- to shorten the first upload and the last download was
    // measured and is slower, 64.2 against 61.7 ms per step: the step is bound by the per-copy cost of 2 x 64 transfers per slice,
    // not by fill and drain; the link alone moves the batch both ways in 47 ms, tools/link_probe.py)
    const size_t nslices = (max_L + slice - 1) / slice;
    for (size_t k = 0; k < nslices; ++k) {
        const int b = (int)(k & 1);
        const size_t slice_k = slice;
        // upload: the staging buffer is free once the conversion of slice k-2 has read it
        if (k >= 2) HIPCHK(e, hipStreamWaitEvent(s_in, comp_done[b], 0));
        for (uint32_t f = 0; f < n_files; ++f) {
            const size_t L = std::min(slice_k, io[f].bytes_per_channel - done[f]);
            dio[f].dsd = e->hb_in[b] + in_stride * f;
            dio[f].bytes_per_channel = L;
            dio[f].pcm = e->hb_out[b] + out_stride * f;
            dio[f].pcm_capacity_bytes = out_stride;
            if (L) HIPCHK(e, hipMemcpyAsync(e->hb_in[b] + in_stride * f, (const uint8_t*)io[f].dsd + done[f] * C, L * C, hipMemcpyHostToDevice, s_in));
            done[f] += L;
        }
        HIPCHK(e, hipEventRecord(in_done[b], s_in));
        // conversion: after its upload, and after the download of slice k-2 has drained the output buffer
        HIPCHK(e, hipStreamWaitEvent(s_c, in_done[b], 0));
        if (k >= 2) HIPCHK(e, hipStreamWaitEvent(s_c, out_done[b], 0));
        int rc = d2d_translate_batch_device(e, dio.data(), n_files, s_c);
        if (rc) { hipDeviceSynchronize(); return rc; }
        HIPCHK(e, hipEventRecord(comp_done[b], s_c));
        // download
        HIPCHK(e, hipStreamWaitEvent(s_out, comp_done[b], 0));
        for (uint32_t f = 0; f < n_files; ++f) {
            const size_t bytes = dio[f].frames_out * fb;
            if (!bytes) continue;
            if ((io[f].frames_out + dio[f].frames_out) * fb > io[f].pcm_capacity_bytes || !io[f].pcm) {
                hipDeviceSynchronize();
                return e->fail(D2D_ERR_CAPACITY, "pcm buffer too small");
            }
            HIPCHK(e, hipMemcpyAsync((uint8_t*)io[f].pcm + io[f].frames_out * fb, e->hb_out[b] + out_stride * f, bytes, hipMemcpyDeviceToHost, s_out));
            io[f].frames_out += dio[f].frames_out;
        }
        HIPCHK(e, hipEventRecord(out_done[b], s_out));
    }
    HIPCHK(e, hipStreamSynchronize(s_out));
    HIPCHK(e, hipStreamSynchronize(s_c));
    e->last_stream = s_c;
    return D2D_OK;
}

int d2d_peak(d2d_engine* e, uint32_t file, uint32_t channel, double* peak_out) {
    if (!e || !peak_out) return D2D_ERR_PARAM;
    if (file >= e->n_files || channel >= e->C) return e->fail(D2D_ERR_PARAM, "file/channel out of range");
    HIPCHK(e, hipSetDevice(e->p.device));
    HIPCHK(e, hipStreamSynchronize(e->last_stream));
    HIPCHK(e, hipMemcpy(peak_out, e->d_peak + (size_t)file * e->C + channel, sizeof(double), hipMemcpyDeviceToHost));
    return D2D_OK;
}

int d2d_peak_dbfs(d2d_engine* e, uint32_t file, float* dbfs_out) {
    if (!e || !dbfs_out) return D2D_ERR_PARAM;
    double m = 0.0;
    for (uint32_t c = 0; c < e->C; ++c) {
        double v = 0.0;
        int rc = d2d_peak(e, file, c, &v);
        if (rc) return rc;
        m = std::max(m, v);
    }
    *dbfs_out = (float)(20.0 * log10(m));   // may be -inf/NaN-free; the CLI skips NaN (dsd_levels main.rs:188)
    return D2D_OK;
}

int d2d_convert_stream(d2d_engine* e, d2d_read_fn read, void* ru, d2d_write_fn write, void* wu,
                       const volatile int* cancel, d2d_progress_fn progress, void* pu,
                       uint64_t total, size_t chunk) {
    if (!e) return D2D_ERR_PARAM;
    if (!read) return e->fail(D2D_ERR_PARAM, "null read callback");
    if (e->n_files != 1) return e->fail(D2D_ERR_STATE, "d2d_convert_stream needs a single-file engine");
    HIPCHK(e, hipSetDevice(e->p.device));
    if (chunk == 0) chunk = 1u << 22;
    if (e->B > 1) chunk = std::max<size_t>(e->B, chunk / e->B * e->B);   // whole planar blocks per read
    // Pinned staging, two deep: while the GPU uploads, converts and downloads chunk k, the host reads
    // chunk k+1 and writes chunk k-1 (the callbacks are the file and sink I/O).
    const size_t fb = d2d_frame_bytes(e);
    const double ratio = e->fc.resamp ? (double)e->fc.resamp->L / (double)e->fc.resamp->Mdn / (double)e->M : 1.0 / (double)e->M;
    const size_t in_cap = chunk * e->Cin, out_cap = ((size_t)((double)chunk * 8.0 * ratio) + 4) * fb;
    struct Pinned {
        uint8_t* in[2] = {nullptr, nullptr}; uint8_t* out[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        ~Pinned() { for (int b = 0; b < 2; ++b) { if (in[b]) hipHostFree(in[b]); if (out[b]) hipHostFree(out[b]); if (ev[b]) hipEventDestroy(ev[b]); } }
    } pin;
    for (int b = 0; b < 2; ++b) {
        HIPCHK(e, hipHostMalloc((void**)&pin.in[b], std::max<size_t>(in_cap, 16), hipHostMallocDefault));
        HIPCHK(e, hipHostMalloc((void**)&pin.out[b], std::max<size_t>(out_cap, 16), hipHostMallocDefault));
        HIPCHK(e, hipEventCreateWithFlags(&pin.ev[b], hipEventDisableTiming));
    }
    // the kernels read and write the pinned buffers themselves when the GPU can address them (they are hipHostMalloc'ed: it can)
    void* vin[2] = {nullptr, nullptr}; void* vout[2] = {nullptr, nullptr};
    const bool direct = !host_staged_forced(e) && device_view(pin.in[0], &vin[0]) && device_view(pin.in[1], &vin[1]) &&
                        device_view(pin.out[0], &vout[0]) && device_view(pin.out[1], &vout[1]);
    int rc = D2D_OK;
    if (!direct) {
        rc = ensure_cap(e, &e->d_in, &e->d_in_cap, std::max<size_t>(in_cap, 16));
        if (rc) return rc;
        rc = ensure_cap(e, &e->d_out, &e->d_out_cap, std::max<size_t>(out_cap, 16));
        if (rc) return rc;
    }
    hipStream_t s = e->own_stream;
    size_t pend_bytes[2] = {0, 0};
    uint64_t pend_in[2] = {0, 0};
    bool pending[2] = {false, false};
    uint64_t done = 0;
    auto retire = [&](int b) -> int {                // wait for chunk in buffer b, hand it to the sink
        if (!pending[b]) return D2D_OK;
        HIPCHK(e, hipEventSynchronize(pin.ev[b]));
        pending[b] = false;
        if (write && pend_bytes[b]) {
            if (write(wu, pin.out[b], pend_bytes[b]) != 0) return e->fail(D2D_ERR_IO, "write callback failed");
        }
        done += pend_in[b];
        if (progress && total) {
            float pct = (float)(100.0 * (double)done / (double)total);
            if (pct >= 100.0f) pct = 99.99f;   // exactly 100 is reserved for the end (src/main.rs:417-418)
            progress(pu, pct);
        }
        return D2D_OK;
    };
    auto drain = [&]() { hipStreamSynchronize(s); pending[0] = pending[1] = false; };
    for (int b = 0;; b ^= 1) {
        if (cancel && *cancel) { drain(); return e->fail(D2D_ERR_CANCELLED, "Conversion cancelled"); }
        rc = retire(b);                              // buffer b was used two chunks ago
        if (rc) { drain(); return rc; }
        long got = read(ru, pin.in[b], chunk);
        if (got < 0) { drain(); return e->fail(D2D_ERR_IO, "read callback failed"); }
        if (got == 0) { rc = retire(b ^ 1); if (rc) { drain(); return rc; } break; }
        const size_t L = (size_t)got;
        if (!direct) HIPCHK(e, hipMemcpyAsync(e->d_in, pin.in[b], L * e->Cin, hipMemcpyHostToDevice, s));
        d2d_file_io io{};
        io.dsd = direct ? vin[b] : e->d_in; io.bytes_per_channel = L;
        io.pcm = direct ? vout[b] : e->d_out; io.pcm_capacity_bytes = direct ? std::max<size_t>(out_cap, 16) : e->d_out_cap;
        rc = d2d_translate_batch_device(e, &io, 1, s);
        if (rc) { drain(); return rc; }
        pend_bytes[b] = io.frames_out * fb;
        pend_in[b] = (uint64_t)got;
        if (!direct && pend_bytes[b]) HIPCHK(e, hipMemcpyAsync(pin.out[b], e->d_out, pend_bytes[b], hipMemcpyDeviceToHost, s));
        HIPCHK(e, hipEventRecord(pin.ev[b], s));
        pending[b] = true;
        rc = retire(b ^ 1);                          // the previous chunk: its write overlaps this chunk's GPU work
        if (rc) { drain(); return rc; }
    }
    if (progress) progress(pu, 100.0f);
    return D2D_OK;
}

int d2d_profile_enable(d2d_engine* e, int on) {
    if (!e) return D2D_ERR_PARAM;
    e->profiling = on != 0;
    return D2D_OK;
}

int d2d_profile_read(d2d_engine* e, double* ms_total, uint64_t* launches) {
    if (!e) return D2D_ERR_PARAM;
    HIPCHK(e, hipSetDevice(e->p.device));
    double tot = 0.0;
    for (size_t i = 0; i < e->prof_used; ++i) {
        HIPCHK(e, hipEventSynchronize(e->prof_pool[i].second));
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->prof_pool[i].first, e->prof_pool[i].second));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = e->prof_used;
    e->prof_used = 0;
    return D2D_OK;
}

int d2d_profile_read_all(d2d_engine* e, double* fir_ms_total, double* step_ms_total, uint64_t* launches) {
    if (!e) return D2D_ERR_PARAM;
    HIPCHK(e, hipSetDevice(e->p.device));
    double tot = 0.0;
    for (size_t i = 0; i < e->step_used; ++i) {
        HIPCHK(e, hipEventSynchronize(e->step_pool[i].second));
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, e->step_pool[i].first, e->step_pool[i].second));
        tot += ms;
    }
    if (step_ms_total) *step_ms_total = tot;
    e->step_used = 0;
    return d2d_profile_read(e, fir_ms_total, launches);
}

size_t d2d_tables_bytes(const d2d_engine* e) {
    return e ? sizeof(TableBlobHeader) + ((e->fir_table_bytes + 15) & ~(size_t)15) + e->resamp_bytes : 0;
}

static TableBlobHeader make_header(const d2d_engine* e) {
    TableBlobHeader h{};
    h.magic = 0x54443244u; h.abi = D2D_ABI_VERSION; h.kernel = e->kernel; h.endianness = e->p.endianness;
    h.ntaps = (uint32_t)e->N; h.M = (uint32_t)e->M; h.scale_bits = (uint32_t)e->S; h.filter_type = (uint32_t)e->fc.fir->type;
    h.table_variant = e->poly ? (e->poly_plain ? 7u : 6u) : e->kernel == D2D_KERNEL_MFMA && e->mfma_v2 ? (e->mfma_pipe ? (uint32_t)e->mfma_pipe : 2u) : 0u;
    if (e->taps32) { h.table_variant = 8u; h.scale_bits = (uint32_t)(e->S + 8); }      // the seven-digit fragments of the 32-bit taps
    if (e->poly) { h.ntaps = (uint32_t)e->poly->NP; h.M = (uint32_t)e->poly->Mp; h.scale_bits = (uint32_t)e->poly->S; h.filter_type = (uint32_t)'P'; }
    h.fir_bytes = e->fir_table_bytes; h.resamp_bytes = e->resamp_bytes;
    return h;
}

int d2d_tables_export_device(d2d_engine* e, void* dst, size_t cap, void* hip_stream) {
    if (!e || !dst) return D2D_ERR_PARAM;
    if (e->fine) return e->fail(D2D_ERR_STATE, "an engine with 32-bit taps holds two tap tables; the blob format carries one");
    if (cap < d2d_tables_bytes(e)) return e->fail(D2D_ERR_CAPACITY, "table blob buffer too small");
    HIPCHK(e, hipSetDevice(e->p.device));
    hipStream_t s = (hipStream_t)hip_stream;
    TableBlobHeader h = make_header(e);
    uint8_t* p = (uint8_t*)dst;
    HIPCHK(e, hipMemcpyAsync(p, &h, sizeof(h), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipStreamSynchronize(s));   // h is a stack object
    p += sizeof(h);
    HIPCHK(e, hipMemcpyAsync(p, e->d_fir_tables, e->fir_table_bytes, hipMemcpyDeviceToDevice, s));
    p += (e->fir_table_bytes + 15) & ~(size_t)15;
    if (e->resamp_bytes) HIPCHK(e, hipMemcpyAsync(p, e->d_resamp, e->resamp_bytes, hipMemcpyDeviceToDevice, s));
    return D2D_OK;
}

int d2d_tables_import_device(d2d_engine* e, const void* src, size_t bytes, void* hip_stream) {
    if (!e || !src) return D2D_ERR_PARAM;
    if (e->fine) return e->fail(D2D_ERR_STATE, "an engine with 32-bit taps holds two tap tables; the blob format carries one");
    if (bytes < d2d_tables_bytes(e)) return e->fail(D2D_ERR_PARAM, "table blob too small");
    HIPCHK(e, hipSetDevice(e->p.device));
    hipStream_t s = (hipStream_t)hip_stream;
    TableBlobHeader h{}, want = make_header(e);
    HIPCHK(e, hipMemcpyAsync(&h, src, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    if (memcmp(&h, &want, sizeof(h)) != 0) return e->fail(D2D_ERR_PARAM, "table blob does not match this engine's configuration");
    const uint8_t* p = (const uint8_t*)src + sizeof(h);
    HIPCHK(e, hipMemcpyAsync(e->d_fir_tables, p, e->fir_table_bytes, hipMemcpyDeviceToDevice, s));
    p += (e->fir_table_bytes + 15) & ~(size_t)15;
    if (e->resamp_bytes) HIPCHK(e, hipMemcpyAsync(e->d_resamp, p, e->resamp_bytes, hipMemcpyDeviceToDevice, s));
    HIPCHK(e, hipStreamSynchronize(s));
    return D2D_OK;
}

int d2d_get_info(const d2d_engine* e, d2d_info* out) {
    if (!e || !out) return D2D_ERR_PARAM;
    memset(out, 0, sizeof(*out));
    out->decimation = (uint32_t)e->M; out->ntaps = (uint32_t)e->N; out->scale_bits = (uint32_t)e->S;
    if (e->fc.resamp) { out->resamp_L = e->fc.resamp->L; out->resamp_M = e->fc.resamp->Mdn; out->resamp_P = e->fc.resamp->P; }
    if (e->poly) {      // one polyphase filter: Lp outputs per Mp bits, NP taps per phase
        out->decimation = 0; out->ntaps = (uint32_t)e->poly->NP; out->scale_bits = (uint32_t)e->poly->S;
        out->resamp_L = (uint32_t)e->poly->Lp; out->resamp_M = (uint32_t)e->poly->Mp; out->resamp_P = (uint32_t)e->poly->NP;
    }
    out->kernel = e->kernel; out->abi_version = D2D_ABI_VERSION;
    strncpy(out->filter_name, e->poly ? e->poly->name : e->fc.fir->name, sizeof(out->filter_name) - 1);
    return D2D_OK;
}

// diagnostic, not part of the public header: per-phase wave-cycle sums of the MFMA kernel (D2D_DBG=16)
void d2d_debug_stamps(unsigned long long* out8) { hipDeviceSynchronize(); mfma_debug_stamps(out8); }
void d2d_debug_stamps2(unsigned long long* out8) { hipDeviceSynchronize(); mfma2_debug_stamps(out8); }
void d2d_debug_stamps3(unsigned long long* out8) { hipDeviceSynchronize(); mfma3_debug_stamps(out8); if (out8[3] == 0) mx_debug_stamps(out8); }

const char* d2d_kernel_name(const d2d_engine* e) {
    if (!e) return "";
    if (!e->launched.empty()) return e->launched.c_str();      // what the last call launched; before the first call: what the dispatch will choose
    if (e->poly) {
        if (e->poly_plain) return "d2d_poly_plain_kernel";
        d2d_engine* m = const_cast<d2d_engine*>(e);
        const bool intq = e->epi.gain == 1.0 && (e->epi.bits == 24 || e->epi.bits == 16) && e->epi.dither != 'F';
        const int kind = e->noise_shape ? 4 : !intq ? 3 : e->epi.dither == 'T' ? 1 : e->epi.dither == 'R' ? 2 : 0;
        m->kname = "d2d_fir_px_kernel<" + std::to_string(e->poly->Lp) + ", " + std::to_string(e->poly->Mp) + ", " + std::to_string(e->poly->NP) + ", " +
                   std::to_string(px_groups(*e->poly)) + ", " + std::to_string(kind) + ">";
        return m->kname.c_str();
    }
    if (e->kernel == D2D_KERNEL_MFMA && e->mfma_v2) {
        d2d_engine* m = const_cast<d2d_engine*>(e);
        if (e->mfma_pipe == 5) {
            const int kind = e->epi.dither == 'T' ? 1 : e->epi.dither == 'R' ? 2 : 0;
            const bool scr = e->fc.resamp || e->noise_shape;
            m->kname = "d2d_fir_mx_kernel<" + std::to_string(e->Mb) + ", " + std::to_string(e->N) + ", " + std::to_string(mx_groups(e->Mb)) + ", " +
                       std::to_string(scr || e->epi.sample_bytes == 4 ? 0 : kind) + ", " + std::to_string(scr ? 0u : e->epi.sample_bytes) + ">";
            return m->kname.c_str();
        }
        if (e->mfma_pipe) {
            const int kind = e->epi.dither == 'T' ? 1 : e->epi.dither == 'R' ? 2 : 0;
            const bool scr = e->fc.resamp || e->noise_shape;
            m->kname = "d2d_fir_mfma3_kernel<" + std::to_string(e->Mb) + ", " + std::to_string(mfma2_pairs(e->M, e->N)) + ", " +
                       std::to_string(e->mfma_pipe == 4 ? e->N : 0) + ", " + std::to_string(scr ? 0 : kind) + ", " + std::to_string(scr ? 0u : e->epi.sample_bytes) + ">";
            return m->kname.c_str();
        }
        m->kname = "d2d_fir_mfma2_kernel<" + std::to_string(e->Mb) + ", " + std::to_string(mfma2_pairs(e->M, e->N)) + ", " +
                   std::to_string(e->C == 1 ? 1 : 2) + ">";
        return m->kname.c_str();
    }
    return e->kernel == D2D_KERNEL_LUT ? lut_kernel_name(e->Mb) : mfma_kernel_name(e->mfma);
}

}  // extern "C"
