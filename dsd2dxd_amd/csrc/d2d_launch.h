// d2d_launch.h -- host-callable launchers of the device kernels (defined in d2d_kernels*.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>

#include <vector>

#include "../../filters/filter_tables.inc"
#include "d2d_internal.h"

namespace d2d {

// Per-kernel launch preparation.  hipFuncSetAttribute acts on the CURRENT device and engines of one
// process may sit on different GPUs and be driven from different threads (one Rdsd2Pcm per Rayon
// worker, src/main.rs:361-394), so the "already done" state is kept per device under a lock.
struct KernelPrep {
    static constexpr int MAX_DEV = 64;
    std::mutex mu;
    bool attr_done[MAX_DEV] = {};
    // occupancy cache of the persistent MFMA launch
    int blocks_per_cu[MAX_DEV] = {}, ncu[MAX_DEV] = {};
    size_t smem_seen[MAX_DEV] = {};
    size_t smem_used[MAX_DEV] = {};
    uint32_t nwaves_seen[MAX_DEV] = {};
    uint32_t nwaves_used[MAX_DEV] = {};

    hipError_t max_dynamic_lds(const void* fn, int bytes, int* dev_out = nullptr) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
        if (dev_out) *dev_out = dev;
        std::lock_guard<std::mutex> g(mu);
        if (attr_done[dev]) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) attr_done[dev] = true;
        return e;
    }
};

// The FIR kernel a launcher has just enqueued, spelled the way rocprofv3 prints it: set by every launch_fir_* at the point of the launch
// (per thread: engines are driven from one thread each), copied into the engine by d2d_translate_batch_device, so that
// d2d_kernel_name() reports what ran and not what the dispatch logic predicts.
extern thread_local const char* d2d_last_launched_kernel;
// (one string per (kernel family, instantiation): the family is part of the key -- a function-local static keyed by the int pack alone
// was shared by every family with the same pack, and the first caller's name won)
template <int... V>
inline const char* launched_name(const char* base) {
    static std::mutex mu;
    static std::map<std::string, std::string> names;
    std::lock_guard<std::mutex> g(mu);
    auto it = names.find(base);
    if (it == names.end()) {
        std::string s = std::string(base) + "<";
        const int v[] = {V...};
        for (size_t i = 0; i < sizeof...(V); ++i) s += (i ? ", " : "") + std::to_string(v[i]);
        it = names.emplace(base, s + ">").first;
    }
    return it->second.c_str();       // (map nodes do not move)
}

size_t lut_smem_bytes(const FirArgs& a, int MB);
uint32_t lut_outputs_per_tile(int MB);
const char* lut_kernel_name(int MB);
hipError_t launch_fir_lut(const FirArgs& a, int MB, uint32_t max_tiles, uint32_t nstreams, hipStream_t s);
hipError_t launch_resample2(Rs2Args& a, const d2d_resamp_def& r, uint32_t max_out, uint32_t nfiles, hipStream_t s);
std::vector<int8_t> build_resamp2_table(const d2d_resamp_def& r);
hipError_t launch_deinterleave(const StreamJob* jobs, uint32_t nfiles, uint32_t C, uint32_t streams_per_file, uint32_t max_L, hipStream_t s);
hipError_t launch_noise_shape(const NoiseShapeArgs& a, hipStream_t s);
// tap_bits = 32: frames from the two scratch halves, v = 256 v_hi + v_lo + lo_bias, y = v * 2^-sbits (d2d_kernels.hip)
hipError_t launch_fine_combine(const StreamJob* jobs, uint32_t nstreams, uint32_t max_nout, size_t lo_off, int64_t lo_bias, int sbits,
                               const Epilogue& epi, hipStream_t s);
hipError_t launch_history(const StreamJob* jobs, uint32_t nstreams, uint32_t C, uint32_t B, uint32_t keep, hipStream_t s);
hipError_t launch_xhist(const StreamJob* jobs, uint32_t nstreams, uint32_t P, hipStream_t s);

}  // namespace d2d
