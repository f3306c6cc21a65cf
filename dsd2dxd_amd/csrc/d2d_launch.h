// d2d_launch.h -- host-callable launchers of the device kernels (defined in d2d_kernels*.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "d2d_internal.h"

namespace d2d {

size_t lut_smem_bytes(const FirArgs& a, int MB);
uint32_t lut_outputs_per_tile(int MB);
const char* lut_kernel_name(int MB);
hipError_t launch_fir_lut(const FirArgs& a, int MB, uint32_t max_tiles, uint32_t nstreams, hipStream_t s);
hipError_t launch_resample(const ResampArgs& a, uint32_t max_out, uint32_t nstreams, hipStream_t s);
hipError_t launch_deinterleave(const StreamJob* jobs, uint32_t nfiles, uint32_t C, uint32_t max_L, hipStream_t s);
hipError_t launch_history(const StreamJob* jobs, uint32_t nstreams, uint32_t C, uint32_t B, uint32_t keep, hipStream_t s);
hipError_t launch_xhist(const StreamJob* jobs, uint32_t nstreams, uint32_t P, hipStream_t s);

}  // namespace d2d
