#!/bin/bash
# A/B helper: ab/<name>/libdsd2dxd_amd.so with d2d_kernels.hip (LUT, resampler, de-interleave, noise shaping) compiled with extra flags
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/dsd2dxd_amd/csrc
mkdir -p $ROOT/ab/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -I$CS -I$ROOT/filters "$@" -c $CS/d2d_kernels.hip -o $ROOT/ab/$NAME/d2d_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/$NAME/libdsd2dxd_amd.so $ROOT/ab/$NAME/d2d_kernels.o $CS/d2d_kernels_mfma.o $CS/d2d_kernels_mfma2.o $CS/d2d_kernels_mfma3.o $CS/d2d_kernels_mfma3b.o \
  $CS/d2d_engine.o $CS/host/dsd_reader.o $CS/host/pcm_sink.o $CS/host/id3_tag.o $CS/host/rdsd2pcm.o $CS/host/rdsd2pcm_c.o -lpthread
echo built ab/$NAME
