#!/bin/bash
# A/B helper: build a variant library ab/<name>/libdsd2dxd_amd.so from an alternative d2d_kernels_mfma2.hip
# (default: the one in the tree) plus the tree's other objects.  Extra hipcc flags after the file name.
# Select it at run time with D2D_AMD_LIB=ab/<name>/libdsd2dxd_amd.so (dsd2dxd_amd/_capi.py, development only).
set -e
NAME=$1; SRC=${2:-dsd2dxd_amd/csrc/d2d_kernels_mfma2.hip}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/dsd2dxd_amd/csrc
mkdir -p $ROOT/ab/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
  -I$CS -I$ROOT/filters "$@" -c $SRC -o $ROOT/ab/$NAME/d2d_kernels_mfma2.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/$NAME/libdsd2dxd_amd.so $CS/d2d_kernels.o $CS/d2d_kernels_mfma.o $ROOT/ab/$NAME/d2d_kernels_mfma2.o \
  $CS/d2d_engine.o $CS/host/dsd_reader.o $CS/host/pcm_sink.o $CS/host/id3_tag.o $CS/host/rdsd2pcm.o $CS/host/rdsd2pcm_c.o -lpthread
echo built ab/$NAME
