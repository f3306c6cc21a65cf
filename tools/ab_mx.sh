#!/bin/bash
# A/B helper: build a variant library ab/<name>/libdsd2dxd_amd.so: the fp6 x fp4 kernel (d2d_kernels_mx.hip, E_M32 shape only) compiled with
# extra hipcc flags (e.g. -DD2D_MX_ABL=2 -DD2D_MX_STAMPS=1 -DD2D_MX_G4=3), linked with the tree's other objects (make first).
# Select it at run time with D2D_AMD_LIB=$PWD/ab/<name>/libdsd2dxd_amd.so (dsd2dxd_amd/_capi.py, development only).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/dsd2dxd_amd/csrc
mkdir -p $ROOT/ab/$NAME
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -I$CS -I$ROOT/filters"
/opt/rocm/bin/hipcc $FL -DD2D_MX_DEV=1 "$@" -c $CS/d2d_kernels_mx.hip -o $ROOT/ab/$NAME/d2d_kernels_mx.o
# the engine and the dispatcher see the geometry macros (groups per column) too
/opt/rocm/bin/hipcc $FL -x hip "$@" -c $CS/d2d_engine.cpp -o $ROOT/ab/$NAME/d2d_engine.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/$NAME/libdsd2dxd_amd.so $CS/d2d_kernels.o $CS/d2d_kernels_mfma.o $CS/d2d_kernels_mfma2.o $CS/d2d_kernels_mfma3.o $CS/d2d_kernels_mfma3b.o \
  $ROOT/ab/$NAME/d2d_kernels_mx.o $ROOT/ab/$NAME/d2d_engine.o $CS/host/dsd_reader.o $CS/host/pcm_sink.o $CS/host/id3_tag.o $CS/host/rdsd2pcm.o $CS/host/rdsd2pcm_c.o -lpthread
echo built ab/$NAME
