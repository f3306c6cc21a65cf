#!/bin/bash
# A/B helper: build a variant library ab/<name>/libdsd2dxd_amd.so: the tree's pipelined kernel (d2d_kernels_mfma3.hip) compiled with
# extra hipcc flags (e.g. -DD2D_M3_ABL=2 -DD2D_M3_STAMPS=1), linked with the tree's other objects (make first).
# Select it at run time with D2D_AMD_LIB=$PWD/ab/<name>/libdsd2dxd_amd.so (dsd2dxd_amd/_capi.py, development only).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/dsd2dxd_amd/csrc
mkdir -p $ROOT/ab/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -I$CS -I$ROOT/filters "$@" -c $CS/d2d_kernels_mfma3.hip -o $ROOT/ab/$NAME/d2d_kernels_mfma3.o
M2O=$CS/d2d_kernels_mfma2.o
if [ -n "$AB_M2" ]; then    # AB_M2=1: the two-group kernel's file too (it builds the tap tables both kernels read)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -I$CS -I$ROOT/filters -DD2D_M2_DEV "$@" -c $CS/d2d_kernels_mfma2.hip -o $ROOT/ab/$NAME/d2d_kernels_mfma2.o
  M2O=$ROOT/ab/$NAME/d2d_kernels_mfma2.o
fi
P1O=$CS/d2d_kernels_mfma3b.o
if [ -n "$AB_P1" ]; then    # AB_P1=1: the second half of the pipelined kernel's instantiations too (16-bit, float, scratch)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -I$CS -I$ROOT/filters -DD2D_M3_PART=1 "$@" -c $CS/d2d_kernels_mfma3.hip -o $ROOT/ab/$NAME/d2d_kernels_mfma3b.o
  P1O=$ROOT/ab/$NAME/d2d_kernels_mfma3b.o
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/$NAME/libdsd2dxd_amd.so $CS/d2d_kernels.o $CS/d2d_kernels_mfma.o $M2O $ROOT/ab/$NAME/d2d_kernels_mfma3.o $P1O \
  $CS/d2d_engine.o $CS/host/dsd_reader.o $CS/host/pcm_sink.o $CS/host/id3_tag.o $CS/host/rdsd2pcm.o $CS/host/rdsd2pcm_c.o -lpthread
echo built ab/$NAME
