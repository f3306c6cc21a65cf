#!/bin/bash
# A/B helper: ab/<name>/libdsd2dxd_amd.so = the tree's library with ONE translation unit recompiled with extra hipcc flags
# (make the tree first).  Select it at run time with D2D_AMD_LIB=$PWD/ab/<name>/libdsd2dxd_amd.so (dsd2dxd_amd/_capi.py,
# development only; bench.py names such a library in its line and cites no counter traffic for it).
#   tools/ab_build.sh <name> <unit> [flags...]     unit: mx | mfma3 | kernels
#     mx       d2d_kernels_mx.hip, E_M32 shape only (-DD2D_MX_DEV): -DD2D_MX_ABL=<mask> -DD2D_MX_STAMPS=1 -DD2D_MX_G4=<groups> -DD2D_MX_NOFLAT=1
#     mfma3    d2d_kernels_mfma3.hip (both parts):                   -DD2D_M3_ABL=<mask> -DD2D_M3_STAMPS=1
#     kernels  d2d_kernels.hip (LUT, resampler, de-interleave, noise shaping)
#     mxm      d2d_kernels_mx.hip, the whole-frame multichannel flavours of the E_M32 shape (D2D_MX_MPART=0): -DD2D_MX_ABL=<mask>
#     px       d2d_kernels_px.hip, every shape:                       -DD2D_PX_ABL=<mask> -DD2D_PX_THREADS=768
set -e
NAME=$1; UNIT=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/dsd2dxd_amd/csrc; O=$ROOT/ab/$NAME
mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -I$CS -I$ROOT/filters"
PX="$(echo $CS/d2d_kernels_px{0,1,2,3,4,5,6,7}.o)"
K=$CS/d2d_kernels.o; M3=$CS/d2d_kernels_mfma3.o; M3B=$CS/d2d_kernels_mfma3b.o; MX="$CS/d2d_kernels_mx.o $(echo $CS/d2d_kernels_mx{1,2,3,4,5,6,7}.o)"; MXG="$(echo $CS/d2d_kernels_mxg{0,1,2,3,4,5}.o)"; MXM="$(echo $CS/d2d_kernels_mxm{0,1,2,3,4,5,6}.o)"
case $UNIT in
  mx)      /opt/rocm/bin/hipcc $FL -DD2D_MX_DEV=1 "$@" -c $CS/d2d_kernels_mx.hip -o $O/d2d_kernels_mx.o; MX=$O/d2d_kernels_mx.o
           /opt/rocm/bin/hipcc $FL -DD2D_MX_DEV=1 -DD2D_MX_PART=99 -DD2D_MX_GPART=0 "$@" -c $CS/d2d_kernels_mx.hip -o $O/d2d_kernels_mxg0.o; MXG="$O/d2d_kernels_mxg0.o $(echo $CS/d2d_kernels_mxg{1,2,3,4,5}.o)" ;;
  mxm)     /opt/rocm/bin/hipcc $FL -DD2D_MX_PART=99 -DD2D_MX_MPART=0 "$@" -c $CS/d2d_kernels_mx.hip -o $O/d2d_kernels_mxm0.o; MXM="$O/d2d_kernels_mxm0.o $(echo $CS/d2d_kernels_mxm{1,2,3,4,5,6}.o)" ;;
  mfma3)   /opt/rocm/bin/hipcc $FL "$@" -c $CS/d2d_kernels_mfma3.hip -o $O/d2d_kernels_mfma3.o & /opt/rocm/bin/hipcc $FL -DD2D_M3_PART=1 "$@" -c $CS/d2d_kernels_mfma3.hip -o $O/d2d_kernels_mfma3b.o; wait
           M3=$O/d2d_kernels_mfma3.o; M3B=$O/d2d_kernels_mfma3b.o ;;
  px)      for i in 0 1 2 3 4 5 6 7; do /opt/rocm/bin/hipcc $FL -DD2D_PX_PART=$i "$@" -c $CS/d2d_kernels_px.hip -o $O/d2d_kernels_px$i.o & done; wait
           PX="$(echo $O/d2d_kernels_px{0,1,2,3,4,5,6,7}.o)" ;;
  kernels) /opt/rocm/bin/hipcc $FL "$@" -c $CS/d2d_kernels.hip -o $O/d2d_kernels.o; K=$O/d2d_kernels.o ;;
  *) echo "unit: mx | mxm | mfma3 | kernels | px"; exit 2 ;;
esac
# (the engine sees the geometry macros too: groups per column name the kernel)
/opt/rocm/bin/hipcc $FL -x hip "$@" -c $CS/d2d_engine.cpp -o $O/d2d_engine.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libdsd2dxd_amd.so $K $CS/d2d_kernels_rs.o $CS/d2d_kernels_mfma.o $CS/d2d_kernels_mfma2.o $M3 $M3B $MX $MXG $MXM $PX $O/d2d_engine.o \
  $CS/host/dsd_reader.o $CS/host/pcm_sink.o $CS/host/id3_tag.o $CS/host/rdsd2pcm.o $CS/host/rdsd2pcm_c.o -lpthread
echo built ab/$NAME
