#!/bin/bash
# as tools/pmc_kernel.sh with a variant library: tools/pmc_kernel_lib.sh <lib name under ab/> <tag> <kernel substring> "<counters>" [bench args]
LIB=$1; shift
export D2D_AMD_LIB=$PWD/ab/$LIB/libdsd2dxd_amd.so
exec bash tools/pmc_kernel.sh "$@"
