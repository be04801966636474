#!/bin/bash
# builds tools/eigh/eigh_probe.bin (development harness of csrc/kernels_eigh.hpp)
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize "$@" -o eigh_probe.bin eigh_probe.hip
