#!/bin/bash
# experiment harness of the KHT host linker (not part of the product)
set -e
cd "$(dirname "$0")"
HIPCC=/opt/rocm/bin/hipcc
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip -c ../../compv_amd/csrc/kht_host.cpp -o kht_host.o
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -x hip -c link_bench.cpp -o link_bench.o
$HIPCC link_bench.o kht_host.o -o link_bench
