#!/usr/bin/env bash
# Builds the real-CompV plugin and its headless driver against a CompV checkout (COMPV_ROOT, default /root/reference)
# and the CompV library built by oracle/build_ref.sh.  Outputs go to integration/_build/ (git-ignored; the binaries
# travel to the GPU box with the snapshot).  No-op when COMPV_ROOT is absent.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${COMPV_ROOT:-/root/reference}"
OUT="$HERE/_build"
mkdir -p "$OUT"
# the C++ multi-GPU batch driver: C ABI + RCCL only, no CompV checkout needed
ROCM="${ROCM_PATH:-/opt/rocm}"
if [ -f "$ROCM/lib/librccl.so" ] && [ -f "$ROCM/lib/libamdhip64.so" ] && [ -f "$ROOT/compv_amd/lib/libcompv_hip.so" ]; then
  g++ -std=c++14 -O2 -w -D__HIP_PLATFORM_AMD__ -I"$ROCM/include" -o "$OUT/multi_gpu_batch" "$HERE/multi_gpu_batch.cxx" \
    -L"$ROOT/compv_amd/lib" -lcompv_hip -L"$ROCM/lib" -lamdhip64 -lrccl \
    -Wl,-rpath,'$ORIGIN/../../compv_amd/lib' -Wl,-rpath,"$ROCM/lib" -lpthread
  echo "integration/build.sh: OK -> $OUT/multi_gpu_batch"
else
  echo "integration/build.sh: no ROCm/RCCL or libcompv_hip.so not built yet -> multi_gpu_batch skipped"
fi
if [ ! -d "$REF/base/include" ] || [ ! -f "$ROOT/oracle/_ref/libcompv_ref.so" ]; then
  echo "integration/build.sh: no CompV checkout / library -> skipping"; exit 0
fi
mkdir -p "$OUT"
INC="-I$REF/base/include -I$REF/core/include -I$REF/gpu/include"
FLAGS="-include limits -std=c++11 -O2 -fPIC -w -DCOMPV_ASM=0"
g++ $FLAGS $INC -shared -o "$OUT/libcompv_hip_plugin.so" "$HERE/compv_hip_plugin.cxx" \
  -L"$ROOT/compv_amd/lib" -lcompv_hip -L"$ROOT/oracle/_ref" -lcompv_ref \
  -Wl,-rpath,'$ORIGIN/../../compv_amd/lib' -Wl,-rpath,'$ORIGIN/../../oracle/_ref'
g++ $FLAGS $INC -o "$OUT/headless_samples" "$HERE/headless_samples.cxx" \
  -L"$OUT" -lcompv_hip_plugin -L"$ROOT/oracle/_ref" -lcompv_ref -L"$ROOT/compv_amd/lib" -lcompv_hip \
  -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../compv_amd/lib' -Wl,-rpath,'$ORIGIN/../../oracle/_ref' -ldl -lpthread
echo "integration/build.sh: OK -> $OUT/libcompv_hip_plugin.so, $OUT/headless_samples"
