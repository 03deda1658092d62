#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *real* CompV reference (CPU, AVX2/SSE intrinsics path,
# COMPV_ASM=0 because yasm is absent) from the sources where they lie under /root/reference into
# oracle/_ref/.  Nothing is copied into the repo: oracle/_ref/ is git-ignored (but it travels to the
# GPU box with gpurun like any other built .so).
#
# The reference's own build system (cmake + yasm) is NOT run; the .cxx lists are read out of the
# module CMakeLists.txt files and compiled directly with g++ (recipe: SURVEY.md Appendix A).
#
# Outputs:
#   oracle/_ref/libcompv_ref.so      base + core + gpu modules of CompV (no GL/camera/drawing)
#   oracle/_ref/libcompv_refshim.so  oracle/ref_shim.cxx : flat C wrappers over the CompV C++ API
#
# Usage: oracle/build_ref.sh [REF_ROOT]   (default /root/reference); no-op when REF_ROOT is absent.
set -euo pipefail
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
OBJ="$OUT/obj"
if [ ! -d "$REF/base" ]; then
  echo "build_ref: $REF not present -> skipping (prebuilt oracle/_ref is used if it exists)"
  exit 0
fi
mkdir -p "$OBJ"
JOBS="${JOBS:-$(nproc)}"
CXXFLAGS="-include limits -std=c++11 -O2 -fPIC -flax-vector-conversions -w \
  -DCOMPV_ASM=0 -DCOMPV_BASE_EXPORTS -DCOMPV_CORE_EXPORTS -DCOMPV_GPU_EXPORTS \
  -I$REF/base/include -I$REF/core/include -I$REF/gpu/include -I$REF/thirdparties/include/common"

simd() { case "$1" in
  *_intrin_*sse2.cxx) echo "-msse2";; *_intrin_*ssse3.cxx) echo "-mssse3";;
  *_intrin_*sse41.cxx) echo "-msse4.1";; *_intrin_*sse42.cxx) echo "-msse4.2";;
  *_intrin_fma3_avx.cxx) echo "-mavx -mfma -D__FMA3__";; *_intrin_*avx.cxx) echo "-mavx";;
  *_intrin_*avx2.cxx) echo "-mavx2 -mfma -D__FMA3__";; esac; }

CMDS="$OBJ/cmds.txt"; : > "$CMDS"
for m in base core gpu; do
  grep -oE '^\s*[A-Za-z0-9_./-]+\.(cxx|cpp)\s*$' "$REF/$m/CMakeLists.txt" | tr -d ' \t' \
    | grep -v '/arm/\|android/\|ml/compv_base_ml_knn' | sort -u | while read -r f; do
      o="$OBJ/${m}_$(echo "$f" | tr / _).o"
      # incremental: skip objects newer than their source
      if [ ! -f "$o" ] || [ "$REF/$m/$f" -nt "$o" ]; then
        echo "g++ $CXXFLAGS $(simd "$f") -c $REF/$m/$f -o $o" >> "$CMDS"
      fi
    done
done
if [ -s "$CMDS" ]; then
  echo "build_ref: compiling $(wc -l < "$CMDS") reference objects with $JOBS jobs ..."
  # a handful of sources that need absent third-party headers may fail; they are not on the path
  xargs -P "$JOBS" -I{} sh -c '{} || echo "  (skipped) {}" | sed "s/.* -c //; s/ -o.*//" >&2' < "$CMDS" || true
fi
g++ -shared -o "$OUT/libcompv_ref.so" "$OBJ"/*.o -ldl -lpthread
g++ -include limits -std=c++11 -O2 -fPIC -w -mavx2 -msse4.1 -DCOMPV_ASM=0 \
  -I"$REF/base/include" -I"$REF/core/include" -I"$REF/gpu/include" \
  -shared -o "$OUT/libcompv_refshim.so" "$HERE/ref_shim.cxx" \
  -L"$OUT" -lcompv_ref -Wl,-rpath,'$ORIGIN' -ldl -lpthread
echo "build_ref: OK -> $OUT/libcompv_ref.so, $OUT/libcompv_refshim.so"
