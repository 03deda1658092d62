#!/bin/bash
# Builds tools/canny_lab/canny_lab from the PRODUCT kernel file (compv_amd/csrc/canny_swar_kernels.hip) compiled once per variant, each into its own
# namespace (-DCOMPVHIP_SWAR_NS) with its own SWAR_* switches; no switch = the shipped kernel.  Experiment harness, not part of the product.
#   VARIANTS="name:flags name:flags ..."  (flags joined with '+', e.g. "ship: legacy240:-DSWAR_LEGACY240 rows64:-DSWAR_ROWS=64")
set -e
cd "$(dirname "$0")"
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-inline-asm -Wno-unused-function -I../../include"
VARIANTS=${VARIANTS:-"ship:"}
mkdir -p build
decls=""; table=""; objs=""
for v in $VARIANTS; do
	name=${v%%:*}; fl=$(echo "${v#*:}" | tr '+' ' ')
	src=${SRC_OVERRIDE:-../../compv_amd/csrc/canny_swar_kernels.hip}
	case "$fl" in *SRC=*) src=$(echo "$fl" | sed 's/.*SRC=\([^ ]*\).*/\1/'); fl=$(echo "$fl" | sed 's/SRC=[^ ]*//');; esac
	$HIPCC $FLAGS -DCOMPVHIP_SWAR_NS=lab_$name $fl -c $src -o build/$name.o &
	decls="$decls DECL($name)"; table="$table { \"$name\", lab_$name::launch_canny_tiles_swar },"; objs="$objs build/$name.o"
done
wait
$HIPCC $FLAGS -x hip "-DLAB_DECLS=$decls" "-DLAB_TABLE=$table" -c lab_main.cpp -o build/lab_main.o
$HIPCC --offload-arch=gfx950 build/lab_main.o $objs -o canny_lab
echo "built canny_lab: $VARIANTS"
