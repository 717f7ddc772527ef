#!/bin/bash
# an experimental build of the library next to the product: tools/build_variant.sh NAME FILE.hip "-DFLAGS"  -> build_exp/NAME/libstringsext_amd.so
# (SX_LIB=build_exp/NAME/libstringsext_amd.so picks it; the other objects are the product's)
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../stringsext_amd/csrc"
mkdir -p ../../build_exp/$name
base=$(basename $file .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $file -o ../../build_exp/$name/$base.o
objs=$(ls build/*.o | grep -v "build/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_exp/$name/libstringsext_amd.so $objs ../../build_exp/$name/$base.o -lpthread
