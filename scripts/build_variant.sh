#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG ..."   -> build_variants/lib_NAME.so (only prad_api.hip carries the sweep kernels)
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build_variants/obj_$1
cd $R/pyradiomics_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c prad_api.hip -o $R/build_variants/obj_$1/prad_api.o
hipcc --offload-arch=gfx950 -shared -fPIC $R/build_variants/obj_$1/prad_api.o .obj/prad_firstorder.o .obj/prad_features.o .obj/prad_resample.o .obj/prad_filters.o -o $R/build_variants/lib_$1.so
echo built $R/build_variants/lib_$1.so
