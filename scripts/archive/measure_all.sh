#!/bin/bash
# usage: measure_all.sh <tag>  -- the side measurements quoted in DESIGN.md section 5, one log under gpurun_out/
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
L=$R/gpurun_out/measure_$tag.log
{
  echo "## all matrices, device-resident (scripts/bench_all.py)"
  python $R/scripts/bench_all.py --size 256 --dist uniform
  python $R/scripts/bench_all.py --size 256 --dist smooth
  python $R/scripts/bench_all.py --size 512 --dist uniform
  python $R/scripts/bench_all.py --size 512 --dist smooth
  echo "## config 3: filter stack -> binning -> GLCM+GLRLM (scripts/bench_filters.py)"
  python $R/scripts/bench_filters.py --size 256
  echo "## first order (scripts/bench_firstorder.py)"
  python $R/scripts/bench_firstorder.py 256
  echo "## voxel-based GLCM maps (scripts/bench_voxel.py)"
  python $R/scripts/bench_voxel.py
  echo "## fused voxel-based maps of the other texture classes (scripts/bench_voxel_texture.py)"
  python $R/scripts/bench_voxel_texture.py 256
  echo "## whole cases: device-resident vs host-array route (scripts/bench_cases.py)"
  python $R/scripts/bench_cases.py 256 3 smooth
  echo "## config 5: 64 cases through the batch front end (scripts/bench_batch.py)"
  python $R/scripts/bench_batch.py 256 64 1,2,4
} 2>&1 | grep -v "amdgpu.ids" > $L
cat $L
