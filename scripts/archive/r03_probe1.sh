#!/bin/bash
# round-3 probe 1 (no code change): does a pure pack kernel (0 LDS) co-reside with the sweep of the neighbouring volume?
# PRAD_NO_PACKROWS=1 = pack_levels_kernel + sweep_fw_kernel (12 angles) + sweep_fw_rows_kernel (x angle); lanes 1/2/3.
# Also: the fallback path at Ng = 64 / 128 (baseline for widening the fixed-window kernel).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p1
mkdir -p $O
cd $R
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"overlapped_kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"overlapped_pack_ms": [0-9.]*\|"serial_ms_per_step": [0-9.]*' | tr '\n' ' '; echo; }
for d in uniform smooth; do
  for L in 1 2 3; do
    echo "== default dist=$d lanes=$L"; PRAD_LANES=$L python bench.py $BA --dist $d 2>&1 | tail -1 | pick
    echo "== NO_PACKROWS dist=$d lanes=$L"; PRAD_NO_PACKROWS=1 PRAD_LANES=$L python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  done
done
for ng in 44 48 64 128; do
  for n in 256 512; do
    echo "== Ng=$ng size=$n"; python bench.py $BA --levels $ng --size $n 2>&1 | tail -1 | pick
  done
done
echo "== size 640 Ng=32 (rows > 512)"; python bench.py $BA --size 640 2>&1 | tail -1 | pick
