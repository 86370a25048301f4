#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*' | tr '\n' ' '; echo; }
for d in uniform smooth; do
  for w in 1.6 2.0 2.4 2.8 3.2; do
    echo "== rowsrole weight=$w dist=$d lanes=1"; PRAD_FW_ROWS_WEIGHT=$w PRAD_LANES=1 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  done
  for w in 2.0 2.4 2.8; do
  for ev in 0 2; do
    e=""; [ $ev != 0 ] && e="PRAD_PACK_EVERY=$ev"
    echo "== PROTO weight=$w every=$ev dist=$d lanes=1"; env $e PRAD_FW_ROWS_WEIGHT=$w PRAD_FUSEPACK_PROTO=1 PRAD_LANES=1 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  done; done
done
