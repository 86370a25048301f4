#!/bin/bash
# round-3 probe 2: (A) two sweep workgroups per CU (8 waves/SIMD, half-size tables); (B) s_setprio on the sweep waves with a
# co-resident pure pack kernel.
R=$GRAFT_REPO_ROOT
cd $R
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"overlapped_kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"overlapped_pack_ms": [0-9.]*\|"serial_ms_per_step": [0-9.]*' | tr '\n' ' '; echo; }
for d in uniform smooth; do
  echo "== base dist=$d lanes=1"; PRAD_LANES=1 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  for kb in 72 50; do for blk in 42 63; do for pw in 3 6; do
    [ $kb = 72 ] && [ $blk = 63 ] && continue
    [ $kb = 50 ] && [ $blk = 42 ] && continue
    echo "== FW_BUDGET_KB=$kb FW_BLOCKS=$blk PER_WAVE=$pw dist=$d lanes=1"
    PRAD_FW_BUDGET_KB=$kb PRAD_FW_BLOCKS=$blk PRAD_FW_PER_WAVE=$pw PRAD_LANES=1 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  done; done; done
  echo "== prio3 NO_PACKROWS dist=$d lanes=2"; PRAD_LIB=$R/build_variants/lib_prio3.so PRAD_NO_PACKROWS=1 PRAD_LANES=2 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  echo "== prio3 default dist=$d lanes=2"; PRAD_LIB=$R/build_variants/lib_prio3.so PRAD_LANES=2 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
done
