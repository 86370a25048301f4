#!/bin/bash
# round-3 probe 3: rows role inside sweep_fw_kernel + pack side job prototype
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fw.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"overlapped_kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"overlapped_pack_ms": [0-9.]*\|"serial_ms_per_step": [0-9.]*' | tr '\n' ' '; echo; }
for d in uniform smooth; do
  for L in 1 2; do
    echo "== rowsrole dist=$d lanes=$L"; PRAD_LANES=$L python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  done
  for w in 0.5 0.6 0.8 1.0; do
    echo "== rowsrole weight=$w dist=$d lanes=1"; PRAD_FW_ROWS_WEIGHT=$w PRAD_LANES=1 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  done
  for ev in 0 1 2 3 4; do
    e=""; [ $ev != 0 ] && e="PRAD_PACK_EVERY=$ev"
    echo "== PROTO fused pack every=$ev dist=$d lanes=1"; env $e PRAD_FUSEPACK_PROTO=1 PRAD_LANES=1 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  done
done
echo "== 256 rowsrole"; PRAD_LANES=1 python bench.py $BA --size 256 2>&1 | tail -1 | pick
echo "== 256 PROTO"; PRAD_FUSEPACK_PROTO=1 PRAD_LANES=1 python bench.py $BA --size 256 2>&1 | tail -1 | pick
