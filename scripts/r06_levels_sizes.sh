#!/bin/bash
# round 6: is the two-table (64-level) walk bound by the size of its 16-bit level volume?  The 64- / 32-level step ratio at three
# volume sizes: 16-bit volumes of 113, 180 and 268 MB against the 256 MB Infinity Cache.
for sz in 384 448 512; do for lv in 32 64; do
  python bench.py --no-cpu-baseline --no-modes --no-host-boundary --size $sz --levels $lv 2>/dev/null | grep '^{"metric"' > /tmp/l.json
  python - <<PY
import json
d = json.load(open("/tmp/l.json"))
print("size $sz levels $lv: %.0f Mvoxels/s  %.4f ms/step  kernel %.4f  rows %.4f  variant %s" % (d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms") or 0, d["roofline"].get("rows_ms") or 0, d["config"].get("variant")))
PY
done; done
