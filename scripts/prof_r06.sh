#!/bin/bash
# round-6 evidence pack, ONE run from the final source: rocprofv3 kernel stats of the bench command for
# 32 / 64 levels, uniform / smooth (pipeline mode: every launch of a step on ONE stream, so the per-dispatch durations ARE kernel
# speeds), the other kernels the bench's modes report (filters, voxel maps, pairs tier, one 256^3 case), the PMC passes (separate,
# --kernel-trace only) and the counter file bench.py reads (profiles/r06_counters.json).  Everything lands in gpurun_out/r06/;
# copy kernel_stats.md / pmc.md / counters.json to profiles/r06_*.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-modes --no-host-boundary"
{
echo "# r06 -- rocprofv3 --kernel-trace --stats of bench.py (scripts/prof_r06.sh), final source of round 6 (HEAD ${PRAD_HEAD:-?} + working tree)"
echo
echo "Pipeline mode (default): the launches of a step sit on one stream; '<..., true>' of sweep_fw_kernel / sweep_fw2_kernel is the launch"
echo "that walks the 12 line angles of volume N-1 AND packs volume N; '<..., false>' are the synchronous calls and the flush."
echo "bench.py times K steps right after its W warm-up steps first (cold_ms_per_step), then runs 80 untimed device warm-up steps and"
echo "times K steps again (ms_per_step): the all-calls averages below include the process's first launches, which are ~7 % slower than"
echo "the steady state; kernel_ms in the line under each heading is the HIP-event figure of the 20 warm timed launches of the same"
echo "profiled run, and the per-dispatch lists give the rocprofv3 durations of exactly those launches."
echo
} > $O/kernel_stats.md
for lv in 32 64; do
  for d in uniform smooth; do
    rocprofv3 --kernel-trace --stats -d $O/stats_${lv}_$d -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --dist $d --levels $lv > $O/stats_${lv}_$d.log 2>&1
    { echo "## bench.py --steps 20 --warmup 3 --levels $lv --dist $d"; grep '^{"metric"' $O/stats_${lv}_$d.log | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"cold_ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"finalize_ms": [0-9.]*' | tr '\n' ' '; echo; echo;
      python $R/scripts/rocpd_stats.py $O/stats_${lv}_$d/s_results.db | grep -E "prad|rocclr|kernel \||---"; echo;
      if [ $d = uniform ]; then
        echo "per-dispatch durations of the walk + pack launch in launch order (us): the first 8 of the process, then the last 33 = the 3 warm-up + 20 timed + 10 instrumented steps of the WARM loop";
        echo; echo '```';
        python - <<PY
import sqlite3
db = sqlite3.connect("$O/stats_${lv}_$d/s_results.db")
rows = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like '%sweep_fw%kernel%' and name like '%true>%' and name not like '%rows%' order by start")]
print("first 8:", " ".join("%.1f" % v for v in rows[:8]))
print("last 33:", " ".join("%.1f" % v for v in rows[-33:]))
t = rows[-30:-10]
print("the 20 timed launches: mean %.2f  min %.1f  max %.1f" % (sum(t) / len(t), min(t), max(t)))
PY
        echo '```'; echo;
      fi; } >> $O/kernel_stats.md
  done
done
# the other kernels the bench's modes report: LoG + wavelet (config 3 stages), voxel maps (config 4), the pairs tier, GLSZM / neighbourhood kernels (config 2)
cat > /tmp/r06_others.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
import bench
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
lv, msk = bench.make_volume(256, 32, "smooth", 0, dev)
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
for _ in range(3):
    engine.wavelet_images(img)
    engine.log_images(img, (1.0, 1.0, 1.0), (1.0, 2.0, 3.0, 4.0, 5.0))
for kind in ("uniform", "smooth"):
    l2, m2 = bench.make_volume(256, 32, kind, 3, dev)
    for _ in range(3):
        engine.glszm_compact(l2, m2, 32, int(m2.sum().item()))
        engine.gldm(l2, m2, 32)
        engine.ngtdm(l2, m2, 32)
for three_d in (False, True):
    bench.mode_voxel(dev, 0, 1, 512, torch.cuda.synchronize, three_d)
g = torch.Generator(device=dev); g.manual_seed(11)
raw = torch.randint(1, 301, (256, 256, 256), generator=g, device=dev, dtype=torch.int32)
ones = torch.ones((256, 256, 256), dtype=torch.uint8, device=dev)
for _ in range(3):
    engine.glcm((raw - 1) % 32 + 1, ones, 32, (1, 2))
    engine.glcm(raw, ones, 300, (1, 2))
    engine.glcm_glrlm((raw - 1) % 255 + 1, ones, 255, 256)
    engine.glcm_glrlm(raw, ones, 300, 256)
    engine.gldm(raw, ones, 300, 0, (1, 2))
    engine.ngtdm(raw, ones, 300, (1, 2))
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $O/stats_others -o s -- python /tmp/r06_others.py > $O/stats_others.log 2>&1
{ echo "## other kernels: filters at 256^3 (3 x wavelet_images + 3 x log_images of five sigmas), config-2 matrices at 256^3 (3 x GLSZM / GLDM / NGTDM, uniform and smooth), voxel maps at 512^3 (5x5 and 5^3 windows), the pairs tier at 256^3 (3 x each: GLCM d=[1,2] at 32 and 300 levels, GLCM+GLRLM at 255 and 300 levels, GLDM / NGTDM d=[1,2] at 300 levels)"; echo; python $R/scripts/rocpd_stats.py $O/stats_others/s_results.db | grep -E "prad|kernel \||---"; echo; } >> $O/kernel_stats.md
# one host thread of 256^3 cases (config 5): which kernels a case is made of (side streams overlap: durations are not kernel speeds)
rocprofv3 --kernel-trace --stats -d $O/stats_case -o s -- python $R/scripts/r06_case_loop.py 12 > $O/stats_case.log 2>&1
{ echo "## 15 cases of 256^3 (scripts/r06_case_loop.py: Original + 8 wavelet sub-bands, six classes; 3 warm-up + 12 timed), launches of four side streams overlap"; grep "per case" $O/stats_case.log; echo; python $R/scripts/rocpd_stats.py $O/stats_case/s_results.db | grep -E "prad|rocclr|at::|kernel \||---" | head -60; echo; } >> $O/kernel_stats.md
# PMC (separate passes, kernel-trace only)
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python $R/bench.py --steps 4 --warmup 2 --device-warmup-ms 0 $BA > $O/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $O/pmc > $O/pmc.md
python - <<PY > $O/counters.json
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(k, c):
    v = acc.get(k, {}).get(c, [])
    return sum(v) / len(v) if v else 0.0
KiB = 1024.0
fused, walk, rows = "prad::sweep_fw_kernel<true, 8, false, true>", "prad::sweep_fw_kernel<true, 8, false, false>", "prad::sweep_fw_rows_kernel<true>"
# FETCH_SIZE calibration (guide, section HBM + profiles/r02b_ablation.md): 16 B/lane loads count half, the walk's 8 B/lane loads
# 0.524 (one line angle alone reads the 134.2 MB level volume exactly once and reports 70.4 MB)
walk_b = mean(walk, "FETCH_SIZE") * KiB / 0.524                     # the 12 line walks' reads of the level volume
pack_b = max(0.0, mean(fused, "FETCH_SIZE") - mean(walk, "FETCH_SIZE")) * KiB * 2.0     # the side job's int32 + uint8 reads
fused_w = mean(fused, "WRITE_SIZE") * KiB
rows_b = mean(rows, "FETCH_SIZE") * KiB * 2.0 + mean(rows, "WRITE_SIZE") * KiB
fin_b = sum((mean(k, "FETCH_SIZE") + mean(k, "WRITE_SIZE")) * KiB for k in ("prad::finalize_glcm_diag_kernel", "prad::finalize_glrlm_kernel", "prad::multi_check_kernel"))
out = {"workload": [512, 32, "uniform"], "deferred_mode": "pipeline",
       "bytes": round(walk_b + pack_b + fused_w + rows_b + fin_b), "kernel_bytes": round(walk_b + pack_b + fused_w),
       "parts": {"walk_reads": round(walk_b), "pack_reads": round(pack_b), "fused_writes": round(fused_w), "rows_kernel": round(rows_b), "finalize": round(fin_b)},
       "lds": mean(fused, "SQ_INSTS_LDS"), "valu": mean(fused, "SQ_INSTS_VALU"), "salu": mean(fused, "SQ_INSTS_SALU"),
       "lds_bank_conflict": mean(fused, "SQ_LDS_BANK_CONFLICT"), "lds_idx_active": mean(fused, "SQ_LDS_IDX_ACTIVE"),
       "source": "profiles/r06_pmc.md (rocprofv3 --pmc passes over bench.py, scripts/prof_r06.sh, final source of round 6; fabric bytes per volume "
                 "in pipeline mode: walks + inline pack + x-angle kernel + finalize; kernel_bytes: the fused launch alone)"}
print(json.dumps(out, indent=1))
PY
# the sliding-window map kernels' SQ counters (two more passes, 256^3 maps)
bash $R/scripts/r06_voxel_pmc.sh > /dev/null 2>&1
{ echo; echo "## sliding-window map kernels (scripts/r06_voxel_pmc.sh: JointEntropy maps of a 256^3 volume, 5^3 and 5 x 5 windows)"; echo; cat $R/gpurun_out/r06_voxpmc/pmc.md; } >> $O/pmc.md
rm -rf $R/gpurun_out/r06_voxpmc/pmc
cd $R
find $O -name "*.db" -delete
find $O -name "*.csv" -size +200k -delete
du -sh $O
head -40 $O/kernel_stats.md
