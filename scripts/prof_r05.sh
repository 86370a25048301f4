#!/bin/bash
# round-5 evidence pack: rocprofv3 kernel stats of the bench command (pipeline mode: every launch of a step on ONE stream, so
# the averages below ARE what bench.py's roofline block reports), PMC passes (separate, --kernel-trace only) and the counter
# file bench.py reads (profiles/r05_counters.json).  Everything lands in gpurun_out/r05/ (copy the .md / .json to profiles/).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-modes --no-host-boundary"
{
echo "# r05 -- rocprofv3 --kernel-trace --stats of bench.py (scripts/prof_r05.sh), final source of round 5"
echo
echo "Pipeline mode (default): the launches of a step sit on one stream; 'sweep_fw_kernel<..., true>' is the launch that walks the"
echo "12 line angles of volume N-1 AND packs volume N (PACK = true); '<..., false>' are the synchronous calls and the flush."
echo
} > $O/kernel_stats.md
for d in uniform smooth; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$d -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --dist $d > $O/stats_$d.log 2>&1
  { echo "## bench.py --steps 20 --warmup 3 --dist $d"; tail -1 $O/stats_$d.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"finalize_ms": [0-9.]*' | tr '\n' ' '; echo; echo;
    python $R/scripts/rocpd_stats.py $O/stats_$d/s_results.db | grep -E "prad|rocclr|kernel \||---"; echo; } >> $O/kernel_stats.md
done
rocprofv3 --kernel-trace --stats -d $O/stats_lanes -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --deferred-mode lanes > $O/stats_lanes.log 2>&1
{ echo "## bench.py --steps 20 --warmup 3 --deferred-mode lanes (two internal streams: launches overlap, durations are not kernel speeds)"; tail -1 $O/stats_lanes.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; echo; python $R/scripts/rocpd_stats.py $O/stats_lanes/s_results.db | grep -E "prad|rocclr|kernel \||---"; echo; } >> $O/kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/stats_256 -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --size 256 > $O/stats_256.log 2>&1
{ echo "## bench.py --size 256"; tail -1 $O/stats_256.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; echo; python $R/scripts/rocpd_stats.py $O/stats_256/s_results.db | grep -E "prad|kernel \||---"; echo; } >> $O/kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/stats_ng64 -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --levels 64 > $O/stats_ng64.log 2>&1
{ echo "## bench.py --levels 64 (two-table fixed-window kernel, 16-bit levels; deferred calls of this path run on two streams, so launches overlap and these durations are NOT kernel speeds: event-timed numbers in r04_probes.md section 16)"; tail -1 $O/stats_ng64.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; echo; python $R/scripts/rocpd_stats.py $O/stats_ng64/s_results.db | grep -E "prad|kernel \||---"; echo; } >> $O/kernel_stats.md
# the other kernels this round touched: LoG + wavelet (config 3 stages), voxel maps (config 4)
cat > /tmp/r05_others.py <<PY
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
for three_d in (False, True):
    bench.mode_voxel(dev, 0, 1, 512, torch.cuda.synchronize, three_d)
# the pairs tier (kernels_pairs.h, round 5): GLCM distances [1, 2] at 32 levels, GLCM + GLRLM at 255 levels, GLDM / NGTDM at 300
g = torch.Generator(device=dev); g.manual_seed(11)
raw = torch.randint(1, 301, (256, 256, 256), generator=g, device=dev, dtype=torch.int32)
ones = torch.ones((256, 256, 256), dtype=torch.uint8, device=dev)
for _ in range(3):
    engine.glcm((raw - 1) % 32 + 1, ones, 32, (1, 2))
    engine.glcm_glrlm((raw - 1) % 255 + 1, ones, 255, 256)
    engine.gldm(raw, ones, 300, 0, (1, 2))
    engine.ngtdm(raw, ones, 300, (1, 2))
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d $O/stats_others -o s -- python /tmp/r05_others.py > $O/stats_others.log 2>&1
{ echo "## filters at 256^3 (3 x wavelet_images + 3 x log_images of five sigmas) voxel maps at 512^3 (5x5 and 5^3 windows, three maps each) and the pairs tier at 256^3 (3 x GLCM d=[1,2] 32 levels, GLCM+GLRLM 255 levels, GLDM / NGTDM d=[1,2] 300 levels)"; echo; python $R/scripts/rocpd_stats.py $O/stats_others/s_results.db | grep -E "prad|kernel \||---"; echo; } >> $O/kernel_stats.md
# PMC (separate passes, kernel-trace only)
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python $R/bench.py --steps 4 --warmup 2 $BA > $O/pmc_$name.log 2>&1; }
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
       "source": "profiles/r05_pmc.md (rocprofv3 --pmc passes over bench.py, scripts/prof_r05.sh; fabric bytes per volume in pipeline mode: "
                 "walks + inline pack + x-angle kernel + finalize; kernel_bytes: the fused launch alone)"}
print(json.dumps(out, indent=1))
PY
cd $R
find $O -name "*.db" -delete
find $O -name "*.csv" -size +200k -delete
du -sh $O
cat $O/kernel_stats.md | head -60
