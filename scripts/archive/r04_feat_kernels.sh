#!/bin/bash
# round 4: the formula kernels of a case (zone / glcm / mcc / ngtdm features) under rocprofv3 + the feature tests
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_features.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cl
CASE_ONLY=0 CASE_REP=4 rocprofv3 --kernel-trace --stats -d /tmp/cl -o s -- python $R/scripts/case_latency.py 2>&1 | grep enqueueSegment
python $R/scripts/rocpd_stats.py /tmp/cl/s_results.db | grep -E "features_kernel|mcc_kernel|fo_|copyBuffer|fillBuffer" | cut -c1-110
cd $R; python scripts/case_latency.py 2>&1 | grep enqueueSegment=
