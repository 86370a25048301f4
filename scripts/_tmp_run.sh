timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
bash scripts/prof_glszm.sh u512 512 uniform 2>&1 | grep "glszm_\|pack" | head -7
bash scripts/prof_glszm.sh s512 512 smooth 2>&1 | grep "glszm_\|pack" | head -7
bash scripts/prof_glszm.sh s256 256 smooth 2>&1 | grep "glszm_\|pack" | head -7
rm -rf gpurun_out/glszm_u512 gpurun_out/glszm_s512 gpurun_out/glszm_s256
