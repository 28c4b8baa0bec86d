set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_k1_chain.py tests/test_gpu_flash_attention.py tests/test_gpu_paged_prefill.py tests/test_gpu_paged_fuzz.py -x -q --timeout 120 --timeout-method thread > $O/chain_tests.txt 2>&1 || { tail -30 $O/chain_tests.txt; exit 1; }
tail -3 $O/chain_tests.txt
bash scripts/gpu_lab_chain.sh
