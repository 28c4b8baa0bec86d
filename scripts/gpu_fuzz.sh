# randomised paged launches against the generic kernel, default policy and forced variants
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz.txt; : > $O
for s in 11 12 13; do timeout -k 10 170 python scripts/diag/fuzz_paged.py $s 150 >> $O 2>&1 || { tail -3 $O; exit 1; }; tail -1 $O; done
NVLLM_K1_KSPLIT=3 timeout -k 10 170 python scripts/diag/fuzz_paged.py 21 120 >> $O 2>&1 || { tail -3 $O; exit 1; }; tail -1 $O
NVLLM_K1_PACK=0 timeout -k 10 170 python scripts/diag/fuzz_paged.py 22 120 >> $O 2>&1 || { tail -3 $O; exit 1; }; tail -1 $O
NVLLM_PAGED_SPLITS=3 NVLLM_PAGED_MAX_ROWS=48 timeout -k 10 170 python scripts/diag/fuzz_paged.py 23 120 >> $O 2>&1 || { tail -3 $O; exit 1; }; tail -1 $O
