# K1 before (end of round 1) / after (this round) under the matrix-pipe counters, config 2 and B=8
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for shape in "B=1 S=4096" "B=8 S=4096"; do
  tag=$(echo $shape | tr -d ' =' )
  ARGS="$shape check=0 rounds=1 iters=3"
  rm -rf $O/k1pmc_a_$tag $O/k1pmc_b_$tag
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/k1pmc_a_$tag --output-format csv -- $R/scripts/k1_lab/k1_lab $ARGS > $O/k1pmc_a_$tag.log 2>&1 && echo passA-$tag-ok || { tail -5 $O/k1pmc_a_$tag.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace -d $O/k1pmc_b_$tag --output-format csv -- $R/scripts/k1_lab/k1_lab $ARGS > $O/k1pmc_b_$tag.log 2>&1 && echo passB-$tag-ok || { tail -5 $O/k1pmc_b_$tag.log; exit 1; }
  echo "#### $shape causal 16Q/8KV hd128" 
  python3 $R/scripts/pmc_k1_compare.py 6 base_r01,cur_r02 $O/k1pmc_a_$tag $O/k1pmc_b_$tag
done > $O/r02_pmc_k1_before_after.txt
cat $O/r02_pmc_k1_before_after.txt
