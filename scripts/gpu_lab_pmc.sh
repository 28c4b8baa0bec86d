# PMC passes over the K1 lab binary (one variant built): wave-cycle breakdown and LDS counters
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
ARGS="${LAB_ARGS:-B=4 S=4096 causal=0 check=0 rounds=1 iters=3}"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_lab1 $O/pmc_lab2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_lab1 --output-format csv -- $R/scripts/k1_lab/k1_lab $ARGS > $O/pmc_lab1.log 2>&1 && echo pass1-ok
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_lab2 --output-format csv -- $R/scripts/k1_lab/k1_lab $ARGS > $O/pmc_lab2.log 2>&1 && echo pass2-ok
python3 $R/scripts/pmc_summary.py $O/pmc_lab1 $O/pmc_lab2 | grep -i "flash_attn_mfma\|duration" 
