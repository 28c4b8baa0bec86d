# PMC passes over the prefill kernel (B=8): wave cycle breakdown, MFMA busy, LDS conflicts
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_fa1 --output-format csv -- python3 $R/bench.py --workload prefill --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_fa1.json 2> $O/pmc_fa1.err && echo pass1-ok
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_fa2 --output-format csv -- python3 $R/bench.py --workload prefill --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_fa2.json 2> $O/pmc_fa2.err && echo pass2-ok
tail -3 $O/pmc_fa2.err
