cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/p1 -o pass -- python3 $R/tools/pass_pmc_run.py 12500000 > $R/gpurun_out/pmc1.out 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p1 -name "*.db" | head -1) 300 > $R/gpurun_out/pmc_f6_lds_pass1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC -d /tmp/p2 -o pass -- python3 $R/tools/pass_pmc_run.py 12500000 > $R/gpurun_out/pmc2.out 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p2 -name "*.db" | head -1) 300 > $R/gpurun_out/pmc_f6_lds_pass2.txt 2>&1
grep -h "scan_f6_pass_lds\|pipe16" $R/gpurun_out/pmc_f6_lds_pass1.txt $R/gpurun_out/pmc_f6_lds_pass2.txt
tail -3 $R/gpurun_out/pmc2.out
