cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
mkdir -p $O
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/p1 -o pass -- python3 $R/tools/wide_pmc_run.py 12500000 > $O/pmc1.out 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p1 -name "*.db" | head -1) 300 > $O/pmc_wide_sq.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /tmp/p2 -o pass -- python3 $R/tools/wide_pmc_run.py 12500000 > $O/pmc2.out 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p2 -name "*.db" | head -1) 300 > $O/pmc_wide_sq2.txt 2>&1
grep -h "wide\|pipe16" $O/pmc_wide_sq.txt $O/pmc_wide_sq2.txt
tail -2 $O/pmc2.out
