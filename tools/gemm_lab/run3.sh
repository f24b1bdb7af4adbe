mkdir -p gpurun_out/pmc3
L=tools/gemm_lab/lab
$L zero nocold 8192,8192,8192 4096,4096,4096 cfgs=4,11,16 reps=10 > gpurun_out/lab3_zero.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc3/a -o a --output-format csv -- $R/$L nocold 8192,8192,8192 cfgs=4,11,16 reps=2 > $R/gpurun_out/pmc3/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmc3/b -o b --output-format csv -- $R/$L nocold 8192,8192,8192 cfgs=4,11,16 reps=2 > $R/gpurun_out/pmc3/b.log 2>&1
cd $R; cat gpurun_out/lab3_zero.txt; ls -R gpurun_out/pmc3 | head -30; tail -3 gpurun_out/pmc3/a.log gpurun_out/pmc3/b.log
