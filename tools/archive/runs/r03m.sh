#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03m; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
SQ1="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 200 $P --pmc $SQ1 -d $O/sq1 -- python3 tools/run_kernels.py gradfused 128 1024 3 > $O/sq1.log 2>&1; echo "sq1 rc=$?"
timeout -k 10 200 $P --pmc $SQ2 -d $O/sq2 -- python3 tools/run_kernels.py gradfused 128 1024 3 > $O/sq2.log 2>&1; echo "sq2 rc=$?"
timeout -k 10 200 $P --pmc $SQ3 -d $O/sq3 -- python3 tools/run_kernels.py gradfused 128 1024 3 > $O/sq3.log 2>&1; echo "sq3 rc=$?"
python3 tools/pmc_summary.py $O/k2b_pmc.json gradfused_128_1024 $O/sq1 $O/sq2 $O/sq3 > $O/k2b_pmc.txt 2>&1
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
grep -A40 "k_bloch_rfgr_bwd<" $O/k2b_pmc.txt | head -60
