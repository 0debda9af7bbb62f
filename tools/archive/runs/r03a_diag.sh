#!/bin/bash
# round-3 probe A: gradient parity (whose error), K2 PMC (VALU instruction mix, both precision modes),
# one-generation grids: wave timelines + SQ/TCC counters of K1 / K1h / K3 at 64^3 x 2048 and 128^3 x 1024.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03a; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python3 tools/grad_parity.py $O/grad_parity.json > $O/grad_parity.txt 2>&1; echo "grad_parity rc=$?"
timeout -k 10 200 python3 tools/timeline_waves.py 64 2048 $O/timeline_64_2048.json > $O/timeline_64_2048.txt 2>&1; echo "timeline64 rc=$?"
timeout -k 10 200 python3 tools/timeline_waves.py 128 1024 $O/timeline_128_1024.json > $O/timeline_128_1024.txt 2>&1; echo "timeline128 rc=$?"
P="rocprofv3 --kernel-trace --output-format csv"
SQ1="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ3="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQ4="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
TC1="TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_BUSY TCC_REQ"
timeout -k 10 200 $P --pmc $SQ1 -d $O/k2_sq1 -- python3 tools/run_kernels.py k2 128 4096 3 > $O/k2_sq1.log 2>&1; echo "k2 sq1 rc=$?"
timeout -k 10 200 $P --pmc $SQ2 -d $O/k2_sq2 -- python3 tools/run_kernels.py k2 128 4096 3 > $O/k2_sq2.log 2>&1; echo "k2 sq2 rc=$?"
python3 tools/pmc_summary.py $O/k2_pmc.json k2_128_4096 $O/k2_sq1 $O/k2_sq2 > $O/k2_pmc.txt 2>&1
for W in "64 2048" "128 1024"; do
  L=$(echo $W | tr ' ' '_')
  timeout -k 10 200 $P --stats -d $O/g_stats_$L -- python3 tools/run_kernels.py grad $W 4 > $O/g_stats_$L.log 2>&1; echo "stats $L rc=$?"
  timeout -k 10 200 $P --pmc $SQ3 -d $O/g_sq3_$L -- python3 tools/run_kernels.py grad $W 3 > $O/g_sq3_$L.log 2>&1; echo "sq3 $L rc=$?"
  timeout -k 10 200 $P --pmc $SQ4 -d $O/g_sq4_$L -- python3 tools/run_kernels.py grad $W 3 > $O/g_sq4_$L.log 2>&1; echo "sq4 $L rc=$?"
  timeout -k 10 200 $P --pmc $TC1 -d $O/g_tc1_$L -- python3 tools/run_kernels.py grad $W 3 > $O/g_tc1_$L.log 2>&1; echo "tc1 $L rc=$?"
  timeout -k 10 200 $P --pmc FETCH_SIZE -d $O/g_fetch_$L -- python3 tools/run_kernels.py grad $W 3 > $O/g_fetch_$L.log 2>&1; echo "fetch $L rc=$?"
  timeout -k 10 200 $P --pmc WRITE_SIZE -d $O/g_write_$L -- python3 tools/run_kernels.py grad $W 3 > $O/g_write_$L.log 2>&1; echo "write $L rc=$?"
  python3 tools/pmc_summary.py $O/onegen_pmc.json grad_$L $O/g_sq3_$L $O/g_sq4_$L $O/g_tc1_$L $O/g_fetch_$L $O/g_write_$L > $O/onegen_pmc_$L.txt 2>&1
  python3 tools/kstats.py $O/g_stats_$L grad_$L >> $O/kstats.txt 2>&1
done
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete; find $O -name '*kernel_trace.csv' -size +2M -delete
du -sh $O; cat $O/kstats.txt | head -40
