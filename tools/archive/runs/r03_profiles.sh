#!/bin/bash
# Round-3 evidence from the final tree: GPU suite (+ parity ledger), bench lines, rocprofv3 kernel stats,
# PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes), K2 instruction mix.  Condensed into profiles/r03_*
# by tools/collect_r03.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03p; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
export MRPHY_PARITY_LEDGER=$O/parity_ledger.json
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -n 4 $O/pytest_gpu.txt > $O/pytest_gpu_tail.txt
unset MRPHY_PARITY_LEDGER
P="rocprofv3 --kernel-trace --output-format csv"
# K2 instruction mix first (bench.py reads profiles/r03_k2_pmc.json; collect it, then bench)
SQ1="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
timeout -k 10 200 $P --pmc $SQ1 -d $O/k2_sq1 -- python3 tools/run_kernels.py k2 128 4096 3 > $O/k2_sq1.log 2>&1; echo "k2 sq1 rc=$?"
timeout -k 10 200 $P --pmc $SQ2 -d $O/k2_sq2 -- python3 tools/run_kernels.py k2 128 4096 3 > $O/k2_sq2.log 2>&1; echo "k2 sq2 rc=$?"
python3 tools/collect_r03.py $O > $O/collect1.txt 2>&1
timeout -k 10 500 python3 bench.py --steps 20 --warmup 2 > $O/bench_fwd.json 2> $O/bench_fwd.log; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --cube 64 --nT 1024 --steps 20 --warmup 2 > $O/bench_fwd_cfg1.json 2> $O/bench_fwd_cfg1.log; echo "bench cfg1 rc=$?"
timeout -k 10 500 python3 bench.py --steps 10 --warmup 2 --no-cpu --shard-of 8 > $O/bench_fwd_shard8.json 2> $O/bench_fwd_shard8.log; echo "bench shard rc=$?"
F="python3 bench.py --steps 20 --warmup 2 --no-cpu"
timeout -k 10 300 $P --stats -d $O/prof_fwd -- $F > $O/bench_fwd_prof.json 2> $O/prof_fwd.log; echo "prof fwd rc=$?"
timeout -k 10 300 $P --pmc FETCH_SIZE -d $O/pmc_fetch_fwd -- python3 tools/run_kernels.py fwd 128 4096 3 > $O/pmc_fetch_fwd.log 2>&1; echo "fetch fwd rc=$?"
timeout -k 10 300 $P --pmc WRITE_SIZE -d $O/pmc_write_fwd -- python3 tools/run_kernels.py fwd 128 4096 3 > $O/pmc_write_fwd.log 2>&1; echo "write fwd rc=$?"
G="python3 bench.py --mode grad --cube 128 --nT 1024 --no-interp --steps 5 --warmup 2"
G4="python3 bench.py --mode grad --steps 10 --warmup 2"
timeout -k 10 300 $G > $O/bench_grad128.json 2> $O/bench_grad128.log; echo "grad128 rc=$?"
timeout -k 10 300 $P --stats -d $O/prof_grad128 -- $G > /dev/null 2> $O/prof_grad128.log; echo "prof grad128 rc=$?"
timeout -k 10 300 $G4 > $O/bench_grad_cfg4.json 2> $O/bench_grad_cfg4.log; echo "grad cfg4 rc=$?"
timeout -k 10 300 $P --stats -d $O/prof_grad_cfg4 -- $G4 > /dev/null 2> $O/prof_grad_cfg4.log; echo "prof grad cfg4 rc=$?"
timeout -k 10 300 $P --pmc FETCH_SIZE -d $O/pmc_fetch_grad128 -- python3 tools/run_kernels.py grad 128 1024 3 > $O/pmc_fetch_grad128.log 2>&1; echo "fetch grad128 rc=$?"
timeout -k 10 300 $P --pmc WRITE_SIZE -d $O/pmc_write_grad128 -- python3 tools/run_kernels.py grad 128 1024 3 > $O/pmc_write_grad128.log 2>&1; echo "write grad128 rc=$?"
timeout -k 10 300 $P --pmc FETCH_SIZE -d $O/pmc_fetch_grad64 -- python3 tools/run_kernels.py grad 64 2048 3 > $O/pmc_fetch_grad64.log 2>&1; echo "fetch grad64 rc=$?"
timeout -k 10 300 $P --pmc WRITE_SIZE -d $O/pmc_write_grad64 -- python3 tools/run_kernels.py grad 64 2048 3 > $O/pmc_write_grad64.log 2>&1; echo "write grad64 rc=$?"
(tools/dbg/dpp_rate; tools/dbg/sgpr_rate; tools/dbg/pk_rate; tools/dbg/sgpr_mix) > $O/valu_operand_rates.txt 2>&1
timeout -k 10 200 python3 tools/ptx_timing.py $O/ptx_timing.json > $O/ptx_timing.log 2>&1; echo "ptx rc=$?"
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete; find $O -name '*kernel_trace.csv' -size +3M -delete
python3 tools/collect_r03.py $O > $O/collect2.txt 2>&1; cat $O/collect2.txt
for d in prof_fwd prof_grad128 prof_grad_cfg4; do python3 tools/kstats.py $O/$d $d >> $O/kstats.txt 2>&1; done
cat $O/kstats.txt; du -sh $O; cat $O/bench_fwd.json
