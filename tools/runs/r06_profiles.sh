#!/bin/bash
# Round-6 evidence from the final tree: GPU suite (+ parity ledger), K2 / K2b instruction mix (bench.py reads them),
# the bench lines of the driver's command and of configs 1, 2 (+ shard rehearsal), 4, rocprofv3 kernel traces of the same
# commands -- condensed PER KERNEL AND GRID SIZE by tools/collect_profiles.py --, PMC traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), the history-in-parts policy in fresh processes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06p; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
# a step that was killed at its limit ends the run: no further GPU step after a hang
chk() { if [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; then echo "step killed (rc=$1): stopping"; exit 1; fi; }
export MRPHY_PARITY_LEDGER=$O/parity_ledger.json
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest_gpu.txt; chk $rc
tail -n 4 $O/pytest_gpu.txt > $O/pytest_gpu_tail.txt
unset MRPHY_PARITY_LEDGER
P="rocprofv3 --kernel-trace --output-format csv"
SQ1="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
timeout -k 10 200 $P --pmc $SQ1 -d $O/k2_sq1 -- python3 tools/run_kernels.py k2 128 4096 3 > $O/k2_sq1.log 2>&1; rc=$?; echo "k2 sq1 rc=$rc"; chk $rc
timeout -k 10 200 $P --pmc $SQ2 -d $O/k2_sq2 -- python3 tools/run_kernels.py k2 128 4096 3 > $O/k2_sq2.log 2>&1; rc=$?; echo "k2 sq2 rc=$rc"; chk $rc
timeout -k 10 200 $P --pmc $SQ1 -d $O/k2b_sq1 -- python3 tools/run_kernels.py gradfused 64 2048 3 > $O/k2b_sq1.log 2>&1; rc=$?; echo "k2b sq1 rc=$rc"; chk $rc
timeout -k 10 200 $P --pmc $SQ2 -d $O/k2b_sq2 -- python3 tools/run_kernels.py gradfused 64 2048 3 > $O/k2b_sq2.log 2>&1; rc=$?; echo "k2b sq2 rc=$rc"; chk $rc
python3 tools/collect_profiles.py $O r06 > $O/collect1.txt 2>&1
# bench lines: the driver's command (all single-GPU configs in one line), then each config alone
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; rc=$?; echo "bench default rc=$rc"; chk $rc
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --shard-of 8 --no-cpu > $O/bench_cfg2.json 2> $O/bench_cfg2.log; rc=$?; echo "bench cfg2 + shard rc=$rc"; chk $rc
timeout -k 10 300 python3 bench.py --config 1 --steps 20 --warmup 2 > $O/bench_cfg1.json 2> $O/bench_cfg1.log; rc=$?; echo "bench cfg1 rc=$rc"; chk $rc
timeout -k 10 300 python3 bench.py --config 4 --steps 10 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.log; rc=$?; echo "bench cfg4 rc=$rc"; chk $rc
# rocprofv3 kernel traces of the same commands; the headline ALONE (--no-extra-configs) and the driver's command
timeout -k 10 300 $P --stats -d $O/prof_cfg2 -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-extra-configs > /dev/null 2> $O/prof_cfg2.log; rc=$?; echo "prof cfg2 rc=$rc"; chk $rc
timeout -k 10 400 $P --stats -d $O/prof_default -- python3 bench.py --steps 20 --warmup 2 --no-cpu > /dev/null 2> $O/prof_default.log; rc=$?; echo "prof default rc=$rc"; chk $rc
timeout -k 10 300 $P --stats -d $O/prof_cfg1 -- python3 bench.py --config 1 --steps 20 --warmup 2 --no-cpu > /dev/null 2> $O/prof_cfg1.log; rc=$?; echo "prof cfg1 rc=$rc"; chk $rc
timeout -k 10 300 $P --stats -d $O/prof_shard -- python3 tools/run_kernels.py fwd 128 4096 20 262144 > /dev/null 2> $O/prof_shard.log; rc=$?; echo "prof shard rc=$rc"; chk $rc
timeout -k 10 300 $P --stats -d $O/prof_cfg4 -- python3 bench.py --config 4 --steps 30 --warmup 2 --no-cpu --grad-route allocator > /dev/null 2> $O/prof_cfg4.log; rc=$?; echo "prof cfg4 rc=$rc"; chk $rc
timeout -k 10 300 $P --stats -d $O/prof_cfg4_ws -- python3 bench.py --config 4 --steps 30 --warmup 2 --no-cpu --grad-route workspace > /dev/null 2> $O/prof_cfg4_ws.log; rc=$?; echo "prof cfg4 ws rc=$rc"; chk $rc
timeout -k 10 300 $P --stats -d $O/prof_f64 -- python3 tools/run_kernels.py grad64 64 1024 10 > /dev/null 2> $O/prof_f64.log; rc=$?; echo "prof f64 rc=$rc"; chk $rc
timeout -k 10 300 $P --stats -d $O/prof_f64fwd -- python3 tools/run_kernels.py fwd64 64 1024 10 > /dev/null 2> $O/prof_f64fwd.log; rc=$?; echo "prof f64 fwd rc=$rc"; chk $rc
# PMC traffic, separate passes
for w in "fwd 64 1024 3:cfg1" "fwd 128 4096 3 262144:shard" "fwd 128 4096 3:cfg2" "grad 64 2048 3:cfg4" "fwd64 64 1024 3:f64fwd" "grad64 64 1024 3:f64grad"; do
  a=${w%%:*}; t=${w##*:}
  timeout -k 10 300 $P --pmc FETCH_SIZE -d $O/pmc_fetch_$t -- python3 tools/run_kernels.py $a > $O/pmc_fetch_$t.log 2>&1; rc=$?; echo "fetch $t rc=$rc"; chk $rc
  timeout -k 10 300 $P --pmc WRITE_SIZE -d $O/pmc_write_$t -- python3 tools/run_kernels.py $a > $O/pmc_write_$t.log 2>&1; rc=$?; echo "write $t rc=$rc"; chk $rc
done
# the history in parts, fresh processes (nothing probed): one block against the policy's four parts
timeout -k 10 300 python3 tools/hist_policy_processes.py $O/hist_policy.json --procs 6 --variants parts1,parts4 > $O/hist_policy.log 2>&1; rc=$?; echo "hist policy rc=$rc"; chk $rc
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete; python3 tools/collect_profiles.py $O r06 > $O/collect2a.txt 2>&1; find $O -name '*kernel_trace.csv' -size +3M -delete
python3 tools/collect_profiles.py $O r06 > $O/collect2.txt 2>&1; cat $O/collect2.txt
for d in prof_cfg2 prof_cfg1 prof_shard prof_cfg4 prof_cfg4_ws prof_f64 prof_f64fwd; do python3 tools/kstats.py $O/$d $d >> $O/kstats.txt 2>&1; done
cat $O/kstats.txt; du -sh $O
