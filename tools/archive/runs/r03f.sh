#!/bin/bash
# placement modes of K1h / K3 seen through the L2's memory-side counters: local DRAM vs GMI (other die) traffic,
# credit stalls, TLB misses -- per dispatch, several placements per pass
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03f; rm -rf $O; mkdir -p $O; cd $R; export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
run() { timeout -k 10 300 $P --pmc $2 -d $O/$1 -- python3 tools/placement_pmc_run.py 128 1024 6 > $O/$1.log 2>&1; echo "$1 rc=$?"; }
run p1 "TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RDREQ_GMI_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_GMI_32B"
run p2 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_GMI_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_GMI_CREDIT_STALL"
run p3 "TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_BUSY TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT"
run p4 "TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ"
python3 tools/placement_pmc_table.py $O/placement_pmc.json $O/p1 $O/p2 $O/p3 $O/p4 > $O/placement_pmc.txt 2>&1
find $O -name '*.db' -delete; find $O -name '*agent_info*' -delete
cat $O/placement_pmc.txt
