#!/bin/bash
# PMC passes over the BM25 kernel, one counter set per rocprofv3 run:  scripts/pmc_bm25.sh OUTDIR [bench|df256|dffull]
set -e
D=$1; MIX=${2:-df256}
mkdir -p $D
export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $D -o $1 -- python3 scripts/pmc_bm25.py 4 50 $MIX > $D/$1.log 2>&1; }
run sqA "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES"
run sqB "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
run sqC "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_BRANCH"
run fetch "FETCH_SIZE"
# the wave walk (round 4: every OR query of <= 8 terms, k <= 64 -- all of the survey mixes)
python3 scripts/pmc_counters.py $D bm25_walk_wave_kernel $D/counters_walk_wave.json
# the workgroup walk (what is left to it: AND, > 8 terms, k > 64; everything under THR_BM25_WALK=block): the posting walk -- ordinary items and stage-A slices in ONE launch (template
# argument 2) when an eighth of the batch holds dense terms, else a launch each (0 / 1) -- and stage B
python3 scripts/pmc_counters.py $D bm25_topk_kernel $D/counters_walk_fused.json ", 2>" || true
python3 scripts/pmc_counters.py $D bm25_topk_kernel $D/counters_walk.json ", 0>" || true
python3 scripts/pmc_counters.py $D bm25_topk_kernel $D/counters_stage_a.json ", 1>" || true
python3 scripts/pmc_counters.py $D bm25_window_kernel $D/counters_stage_b.json
