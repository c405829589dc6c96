#!/bin/bash
# The measurements a round's profiles/ are made of (one GPU, ~10 min):  scripts/collect_profiles.sh OUTDIR
#   default bench line + rocprofv3 kernel stats of the same command; dense-only run whose scan
#   launches all have the bench shape; shard proxies (125 K and 500 K rows); dim 1024; PMC passes
#   over the default scan folded into the JSON bench.py quotes as roofline.traffic.
set -e
D=${1:-gpurun_out/collect}
mkdir -p $D
export TMPDIR=/tmp
python3 bench.py > $D/bench_default.json.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_default -o bench -- python3 bench.py --no-cpu-baseline > $D/bench_default_profiled.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_dense -o dense -- python3 bench.py --config dense --no-extras --no-cpu-baseline > $D/bench_dense_only.json.log 2>&1
python3 bench.py --docs 125000 --config dense --no-extras --no-cpu-baseline > $D/bench_shard125k.json.log 2>&1
python3 bench.py --docs 500000 --config dense --no-extras --no-cpu-baseline > $D/bench_shard500k.json.log 2>&1
python3 bench.py --dim 1024 --config dense --no-extras --no-cpu-baseline > $D/bench_dim1024.json.log 2>&1
python3 bench.py --docs 1250000 --config triple --no-extras --no-cpu-baseline > $D/bench_shard1250k_triple.json.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_triple -o triple -- python3 bench.py --config triple --no-extras --no-cpu-baseline --steps 6 > $D/bench_triple_profiled.log 2>&1
python3 scripts/step_trace.py $D/prof_triple/triple_kernel_trace.csv > $D/triple_step_trace.md
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_dense_bm25 -o db -- python3 bench.py --config dense_bm25 --no-extras --no-cpu-baseline --steps 6 > $D/bench_dense_bm25_profiled.log 2>&1
python3 scripts/step_trace.py $D/prof_dense_bm25/db_kernel_trace.csv > $D/dense_bm25_step_trace.md
# a document shard's share of a dense step, with and without the shard floor (all shards in this process)
python3 bench.py --config dense --no-extras --no-cpu-baseline --shard-proxy 8 > $D/bench_shard_proxy8.json.log 2>&1
python3 bench.py --config dense --no-extras --no-cpu-baseline --shard-proxy 2 > $D/bench_shard_proxy2.json.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_floor -o floor -- python3 scripts/floor_probe.py floor 8 > $D/floor_probe_floor.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_floor -o classic -- python3 scripts/floor_probe.py classic 8 > $D/floor_probe_classic.log 2>&1
python3 bench.py --lexical-mix no-stopwords --no-cpu-baseline > $D/bench_no_stopwords.json.log 2>&1
bash scripts/pmc_passes.sh $D/pmc_scan 2048 > $D/pmc_scan.log 2>&1
python3 scripts/pmc_counters.py $D/pmc_scan dense_scan_f16qs $D/scan_f16qs_counters.json "768, 1" > $D/pmc_fold.log 2>&1
# BM25 alone: three query mixes, kernel stats and counters of its three kernels on 2048 df-proportional queries
python3 scripts/bench_bm25.py > $D/bench_bm25.json 2> $D/bench_bm25.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_bm25 -o bm25 -- python3 scripts/pmc_bm25.py 4 50 dffull > $D/prof_bm25.log 2>&1
bash scripts/pmc_bm25.sh $D/pmc_bm25 dffull > $D/pmc_bm25.log 2>&1
# (the raw traces are tens of MB each: the stats files and the step tables above are what is kept)
rm -f $D/prof_*/*_kernel_trace.csv
ls $D
