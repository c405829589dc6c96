# A/B runs of scripts/bench_bm25.py under tuning knobs: bash scripts/exp_bm25.sh "name:ENV=VAL" ...
mkdir -p gpurun_out/exp
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="X=1"
  env $envs timeout -k 10 200 python scripts/bench_bm25.py 1000000 2048 > gpurun_out/exp/$name.json 2>&1 || { tail -5 gpurun_out/exp/$name.json; exit 1; }
  echo "$name $(tail -1 gpurun_out/exp/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print([ (k[:12], v["ms_with_bounds"], v["ms_every_posting_scored"], v["bit_equal_to_oracle"]) for k,v in d.items() if isinstance(v,dict)])')"
done
