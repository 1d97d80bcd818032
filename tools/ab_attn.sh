for v in "" "PFO_ATTN_FWD_BLOCKS=2048" "PFO_ATTN_FWD_BLOCKS=4096" "PFO_ATTN_BWD_BLOCKS=1280" "PFO_ATTN_BWD_BLOCKS=1024" "PFO_ATTN_BWD_BLOCKS=4096" "PFO_ATTN_BWD_BLOCKS=13440"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 0.5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'], {k:v for k,v in d['roofline']['families_ms_per_step'].items() if 'attn' in k})"
done
