# the per-GPU batch sizes of --scaling strong at 8 / 4 / 2 GPUs (65 536 x 32 split over the ranks) on one GPU: batches in flight
cd "$GRAFT_REPO_ROOT"
for cfg in "8192 160 8" "8192 160 12" "8192 160 16" "16384 80 8" "16384 80 12" "32768 40 5" "32768 40 8" "65536 20 5"; do
  set -- $cfg
  python3 bench.py --targets $1 --steps $2 --warmup 8 --streams $3 --no-cpu-baseline --no-secondary --per-span-steps 0 --repeats 2 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('targets $1 steps $2 in flight $3:', '%.4g dec/s' % d['value'], 'frac %.3f' % d['roofline']['frac'])"
done
