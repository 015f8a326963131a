#!/bin/bash
# GPU box: does handing freed result arrays back to the OS (munmap / heap trimming -> MMU notifiers on memory the runtime had pinned)
# cost the benches anything?  glibc told to keep everything (no mmap for big blocks, no trimming) against the default, interleaved.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_malloc_ab; mkdir -p $O
show() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$2', '%.4g /s  %.3f ms/step [%.3f..%.3f] frac %.3f' % (d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], r['frac']))"; }
for rep in 1 2; do for keep in 0 1; do
  if [ $keep = 1 ]; then export MALLOC_MMAP_THRESHOLD_=33554432 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_TOP_PAD_=268435456; else unset MALLOC_MMAP_THRESHOLD_ MALLOC_TRIM_THRESHOLD_ MALLOC_TOP_PAD_; fi
  for w in cfg3 cfg4 cfg2 cfg5; do
    python3 bench.py --workload $w --no-cpu-baseline --no-secondary --per-span-steps 0 > $O/${w}_$keep.json 2>> $O/err.txt; show $O/${w}_$keep.json "keep $keep $w"
  done
  python3 bench.py --api-only 2>> $O/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('keep $keep api', d['wall_ms_all'], 'api_large', d['api_large']['wall_ms_all'])"
done; done
