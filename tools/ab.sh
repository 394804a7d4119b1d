#!/bin/bash
# A/B of library builds on the bench workload: bash tools/ab.sh lib1.so lib2.so ...   (paths relative to the repo root)
# prints ms/step and the per-kernel-class times of each build
for L in "$@"; do
  MSFL_LIB=$PWD/$L python bench.py --steps ${STEPS:-50} --warmup 5 --no-h2d --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('%-36s step %.4f ms  assoc %.4f fit %.4f solve %.4f index %.4f  failed %d' % ('$L', d['ms_per_step'], k['assoc'], k['fit'], k['solve'], k['index_build'], d['n_failed']))"
done
