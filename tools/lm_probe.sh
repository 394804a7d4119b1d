for n in 256 512 1024 2048; do
  python bench.py --steps 30 --warmup 5 --scans $n --no-h2d --cpu-sample 0 --no-stages --no-worlds --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']
print('scans $n  step %.4f ms  assoc %.4f fit %.4f solve %.4f index %.4f' % (d['ms_per_step'], k['assoc'], k['fit'], k['solve'], k['index_build']))"
done
echo PROFILE
MSFL_LIB=$PWD/build_ab/libmsfl_hip_lmprof128.so python bench.py --steps 10 --warmup 2 --no-h2d --cpu-sample 0 --no-stages --no-worlds --steady-steps 0 2>&1 >/dev/null | grep "lm profile"
