"""(CPU; run on the GPU box's host) thread scaling of the CPU baseline: the AVX-512 intrinsics twin of the oracle's fp32 pass and
the compiler-vectorised scalar oracle, 1080p -> 4K, by OpenMP thread count and wait policy.  The box's container is capped by a
cgroup CPU quota (16 CPUs of a 64-core part): spinning OpenMP workers burn quota while they wait, so OMP_WAIT_POLICY matters."""
import os, subprocess, sys
code = '''
import sys, time, os, ctypes
sys.path[:0] = ['oracle', 'video-super-resolution-library_amd']
import numpy as np, oracle_py as O, synth
y = synth.natural_y(1920, 1080, 8, seed=1)
p1 = O.make_pass(O.Model('filters_2x/filters_highres', 8, 1), 8, False, 2)
for name, fn in (('intrinsics', O.process_y_intrinsics), ('scalar', O.process_y)):
    if name == 'intrinsics' and O.lib512() is None: continue
    fn(y, 3840, 2160, p1); best = 1e9
    for i in range(4):
        t = time.perf_counter(); fn(y, 3840, 2160, p1); best = min(best, time.perf_counter() - t)
    print(os.environ.get('OMP_NUM_THREADS'), os.environ.get('OMP_WAIT_POLICY'), name, '%.1f ms/frame  %.0f MP/s' % (best * 1e3, 8.2944 / best))
'''
for pol in ("passive", "active"):
    for t in (1, 4, 8, 16, 32):
        env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_WAIT_POLICY=pol, RAISR_ORACLE_ISA="auto")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
