"""One-off: tests/test_gpu_fuzz.py::test_random_shape_against_oracle over seeds 160 ... 1359.
Run from the repository root: python tools/stress/loss_more_seeds.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_fuzz as f
bad = 0
t0 = time.time()
for seed in range(160, 1360):
    try:
        f.test_random_shape_against_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:300], flush=True)
    if seed % 100 == 0:
        print("seed", seed, "failures", bad, "%.0f s" % (time.time() - t0), flush=True)
print("done: 1200 seeds, failures", bad)
sys.exit(1 if bad else 0)
