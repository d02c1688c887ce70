"""More seeds of tests/test_gpu_fuzz.py's generator than the test-suite runs (diagnostic, GPU).  Usage: fuzz_more.py [first] [count] [model_lights]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as f
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model_lights = len(sys.argv) > 3 and sys.argv[3] == "model_lights"
bad = 0
for seed in range(first, first + count):
    try:
        f.test_random_scene(seed, model_lights)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:200], flush=True)
print(f"{count - bad} of {count} seeds pass", flush=True)
sys.exit(1 if bad else 0)
