"""One-off campaign: tests/test_gpu_random_scenes.py::test_scheduler_paths_agree_on_random_mesh_worlds over many seeds."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random_scenes as T
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    try:
        T.test_scheduler_paths_agree_on_random_mesh_worlds(seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, str(e)[:200], flush=True)
print("scheduler seeds %d..%d: %d mismatches, %.1fs" % (first, first + count - 1, bad, time.time() - t0), flush=True)
