"""More seeds of tests/test_gpu_fuzz.py than the suite runs: python tools/fuzz_long.py [first last]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import test_gpu_fuzz as F
a = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = int(sys.argv[2]) if len(sys.argv) > 2 else 400
t = time.time(); bad = []
for seed in range(a, b):
    try:
        F.test_random_configuration(seed)
    except Exception as e:                      # keep going: report every failing seed
        bad.append((seed, repr(e)[:200]))
for seed in range(a, b):
    try:
        F.test_random_configuration_persistent_kernel(seed)
    except Exception as e:
        bad.append(("persistent", seed, repr(e)[:200]))
for seed in range(a, a + (b - a) // 10):
    try:
        F.test_random_large_catalogue_compressed(seed)
    except Exception as e:
        bad.append(("large", seed, repr(e)[:200]))
for seed in range(a, a + (b - a) // 5):
    try:
        F.test_random_configuration_cells(seed)
    except Exception as e:
        bad.append(("cells", seed, repr(e)[:200]))
print("seeds %d..%d: %d failures in %.0f s" % (a, b, len(bad), time.time() - t), flush=True)
for x in bad:
    print(x)
