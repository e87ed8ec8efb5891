import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import faulthandler; faulthandler.enable()
import test_gpu_dist2 as T
root = tempfile.mkdtemp()
T._make_dataset(root + "/data")
res = {}
T._bulk_worker(0, 1, 36999, root + "/data", root + "/out", False, res)
print("worker done", len(res[0]), flush=True)
