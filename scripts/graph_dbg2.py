import faulthandler, os, sys
faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "mt"
if mode == "st":
    torch.autograd.set_multithreading_enabled(False)
import test_gpu_graph as t
t.test_roi_head_half_with_backward_replays_identically()
print("OK", mode)
