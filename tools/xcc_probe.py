import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.environ["GENIE_LIB_PATH"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
lib.genie_debug_xcc_map(ctypes.c_void_p(out.data_ptr()), n, ctypes.c_void_p(0))
torch.cuda.synchronize()
ids = out.cpu()
print("XCC id of blocks 0..31:", ids[:32].tolist())
import torch as t
print("blocks whose XCC id == block %% 8: %d of %d" % (int((ids == (t.arange(n) % 8)).sum()), n))
