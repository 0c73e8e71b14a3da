import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.environ["GENIE_LIB_PATH"])
out = torch.full((64,), -1, dtype=torch.int32, device="cuda:0")
lib.genie_debug_xcc_map(ctypes.c_void_p(out.data_ptr()), 64, ctypes.c_void_p(0))
torch.cuda.synchronize()
print("XCC id of blocks 0..63:", out.cpu().tolist())
