"""Which XCD the workgroups of a launch land on (genie_where_am_i): workgroup b -> XCD b % 8 is what the XCD-chunked sweeps assume."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = torch.full((n, 2), -1, dtype=torch.int32, device="cuda:0")
_lib.check(lib.genie_where_am_i(ctypes.c_void_p(out.data_ptr()), n, ctypes.c_void_p(0)), "where")
torch.cuda.synchronize()
ids = out.cpu()[:, 0]
print("XCC id of blocks 0..31:", ids[:32].tolist())
print("blocks whose XCC id == block %% 8: %d of %d" % (int((ids == (torch.arange(n) % 8)).sum()), n))
