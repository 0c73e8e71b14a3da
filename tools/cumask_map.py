#!/usr/bin/env python
"""What does a stream CU mask select on this GPU? For one-bit masks and a few word patterns: the set of (XCD, CU) a probe
launch of many workgroups lands on (genie_stream_create_masked + genie_where_am_i)."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie_amd import _lib
lib = _lib.load()
torch.zeros(1, device="cuda:0")


def where(bits, n_words=8, nblk=2048):
    words = (ctypes.c_uint32 * n_words)(*[(bits >> (32 * i)) & 0xffffffff for i in range(n_words)])
    st = ctypes.c_void_p()
    _lib.check(lib.genie_stream_create_masked(words, n_words, ctypes.byref(st)), "create")
    out = torch.full((nblk, 2), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    _lib.check(lib.genie_where_am_i(ctypes.c_void_p(out.data_ptr()), nblk, st), "probe")
    torch.cuda.synchronize()
    lib.genie_stream_destroy(st)
    o = out.cpu().numpy()
    xcc = o[:, 0]
    hw = o[:, 1].astype(np.uint32)
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7      # HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    pairs = sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    return pairs


for name, bits in (("all", (1 << 256) - 1), ("bit 0", 1), ("bit 1", 2), ("bit 5", 1 << 5), ("bit 31", 1 << 31), ("bit 32", 1 << 32),
                   ("bit 40", 1 << 40), ("bit 255", 1 << 255), ("bits 0-7", 0xff), ("word0", 0xffffffff), ("word1", 0xffffffff << 32)):
    p = where(bits)
    xs = sorted(set(a for a, _, _, _ in p))
    print("%-9s -> %3d distinct CUs on XCDs %s; first: %s" % (name, len(p), xs, p[:4]))


def block_map(bits, nblk=64, n_words=8):
    words = (ctypes.c_uint32 * n_words)(*[(bits >> (32 * i)) & 0xffffffff for i in range(n_words)])
    st = ctypes.c_void_p()
    _lib.check(lib.genie_stream_create_masked(words, n_words, ctypes.byref(st)), "create")
    out = torch.full((nblk, 2), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    _lib.check(lib.genie_where_am_i(ctypes.c_void_p(out.data_ptr()), nblk, st), "probe")
    torch.cuda.synchronize()
    lib.genie_stream_destroy(st)
    return out.cpu().numpy()[:, 0].tolist()


print("block -> XCD, unmasked-equivalent (all bits):", block_map((1 << 256) - 1)[:32])
print("block -> XCD, main mask (bits 8..255):      ", block_map(((1 << 256) - 1) ^ 0xff)[:32])
print("block -> XCD, main mask (bits 16..255):     ", block_map(((1 << 256) - 1) ^ 0xffff)[:32])
print("block -> XCD, tail mask (bits 0..7):        ", block_map(0xff)[:32])
