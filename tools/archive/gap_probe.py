#!/usr/bin/env python
"""What does an event record / a cross-stream wait cost between two back-to-back kernels of one stream? (the 12-us gap of
the window pipeline after stage 2)"""
import time
import torch
dev = "cuda:0"
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
small = torch.zeros(1024, device=dev)


def run(mode, n=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = None
    for i in range(n):
        torch.mul(x, 1.0001, out=y)
        if mode in ("record", "record+side", "record+side+wait"):
            e = torch.cuda.Event()
            e.record(main)
            if mode != "record":
                side.wait_event(e)
                with torch.cuda.stream(side):
                    small.add_(1.0)
                    d = torch.cuda.Event()
                    d.record(side)
                if mode == "record+side+wait" and prev is not None:
                    main.wait_event(prev)
                prev = d
        torch.mul(y, 0.9999, out=x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for _ in range(2):
    for m in ("none", "record", "record+side", "record+side+wait"):
        print("%-18s %.1f us per pair" % (m, run(m)))
