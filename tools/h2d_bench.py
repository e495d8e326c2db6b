#!/usr/bin/env python3
"""The H2D-inclusive leg of bench.py alone (page-locked host frames in, all results back on the host every step), next to the
raw link rate of this box: one 59 MB pinned upload (hipMemcpyAsync) timed alone and with a concurrent download.
usage: python tools/h2d_bench.py [--handles N] [--h2d-steps K]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import numpy as np
    import torch
    import orb_slam3_fast_amd as orbx
    a = bench.parse(sys.argv[1:])
    wl = bench.Workload(a)
    for _ in range(6):
        wl.step()
    wl.sync()
    out = bench.h2d_leg(a, wl, orbx, np, torch)
    # raw link: the same 2B frames as one linear pinned copy, alone and against a concurrent 7.3 MB download
    B, W, H = a.pairs, a.width, a.height
    host = torch.empty(2 * B * W * H, dtype=torch.uint8).pin_memory()
    dev = torch.empty_like(host, device="cuda")
    back_d = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    back_h = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for both in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            with torch.cuda.stream(s1):
                dev.copy_(host, non_blocking=True)
            if both:
                with torch.cuda.stream(s2):
                    back_h.copy_(back_d, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        out["raw_upload_ms" + ("_with_download" if both else "")] = round(1e3 * dt, 4)
        out["raw_upload_GBps" + ("_with_download" if both else "")] = round(host.numel() / dt / 1e9, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
