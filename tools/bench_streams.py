#!/usr/bin/env python3
"""Experiment: the 256-frame step split over S contexts (S HIP streams, 256/S frames each) enqueued from one host thread.
Latency-bound kernels of one chunk (quadtree, small pyramid levels) can then overlap the vector-ALU-bound ones of another.
Same work per step as bench.py (256 frames, 255 pair matches; a chunk's extra pair is its last frame against its first)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from motioncheck_ccm_slam_amd import _lib, synth
from motioncheck_ccm_slam_amd.orb import ORBextractor
FRAMES, W, H = 256, 752, 480
lib = _lib.load()
frames = synth.frames(0, FRAMES)
img = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
for S in (1, 2, 4, 8):
    n = FRAMES // S
    ctxs = [_lib.Context(0) for _ in range(S)]
    exs = [ORBextractor(1000, 1.2, 8, 20, 7, ctx=c) for c in ctxs]
    m = exs[0].max_per_image
    outs = [tuple(torch.empty((n, m), dtype=torch.int32, device="cuda") for _ in range(3)) for _ in range(S)]
    def step():
        for k in range(S):
            exs[k].extract_dev(img.data_ptr() + k * n * W * H, W, H, W, W * H, n)
        for k in range(S):
            d, c, mm = exs[k].result_dev()
            bi, bd, sd = outs[k]
            np_in = n - 1
            ctxs[k].check(lib.ccm_hamming_match_dev(ctxs[k].handle, C.c_void_p(d), mm, C.c_size_t(mm), C.c_void_p(d + mm * 32), mm, C.c_size_t(mm), np_in,
                                                    C.c_void_p(c), C.c_void_p(c + 4), C.c_void_p(bi.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(sd.data_ptr())))
            if S > 1:    # the chunk's 128th pair
                ctxs[k].check(lib.ccm_hamming_match_dev(ctxs[k].handle, C.c_void_p(d + (n - 1) * mm * 32), mm, C.c_size_t(mm), C.c_void_p(d), mm, C.c_size_t(mm), 1,
                                                        C.c_void_p(c + 4 * (n - 1)), C.c_void_p(c), C.c_void_p(bi.data_ptr() + 4 * m * (n - 1)), C.c_void_p(bd.data_ptr() + 4 * m * (n - 1)), C.c_void_p(sd.data_ptr() + 4 * m * (n - 1))))
    def sync():
        for c in ctxs: c.sync()
    for _ in range(3): step()
    sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20): step()
        sync()
        best = min(best, (time.perf_counter() - t0) / 20)
    feats = sum(int(e.fetch()[2].sum()) for e in exs)
    print("streams=%d  %.4f ms/step  %.1f Mfeatures/s  (%d features)" % (S, best * 1e3, feats / best / 1e6, feats), flush=True)
    for c in ctxs: c.close()
