#!/usr/bin/env python3
"""Host-to-device rate of this box's link, to put the host-fed bench number in proportion.  Pinned host memory -> HBM with
hipMemcpyAsync (DMA engines), best of 5: one 1 GiB copy; then the shape of the bench's input ring — 78.6 MB pieces (256 frames of
640x480) — back to back on ONE stream, and side by side on 4 and 16 streams.  Prints GB/s."""
import time
import torch

n = 1 << 30
src = torch.empty(n, dtype=torch.uint8, pin_memory=True)
src.fill_(7)
dst = torch.empty(n, dtype=torch.uint8, device="cuda:0")


def best_of(fn, bytes_moved, reps=5):
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = max(best, bytes_moved / (time.perf_counter() - t0) / 1e9)
    return best


print("pinned host -> HBM, 1 GiB, DMA: %.1f GB/s" % best_of(lambda: dst.copy_(src, non_blocking=True), n))
piece = 256 * 640 * 480
k = n // piece
for n_streams in (1, 4, 16):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]

    def pieces():
        for i in range(k):
            with torch.cuda.stream(streams[i % n_streams]):
                dst[i * piece:(i + 1) * piece].copy_(src[i * piece:(i + 1) * piece], non_blocking=True)
    print("pinned host -> HBM, %d pieces of %.1f MB on %d stream(s): %.1f GB/s" % (k, piece / 1e6, n_streams, best_of(pieces, k * piece)))

# ---- the same 78.6 MB pieces on one stream while the GPU is busy: (a) kernels only (HBM-bound elementwise work on another stream),
# (b) kernels + many small pinned -> HBM copies on a third stream (the job records every launch of the tracker sends down),
# (c) small copies only — which of them takes the link's rate away from the big transfers?
import threading

work = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")
small_src = torch.empty(64 * 1024, dtype=torch.uint8, pin_memory=True)
small_dst = torch.empty(64 * 1024, dtype=torch.uint8, device="cuda:0")
stop = False


def kernels():
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        while not stop:
            work.mul_(1.0001)
            if torch.cuda.current_stream().query():
                pass
            time.sleep(0.0002)


def small_copies():
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(8):
                small_dst.copy_(small_src, non_blocking=True)
            st.synchronize()


one = [torch.cuda.Stream()]


def pieces1():
    for i in range(k):
        with torch.cuda.stream(one[0]):
            dst[i * piece:(i + 1) * piece].copy_(src[i * piece:(i + 1) * piece], non_blocking=True)


for label, fns in (("kernels on another stream", [kernels]), ("kernels + small copies", [kernels, small_copies]), ("small copies only", [small_copies])):
    stop = False
    ts = [threading.Thread(target=f) for f in fns]
    for t in ts:
        t.start()
    time.sleep(0.3)
    rates = []
    for _ in range(5):
        one[0].synchronize()
        t0 = time.perf_counter()
        pieces1()
        one[0].synchronize()
        rates.append(k * piece / (time.perf_counter() - t0) / 1e9)
    stop = True
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    print("pinned host -> HBM, %d pieces of %.1f MB on 1 stream, %s: %.1f GB/s (best), %.1f (worst)" % (k, piece / 1e6, label, max(rates), min(rates)))
