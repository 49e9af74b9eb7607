"""Where does the streaming feed's time go?  Aggregate GB/s of its stages in isolation, per reader-thread count:
(a) positional reads of a page-cache-hot file into pinned staging, (b) the same + the asynchronous H2D copy.

    python tools/feed_probe.py --gb 8 --threads 1 4 8 16 [--root /dev/shm]
"""
import argparse
import os
import tempfile
import threading
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    ap.add_argument("--chunk-mb", type=int, default=256)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 4, 8, 16])
    ap.add_argument("--root", default=None)
    ap.add_argument("--pageable", action="store_true", help="stage in pageable memory instead of pinned")
    a = ap.parse_args()
    chunk = a.chunk_mb << 20
    n_chunks = int(a.gb * 1e9) // chunk
    fd, path = tempfile.mkstemp(prefix="feedprobe_", dir=a.root)
    try:
        blk = np.random.default_rng(0).integers(0, 255, size=chunk, dtype=np.uint8)
        for _ in range(n_chunks):
            os.write(fd, blk)
        os.close(fd)
        dev = torch.device("cuda:0")
        for mode in ("read", "read+h2d"):
            for nt in a.threads:
                rfd = os.open(path, os.O_RDONLY)
                nxt = [0]
                lock = threading.Lock()

                def worker():
                    stage = [torch.empty(chunk, dtype=torch.uint8, pin_memory=not a.pageable) for _ in range(2)]
                    dst = [torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(2)]
                    stream = torch.cuda.Stream(dev)
                    evs = [None, None]
                    i = 0
                    while True:
                        with lock:
                            c = nxt[0]
                            nxt[0] += 1
                        if c >= n_chunks:
                            break
                        j = i & 1
                        i += 1
                        if evs[j] is not None:
                            evs[j].synchronize()
                        mv = memoryview(stage[j].numpy())
                        pos, off = 0, c * chunk
                        while pos < chunk:
                            got = os.preadv(rfd, [mv[pos:]], off + pos)
                            pos += got
                        if mode == "read+h2d":
                            with torch.cuda.stream(stream):
                                dst[j].copy_(stage[j], non_blocking=True)
                                evs[j] = torch.cuda.Event()
                                evs[j].record()
                    for e in evs:
                        if e is not None:
                            e.synchronize()

                ts = [threading.Thread(target=worker) for _ in range(nt)]
                t0 = time.perf_counter()
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                dt = time.perf_counter() - t0
                os.close(rfd)
                print(f"{mode:9s} {nt:2d} thread(s): {n_chunks * chunk / dt / 1e9:6.2f} GB/s", flush=True)
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
