"""Host <-> HBM transfers of whole fields (the boundary hands over host arrays; SURVEY 8d: once the kernels take ~0.1 s the
wall time is storage -> HBM).  A pageable, strided NumPy slice goes through the driver's bounce buffers at a few GB/s;
here it is cut into chunks of rows that a few host threads copy into pinned staging buffers while the DMA engine moves the
previous chunk on its own HIP stream -- both directions, any dtype.  Plumbing only: no arithmetic happens here."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional

import numpy as np


class PinnedPipe:
    def __init__(self, device, chunk_bytes: int = 256 << 20, nbuf: int = 3, threads: int = 6):
        import torch

        self.torch = torch
        self.device = device
        self.chunk_bytes, self.nbuf = int(chunk_bytes), int(nbuf)
        self.pool = ThreadPoolExecutor(max_workers=threads)
        self.threads = threads
        self.stream = torch.cuda.Stream(device=device)
        self._bufs: Optional[List] = None

    def _buffers(self):
        if self._bufs is None:  # pinned once, reused by every transfer
            self._bufs = [self.torch.empty(self.chunk_bytes, dtype=self.torch.uint8, pin_memory=True) for _ in range(self.nbuf)]
        return self._bufs

    def _host_copy(self, dst: np.ndarray, src: np.ndarray) -> None:
        """dst[...] = src with the rows dealt out to the pool (NumPy releases the GIL while it copies)."""
        n = dst.shape[0]
        if n < 2 * self.threads or dst.nbytes < (8 << 20):
            np.copyto(dst, src, casting="unsafe")
            return
        step = (n + self.threads - 1) // self.threads
        jobs = [self.pool.submit(np.copyto, dst[a:a + step], src[a:a + step], "unsafe") for a in range(0, n, step)]
        for j in jobs:
            j.result()

    def upload(self, src: np.ndarray, dtype):
        """2-D host array (rows may be strided, any float dtype) -> contiguous device tensor of ``dtype``."""
        torch = self.torch
        T, n = src.shape
        tdt = getattr(torch, np.dtype(dtype).name)
        dev = torch.empty((T, n), dtype=tdt, device=self.device)
        if T == 0 or n == 0:
            return dev
        item = np.dtype(dtype).itemsize
        rows = max(1, self.chunk_bytes // (n * item))
        if n * item > self.chunk_bytes:  # a single row larger than a staging buffer: let torch handle it
            dev.copy_(torch.from_numpy(np.ascontiguousarray(src, dtype=dtype)))
            return dev
        bufs, events = self._buffers(), [None] * self.nbuf
        for i, t0 in enumerate(range(0, T, rows)):
            t1, b = min(T, t0 + rows), i % self.nbuf
            if events[b] is not None:
                events[b].synchronize()
            stage = bufs[b][: (t1 - t0) * n * item].view(tdt).reshape(t1 - t0, n)
            self._host_copy(stage.numpy(), src[t0:t1])
            with torch.cuda.stream(self.stream):
                dev[t0:t1].copy_(stage, non_blocking=True)
                events[b] = torch.cuda.Event()
                events[b].record(self.stream)
        self.stream.synchronize()
        return dev

    def download(self, t, dst: np.ndarray) -> None:
        """Device tensor (1-D or 2-D, possibly a strided view; the producing stream must have been synchronised) -> the
        host array ``dst`` of the same shape (a strided view of a larger array is fine)."""
        torch = self.torch
        if t.dim() == 1:
            t, dst = t.reshape(1, -1), dst.reshape(1, -1)
        T, n = t.shape
        if T == 0 or n == 0:
            return
        item = t.element_size()
        if n * item > self.chunk_bytes:
            np.copyto(dst, t.cpu().numpy(), casting="unsafe")
            return
        rows = max(1, self.chunk_bytes // (n * item))
        bufs = self._buffers()
        pending = []  # (event, staging view, t0, t1)

        def drain(k):
            while len(pending) > k:
                ev, stage, a, b = pending.pop(0)
                ev.synchronize()
                self._host_copy(dst[a:b], stage.numpy())

        for i, t0 in enumerate(range(0, T, rows)):
            t1, b = min(T, t0 + rows), i % self.nbuf
            drain(self.nbuf - 1)  # the buffer about to be reused has been copied out
            stage = bufs[b][: (t1 - t0) * n * item].view(t.dtype).reshape(t1 - t0, n)
            with torch.cuda.stream(self.stream):
                stage.copy_(t[t0:t1].contiguous(), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            pending.append((ev, stage, t0, t1))
        drain(0)
