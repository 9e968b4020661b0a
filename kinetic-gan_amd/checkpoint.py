"""Checkpoints without stalling the training stream.

The reference saves ``generator.state_dict()`` / ``discriminator.state_dict()`` with a blocking ``torch.save`` inside the
training loop (kinetic-gan.py:189-192: a device -> host copy per tensor on the compute stream, then serialisation).
Here a snapshot is a device-side copy of every state tensor into ONE staging buffer, enqueued ON THE TRAINING STREAM
(a few multi-tensor copy kernels: the iteration enqueued behind save() - or a hipGraph replay, which rewrites the
state in place through raw pointers - starts only after them, so the snapshot cannot be torn; round-3 ADVICE), then one
device -> pinned-host copy on a side stream that waits for the staging copies (the training stream does NOT wait for
it), and a worker thread that serialises the host copy.  The files are what the reference writes: ``torch.save`` of an ordered dict with the
reference's ``state_dict`` keys, loadable by ``generate.py:66`` and by ``Module.load_state_dict``.
"""
import collections
import os
import queue
import threading
from typing import Optional

import torch


class AsyncCheckpointWriter:
    def __init__(self, max_pending: int = 2):
        self._q: "queue.Queue" = queue.Queue(maxsize=max_pending)
        self._err: Optional[BaseException] = None
        self._stream = None
        self._t = threading.Thread(target=self._run, name="kg-checkpoint", daemon=True)
        self._t.start()

    # ---- training-thread side -----------------------------------------------------------------------------------
    def save(self, module: torch.nn.Module, path: str) -> None:
        """Snapshot ``module.state_dict()`` as of the work enqueued so far and write it to ``path`` in the background."""
        self._raise_pending()
        sd = module.state_dict()
        items = [(k, v.detach()) for k, v in sd.items()]
        dev = items[0][1].device if items else torch.device("cpu")
        total = sum(-(-v.numel() * v.element_size() // 16) * 16 for _, v in items)       # every tensor starts 16-byte aligned
        if dev.type == "cuda":
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(dev)
            # staging copies on the training stream: ordered behind the work enqueued so far AND in front of whatever is
            # enqueued next (in-place updates of a replayed graph included)
            stage = torch.empty(total, dtype=torch.uint8, device=dev)
            off, layout, dsts, srcs = 0, [], [], []
            for k, v in items:
                n = v.numel() * v.element_size()
                dsts.append(stage[off:off + n].view(v.dtype).view(v.shape))
                srcs.append(v)
                layout.append((k, v.dtype, tuple(v.shape), off, n))
                off += -(-n // 16) * 16
            by_dtype = collections.OrderedDict()
            for dst, src in zip(dsts, srcs):
                by_dtype.setdefault(src.dtype, ([], []))
                by_dtype[src.dtype][0].append(dst)
                by_dtype[src.dtype][1].append(src)
            for dl, sl in by_dtype.values():
                torch._foreach_copy_(dl, sl)                 # one multi-tensor launch per dtype instead of one per tensor
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
                host.copy_(stage, non_blocking=True)
                stage.record_stream(self._stream)
                done = torch.cuda.Event()
                done.record(self._stream)
        else:
            host = torch.empty(total, dtype=torch.uint8)
            off, layout, done = 0, [], None
            for k, v in items:
                n = v.numel() * v.element_size()
                host[off:off + n].view(v.dtype).view(v.shape).copy_(v)
                layout.append((k, v.dtype, tuple(v.shape), off, n))
                off += -(-n // 16) * 16
        self._q.put((path, host, layout, done))

    def wait(self) -> None:
        """Block until every queued checkpoint is on disk (end of training, tests)."""
        self._q.join()
        self._raise_pending()

    def close(self) -> None:
        self.wait()
        self._q.put(None)
        self._t.join()

    # ---- worker ----------------------------------------------------------------------------------------------------
    def _run(self):
        while True:
            job = self._q.get()
            if job is None:
                self._q.task_done()
                return
            try:
                path, host, layout, done = job
                if done is not None:
                    done.synchronize()
                sd = collections.OrderedDict()
                for k, dtype, shape, off, n in layout:
                    sd[k] = host[off:off + n].view(dtype).view(shape).clone()
                tmp = path + ".tmp"
                torch.save(sd, tmp)
                os.replace(tmp, path)                  # a reader never sees a half-written file
            except BaseException as e:                 # surfaced on the training thread at the next save() / wait()
                self._err = e
            finally:
                self._q.task_done()

    def _raise_pending(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise RuntimeError("checkpoint writer failed") from e
