"""Dynamic micro-batching of concurrent `generate` calls (SURVEY §8(b) conventions, §8(e), §8(f)2).

The reference executes requests strictly one at a time (every endpoint blocks the event loop inside `do_whisper`,
main.py:1205,1338), so its replica pool is never fed concurrently.  Here each model owns ONE queue of pending
utterances; every GPU replica runs a worker that, whenever its GPU becomes free, takes the oldest pending utterance and every
other queued utterance with the same batch key (prompt length, beam size, decoding options, input kind) up to the device
batch capacity, and runs them as one `wis_generate`.  A lone request is never delayed (no timer: batches form only from work
that queued up while the GPU was busy), under load the batch grows by itself - config 4 of BASELINE.json (64 concurrent
utterances, 8 per GPU) is exactly this.
"""
import threading
from collections import deque


class _Item:
    __slots__ = ("key", "payload", "done", "result", "error", "affinity")

    def __init__(self, key, payload, affinity=None):
        self.key, self.payload, self.affinity = key, payload, affinity
        self.done = threading.Event()
        self.result = self.error = None


class MicroBatcher:
    """run(worker_ctx, key, [payload, ...]) -> [result, ...] is called on a worker thread, one thread per worker_ctx
    (= GPU replica).  capacity(key) bounds the number of payloads per call."""

    def __init__(self, workers, run, capacity):
        self._run, self._capacity = run, capacity
        self._q = deque()
        self._cv = threading.Condition()
        self._stop = False
        self.batches = []                      # (worker index, batch size) log, bounded; for tests and stats
        self._threads = [threading.Thread(target=self._loop, args=(i, w), daemon=True, name=f"wis-batcher-{i}") for i, w in enumerate(workers)]
        for t in self._threads:
            t.start()

    def submit(self, key, payloads, affinity=None):
        """Blocks until every payload has a result; results come back in payload order.  `affinity`: the worker context (GPU
        replica) that must run these payloads - e.g. features that already live in that GPU's memory; None = any worker."""
        items = [_Item(key, p, affinity) for p in payloads]
        with self._cv:
            if self._stop:
                raise RuntimeError("batcher is closed")
            self._q.extend(items)
            self._cv.notify_all()
        for it in items:
            it.done.wait()
        for it in items:
            if it.error is not None:
                raise it.error
        return [it.result for it in items]

    def _takeable(self, ctx):
        return any(it.affinity is None or it.affinity is ctx for it in self._q)

    def _take(self, ctx):
        """Oldest item this worker may run + every queued item with the same key it may run, up to capacity (queue order
        preserved for the rest)."""
        first, batch, rest, cap = None, [], deque(), 1
        while self._q:
            it = self._q.popleft()
            mine = it.affinity is None or it.affinity is ctx
            if first is None and mine:
                first, batch = it, [it]
                cap = max(1, int(self._capacity(first.key)))
            elif first is not None and mine and it.key == first.key and len(batch) < cap:
                batch.append(it)
            else:
                rest.append(it)
        self._q = rest
        return batch

    def _loop(self, idx, ctx):
        while True:
            with self._cv:
                while not self._takeable(ctx) and not self._stop:
                    self._cv.wait()
                if self._stop and not self._takeable(ctx):
                    return
                batch = self._take(ctx)
            try:
                out = self._run(ctx, batch[0].key, [it.payload for it in batch])
                if len(out) != len(batch):
                    raise RuntimeError(f"batch of {len(batch)} returned {len(out)} results")
                for it, r in zip(batch, out):
                    it.result = r
            except BaseException as e:             # every caller of the failed batch sees the error
                for it in batch:
                    it.error = e
            finally:
                if len(self.batches) < 4096:
                    self.batches.append((idx, len(batch)))
                for it in batch:
                    it.done.set()

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        for t in self._threads:
            if t is not threading.current_thread():
                t.join(timeout=5)
