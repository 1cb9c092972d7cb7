"""Dynamic micro-batching of concurrent `generate` calls (SURVEY §8(b) conventions, §8(e), §8(f)2).

The reference executes requests strictly one at a time (every endpoint blocks the event loop inside `do_whisper`,
main.py:1205,1338), so its replica pool is never fed concurrently.  Here each model owns ONE queue of pending
utterances and one worker thread per GPU replica (several replicas may share a GPU: `wis_model_clone`).  A worker that takes
work runs the oldest pending utterance and every other queued utterance with the same batch key (prompt length, beam size,
decoding options, input kind) up to the device batch capacity as one `wis_generate`.

Burst behaviour (BASELINE.json configs[3]: 64 utterances arriving together on 8 GPUs x 4 replicas).  With "every idle worker
grabs what is there" the first arrivals of a burst are taken alone by 32 idle workers - 32 device batches of 1-2.  So:

* ONE taker per arrival: `submit` wakes a single idle worker (on the least loaded GPU) per batch worth of unclaimed work, never
  the whole pool; idle replicas of a GPU that already has a taker filling its batch stay asleep.
* A taker LINGERS for its batch to fill, but only when more work is plausibly on its way: other requests are in flight (so this
  is not a lone client) and the queue grew within the last `QUIET_S` (300 us; 1 ms when the GPU is busy).  It goes as soon as the
  batch is full or the arrivals stop, at the latest after `LINGER_IDLE_S` (0.5 ms) when its GPU is idle, `LINGER_BUSY_S` (2 ms) when other batches are
  running on that GPU anyway (the wait then costs the GPU nothing).
* A lone request is never delayed: nothing else in flight => it is taken at once (tests/test_server_cpu.py).
"""
import threading
import time
from collections import deque

_IDLE, _WOKEN, _LINGER, _RUN = range(4)


class _Item:
    __slots__ = ("key", "payload", "done", "result", "error", "affinity", "call")

    def __init__(self, key, payload, affinity=None, call=None):
        self.key, self.payload, self.affinity, self.call = key, payload, affinity, call
        self.done = threading.Event()
        self.result = self.error = None


class _Worker:
    __slots__ = ("idx", "ctx", "group", "state", "cv", "cap", "thread")

    def __init__(self, idx, ctx, group, lock):
        self.idx, self.ctx, self.group = idx, ctx, group
        self.state, self.cap = _IDLE, 0
        self.cv = threading.Condition(lock)
        self.thread = None


class MicroBatcher:
    """run(worker_ctx, key, [payload, ...]) -> [result, ...] is called on a worker thread, one thread per worker_ctx
    (= GPU replica).  capacity(key) bounds the number of payloads per call.  groups[i] = the GPU worker i runs on (default:
    `worker_ctx.device` when it has one, else the worker is a GPU of its own)."""

    QUIET_S = 300e-6          # no arrival for this long: a lingering taker stops waiting (idle GPU)
    QUIET_BUSY_S = 1e-3       # ... on a GPU that is running other batches (their clients resubmit about a millisecond apart)
    LINGER_IDLE_S = 0.5e-3    # longest wait for a batch to fill on an otherwise idle GPU
    LINGER_BUSY_S = 2e-3      # ... on a GPU that is running other device batches anyway

    def __init__(self, workers, run, capacity, groups=None):
        self._run, self._capacity = run, capacity
        self._q = deque()
        self._lock = threading.Lock()
        self._stop = False
        self._calls = 0                        # submit() calls not yet answered
        self._last_arrival = 0.0
        self.batches = []                      # (worker index, batch size) log, bounded; for tests and stats
        self.lingers = 0                       # batches whose taker waited for more work (stats)
        if groups is None:
            groups = [getattr(w, "device", None) for w in workers]
            groups = [g if g is not None else ("w", i) for i, g in enumerate(groups)]
        self._workers = [_Worker(i, w, g, self._lock) for i, (w, g) in enumerate(zip(workers, groups))]
        self._running = {}                     # group -> device batches running
        for w in self._workers:
            self._running.setdefault(w.group, 0)
            w.thread = threading.Thread(target=self._loop, args=(w,), daemon=True, name=f"wis-batcher-{w.idx}")
        self._threads = [w.thread for w in self._workers]
        for t in self._threads:
            t.start()

    # ---- client side ------------------------------------------------------------------
    def submit(self, key, payloads, affinity=None):
        """Blocks until every payload has a result; results come back in payload order.  `affinity`: a worker context (GPU replica)
        that must run these payloads, or ("device", g): any worker of GPU g - e.g. features that already live in that GPU's memory;
        None = any worker."""
        call = object()
        items = [_Item(key, p, affinity, call) for p in payloads]
        with self._lock:
            if self._stop:
                raise RuntimeError("batcher is closed")
            self._calls += 1
            self._q.extend(items)
            self._last_arrival = time.perf_counter()
            self._dispatch()
        try:
            for it in items:
                it.done.wait()
        finally:
            with self._lock:
                self._calls -= 1
        for it in items:
            if it.error is not None:
                raise it.error
        return [it.result for it in items]

    def load(self, group=None):
        """(items queued, device batches running[, on GPU `group`]) - what optional work (a streaming session's speculative interim decodes)
        looks at before it adds itself to the queue"""
        with self._lock:
            return len(self._q), (sum(self._running.values()) if group is None else self._running.get(group, 0))

    # ---- scheduling (all under self._lock) ------------------------------------------------
    def _mine(self, w, it):
        a = it.affinity
        return a is None or a is w.ctx or (isinstance(a, tuple) and len(a) == 2 and a[0] == "device" and a[1] == w.group)

    def _takeable(self, w):
        return any(self._mine(w, it) for it in self._q)

    def _unclaimed_free(self):
        """queued items any worker may run, minus what the workers already woken / filling a batch will take"""
        n = sum(1 for it in self._q if it.affinity is None)
        return n - sum(w.cap for w in self._workers if w.state in (_WOKEN, _LINGER))

    def _dispatch(self):
        """Wake exactly the workers the queue needs: a lingering taker whose batch may be complete now, the owner of rows bound to
        one replica, and ONE idle worker per batch of unclaimed free work (least loaded GPU first, GPUs without a taker first)."""
        for w in self._workers:
            if w.state == _LINGER:
                w.cv.notify()
        for it in self._q:
            if it.affinity is not None:
                mine = [w for w in self._workers if self._mine(w, it)]
                if not any(w.state in (_WOKEN, _LINGER) for w in mine):      # (one taker for the rows bound to a replica / a GPU)
                    idle = [w for w in mine if w.state == _IDLE]
                    if idle:
                        idle[0].state, idle[0].cap = _WOKEN, 0
                        idle[0].cv.notify()
        while self._unclaimed_free() > 0:
            idle = [w for w in self._workers if w.state == _IDLE]
            if not idle:
                return
            taker_groups = {w.group for w in self._workers if w.state in (_WOKEN, _LINGER)}
            w = min(idle, key=lambda w: (w.group in taker_groups, self._running[w.group], w.idx))
            first = next(it for it in self._q if it.affinity is None)
            w.state, w.cap = _WOKEN, max(1, int(self._capacity(first.key)))
            w.cv.notify()

    def _bound_rows_waiting(self, w):
        """rows bound to this worker's replica / GPU that no other worker is about to take"""
        for it in self._q:
            if it.affinity is not None and self._mine(w, it):
                if not any(o is not w and o.state in (_WOKEN, _LINGER) and self._mine(o, it) for o in self._workers):
                    return True
        return False

    def _take(self, w):
        """Oldest item this worker may run + every queued item with the same key it may run, up to capacity (queue order
        preserved for the rest)."""
        first, batch, rest, cap = None, [], deque(), 1
        while self._q:
            it = self._q.popleft()
            mine = self._mine(w, it)
            if first is None and mine:
                first, batch = it, [it]
                cap = max(1, int(self._capacity(first.key)))
            elif first is not None and mine and it.key == first.key and len(batch) < cap:
                batch.append(it)
            else:
                rest.append(it)
        self._q = rest
        return batch

    def _fill(self, w):
        """-> (items of the batch this worker would take now, its capacity)"""
        first, n, cap = None, 0, 1
        for it in self._q:
            if not self._mine(w, it):
                continue
            if first is None:
                first, n, cap = it, 1, max(1, int(self._capacity(it.key)))
            elif it.key == first.key:
                n += 1
        return min(n, cap), cap, first

    def _linger(self, w):
        """wait (lock released inside cv.wait) for this taker's batch to fill while more work is plausibly arriving"""
        t0 = time.perf_counter()
        waited = False
        while not self._stop:
            n, cap, first = self._fill(w)
            if first is None or n >= cap:
                break
            w.cap = cap
            calls_in_batch = len({it.call for it in self._q if self._mine(w, it) and it.key == first.key})
            others_in_flight = self._calls - calls_in_batch > 0
            busy = self._running[w.group] > 0
            now = time.perf_counter()
            quiet_left = (self.QUIET_BUSY_S if busy else self.QUIET_S) - (now - self._last_arrival)
            limit_left = (self.LINGER_BUSY_S if busy else self.LINGER_IDLE_S) - (now - t0)
            if not (others_in_flight or busy) or quiet_left <= 0 or limit_left <= 0:
                break
            waited = True
            w.cv.wait(min(quiet_left, limit_left))
        if waited:
            self.lingers += 1

    def _loop(self, w):
        lock = self._lock
        while True:
            with lock:
                while True:
                    if self._stop and not self._takeable(w):
                        w.state = _IDLE
                        return
                    if self._takeable(w) and (w.state == _WOKEN or self._unclaimed_free() > 0 or self._bound_rows_waiting(w)):
                        break
                    w.state, w.cap = _IDLE, 0
                    # (while shutting down: re-check on a timer - close() notifies once, and a worker that woke while another one was
                    # lingering over the same rows would otherwise sleep through the end of the queue and hold close() in join())
                    w.cv.wait(0.05 if self._stop else None)
                w.state = _LINGER
                w.cap = self._fill(w)[1]
                self._linger(w)
                batch = self._take(w)
                if not batch:                      # (another worker took the rows while this one waited)
                    w.state, w.cap = _IDLE, 0
                    continue
                w.state, w.cap = _RUN, 0
                self._running[w.group] += 1
                self._dispatch()                   # leftovers: the next taker
            try:
                out = self._run(w.ctx, batch[0].key, [it.payload for it in batch])
                if len(out) != len(batch):
                    raise RuntimeError(f"batch of {len(batch)} returned {len(out)} results")
                for it, r in zip(batch, out):
                    it.result = r
            except BaseException as e:             # every caller of the failed batch sees the error
                for it in batch:
                    it.error = e
            finally:
                with lock:
                    self._running[w.group] -= 1
                    w.state = _IDLE
                    if len(self.batches) < 4096:
                        self.batches.append((w.idx, len(batch)))
                for it in batch:
                    it.done.set()

    def close(self):
        with self._lock:
            self._stop = True
            for w in self._workers:
                w.cv.notify()
        for t in self._threads:
            if t is not threading.current_thread():
                t.join(timeout=5)
