"""Host-side overlap for the band loops: decode of the next chunk and encode / file writes of the previous one run
on worker threads while the GPU works on the current one.  The reference does all three one frame at a time on one thread
(bands/depth_anything.py:203-225, bands/common/io.py:246-305); at the engine's rates decode and libx264 become the bottleneck.
The engine calls (ctypes), numpy copies, decord and PyAV all release the GIL, so plain threads overlap for real.
Order is preserved: chunks come out, and writes go in, exactly in submission order; a worker's exception is re-raised in the caller."""
import queue
import threading


def prefetch(fn, items, depth=2):
    """yield (item, fn(item)) for item in items, with up to `depth` results computed ahead on ONE worker thread (so a decoder that is
    not thread-safe is only ever touched by that thread while the loop runs)."""
    items = list(items)
    q = queue.Queue(maxsize=max(1, depth))
    stop = threading.Event()

    def work():
        try:
            for it in items:
                if stop.is_set():
                    return
                q.put((it, fn(it), None))
        except BaseException as e:          # noqa: BLE001 - handed to the consumer
            q.put((None, None, e))

    t = threading.Thread(target=work, name="band-prefetch", daemon=True)
    t.start()
    try:
        for _ in items:
            it, val, err = q.get()
            if err is not None:
                raise err
            yield it, val
    finally:
        stop.set()
        while t.is_alive():                  # unblock a producer waiting on a full queue
            try:
                q.get_nowait()
            except queue.Empty:
                t.join(0.01)


class AsyncSink:
    """submit(fn, *args) runs the calls on one worker thread in submission order, at most `depth` pending (back-pressure);
    close() waits for them and re-raises the first exception."""

    def __init__(self, depth=2):
        self._q = queue.Queue(maxsize=max(1, depth))
        self._err = None
        self._t = threading.Thread(target=self._work, name="band-sink", daemon=True)
        self._t.start()

    def _work(self):
        while True:
            job = self._q.get()
            if job is None:
                return
            if self._err is None:
                try:
                    job[0](*job[1])
                except BaseException as e:      # noqa: BLE001 - re-raised by submit / close
                    self._err = e

    def submit(self, fn, *args):
        if self._err is not None:
            self.close()
        self._q.put((fn, args))

    def close(self):
        if self._t.is_alive():
            self._q.put(None)
            self._t.join()
        if self._err is not None:
            err, self._err = self._err, None
            raise err
