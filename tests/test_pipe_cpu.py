"""CPU: the band loops' host-side overlap helpers keep order, apply back-pressure and surface worker exceptions."""
import threading
import time

import pytest

from bands.common.pipe import AsyncSink, prefetch


def test_prefetch_keeps_order_and_runs_ahead():
    started = []

    def load(i):
        started.append(i)
        time.sleep(0.01)
        return i * i

    seen = []
    for i, v in prefetch(load, range(8), depth=2):
        if i == 0:
            time.sleep(0.08)                    # the worker gets ahead, but never more than depth (+ the one in flight)
            assert 1 in started and len(started) <= 4
        seen.append((i, v))
    assert seen == [(i, i * i) for i in range(8)]


def test_prefetch_single_worker_thread():
    names = set()
    list(prefetch(lambda i: names.add(threading.current_thread().name), range(5)))
    assert names == {"band-prefetch"}


def test_prefetch_raises_in_consumer_and_stops():
    def load(i):
        if i == 2:
            raise ValueError("bad frame")
        return i

    got = []
    with pytest.raises(ValueError, match="bad frame"):
        for i, v in prefetch(load, range(6)):
            got.append(i)
    assert got == [0, 1]


def test_prefetch_early_exit_does_not_hang():
    for i, v in prefetch(lambda i: i, range(100), depth=1):
        if i == 3:
            break
    assert threading.active_count() < 8


def test_sink_runs_in_order_and_close_waits():
    out = []
    sink = AsyncSink(depth=2)
    for i in range(10):
        sink.submit(lambda k: (time.sleep(0.002), out.append(k)), i)
    sink.close()
    assert out == list(range(10))


def test_sink_reraises():
    sink = AsyncSink()
    sink.submit(lambda: (_ for _ in ()).throw(RuntimeError("disk full")))
    with pytest.raises(RuntimeError, match="disk full"):
        for _ in range(4):
            sink.submit(lambda: None)
        sink.close()
