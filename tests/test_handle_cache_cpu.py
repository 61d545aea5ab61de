"""CPU: the bounded LRU cache of parked roots handles (lightzero_amd/mcts/ctree/_tree_common.py) with the library calls
replaced by a recorder: at most LZ_HANDLE_CACHE_MAX handles stay parked, at most two per shape, the least recently parked one is
destroyed first, unpark returns the most recent handle of the shape, flush destroys everything (optionally of one engine)."""
import types


def test_park_unpark_evict_flush(monkeypatch):
    from lightzero_amd.mcts.ctree import _tree_common as tc
    destroyed = []
    fake = types.SimpleNamespace(lz_roots_destroy=lambda h: destroyed.append(h))
    monkeypatch.setattr(tc.L, "lib", lambda: fake)
    monkeypatch.setattr(tc, "_HANDLE_CACHE", {})
    monkeypatch.setattr(tc, "_PARK_ORDER", tc.collections.OrderedDict())
    monkeypatch.setattr(tc, "_HANDLE_CACHE_MAX", 4)
    key = lambda eng, b: (eng, 0, b, 6, 50)
    assert tc._park(key(1, 10), 101, 7) and tc._park(key(1, 10), 102, 8)
    assert not tc._park(key(1, 10), 103, 9)            # two per shape: the caller destroys the third
    assert tc._unpark(key(1, 10)) == (102, 8)          # most recent first
    assert tc._park(key(1, 10), 102, 8)
    assert tc._park(key(1, 11), 111, 1) and tc._park(key(2, 12), 121, 2)
    assert tc.handle_cache_size() == 4 and destroyed == []
    assert tc._park(key(2, 13), 131, 3)                # a fifth handle: the least recently parked one (101) goes
    assert destroyed == [101] and tc.handle_cache_size() == 4
    assert tc._unpark(key(1, 10)) == (102, 8) and tc._unpark(key(1, 10)) is None
    assert tc.flush_handle_cache(2) == 2 and sorted(destroyed[1:]) == [121, 131]   # engine 2's handles only
    assert tc.handle_cache_size() == 1 and tc._unpark(key(1, 11)) == (111, 1)
    assert tc.flush_handle_cache() == 0 and tc.handle_cache_size() == 0
    # a disabled cache parks nothing
    monkeypatch.setattr(tc, "_HANDLE_CACHE_MAX", 0)
    assert not tc._park(key(1, 10), 201, 0)


def test_thread_local_random_source():
    """_lib.rs(): the global np.random unless THIS thread installed a stream of its own (pipelined collector groups)"""
    import threading
    import numpy as np
    from lightzero_amd import _lib as L
    assert L.rs() is np.random
    seen = {}

    def worker():
        with L.random_source(np.random.RandomState(5)) as rs:
            seen["inside"] = L.rs() is rs
            seen["draw"] = float(L.rs().rand())
        seen["after"] = L.rs() is np.random
    t = threading.Thread(target=worker)
    t.start(); t.join()
    assert seen == {"inside": True, "draw": float(np.random.RandomState(5).rand()), "after": True}
    assert L.rs() is np.random   # the main thread never saw the worker's stream
