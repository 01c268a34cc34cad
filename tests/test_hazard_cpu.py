"""Host logic of the capture-time hazard checker (propainter_amd/hazard.py) on hand-written submission programs: streams are plain
integers here, allocations / launches / events are fed through the same entry points the patched torch.cuda calls use on the GPU
(tests/test_modules_gpu.py runs it on real captures)."""
from propainter_amd.hazard import Recorder, _merge, _overlap

A, B = 0x1000, 0x9000


def _rec():
    return Recorder("cpu")


def test_recycled_block_read_on_another_stream_needs_an_edge():
    """The round-5 failure class: a block allocated on stream 1 is read on stream 2; the tensor dies; the caching allocator hands the
    block to the next allocation of stream 1, whose first write is not ordered behind stream 2's read."""
    r = _rec()
    r.on_alloc(A, A + 256, 1, "x")
    r.access("produce_x", [], [(A, A + 256, False)], sid=1)
    e = object()
    r.on_record(e, 1)
    r.on_wait(e, 2)
    r.access("consume_x", [(A, A + 256, False)], [], sid=2)
    r.on_alloc(A, A + 128, 1, "y")                      # recycled (a split of the block)
    r.access("produce_y", [], [(A, A + 128, False)], sid=1)
    rep = r.report()
    assert len(rep["alias"]) == 1 and "consume_x" in rep["alias"][0] and "produce_y" in rep["alias"][0], rep
    assert not rep["race"]


def test_the_same_program_with_a_join_is_clean_and_inherits_over_generations():
    r = _rec()
    r.on_alloc(A, A + 256, 1, "x")
    r.access("produce_x", [], [(A, A + 256, False)], sid=1)
    e = object(); r.on_record(e, 1); r.on_wait(e, 2)
    r.access("consume_x", [(A, A + 256, False)], [], sid=2)
    j = object(); r.on_record(j, 2); r.on_wait(j, 1)   # stream 1 waits for the reader before the block is reused
    r.on_alloc(A, A + 256, 1, "y")
    r.access("produce_y", [], [(A, A + 256, False)], sid=1)
    # a third life on stream 3 that never synchronised with anything: both earlier lives constrain it
    r.on_alloc(A, A + 256, 3, "z")
    r.access("produce_z", [], [(A, A + 256, False)], sid=3)
    rep = r.report()
    assert len(rep["alias"]) >= 1 and all("produce_z" in m for m in rep["alias"]), rep


def test_same_stream_reuse_and_host_sync_are_safe():
    r = _rec()
    r.on_alloc(A, A + 64, 1, "x")
    r.access("k1", [], [(A, A + 64, False)], sid=1)
    r.on_alloc(A, A + 64, 1, "y")                       # stream order protects a same-stream reuse
    r.access("k2", [], [(A, A + 64, False)], sid=1)
    r.access("k3", [(A, A + 64, False)], [], sid=2)     # unordered read of y on stream 2: a race with k2 (and unordered with the block's first life too)
    r.on_host_sync(None)
    r.on_alloc(A, A + 64, 2, "z")                       # after a device synchronise anything goes
    r.access("k4", [], [(A, A + 64, False)], sid=2)
    rep = r.report()
    assert all(m.startswith("k3") for m in rep["alias"]) and len(rep["race"]) == 1 and "k3" in rep["race"][0], rep
    assert not any("k2" in m.split(" touches ")[0] or "k4" in m.split(" touches ")[0] for m in rep["alias"]), rep


def test_race_inside_one_generation_and_disjoint_slices():
    r = _rec()
    r.on_alloc(B, B + 1024, 1, "buf")
    r.access("w_lo", [], [(B, B + 512, False)], sid=1)
    r.access("w_hi", [], [(B + 512, B + 1024, False)], sid=2)        # disjoint halves on two lanes: fine
    assert not r.report()["race"]
    r.access("r_all", [(B, B + 1024, False)], [], sid=3)              # reads both halves without waiting for either lane
    rep = r.report()
    assert len(rep["race"]) == 2, rep
    # two strided channel windows of one NHWC buffer overlap as byte ranges: reported apart, as "race?"
    r2 = _rec()
    r2.on_alloc(B, B + 1024, 1, "nhwc")
    r2.access("conv_a", [], [(B, B + 1024, True)], sid=1)
    r2.access("conv_b", [], [(B + 16, B + 1024, True)], sid=2)
    rep2 = r2.report()
    assert not rep2["race"] and len(rep2["race?"]) == 1


def test_graph_launches_in_flight_together():
    r = _rec()
    g1 = {"r": _merge([(A, A + 64)]), "w": _merge([(B, B + 64)]), "n": 3}
    g2 = {"r": _merge([(B, B + 64)]), "w": _merge([(A + 4096, A + 5000)]), "n": 2}
    r.names.update({1: "s1", 2: "s2"})
    r._sid = lambda stream=None: r._cur                 # (the GPU form asks torch for the current stream)
    r._cur = 1; r.graph_access(g1)
    r._cur = 2; r.graph_access(g2)                      # reads what g1 writes, nothing orders the two launches
    rep = r.report()
    assert len(rep["race"]) == 1 and "in flight together" in rep["race"][0], rep
    r = _rec(); r.names.update({1: "s1", 2: "s2"}); r._sid = lambda stream=None: r._cur
    r._cur = 1; r.graph_access(g1)
    e = object(); r.on_record(e, 1); r.on_wait(e, 2)
    r._cur = 2; r.graph_access(g2)
    assert not r.report()["race"]
    assert _overlap(_merge([(0, 4), (4, 8), (20, 30)]), _merge([(8, 20)])) is None and _merge([(0, 4), (4, 8)]) == [[0, 8]]


def test_uninitialised_read_and_capture_scoped_accesses():
    r = _rec()
    r.on_alloc(A, A + 256, 1, "torch.empty")
    r.access("reader", [(A, A + 128, False)], [], sid=1)               # nothing has written this life of the block
    r.access("writer", [], [(A, A + 256, False)], sid=1)
    r.access("reader2", [(A + 64, A + 128, False)], [], sid=1)
    rep = r.report()
    assert len(rep["uninit"]) == 1 and rep["uninit"][0].startswith("reader "), rep
    # accesses recorded under a graph capture describe the graph: after the capture they neither race with nor order eager work
    r = _rec()
    r.on_alloc(B, B + 64, 1, "static out")
    r.cap_id = r.ncap = 1
    r.capturing = {"r": [], "w": [], "n": 0}
    r.access("captured_writer", [], [(B, B + 64, False)], sid=7)        # the capture stream
    r.capturing = None
    r.end_capture()
    r.access("eager_reader", [(B, B + 64, False)], [], sid=1)           # e.g. torch.equal(graph.replay(), ref) on the default stream
    rep = r.report()
    assert not rep["race"] and not rep["alias"] and not rep["uninit"], rep
