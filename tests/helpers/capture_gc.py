"""Child process of tests/test_engine_gpu.py::test_graph_capture_survives_garbage_of_an_earlier_plan.

Round 3 saw `Fatal Python error: Aborted ... Garbage-collecting` under Engine.predict_step once: Python's cyclic
collector ran DURING a hipGraph capture and finalised an earlier engine's graph / events (destroying those is not a
capturable operation).  engine._capture collects first and keeps the collector off while the stream is capturing.  This
script builds that situation on purpose -- an earlier plan with a captured graph that is only reachable through a
reference cycle, and a collector that triggers on nearly every allocation -- and captures a second plan.  It runs as its
own process so that an abort is a return code, not a dead pytest session.  Prints CAPTURE_GC_OK on success."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from spnet_amd import engine as E  # noqa: E402


def main():
    import weakref
    H, W, B = 96, 128, 2
    x = torch.rand(B, H, W, 1, device="cuda") * 2 - 1
    gc.disable()                                 # the garbage made below must still exist when the capture is asked for
    a = E.Engine(H, W, B, device="cuda:0", seed=1, train=False)
    a.x_in.copy_(x)
    a.predict_step()
    y_a = a.predict_step().clone()               # second call: captured and replayed
    assert a._igraph is not None
    # a training plan as well: its side stream, events and pinned staging ring become garbage with it
    t = E.Engine(H, W, B, device="cuda:0", seed=1, train=True)
    t.train_step(x, torch.rand(B, 576, device="cuda"), 1e-5)
    torch.cuda.synchronize()
    dead = [weakref.ref(a), weakref.ref(t)]
    a.cycle, t.cycle = a, t                      # reachable only through reference cycles from here on
    del a, t
    b = E.Engine(H, W, B, device="cuda:0", seed=1, train=False)
    b.x_in.copy_(x)
    b.forward(None, training=False)              # eager: BatchNorm inference coefficients of this plan
    assert all(r() is not None for r in dead), "the garbage was collected too early for this test to mean anything"
    seen = {}
    fwd = b.forward

    def spying_forward(*args, **kw):             # what the capture sees
        if torch.cuda.is_current_stream_capturing():
            seen["gc_enabled"] = gc.isenabled()
            seen["garbage_alive"] = [r() is not None for r in dead]
            junk = [[i] for i in range(2000)]    # allocation pressure that would trigger a collection
            del junk
        return fwd(*args, **kw)

    gc.set_threshold(1, 1, 1)                    # the collector wants to run on almost every allocation ...
    gc.enable()                                  # ... from here on
    graph = E._capture(lambda: spying_forward(None, training=False))
    assert seen.get("gc_enabled") is False, "the cyclic collector was enabled while the stream was capturing"
    assert seen.get("garbage_alive") == [False, False], "earlier plans must be finalised BEFORE the capture starts"
    assert gc.isenabled(), "_capture must restore the collector"
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_a, b.out)
    # and through the public path (predict_step captures its own graph the same way)
    gc.set_threshold(700, 10, 10)
    y_b = b.predict_step()
    y_b = b.predict_step().clone()
    assert torch.equal(y_a, y_b)
    gc.collect()
    torch.cuda.synchronize()
    print("CAPTURE_GC_OK")


if __name__ == "__main__":
    main()
