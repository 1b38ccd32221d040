"""What does hipEventQuery (torch.cuda.Event.query) from ANOTHER thread return while a stream is being captured?  (ProcessGroupNCCL's watchdog polls
the end events of its outstanding Work objects every 100 ms; round 4's driver run lost the suite to `hipErrorCapturedEvent` thrown there.)
One capture per case: a failed query may invalidate the capture it lands in."""
import threading
import torch

def q(ev):
    out = []
    def f():
        try:
            out.append(("ok", ev.query()))
        except Exception as e:
            out.append(("ERR", str(e).splitlines()[0][:100]))
    t = threading.Thread(target=f); t.start(); t.join()
    return out[0]

x = torch.zeros(1 << 20, device="cuda")
for mode in ("global", "thread_local", "relaxed"):
    print("capture_error_mode =", mode, flush=True)
    for case in ("eager event of the stream that is now capturing", "eager event of an unrelated stream",
                 "event captured earlier, re-recorded eagerly on another stream", "eager event of a stream that JOINS the capture"):
        s, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        e = torch.cuda.Event()
        if case.startswith("eager event of the stream that is now"):
            with torch.cuda.stream(s):
                x += 1; e.record(s)
        elif case.startswith("event captured earlier"):
            g0 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g0, stream=s, capture_error_mode=mode):
                x += 1
                e.record(s)
            with torch.cuda.stream(s2):
                x += 1; e.record(s2)
        else:
            with torch.cuda.stream(s2):
                x += 1; e.record(s2)
        torch.cuda.synchronize()
        before = q(e)
        g = torch.cuda.CUDAGraph()
        res = end = None
        try:
            with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
                x += 1
                if "JOINS" in case:
                    ev = torch.cuda.Event(); ev.record(s); s2.wait_event(ev)
                    with torch.cuda.stream(s2):
                        x += 1
                res = q(e)
                if "JOINS" in case:
                    ev2 = torch.cuda.Event(); ev2.record(s2); s.wait_event(ev2)
            end = "capture ok"
        except Exception as ex:
            end = "capture FAILED: " + str(ex).splitlines()[0][:80]
        torch.cuda.synchronize()
        print("    %-62s before %s | during %s | %s" % (case, before, res, end), flush=True)
