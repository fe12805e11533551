"""What a boundary between two launches costs on this stack (ROCm 7.2, torch CUDAGraph.replay), with kernels long enough
(~35 us) that the dispatcher is not the limit: chains of 10 kernels on one stream, cut in the middle in different ways.
Prints the extra time per cut against the uncut chain."""
import torch
dev = "cuda:0"
x = torch.zeros(48_000_000, device=dev)            # 192 MB read + write per kernel: ~35 us
y = torch.zeros(1_000_000, device=dev)
def k(t=x):
    t.mul_(1.0001)
for _ in range(5): k()
torch.cuda.synchronize()
def timed(fn, n=40):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def graph_of(n, t=x):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): k(t)
    return g
side = torch.cuda.Stream()
ev = torch.cuda.Event()
base_eager = timed(lambda: [k() for _ in range(10)])
g10 = graph_of(10)
base = timed(lambda: g10.replay())
print(f"10 kernels eager {base_eager:7.1f} us, as one graph {base:7.1f} us ({base / 10:.1f} us per kernel)")
ga, gb = graph_of(5), graph_of(5)
def report(name, fn, cuts=1):
    t = timed(fn)
    print(f"{name:70s} {t:7.1f} us  -> {(t - base) / cuts:6.1f} us per cut")
main = torch.cuda.current_stream()
report("graph(5) | graph(5)", lambda: (ga.replay(), gb.replay()))
report("graph(5) | event record | graph(5)", lambda: (ga.replay(), ev.record(main), gb.replay()))
def with_fork():
    ga.replay(); side.wait_stream(main); gb.replay()
report("graph(5) | side.wait_stream(main) | graph(5)", with_fork)
def with_join_done():
    with torch.cuda.stream(side):
        k(y)                                         # a tiny kernel on the side stream, long finished when main gets there
    ga.replay(); main.wait_stream(side); gb.replay()
report("graph(5) | main.wait_stream(side: finished long ago) | graph(5)", with_join_done)
gs = None
with torch.cuda.stream(side):
    pass
g_side = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_side, stream=side):
    for _ in range(5): k(y)
def with_join_late():
    side.wait_stream(main)
    ga.replay()
    with torch.cuda.stream(side):
        g_side.replay(); k(x)                        # the side stream ends ~35 us AFTER graph a
    main.wait_stream(side); gb.replay()
t = timed(with_join_late)
print(f"{'graph(5) | main.wait_stream(side ending one kernel later) | graph(5)':70s} {t:7.1f} us  -> {t - base - base / 10:6.1f} us beyond the extra kernel")
report("graph(5) | eager kernel x5", lambda: (ga.replay(), [k() for _ in range(5)]))
report("eager x5 | graph(5)", lambda: ([k() for _ in range(5)], gb.replay()))
def eager_join():
    with torch.cuda.stream(side):
        k(y)
    for _ in range(5): k()
    main.wait_stream(side)
    for _ in range(5): k()
report("eager x5 | main.wait_stream(side: finished) | eager x5 (vs eager chain)", eager_join)
print(f"   (eager chain: {base_eager:.1f} us)")
# ---- can a tiny eager kernel on either side of the event operation shield the graphs from it?
z = torch.zeros(64, device=dev)
def tiny():
    z.add_(1.0)
def fork_shielded():
    ga.replay(); tiny(); side.wait_stream(main); tiny(); gb.replay()
report("graph(5) | tiny | side.wait_stream(main) | tiny | graph(5)", fork_shielded)
def fork_shield_before():
    ga.replay(); tiny(); side.wait_stream(main); gb.replay()
report("graph(5) | tiny | side.wait_stream(main) | graph(5)", fork_shield_before)
def fork_shield_after():
    ga.replay(); side.wait_stream(main); tiny(); gb.replay()
report("graph(5) | side.wait_stream(main) | tiny | graph(5)", fork_shield_after)
def join_shielded():
    with torch.cuda.stream(side):
        k(y)
    ga.replay(); tiny(); main.wait_stream(side); tiny(); gb.replay()
report("graph(5) | tiny | main.wait_stream(side: finished) | tiny | graph(5)", join_shielded)
def join_shield_after():
    with torch.cuda.stream(side):
        k(y)
    ga.replay(); main.wait_stream(side); tiny(); gb.replay()
report("graph(5) | main.wait_stream(side: finished) | tiny | graph(5)", join_shield_after)
report("graph(5) | tiny | graph(5)", lambda: (ga.replay(), tiny(), gb.replay()))
# ---- event record / wait as NODES of the graphs (external events) instead of stream operations between the launches
try:
    ev_x = torch.cuda.Event(external=True)
    ev_f = torch.cuda.Event(external=True)
    g_a2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_a2):                      # main: 5 kernels, then the fork event as a node
        for _ in range(5): k()
        ev_f.record(torch.cuda.current_stream())
    g_s2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_s2, stream=side):         # side: wait for the fork node, 5 small kernels, record the join event
        torch.cuda.current_stream().wait_event(ev_f)
        for _ in range(5): k(y)
        ev_x.record(torch.cuda.current_stream())
    g_b2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_b2):                      # main: wait for the join node, 5 kernels
        torch.cuda.current_stream().wait_event(ev_x)
        for _ in range(5): k()
    def ext():
        g_a2.replay()
        with torch.cuda.stream(side):
            g_s2.replay()
        g_b2.replay()
    report("graph(5 + record node) | side graph(wait node .. record node) | graph(wait node + 5)", ext)
    x.zero_(); y.zero_()
    ext(); torch.cuda.synchronize()
    print("   values after one pass:", float(x[0]), float(y[0]), "(expected 0 stays 0: ordering not checked here)")
except Exception as ex:
    print("external events under capture: not available here:", repr(ex)[:300])
# ---- the same fork / join with PERSISTENT event objects (Stream.wait_stream creates and destroys an event per call)
evp, evq = torch.cuda.Event(), torch.cuda.Event()
def fork_persistent():
    ga.replay(); evp.record(main); side.wait_event(evp); gb.replay()
report("graph(5) | persistent event: record(main), side.wait_event | graph(5)", fork_persistent)
def join_persistent():
    with torch.cuda.stream(side):
        k(y); evq.record(side)
    ga.replay(); main.wait_event(evq); gb.replay()
report("graph(5) | persistent event recorded on side long ago: main.wait_event | graph(5)", join_persistent)
