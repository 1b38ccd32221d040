"""Where does an UNPROFILED replay of the captured training step spend its time?  Tiny stamp kernels (tools/probes/stamp.hip: the 100 MHz device clock)
are captured into the step at the forks / joins of its streams; after N replays the last step's stamps are printed relative to the step's first node.
    python tools/debug/step_stamps.py [steps]"""
import ctypes, os, subprocess, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
so = "/tmp/libstamp.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(R, "tools/probes/stamp.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.probe_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
from bench import build_model
from sam_textvqa_amd import autograd as ag, modules, ops, trainer as trmod
from sam_textvqa_amd.synthetic import clone_batch, make_batch
from sam_textvqa_amd.trainer import Trainer

slots = torch.zeros(64, dtype=torch.int64, device="cuda")
names = []
def stamp(name):
    if name not in names:
        names.append(name)
    i = names.index(name)
    rc = lib.probe_stamp(slots.data_ptr() + 8 * i, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc

def wrap(obj, attr, before=None, after=None):
    f = getattr(obj, attr)
    def g(*a, **k):
        if before: stamp(before)
        r = f(*a, **k)
        if after: stamp(after)
        return r
    setattr(obj, attr, g)

M = modules.SAM4C
wrap(M, "_forward_text_bert", "textbert fwd begin", "textbert fwd end")
wrap(M, "_forward_obj_encoding", "obj enc begin", None)
wrap(M, "_forward_ocr_encoding", None, "ocr enc end")
wrap(M, "_forward_mmt", "mmt fwd begin (before join)", "mmt fwd end")
wrap(M, "_forward_output", None, "heads fwd end")
wrap(modules.MMT, "forward", None, None)
_gb = ag.GradBarrierFn.backward
def gb(ctx, g):
    stamp("bwd reaches %s encoder" % ctx.name)
    return _gb(ctx, g)
ag.GradBarrierFn.backward = staticmethod(gb)
_fl = ag.DeferredWgrads.flush.__func__
def fl(cls, late=False):
    if cls.jobs:
        stamp("wgrad flush %d begin%s" % (fl.n, " (late)" if late else ""))
        _fl(cls, late)
        if not late: stamp("wgrad flush %d end" % fl.n)
        fl.n += 1
    else:
        _fl(cls, late)
fl.n = 0
ag.DeferredWgrads.flush = classmethod(fl)
trmod.DeferredWgrads = ag.DeferredWgrads
_ss = ops.sumsq
def ss(*a, **k):
    fl.n = 0
    stamp("joined: sumsq begin")
    return _ss(*a, **k)
ops.sumsq = ss
_sa = ops.step_advance
def sa(*a, **k):
    if sa.n == 0: stamp("step begin")           # (the unpipelined capture opens with step_advance; the pipelined one with _issue_pending_update)
    r = _sa(*a, **k)
    stamp("step_advance %d done" % sa.n)
    sa.n ^= 1
    return r
sa.n = 0
ops.step_advance = sa
_ip = trmod.Trainer._issue_pending_update
def ip(self, bd):
    if "step begin" not in names: stamp("step begin")
    return _ip(self, bd)
trmod.Trainer._issue_pending_update = ip
_ap = trmod.Trainer._adam_piece
def ap(self, lo, hi, gate, max_blocks=0):
    r = _ap(self, lo, hi, gate, max_blocks)
    if gate is not None:
        stamp("adam piece [%d..] end" % lo)
    return r
trmod.Trainer._adam_piece = ap

_ad = ops.adam_step_dev
def ad(*a, **k):
    stamp("adam begin")
    r = _ad(*a, **k)
    stamp("adam end")
    return r
ops.adam_step_dev = ad
def _wrap_op(nm):
    f = getattr(ops, nm)
    state = {"n": 0}
    def g(*a, **k):
        i = state["n"] % 2
        state["n"] += 1
        stamp("%s #%d begin" % (nm, i))
        r = f(*a, **k)
        stamp("%s #%d end" % (nm, i))
        return r
    setattr(ops, nm, g)
_wrap_op("embedding_bwd")
_wrap_op("embed_sum_bwd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
model = build_model(3, ("n", "n", "s", "s", "s", "s"), 5000)
tr = Trainer(model, seed=1, use_graph=True)
batch = make_batch(64, device="cuda", seed=1)
for _ in range(4):
    tr.step(clone_batch(batch))
staged = tr.input_buffers() or batch
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(staged)
stamp("after the replay (eager stream)")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
v = slots.cpu().tolist()
t_begin = v[names.index("step begin")]
print("ms per step %.3f; stamps of the last replay, us after the step's first node:" % (dt * 1e3))
for n, t in sorted(zip(names, v), key=lambda x: x[1]):
    print("%9.1f  %s" % ((t - t_begin) / 100.0, n))
