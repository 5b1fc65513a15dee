"""Does the forward pass gain from a partner?  One plain-DFCNN engine at B = 32 against TWO engines at B = 16 whose forward passes are
enqueued on two streams (a persistent Winograd launch gives its CUs back one by one at its ragged end; in the backward pass the other
stream's launch takes them at once -- 0.40 ms of the step, profiles/r05_transformer_second_stream_probe.txt addendum -- the forward
pass runs alone).  Forward only, no fbank; usage (GPU box): python3 tools/exp_split_forward.py [model]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from asr_dfcnn_transformer_amd.engine import DFCNNEngine

model = sys.argv[1] if len(sys.argv) > 1 else 'm1'
T, F, V = 1600, 200, 1536
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(32, T, F, device=dev, generator=g)
one = DFCNNEngine(model=model, vocab=V, B=32, T=T, F=F, seed=0, device=dev)
halves = [DFCNNEngine(model=model, vocab=V, B=16, T=T, F=F, seed=0, device=dev) for _ in range(2)]
xs = [x[:16].contiguous(), x[16:].contiguous()]
s2 = torch.cuda.Stream(device=dev)


def run_one():
    one.forward(x)


def run_split():
    ev = torch.cuda.Event(); ev.record()
    halves[0].forward(xs[0])
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        halves[1].forward(xs[1])
        done = torch.cuda.Event(); done.record()
    torch.cuda.current_stream().wait_event(done)


def run_serial():
    halves[0].forward(xs[0]); halves[1].forward(xs[1])


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(3):
    print('round %d: one engine B 32: %.3f ms   two engines B 16, two streams: %.3f ms   two engines B 16, one stream: %.3f ms' %
          (rnd, timed(run_one), timed(run_split), timed(run_serial)), flush=True)
a = one.forward(x).clone()
b0, b1 = halves[0].forward(xs[0]).clone(), halves[1].forward(xs[1]).clone()
torch.cuda.synchronize()
print('same bits:', bool(torch.equal(a[:, :16], b0) and torch.equal(a[:, 16:], b1)))
