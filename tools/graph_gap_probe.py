"""Dependent small kernels issued eagerly from Python vs replayed from a captured HIP graph (torch.cuda.CUDAGraph):
   python tools/graph_gap_probe.py
Measured on the MI355X box: 5.5 us per kernel eager (host-issue bound), 1.7 us per kernel in graph replay, 246 us of
host time for a 400-node replay."""
import torch, time
dev = 'cuda:0'
x = torch.randn(64 * 1024, device=dev)
def chain(n):
    y = x
    for _ in range(n):
        y = y * 1.0001
    return y
N = 400
for _ in range(3): chain(N)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); chain(N); e.record(); torch.cuda.synchronize()
print("eager: %.2f us per dependent small kernel (GPU time between events)" % (s.elapsed_time(e) * 1e3 / N))
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    chain(8)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        out = chain(N)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
print("graph replay: %.2f us per dependent small kernel" % (s.elapsed_time(e) * 1e3 / N))
t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize()
print("graph replay host time: %.1f us for %d nodes" % ((t1 - t0) * 1e6, N))
