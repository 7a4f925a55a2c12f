import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from simfire_amd import workloads
from simfire_amd.engine import FireEngine
E, K = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 4
light = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = workloads.c3(1024, E)
eng = FireEngine(M_f=w.M_f, **w.engine_kwargs())
eng.set_layers(*w.layers()); eng.reset(w.init_xy); eng.step(12)
a = torch.randn(4096, 4096, dtype=torch.float16, device="cuda"); b = torch.randn(4096, 4096, dtype=torch.float16, device="cuda")
x = torch.zeros(1024, device="cuda")
(a @ b); x.add_(1); torch.cuda.synchronize()
side = torch.cuda.Stream()
W1 = torch.randn(512, 512, dtype=torch.float16, device="cuda"); W2 = torch.randn(512, 512, dtype=torch.float16, device="cuda"); W3 = torch.randn(512, 64, dtype=torch.float16, device="cuda")
obs = torch.randn(E, 512, dtype=torch.float16, device="cuda")
def policy():
    return torch.relu(torch.relu(obs @ W1) @ W2) @ W3
policy(); torch.cuda.synchronize()
eng.set_tuning(loop_light=light)
eng.loop_start(K)
pts = np.zeros((E, K, 3), dtype=np.int32)
for i in range(5): eng.loop_step(pts)
def timed(fn, stream=None):
    t0 = time.perf_counter()
    if stream is None:
        fn(); torch.cuda.current_stream().synchronize()
    else:
        with torch.cuda.stream(stream):
            fn()
        stream.synchronize()
    return (time.perf_counter() - t0) * 1e6
print("E", E, "light", light)
print("tiny kernel, default stream: %.0f us" % timed(lambda: x.add_(1)), "restarts", eng.loop_restarts())
for i in range(3): eng.loop_step(pts)
print("policy MLP (3 layers of 512), default stream: %.0f us" % timed(policy), "restarts", eng.loop_restarts())
for i in range(3): eng.loop_step(pts)
print("policy MLP again:            %.0f us" % timed(policy), "restarts", eng.loop_restarts())
for i in range(3): eng.loop_step(pts)
print("matmul, default stream:      %.0f us" % timed(lambda: a @ b), "restarts", eng.loop_restarts())
for i in range(3): eng.loop_step(pts)
print("after steps: restarts", eng.loop_restarts())
eng.loop_stop()
print("matmul with no loop:         %.0f us" % timed(lambda: a @ b), " policy with no loop: %.0f us" % timed(policy))
