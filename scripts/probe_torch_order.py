"""Experiment: does torch's HIP runtime initialise after the engine's (libamdhip64 of /opt/rocm vs torch's own copy)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
es = []
for i in range(n):
    e = engine.Engine(0)
    if i % 2: e.close()
    else: es.append(e)
import torch
print("engines", n, "available", torch.cuda.is_available(), flush=True)
try:
    x = torch.zeros(4, device="cuda:0")
    print("tensor ok", x.sum().item(), flush=True)
except Exception as ex:
    print("tensor FAILED", ex, flush=True)
print(os.popen("ls /proc/self/fd | wc -l").read().strip(), "fds", flush=True)
