"""Host time of one asynchronous render call: a frame so small that the device is never the limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import _native
from raytracing_amd.flatten import make_opts
from raytracing_amd.workloads import get_flat_workload
flat, opt = get_flat_workload("cornell_1080p_b4")
opt = dict(opt, width=64, height=8)
ctx = _native.Context(0); ctx.upload(flat); ctx.set_timing(False)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(dev); torch.cuda.set_stream(s)
out = torch.empty((8, 64, 3), dtype=torch.uint8, device=dev)
for _ in range(20):
    ctx.render_rows_device(opt, 0, 0, 8, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
torch.cuda.synchronize()
K = 2000
t0 = time.perf_counter()
for _ in range(K):
    ctx.render_rows_device(opt, 0, 0, 8, out.data_ptr(), stream=s.cuda_stream, want_stats=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e6 * (t1 - t0) / K:.1f} us per call, with the device's tail {1e6 * (t2 - t0) / K:.1f} us per call")
t0 = time.perf_counter()
for _ in range(K):
    make_opts(opt, 0)
print(f"make_opts alone {1e6 * (time.perf_counter() - t0) / K:.1f} us")
