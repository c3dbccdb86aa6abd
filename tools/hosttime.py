import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sdrplusplus_amd import capi, workloads
push = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
dev = torch.device("cuda:0")
x = torch.randn(push * 2, device=dev, dtype=torch.float32) * 0.1
ctx = capi.Context(0, max_push=push)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
info = workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
for i in range(5):
    ctx.push_device(x.data_ptr(), push)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter(); ts = []
    for i in range(20):
        a = time.perf_counter(); ctx.push_device(x.data_ptr(), push); ts.append(time.perf_counter() - a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(" ".join("%.0f" % (1e6 * t) for t in ts))
    print("push=%d enqueue avg %.1f us (min %.1f max %.1f), enqueue total %.2f ms, with sync %.2f ms -> %.1f us/step" % (push, 1e6 * sum(ts) / len(ts), 1e6 * min(ts), 1e6 * max(ts), 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e6 * (t2 - t0) / 20))
ctx.close()
