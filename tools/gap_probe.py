import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sdrplusplus_amd import capi, radio, workloads
push = 50000
for fft in (True, False):
    ctx = capi.Context(0, max_push=push)
    if os.environ.get("SDRPP_TOOL_PIPELINE"):
        ctx.set_backend_pipeline(int(os.environ["SDRPP_TOOL_PIPELINE"]))
    if fft:
        workloads.setup(ctx, 3, dense_fft=True, data_width=1024, nvfo=32)
    else:
        for mode, if_rate, bw, centre, _ in workloads.vfo_plan(3, 32):
            d, keep = radio.vfo_desc(10e6, if_rate, bw, centre, mode)
            ctx.vfo_add(d, keep)
    x = torch.randn(push * 2, device="cuda:0", dtype=torch.float32) * 0.1
    for _ in range(10):
        ctx.push_device(x.data_ptr(), push)
    ctx.sync()
    for trial in range(2):
        t0 = time.perf_counter()
        for _ in range(400):
            ctx.push_device(x.data_ptr(), push)
        ctx.sync()
        print("fft", fft, "us per push %.1f" % (1e6 * (time.perf_counter() - t0) / 400))
    ctx.close()
