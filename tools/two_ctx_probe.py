#!/usr/bin/env python3
"""Two independent pipelined graphs (two contexts, two HIP streams, two pushing threads) on ONE GPU against one: how much of a tick's
launch gap / start ramp / tail the hardware recovers when another kernel is there to fill it.  The upper bound for any scheme that lets
consecutive ticks of ONE graph overlap.  usage: two_ctx_probe.py [cfg] [block]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from sdrplusplus_amd import capi, workloads

    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    nvfo = workloads.CFG[cfg]["nvfo"]
    dev = torch.device("cuda", 0)
    nb = 4

    def make():
        ctx = capi.Context(0, max_push=B)
        workloads.setup(ctx, cfg, dense_fft=True, data_width=1024, nvfo=nvfo or None, fft=True)
        xd = [torch.from_numpy(workloads.synth(cfg, B, seed=7 + i, nvfo=nvfo or None).view(np.float32)).to(dev) for i in range(nb)]
        ctx.set_pipelined(True, 0)
        return ctx, xd

    graphs = [make(), make()]
    npush = max(24, min(2000, (1 << 28) // B))

    def run(g, n):
        ctx, xd = g
        for i in range(n):
            ctx.push_device(xd[i % nb].data_ptr(), B)
        ctx.sync()

    out = {"cfg": cfg, "push": B, "pushes_per_graph": npush}
    for g in graphs:
        run(g, 16)
    for trial in range(3):
        t0 = time.perf_counter()
        run(graphs[0], npush)
        one = B * npush / (time.perf_counter() - t0) / 1e6
        th = [threading.Thread(target=run, args=(g, npush)) for g in graphs]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        two = 2 * B * npush / (time.perf_counter() - t0) / 1e6
        out.setdefault("one_graph_MSps", []).append(round(one, 1))
        out.setdefault("two_graphs_aggregate_MSps", []).append(round(two, 1))
    for ctx, _ in graphs:
        ctx.set_pipelined(False)
        ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
