#!/usr/bin/env python3
"""What ONE rank of an N-GPU run costs per frame, measured on one GPU: renders part 0 of N of config 3 in batches of B
frames with F batches in flight (render + assemble of a part-major buffer; the RCCL gather itself is not in it).
Gives the per-rank floor of the strong-scaling bench before an 8-GPU node is available.
usage: gpu_rank_share.py [N ...]   env RANKSHARE_F=1,2,3  RANKSHARE_B=1,2,4,8  RANKSHARE_K=frames
RANKSHARE_ROTATE=1: the rank assembles every N-th call only (bench.py's default, --gather-dst rotate: call k goes to rank k % N)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import device, host, scenes
import torch
W, H, MR = 1920, 1080, 3
sd, _ = scenes.config(3)
hs = host.HostScene(sd); desc = hs.flatten(); cam = hs.camera(W, H)
Fs = [int(f) for f in os.environ.get("RANKSHARE_F", "1,2,3").split(",")]
Bs = [int(f) for f in os.environ.get("RANKSHARE_B", "1,2,4,8").split(",")]  # 0 = 8 per part (the bench's default)
K = int(os.environ.get("RANKSHARE_K", "480"))
ROT = os.environ.get("RANKSHARE_ROTATE", "0") == "1"
REP = int(os.environ.get("RANKSHARE_REPEAT", "1"))
for N in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    prm = device.make_params(W, H, MR, part_index=0, part_count=N, layout=1)
    px = device.part_pixels(prm)
    for B in [b if b > 0 else min(64, 8 * N) for b in Bs]:
        cams = (device.Camera * B)(*([cam] * B))
        for F in Fs:
            ctxs = [device.Context(0) for _ in range(F)]
            for c in ctxs: c.upload(desc)
            mine = [torch.zeros(px * B, dtype=torch.int32, device="cuda") for _ in range(F)]
            gath = [torch.zeros(px * N * B, dtype=torch.int32, device="cuda") for _ in range(F)]
            frm = [torch.zeros(W * H * B, dtype=torch.int32, device="cuda") for _ in range(F)]
            def step(k):
                j = k % F
                ctxs[j].render_batch_device(cams, prm, mine[j].data_ptr(), sync=False)
                if not ROT or k % N == 0:
                    ctxs[j].assemble_batch(prm, B, gath[j].data_ptr(), frm[j].data_ptr(), sync=False)
            def sync():
                for c in ctxs: c.synchronize()
            for k in range(4 * F): step(k)
            dts = []
            for _ in range(REP):   # RANKSHARE_REPEAT bursts of K frames each (a 20-frame burst is ~1 ms: one sample says little)
                sync(); t0 = time.perf_counter()
                for k in range(K // B): step(k)
                sync(); dts.append((time.perf_counter() - t0) / (K // B * B) * 1e3)
            dts.sort(); dt = dts[len(dts) // 2]
            print(f"parts={N} batch={B} inflight={F} burst={K}{' assemble every N-th call' if ROT else ''}: {dt:.4f} ms/frame median of {REP} "
                  f"(min {dts[0]:.4f} max {dts[-1]:.4f})  (0.421/N = {0.421 / N:.4f})", flush=True)
            for c in ctxs: c.close()
