"""How much of a one-tile forward is launch gaps?  Captures HipVisionTower.forward_tiles (about 130 kernel launches) in a HIP graph through
torch.cuda.graph and replays it: python tools/graph_probe.py [tiles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import weights as Wt
from clip_decontamination_amd.engine import HipVisionTower

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = Wt.vit_config("ViT-L-14")
tower = HipVisionTower(cfg, Wt.make_vit_weights(cfg, seed=0), precision="bf16", device="cuda:0")
scene = torch.from_numpy(Wt.make_tiles_u8(1, 2304, seed=1, smooth=True)[0]).cuda()
wins = torch.tensor([[0, 512, 256 * i, 256 * i + 512] for i in range(T)], dtype=torch.int32, device="cuda:0")
opts = tower.forward_opts("SegEarth", True)


def run():
    return tower.forward_tiles(scene, wins, (512, 512), opts)


for _ in range(3):
    ref = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 10
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        run()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 10
same = torch.equal(out[1], ref[1])
print(f"{T} tile(s): eager {eager * 1e3:.2f} ms, graph replay {graph * 1e3:.2f} ms, same tokens: {same}", flush=True)
