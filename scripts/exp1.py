import sys, time, torch
sys.path.insert(0, '.')
from ripcurrents_amd import synth
from ripcurrents_amd.api import Context
W, H = 1920, 1080
P = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)
frames = synth.surf_clip(W, H, 17, device=torch.device("cuda"))
flows = torch.empty((16, H, W, 2), dtype=torch.float32, device="cuda")
ctx = Context(W, H)
def run(tag, **opts):
    for k, v in opts.items(): ctx.set_option(k, v)
    for _ in range(2): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ctx.farneback_clip(frames, flows, **P)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5 / 16
    print("%-40s %.1f us/frame  %.0f fps" % (tag, dt * 1e6, 1 / dt), flush=True)
run("default chunk4 fused remap")
run("no remap", xcd_remap=0)
run("no remap, unfused", fuse_iters=0)
run("remap, unfused", xcd_remap=1)
run("chunk1 fused remap", fuse_iters=1, chunk=1)
run("chunk2", chunk=2)
run("chunk8", chunk=8)
run("chunk16", chunk=16)
run("chunk16 noremap", xcd_remap=0)
