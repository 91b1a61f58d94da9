import importlib, time, sys
sys.path.insert(0, "/root/repo")
trt = importlib.import_module("tiny-raytracer_amd")
for (w, h, spp) in ((300, 300, 300), (300, 300, 30), (64, 64, 4)):
    desc = trt.scenes.cornell(w, h)
    world, cam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 8, 20, False, desc["background"])
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); img = r.render(cam, world); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{w}x{h}x{spp}spp: kernel {r.last_stats['kernel_ms']:.2f} ms; call ms: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
