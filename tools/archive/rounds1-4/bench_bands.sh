#!/bin/bash
# What one of N ranks does (its bands of the 2048^2 Cornell frame), timed on one GPU for N = 1, 2, 4, 8 and both
# megakernel-class backends: bench.py cannot fake N ranks, so this renders rank 0's rows directly.
python - <<'PY'
import importlib, time, os
import torch
trt = importlib.import_module("tiny-raytracer_amd"); tiles = importlib.import_module("tiny-raytracer_amd.tiles")
desc = trt.scenes.cornell(2048, 2048); world, cam = trt.world_from_description(desc); scene = world.get_bvh()
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream()
for n in (1, 2, 4, 8):
    lay = tiles.band_layout(2048, n, 0)
    band = dict(band_rows=lay["band_rows"], band_stride=lay["band_stride"], band_offset=lay["band_offset"], rows_local=lay["rows_local"]) if n > 1 else {}
    acc = torch.zeros((lay["rows_local"], 2048, 3), device=dev); ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    for name, backend in (("megakernel", 0), ("streamed", 3)):
        r = trt.Renderer(4096, 1, 50, False, desc["background"], backend=backend)
        res = []
        for rep in range(3):
            ctr.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            r.render_device(cam, scene, acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=rep*256, sample_end=(rep+1)*256, accumulate=1, **band)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res.append(int(ctr[1]) / dt / 1e9)
        print(f"N={n} rows/GPU={lay['rows_local']:4d} {name:>10}: {max(res):6.2f} Gray/s per GPU  -> x{n} = {max(res)*n:6.1f}", flush=True)
PY
