#!/bin/bash
# batch size sweep of the streamed backend: whole frame (bench.py) and one GPU's eighth (bands of rank 0 of 8)
for b in 1 2 4 8 16 32 64; do
  export TRT_STREAM_BATCH_SPP=$b
  python - <<'PY'
import importlib, time, os
import torch
trt = importlib.import_module("tiny-raytracer_amd"); tiles = importlib.import_module("tiny-raytracer_amd.tiles")
desc = trt.scenes.cornell(2048, 2048); world, cam = trt.world_from_description(desc); scene = world.get_bvh()
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream()
out = []
for n in (1, 8):
    lay = tiles.band_layout(2048, n, 0)
    band = dict(band_rows=lay["band_rows"], band_stride=lay["band_stride"], band_offset=lay["band_offset"], rows_local=lay["rows_local"]) if n > 1 else {}
    acc = torch.zeros((lay["rows_local"], 2048, 3), device=dev); ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    r = trt.Renderer(4096, 1, 50, False, desc["background"], backend=4)
    res = []
    for rep in range(3):
        ctr.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_device(cam, scene, acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=rep*256, sample_end=(rep+1)*256, accumulate=1, **band)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.append(int(ctr[1]) / dt / 1e9)
    out.append(max(res))
print("batch_spp %3s: N=1 %.2f Gray/s, one eighth (N=8) %.2f Gray/s per GPU" % (os.environ["TRT_STREAM_BATCH_SPP"], out[0], out[1]), flush=True)
PY
done
