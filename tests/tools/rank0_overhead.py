"""Host-side cost of what rank 0 of an 8-GPU frame does per frame besides its Trace (bench.py, one process per GPU):
export of its strip, Reset + merge of the assembled frame + tone-map.  Run on one MI355X."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
torch.cuda.init()
from polaris_amd import scenes, ctypes_api as T
from polaris_amd.tracer import HipTracer, UpdateMode, ChangeType
W = H = 512; spp = 128; B = 5; bh = 64
sc = scenes.SCENES['cornell'](1.0)
seeds = scenes.make_seeds(spp, B)
tr = HipTracer('t', 0); tr.Init()
tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)
def req(y, h):
    r = T.BlockRequest(); r.frame_w, r.frame_h, r.block_x, r.block_y, r.block_w, r.block_h = W, H, 0, y, W, h
    r.samples_per_pixel, r.num_bounces, r.min_bounces_for_rr = spp, B, 3; r.exposure, r.seed, r.accumulated_samples = 1.2, 0, 0
    return r
frame = torch.zeros((H * W, 4), dtype=torch.float32, device='cuda')
acc = {k: [] for k in ('trace', 'export', 'stream_sync', 'reset', 'merge', 'tonemap')}
for it in range(30):
    t = [time.perf_counter()]
    tr.Trace(req(192, bh), seeds); t.append(time.perf_counter())
    tr.export_block(req(192, bh), frame[192 * W:].data_ptr()); t.append(time.perf_counter())
    torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
    tr.reset_frame(); t.append(time.perf_counter())
    tr.merge_device(frame.data_ptr(), req(0, H)); t.append(time.perf_counter())
    tr.SyncFramebuffer(req(0, H)); t.append(time.perf_counter())
    if it >= 5:
        for k, a, b in zip(acc, t[:-1], t[1:]): acc[k].append((b - a) * 1e6)
print({k: round(float(np.mean(v)), 1) for k, v in acc.items()}, 'us; device ms of the Trace', round(tr.last_trace_stats.device_ms, 3))
tr.Close()
