import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from polaris_amd import scenes, ctypes_api as T
from polaris_amd.tracer import HipTracer, UpdateMode, ChangeType
W=H=512; spp=128; B=5
sc = scenes.SCENES['cornell'](1.0)
seeds = scenes.make_seeds(spp, B)
tr = HipTracer('t', 0); tr.Init()
tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)
def req():
    r = T.BlockRequest(); r.frame_w, r.frame_h, r.block_x, r.block_y, r.block_w, r.block_h = W, H, 0, 0, W, H
    r.samples_per_pixel, r.num_bounces, r.min_bounces_for_rr = spp, B, 3; r.exposure, r.seed, r.accumulated_samples = 1.2, 0, 0
    return r
for _ in range(5):
    r = req(); tr.Trace(r, seeds); tr.MergeOutput(tr, r); tr.SyncFramebuffer(req())
tt = []; dd = []; tm = []; ts = []
for _ in range(20):
    r = req()
    t0 = time.perf_counter(); tr.Trace(r, seeds); t1 = time.perf_counter()
    tr.MergeOutput(tr, r); t2 = time.perf_counter(); tr.SyncFramebuffer(req()); t3 = time.perf_counter()
    tt.append(t1 - t0); dd.append(tr.last_trace_stats.device_ms); tm.append(t2 - t1); ts.append(t3 - t2)
print('trace wall ms %.3f  device ms %.3f  merge call ms %.3f  sync_framebuffer ms %.3f  frame %.3f' % (np.mean(tt)*1e3, np.mean(dd), np.mean(tm)*1e3, np.mean(ts)*1e3, (np.mean(tt)+np.mean(tm)+np.mean(ts))*1e3))
tr.Close()
