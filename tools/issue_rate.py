import sys, time, ctypes, numpy as np
sys.path.insert(0, '.')
import torch
from acl_amd import runtime, synth
import bench
clips, ci, times = bench.build_workload('one_clip', 0)
dev = torch.device('cuda:0')
ctx = runtime.Context(0)
h = np.array([ctx.register_clip(c.blob) for c in clips], dtype=np.uint32)
n = ci.size
d_clips = torch.from_numpy(h[ci].astype(np.int32)).to(dev); d_times = torch.from_numpy(times).to(dev)
d_poses = torch.empty((n, 1200), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream(dev)
params = runtime.default_params()
lib = runtime.load_library()
args = (ctx._handle, d_clips.data_ptr(), d_times.data_ptr(), n, ctypes.byref(params), d_poses.data_ptr(), 4800, stream.cuda_stream)
for _ in range(50): lib.aclhip_decompress_tracks_batch(*args)
torch.cuda.synchronize()
for K in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(K): lib.aclhip_decompress_tracks_batch(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(K, 'issue us/call', (t1-t0)/K*1e6, 'total us/step', (t2-t0)/K*1e6)
# null stream
args0 = args[:-1] + (None,)
t0 = time.perf_counter()
for _ in range(1000): lib.aclhip_decompress_tracks_batch(*args0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('null stream: issue', (t1-t0)/1000*1e6, 'total', (t2-t0)/1000*1e6)
