import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from acl_amd import runtime
from oracle import bindings as ob
from oracle.database import OracleDatabase
import helpers
import test_gpu_full_size as T
ctx = runtime.Context(0); device = torch.device("cuda:0")
case = helpers.load_bench_database()
database = ctx.register_database(case["database"], None, case["bulk_low"])
handles = [ctx.register_clip_with_database(c, database) for c in case["clips"]]
odb = OracleDatabase(case["database"], case["bulk_medium"], case["bulk_low"])
rng = np.random.default_rng(31); n = 32768
which = rng.integers(0, 64, size=n)
dur = np.array([ob.oracle().aclo_finite_duration(c.ctypes.data, ob.LOOP_AS_COMPRESSED) for c in case["clips"]], dtype=np.float32)
times = (rng.uniform(0, 1, size=n).astype(np.float32) * dur[which]).astype(np.float32)
stream = torch.cuda.Stream(device)
for req in [(True, 2, 1), (True, 2, 1), (True, 2, 3), (True, 2, 5), (False, 2, 4), (True, 2, 2), (True, 2, 0xFFFFFFFF), (True, 1, 0xFFFFFFFF), (True, 2, 0xFFFFFFFF), (False, 2, 0xFFFFFFFF), (True, 2, 0xFFFFFFFF)]:
    f = ctx.database_stream_in if req[0] else ctx.database_stream_out
    g = odb.stream_in if req[0] else odb.stream_out
    print(req, f(database, req[1], req[2], stream=stream.cuda_stream), g(req[1], req[2]))
    got = T._decode(ctx, torch, device, handles, which, times, 100, stream=stream).cpu().numpy()
st = (np.floor(times * 30.0) / 30.0).astype(np.float32)
results = {}
for label, policy, obp in (("floor", runtime.ROUND_FLOOR, ob.ROUND_FLOOR), ("none", runtime.ROUND_NONE, ob.ROUND_NONE), ("floor again", runtime.ROUND_FLOOR, ob.ROUND_FLOOR)):
    got = T._decode(ctx, torch, device, handles, which, st, 100, params=runtime.default_params(rounding_policy=policy), stream=stream).cpu().numpy()
    exp = ob.oracle_decompress_tracks_batch(case["clips"], which, st, 100, rounding=obp, options=odb.options())
    bad = np.argwhere(got[..., helpers.XYZ_LANES].view(np.uint32) != exp[..., helpers.XYZ_LANES].view(np.uint32))
    results[label] = got
    print(label, "mismatching instances", len(np.unique(bad[:, 0])), "of", n)
    if len(bad):
        i = bad[0][0]; print(" first", i, "clip", which[i], "time", st[i], "tracks", np.unique(bad[bad[:, 0] == i][:, 1])[:10], got[i, bad[0][1]], exp[i, bad[0][1]])

si = st * np.float32(30.0); exact = si == np.round(si)
d = np.abs(results["floor"] - results["none"]).max(axis=(1, 2))
print("exact", exact.sum(), "max diff on exact", d[exact].max(), "instances differing", (d[exact] > 1e-6).sum())
i = np.argmax(np.where(exact, d, 0)); print(i, which[i], st[i], si[i], dur[which[i]], d[i])
