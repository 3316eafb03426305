import ctypes as C, json, os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from reseq_amd import api, workloads
tmp = tempfile.mkdtemp(prefix="rsq_small_")
ppath = os.path.join(tmp, "p0.rsqp")
arrays = workloads.p0_profile(ppath)
N = 16384
rows, _ = workloads.seq_to_illumina_rows(N, arrays)
W = rows.shape[1]
d_text = api.DeviceArray.from_numpy(0, np.concatenate([rows.reshape(-1), np.zeros(8, np.uint8)]))
prof = api.Profile(ppath)
sim = api.Simulator(prof, None, 0)
sim.prepare(11)
d_out = api.DeviceArray(0, N * 400)
need, k, used = C.c_size_t(0), C.c_uint64(0), C.c_size_t(0)
def call(first, n):
    best = 1e9
    for _ in range(4):
        api._check(api.lib().rsq_sim_error_model_fasta(sim.h, first, C.c_void_p(d_text.ptr.value + first * W), n * W, 1, d_out.ptr, d_out.nbytes, C.byref(need), C.byref(k), C.byref(used), None))
        best = min(best, sim.last_kernel_ms("fill_reads"))
    return best
out = {"all": call(0, N)}
per = [call(c * 128, 128) for c in range(N // 128)]
out["per_128"] = [round(x, 3) for x in per]
out["halves"] = [round(call(0, N // 2), 3), round(call(N // 2, N // 2), 3)]
print(json.dumps(out))
