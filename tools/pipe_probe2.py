"""How long hipMemcpyAsync H2D calls take on the HOST while kernels run on another stream (raw HIP via ctypes)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coper_amd import data as cdata
from coper_amd.models import ConvE
hip = ctypes.CDLL("libamdhip64.so")
md = cdata.model_descriptors("fb15k237_cpg")
dev = torch.device("cuda:0")
m = ConvE(md, device=dev, score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 0)).prepare()
Q = 20480
q = cdata.synthetic_queries(md, Q, seed=0)
dq = {k: torch.as_tensor(v).to(dev) for k, v in q.items()}
nnz = len(q["filt_idx"])
m.reserve(Q, nnz)
N = 1500000
pin = torch.empty(N // 8, dtype=torch.int64).pin_memory()
# a device buffer of its own hipMalloc, and one from torch's pool
raw = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(raw), ctypes.c_size_t(N)) == 0
pool = torch.empty(N // 8, dtype=torch.int64, device=dev)
hraw = ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(hraw), ctypes.c_size_t(N), 0) == 0
s_side = torch.cuda.Stream(device=dev)
st = ctypes.c_void_p()
assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0     # hipStreamNonBlocking
H2D = 1

def kernels(n):
    for _ in range(n):
        m.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], filt_nnz=nnz, want_equal=False)

def probe(name, dst, src, stream_handle, busy):
    torch.cuda.synchronize()
    if busy:
        kernels(4)     # ~2 ms of work queued on the launch stream
    t0 = time.perf_counter()
    rc = hip.hipMemcpyAsync(ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(N), H2D, stream_handle)
    t1 = time.perf_counter()
    assert rc == 0
    hip.hipStreamSynchronize(stream_handle)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%-64s call %.3f ms, copy done after %.3f ms, all done after %.3f ms" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t0) * 1e3), flush=True)

for rep in range(2):
    for busy in (False, True):
        b = "busy" if busy else "idle"
        probe("torch side stream, pool dst, torch pinned src, " + b, pool.data_ptr(), pin.data_ptr(), ctypes.c_void_p(s_side.cuda_stream), busy)
        probe("torch side stream, raw dst, torch pinned src, " + b, raw.value, pin.data_ptr(), ctypes.c_void_p(s_side.cuda_stream), busy)
        probe("raw nonblocking stream, raw dst, hipHostMalloc src, " + b, raw.value, hraw.value, st, busy)
        probe("raw nonblocking stream, pool dst, torch pinned src, " + b, pool.data_ptr(), pin.data_ptr(), st, busy)
        probe("launch stream itself, pool dst, torch pinned src, " + b, pool.data_ptr(), pin.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), busy)
