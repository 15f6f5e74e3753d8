"""Which part of a double-buffered H2D / kernels / D2H pipeline costs what (FB15k-237 shapes, Q = 20,480)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coper_amd import data as cdata
from coper_amd.models import ConvE

md = cdata.model_descriptors("fb15k237_cpg")
dev = torch.device("cuda:0")
m = ConvE(md, device=dev, score_mode="bf16x3").load_parameters(cdata.synthetic_params(md, 0)).prepare()
Q = 20480
q = cdata.synthetic_queries(md, Q, seed=0)
keys = ("e1", "rel", "e2", "filt_indptr", "filt_idx")
sizes = [int(np.asarray(q[k]).size) for k in keys]
offs = np.concatenate([[0], np.cumsum(sizes)])
pin = torch.empty(int(offs[-1]), dtype=torch.int64).pin_memory()
for k, o, n in zip(keys, offs, sizes):
    pin[o:o + n].copy_(torch.as_tensor(np.asarray(q[k], dtype=np.int64)))
nnz = sizes[4]
NS = int(os.environ.get("NSLOT", "2"))
stage = [torch.empty_like(pin, device=dev) for _ in range(NS)]
views = [{k: st[o:o + n] for k, o, n in zip(keys, offs, sizes)} for st in stage]
rd = [torch.empty(Q, dtype=torch.int32, device=dev) for _ in range(NS)]
oh = [torch.empty(Q, dtype=torch.int32).pin_memory() for _ in range(NS)]
m.reserve(Q, nnz)
s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ev_in = [torch.cuda.Event() for _ in range(NS)]; ev_done = [torch.cuda.Event() for _ in range(NS)]; ev_out = [torch.cuda.Event() for _ in range(NS)]

def run(name, h2d_side, d2h_side, kernels=True, n=40, host_gate=False, toggle=False, done_after_d2h=False, warm=6):
    cnt = [0]
    def step():
        sl = cnt[0] % NS; cnt[0] += 1
        cur = torch.cuda.current_stream(dev)
        v = views[sl]
        if h2d_side:
            with torch.cuda.stream(s_in):
                if host_gate:
                    ev_done[sl].synchronize()      # the slot's previous pass (NSLOT passes ago) has finished: no stream-side wait
                else:
                    s_in.wait_event(ev_done[sl])
                stage[sl].copy_(pin, non_blocking=True)
                ev_in[sl].record(s_in)
            cur.wait_event(ev_in[sl])
        else:
            stage[sl].copy_(pin, non_blocking=True)
        if d2h_side:
            cur.wait_event(ev_out[sl])
        if kernels:
            m.rank_pass(v["e1"], v["rel"], v["e2"], v["filt_indptr"], v["filt_idx"], filt_nnz=nnz, want_equal=False, out=rd[sl])
        if toggle:
            m.profile(cnt[0] % 4 == 0)
        if not done_after_d2h:
            ev_done[sl].record(cur)
        if d2h_side:
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_done[sl])
                oh[sl].copy_(rd[sl], non_blocking=True)
                ev_out[sl].record(s_out)
        else:
            oh[sl].copy_(rd[sl], non_blocking=True)
        if done_after_d2h:
            ev_done[sl].record(cur)
    for _ in range(warm): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-44s %.4f ms per pass (host issue %.4f ms)" % (name, dt / n * 1e3, t_issue / n * 1e3), flush=True)

run("all on one stream", False, False)
run("H2D on a copy stream", True, False)
run("D2H on a copy stream", False, True)
run("both on copy streams", True, True)
run("H2D on a copy stream, host-gated slots", True, False, host_gate=True)
run("host-gated, 20 passes", True, False, host_gate=True, n=20)
run("host-gated, 20 passes, 3 warm", True, False, host_gate=True, n=20, warm=3)
run("host-gated, done after D2H", True, False, host_gate=True, n=20, done_after_d2h=True)
run("host-gated, profile toggled", True, False, host_gate=True, n=20, toggle=True)
m.profile(False)
run("copies only, one stream", False, False, kernels=False)
run("copies only, copy streams", True, True, kernels=False)

# ---- the same pipeline with raw HIP calls for the copy-in side (ctypes): stream, events, hipMemcpyAsync
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
st = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
def mkev():
    e = ctypes.c_void_p(); assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0; return e     # hipEventDisableTiming
rin = [mkev() for _ in range(NS)]; rdone = [mkev() for _ in range(NS)]
def run_raw(name, n=40, warm=6, gate=True):
    cnt = [0]
    nbytes = ctypes.c_size_t(pin.numel() * 8)
    def step():
        sl = cnt[0] % NS; cnt[0] += 1
        cur = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if gate and cnt[0] > NS:
            hip.hipEventSynchronize(rdone[sl])
        assert hip.hipMemcpyAsync(ctypes.c_void_p(stage[sl].data_ptr()), ctypes.c_void_p(pin.data_ptr()), nbytes, 1, st) == 0
        hip.hipEventRecord(rin[sl], st)
        hip.hipStreamWaitEvent(cur, rin[sl], 0)
        v = views[sl]
        m.rank_pass(v["e1"], v["rel"], v["e2"], v["filt_indptr"], v["filt_idx"], filt_nnz=nnz, want_equal=False, out=rd[sl])
        oh[sl].copy_(rd[sl], non_blocking=True)
        hip.hipEventRecord(rdone[sl], cur)
    for _ in range(warm): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-44s %.4f ms per pass (host issue %.4f ms)" % (name, dt / n * 1e3, t_issue / n * 1e3), flush=True)
    assert all(np.array_equal(o.numpy(), oh[0].numpy()) for o in oh)
run_raw("raw HIP copy-in stream, host-gated, 40")
run_raw("raw HIP copy-in stream, host-gated, 20", n=20, warm=3)
run("all on one stream", False, False)

def run_raw_timed(n=12):
    cnt = [0]
    nbytes = ctypes.c_size_t(pin.numel() * 8)
    rows = []
    def step():
        sl = cnt[0] % NS; cnt[0] += 1
        cur = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        t = [time.perf_counter()]
        if cnt[0] > NS:
            hip.hipEventSynchronize(rdone[sl])
        t.append(time.perf_counter())
        hip.hipMemcpyAsync(ctypes.c_void_p(stage[sl].data_ptr()), ctypes.c_void_p(pin.data_ptr()), nbytes, 1, st)
        t.append(time.perf_counter())
        hip.hipEventRecord(rin[sl], st)
        t.append(time.perf_counter())
        hip.hipStreamWaitEvent(cur, rin[sl], 0)
        t.append(time.perf_counter())
        v = views[sl]
        m.rank_pass(v["e1"], v["rel"], v["e2"], v["filt_indptr"], v["filt_idx"], filt_nnz=nnz, want_equal=False, out=rd[sl])
        t.append(time.perf_counter())
        oh[sl].copy_(rd[sl], non_blocking=True)
        t.append(time.perf_counter())
        hip.hipEventRecord(rdone[sl], cur)
        t.append(time.perf_counter())
        rows.append([ (b - a) * 1e3 for a, b in zip(t, t[1:]) ])
    torch.cuda.synchronize()
    for _ in range(n): step()
    torch.cuda.synchronize()
    print("host ms per call: evsync  memcpyH2D  evrecord  waitevent  rank_pass  D2Hcopy  evrecord")
    for r in rows: print("   " + "  ".join("%8.3f" % x for x in r))
run_raw_timed()
