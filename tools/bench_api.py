#!/usr/bin/env python3
"""Drop-in entry timed end to end: ranking_and_hits with host batches in, float64 means out (PCIe-inclusive)."""
import sys, time
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE
from coper_amd.metrics import ranking_and_hits
md = cdata.model_descriptors("fb15k237_cpg")
p = cdata.synthetic_params(md, 0)
m = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(p).prepare()
loader = cdata.SyntheticKGLoader("fb15k237_cpg")
ds = loader.eval_dataset(None, "test", batch_size=512)
for _ in range(3): ranking_and_hits(m, None, ds, "test")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): out = ranking_and_hits(m, None, ds, "test")
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("ranking_and_hits (EvalDataset, 20480 queries): %.3f ms  %.2f M triples/s" % (dt * 1e3, 20480 / dt / 1e6), out[:2])
# generic iterable of batches (no as_single_batch): the reference's batch contract with CSR filters
batches = list(ds)
for _ in range(3): ranking_and_hits(m, None, iter(batches), "test")      # (the first call allocates the pinned staging buffers)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): out = ranking_and_hits(m, None, iter(batches), "test")
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("ranking_and_hits (list of 512-batches): %.3f ms  %.2f M triples/s" % (dt * 1e3, 20480 / dt / 1e6))
