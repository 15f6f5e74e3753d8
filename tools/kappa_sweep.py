#!/usr/bin/env python3
"""VERDICT r5 "Next round" 6, step 1: what a single-product (hi*hi only) first pass of the count kernel would cost in band walks.

For each workload: rank_band_kappa swept; per kappa the pass time, the band launch's time (library timers) and the number of
(query, entity) pairs inside the band (counted here from the f32 mode's logits of the pass's own h: |s - t| <= tau_q with
tau_q = 2 kappa (|h_q| max|E_e| + 8 max|bias|), bf16x3_chain.h: x3_band_tau without its absolute term).  And the error a
single-product pass would have to cover: max |fp16(E') . fp16(h') - chain| / (|h_q| max|E_e| + 8 max|bias|) over a pass, with E', h'
scaled as the mode scales them (largest magnitude in [2^14, 2^15)); the kappa that covers it with the mode's usual 4x margin.

    python tools/kappa_sweep.py [workload ...]        (default: fb15k237_cpg wn18rr_cpg)"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from coper_amd import data as cdata
from coper_amd.models import ConvE

KAPPAS = [1e-6, 3e-5, 3e-4, 1e-3, 3e-3]


def pow2_scale(absmax):
    return 2.0 ** (14 - math.floor(math.log2(absmax))) if absmax > 0 else 1.0


def main():
    names = sys.argv[1:] or ["fb15k237_cpg", "wn18rr_cpg"]
    for name in names:
        md = cdata.model_descriptors(name)
        Q = cdata.CONFIGS[name]["queries"]
        p = cdata.synthetic_params(md, 0)
        q = cdata.synthetic_queries(md, Q, seed=0)
        dq = {k: torch.as_tensor(v).cuda() for k, v in q.items()}
        nnz = len(q["filt_idx"])
        # the f32 handle: chain logits of the pass's h, for the pair census and the single-product error
        mf = ConvE(md, device="cuda:0", score_mode="f32").load_parameters(p).prepare()
        mx = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(p).prepare()
        mx.reserve(Q, nnz)
        _, _, h = mx.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], filt_nnz=nnz, want_equal=True, want_h=True)
        E = torch.as_tensor(p["ent_emb"]).cuda()
        bias = torch.as_tensor(p["pred_bias"]).cuda()
        emax = float(E.norm(dim=1).max())
        bmax = float(bias.abs().max())
        hn = h.norm(dim=1)
        unit = hn * emax + 8.0 * bmax                      # tau_q = 2 kappa unit_q
        sE, sh = pow2_scale(float(E.abs().max())), pow2_scale(float(h.abs().max()))
        Eh = (E * sE).half().float()
        hh = (h * sh).half().float()
        census = {k: 0 for k in KAPPAS}
        err1 = 0.0
        CH = 2048
        for lo in range(0, Q, CH):
            hi = min(Q, lo + CH)
            s = mf.score_all(h[lo:hi])                     # [c, |E|] the fp32 chain's logits
            t = s.gather(1, dq["e2"][lo:hi].view(-1, 1))
            dist = (s - t).abs() / unit[lo:hi].view(-1, 1)
            for k in KAPPAS:
                census[k] += int((dist <= 2.0 * k).sum())
            s1 = (hh[lo:hi] @ Eh.t()) / (sE * sh) + bias.view(1, -1)
            err1 = max(err1, float(((s1 - s).abs() / unit[lo:hi].view(-1, 1)).max()))
        mf.close()
        print("%s: Q=%d |E|=%d; single-product (hi*hi) max |s1 - chain| / unit = %.3e -> kappa >= %.1e covers it with the usual 4x margin"
              % (name, Q, md["num_ent"], err1, 4 * err1))
        mx.close()
        print("   %-8s %12s %10s %10s %10s" % ("kappa", "pairs/pass", "pass ms", "band us", "count us"))
        for k in KAPPAS:
            m = ConvE(md, device="cuda:0", score_mode="bf16x3", rank_band_kappa=k).load_parameters(p).prepare()
            m.reserve(Q, nnz)

            def step():
                return m.rank_pass(dq["e1"], dq["rel"], dq["e2"], dq["filt_indptr"], dq["filt_idx"], filt_nnz=nnz, want_equal=False)
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in ev:
                a.record(); step(); b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            m.profile(True)
            for kn in ("score_count", "band_exact"):
                m.profile_read(kn)
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            t_band, n_band = m.profile_read("band_exact")
            t_cnt, n_cnt = m.profile_read("score_count")
            m.profile(False)
            print("   %-8.0e %12d %10.4f %10.1f %10.1f" % (k, census[k], ts[len(ts) // 2], 1e3 * t_band / max(n_band, 1), 1e3 * t_cnt / max(n_cnt, 1)))
            m.close()
        del E, Eh, hh, h
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
