#!/usr/bin/env python3
"""Per-kernel rooflines of ONE training step ("report each one's own roofline ... rocprof per-kernel, not just the
host clock"): reads the per-launch table `tools/trace_train.sh` prints (rocprofv3 kernel trace of tools/bench_train.py) and sets every
launch of at least `--min-us` against its ALGORITHMIC work at FB15k-237 CoPER shapes (B = 512, L = 1000, d = 200, r = 32, F = 10368,
|E| = 14541; 32.4 M trainable parameters):

    tools/train_kernel_roofline.py profiles/r06_train_trace.txt [--json out.json]

bytes against 8 TB/s, flops against the 2.5 PF of the 16-bit matrix cores (the x3 arithmetic spends three hardware products per
product: `hw_frac` = 3 x frac is what the matrix pipe sees)."""
import json
import re
import sys

B, L, d, r, C = 512, 1000, 200, 32, 32
F = 18 * 18 * C                     # 10368
E = 14541
N_PARAM = 32436293
P_ELEMS = r * F * d                 # the projection of the fc_weights generator
HBM, MFMA = 8.0e12, 2.5e15

# kernel -> list of (what, bytes, flops) for its successive launches in a step (the last entry repeats)
WORK = {
    "k_tr_amsgrad": [("AMSGrad + clip: p, g, m, v, v_hat read, p, m, v, v_hat written, every trainable element", 9 * 4 * N_PARAM, 0)],
    "k_gemm_nt_w128_bf16x3": [("T = x P   [B] x [r d], K = F", 4 * (B * F + P_ELEMS + r * B * d), 2.0 * B * r * F * d),
                              ("dP = x^T dT   [F] x [r d], K = B", 4 * (B * F + r * B * d + P_ELEMS), 2.0 * B * r * F * d),
                              ("dx = dT P^T   [B] x [F], K = r d", 4 * (r * B * d + P_ELEMS + B * F), 2.0 * B * r * F * d)],
    "k_gemm_nt_bf16x3": [("dE = S^T h   [E] x [d], K = B", 4 * (B * E + B * d + E * d), 2.0 * B * E * d)],
    "k_pack_frag_both": [("both 16-bit views of the projection from one read: 118 MB in, 2 x 118 MB out", 4 * P_ELEMS * 3, 0)],
    "k_tr_score_loss": [("sampled scorer forward: B L rows of d floats gathered (L2-resident table), loss, ds", 4 * B * L * d, 2.0 * B * L * d)],
    "k_tr_dh_gather4": [("dh = sum_l ds E[row]: the same B L rows gathered again", 4 * B * L * d, 2.0 * B * L * d)],
    "k_tr_conv_bwd": [("conv backward: dx [B, F] read, image gradients and filter gradients", 4 * (B * F + B * 20 * 20), 2.0 * 2 * B * F * 9)],
    "k_tr_bn1_bwd_sums": [("BN1 backward sums: y and dx read, dx (ReLU / dropout applied) written", 4 * 3 * B * F, 0)],
    "k_tr_bn1_bwd_apply": [("BN1 backward apply: y, dx read, dx written", 4 * 3 * B * F, 0)],
    "k_tr_fc_post_bwd": [("dropout / bias backward of the dense layer, dc_b: dz [B, d], the bias projection", 4 * (2 * B * d + r * d), 2.0 * B * r * d)],
    "k_tr_fc_post": [("z0 = sum_rho c T[rho] + bias: the forward partials T [r, B, d] read", 4 * (r * B * d + B * d), 2.0 * B * r * d)],
    "k_tg_reduce": [("split-K partial sums added in slice order", 4 * 2 * B * r * d * 2, 0), ("split-K partial sums (dx)", 4 * 3 * B * F, 0)],
    "k_tr_col_sums": [("BN1 batch statistics: y [B P, C] read", 4 * B * F, 0), ("FCBN batch statistics", 4 * B * d, 0)],
    "k_tr_col_sums_add": [("dbias = column sums of S [B, E]", 4 * B * E, 0)],
    "k_tr_build_S": [("S rows built in LDS, written once", 4 * (B * E + 2 * B * L), 0)],
    "k_tr_conv_fwd": [("conv forward: images gathered, y [B, F] written", 4 * (B * F + B * 20 * 20), 2.0 * B * F * 9)],
    "k_tr_bn1_fwd": [("BN1 + ReLU + dropout: y read, x written", 4 * 2 * B * F, 0)],
}


def main(argv):
    path = argv[0]
    min_us = 10.0
    out_json = None
    if "--min-us" in argv:
        min_us = float(argv[argv.index("--min-us") + 1])
    if "--json" in argv:
        out_json = argv[argv.index("--json") + 1]
    seen, rows, total = {}, [], 0.0
    for ln in open(path):
        m = re.match(r"\s*([0-9.]+)\s+([0-9.]+) us\s+grid\s+\S+\s+(?:coper::)?(\S+)", ln)
        if not m:
            continue
        us, name = float(m.group(2)), m.group(3)
        base = name.split("<")[0]
        total += us
        k = seen.get(base, 0)
        seen[base] = k + 1
        if base not in WORK or us < min_us:
            continue
        what, by, fl = WORK[base][min(k, len(WORK[base]) - 1)]
        t_hbm, t_mfma = by / HBM * 1e6, fl / MFMA * 1e6
        bound = "mfma" if t_mfma > t_hbm else "hbm"
        floor = max(t_hbm, t_mfma)
        e = {"kernel": name, "us": us, "what": what, "bound": bound, "floor_us": round(floor, 1), "frac": round(floor / us, 3),
             "achieved": round(fl / us / 1e6, 1) if bound == "mfma" else round(by / us / 1e3, 1), "unit": "TFLOP/s" if bound == "mfma" else "GB/s"}
        if bound == "mfma" and "x3" in name:
            e["hw_frac"] = round(3 * floor / us, 3)
        rows.append(e)
    rows.sort(key=lambda e: -e["us"])
    print("%-34s %8s %6s %9s %6s  %s" % ("kernel", "us", "bound", "floor us", "frac", "algorithmic work"))
    for e in rows:
        print("%-34s %8.1f %6s %9.1f %6.3f  %s%s" % (e["kernel"][:34], e["us"], e["bound"], e["floor_us"], e["frac"], e["what"],
                                                 "  [hardware products: %.2f of the pipe]" % e["hw_frac"] if "hw_frac" in e else ""))
    covered = sum(e["us"] for e in rows)
    print("launches listed: %.1f us of %.1f us of kernels in the step" % (covered, total))
    if out_json:
        json.dump({"shapes": dict(B=B, L=L, d=d, r=r, F=F, E=E, trainable_parameters=N_PARAM), "kernels_us": total, "listed_us": covered, "kernels": rows,
                   "source": path, "peaks": {"hbm_GBs": 8000.0, "mfma_16bit_TFLOPs": 2500.0}}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
