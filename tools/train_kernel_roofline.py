#!/usr/bin/env python3
"""Per-kernel rooflines of ONE training step ("report each one's own roofline ... rocprof per-kernel, not just the
host clock"): reads the per-launch table `tools/trace_train.sh` prints (rocprofv3 kernel trace of tools/bench_train.py) and sets every
launch of at least `--min-us` against its ALGORITHMIC work at FB15k-237 CoPER shapes (B = 512, L = 1000, d = 200, r = 32, F = 10368,
|E| = 14541; 32.4 M trainable parameters):

    tools/train_kernel_roofline.py profiles/r06_train_trace.txt [--json out.json]

bytes against 8 TB/s, flops against the 2.5 PF of the 16-bit matrix cores.  The x3 arithmetic spends three hardware products per
product: for those kernels `frac` prices the hardware products (what the matrix pipe sees), `alg_frac` the algorithm's flops."""
import json
import re
import sys

B, L, d, r, C = 512, 1000, 200, 32, 32
F = 8 * 18 * C                      # 4608: the 10 x 20 image under a 3 x 3 VALID conv, 32 channels
E = 14541
N_PARAM = 32436293
P_ELEMS = r * F * d                 # the projection of the fc_weights generator (29.5 M of the 32.4 M parameters)
HBM, MFMA = 8.0e12, 2.5e15

# kernel -> list of (what, bytes, flops, hardware products per product) for its successive launches in a step (the last entry repeats)
WORK = {
    "k_tr_amsgrad": [("AMSGrad + clip: p, g, m, v, v_hat read, p, m, v, v_hat written, every trainable element", 9 * 4 * N_PARAM, 0, 1)],
    "k_gemm_nt_w128_bf16x3": [("T = x P   [B] x [r d], K = F (five K slices)", 4 * (B * F + P_ELEMS + 5 * r * B * d), 2.0 * B * r * F * d, 3),
                              ("dx = dT P^T   [B] x [F], K = r d (seven K slices)", 4 * (r * B * d + P_ELEMS + 7 * B * F), 2.0 * B * r * F * d, 3),
                              ("dP = x^T dT   [F] x [r d], K = B (launched behind dx since round 6)", 4 * (B * F + r * B * d + P_ELEMS), 2.0 * B * r * F * d, 3)],
    "k_gemm_nt_bf16x3": [("dE = S^T h   [E] x [d], K = B", 4 * (B * E + B * d + E * d), 2.0 * B * E * d, 3)],
    "k_pack_frag_both": [("both 16-bit views of the projection from one read: 118 MB in, 2 x 118 MB out", 4 * P_ELEMS * 3, 0, 1)],
    "k_tr_score_loss_dh": [("sampled scorer, forward and dh: B L rows of d floats gathered ONCE (table resident in L2 / MALL), loss, ds", 4 * B * L * d, 4.0 * B * L * d, 1)],
    "k_tr_conv_bwd": [("conv backward: dx [B, F] read, image gradients and per-query filter gradients", 4 * (B * F + B * 200 + B * 320), 2.0 * 2 * B * F * 9, 1)],
    "k_tr_bn1_bwd_sums": [("BN1 backward sums: y and the seven K slices of dx read, dx (ReLU / dropout applied) written", 4 * 9 * B * F, 0, 1)],
    "k_tr_bn1_bwd_apply": [("BN1 backward apply: y, dx read, dx written", 4 * 3 * B * F, 0, 1)],
    "k_tr_fc_post_slices": [("z1 and T from the five K slices of x P: 5 x [B, r d] read, T [r, B, d] and z1 written", 4 * (6 * r * B * d + B * d), 2.0 * B * r * d, 1)],
    "k_tr_fc_post_bwd": [("dropout / bias backward of the dense layer: dz [B, d]", 4 * 3 * B * d, 0, 1)],
    "k_tg_reduce": [("the seven K slices of dx added in slice order", 4 * 8 * B * F, 0, 1)],
    "k_tr_col_sums": [("BN1 batch statistics: y [B P, C] read", 4 * B * F, 0, 1), ("FCBN batch statistics", 4 * B * d, 0, 1)],
    "k_tr_col_sums_add": [("dbias = column sums of S [B, E]", 4 * B * E, 0, 1)],
    "k_tr_build_S": [("S rows built in LDS, written once", 4 * (B * E + 2 * B * L), 0, 1)],
    "k_tr_conv_fwd": [("conv forward: images gathered, y [B, F] written", 4 * (B * F + B * 200), 2.0 * B * F * 9, 1)],
    "k_tr_bn1_fwd": [("BN1 + ReLU + dropout: y read, x written", 4 * 2 * B * F, 0, 1)],
    "k_pack_frag": [("an operand's two fp16 planes from its fp32 values", 0, 0, 1)],
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
        what, by, fl, prod = WORK[base][min(k, len(WORK[base]) - 1)]
        if by == 0 and fl == 0:
            continue
        t_hbm, t_alg, t_hw = by / HBM * 1e6, fl / MFMA * 1e6, prod * fl / MFMA * 1e6
        bound = "mfma" if t_hw > t_hbm else "hbm"
        floor = max(t_hbm, t_hw)
        e = {"kernel": name, "us": us, "what": what, "bound": bound, "floor_us": round(floor, 1), "frac": round(floor / us, 3),
             "achieved": round(prod * fl / us / 1e6, 1) if bound == "mfma" else round(by / us / 1e3, 1), "unit": "TFLOP/s" if bound == "mfma" else "GB/s"}
        if prod > 1:      # the x3 arithmetic: three hardware products per product -- `frac` prices the hardware products, `alg_frac` the algorithm's
            e["alg_frac"] = round(max(t_hbm, t_alg) / us, 3)
            e["hardware_products"] = prod
        rows.append(e)
    rows.sort(key=lambda e: -e["us"])
    print("%-34s %8s %6s %9s %6s  %s" % ("kernel", "us", "bound", "floor us", "frac", "algorithmic work"))
    for e in rows:
        print("%-34s %8.1f %6s %9.1f %6.3f  %s%s" % (e["kernel"][:34], e["us"], e["bound"], e["floor_us"], e["frac"], e["what"],
                                                 "  [x%d hardware products; algorithmic %.2f]" % (e["hardware_products"], e["alg_frac"]) if "alg_frac" in e else ""))
    covered = sum(e["us"] for e in rows)
    print("launches listed: %.1f us of %.1f us of kernels in the step" % (covered, total))
    if out_json:
        json.dump({"shapes": dict(B=B, L=L, d=d, r=r, F=F, E=E, trainable_parameters=N_PARAM), "kernels_us": total, "listed_us": covered, "kernels": rows,
                   "source": path, "peaks": {"hbm_GBs": 8000.0, "mfma_16bit_TFLOPs": 2500.0}}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
