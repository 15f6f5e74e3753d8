#!/usr/bin/env python3
"""The reference's train / evaluate / checkpoint loop (`CoPER_ConvE/qa_cpg/run_cpg.py:108-260`) on this engine,
end to end on one MI355X: TSV triples -> loader (ids, `_reverse` augmentation, full-graph filters) -> ConvE ->
AMSGrad steps on sampled labels -> filtered MR / MRR / Hits@k on dev and test -> TensorFlow-format checkpoint
(variables + optimizer slots) -> restore into a second model and re-evaluate.

    python examples/train_eval_loop.py [--data DIR_WITH_train.txt_dev.txt_test.txt] [--steps 400] [--variant cpg|plain|lookup]

Without --data it uses the 400/70/73 split of nell-995 dev triples kept under tests/golden/kg_tsv.
Everything model-side goes through libcoper_hip.so (include/coper_hip.h); there is no CPU fallback."""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from coper_amd import data as cdata, weights  # noqa: E402
from coper_amd.kg_loader import TSVKGLoader  # noqa: E402
from coper_amd.metrics import ranking_and_hits  # noqa: E402
from coper_amd.models import ConvE  # noqa: E402

VARIANTS = {   # the three model families of run_cpg.py:37-56, in model_descriptors form (configs/*.yaml keys)
    "cpg": dict(context_rel_conv=None, context_rel_out=[], rel_emb_size=32),
    "plain": dict(context_rel_conv=None, context_rel_out=None, rel_emb_size=200),
    "lookup": dict(context_rel_conv=None, context_rel_out=[], rel_emb_size=1, do_parameter_lookup=True),
}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=os.path.join(ROOT, "tests", "golden", "kg_tsv"))
    ap.add_argument("--variant", choices=sorted(VARIANTS), default="cpg")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--num-labels", type=int, default=100)
    ap.add_argument("--eval-every", type=int, default=200)
    ap.add_argument("--workdir", default=None)
    args = ap.parse_args(argv)

    work = args.workdir or tempfile.mkdtemp(prefix="coper_loop_")
    for f in ("train.txt", "dev.txt", "test.txt"):
        shutil.copy(os.path.join(args.data, f), work)
    loader = TSVKGLoader(work, os.path.basename(os.path.normpath(args.data)))
    loader.maybe_create_tf_record_files(work)                      # run_cpg.py:108: ids are known from here on

    md = dict(cdata._COMMON)                                       # the keys of models.py:99-130
    md.update(VARIANTS[args.variant])
    md.update(num_ent=loader.num_ent, num_rel=loader.num_rel, ent_emb_size=200, learning_rate=0.003,
              label_smoothing_epsilon=0.1, hidden_dropout=0.2, output_dropout=0.2, batch_norm_train_stats=True,
              batch_norm_momentum=0.1)
    model = ConvE(md, device="cuda:0", score_mode="bf16x3")
    model.load_parameters(cdata.synthetic_params(md, seed=0))      # random init of the named architecture
    model.train_init(seed=0)

    # device=: the negative sampler runs on the GPU (coper_amd.data.DeviceTrainDataset; ~40 x the host sampler at FB15k-237 sizes)
    train = iter(loader.train_dataset(work, batch_size=args.batch_size, num_labels=args.num_labels, prop_negatives=10.0,
                                      one_positive_label_per_sample=True, device=model.device))
    dev = loader.eval_dataset(work, "dev", batch_size=512)
    test = loader.eval_dataset(work, "test", batch_size=512)

    def evaluate(m, dataset, name):
        m.prepare()                                                # rebuild the inference caches from the variables
        mr, mrr, hits = ranking_and_hits(m, os.path.join(work, "eval"), dataset, name)
        print("  %-5s MR %8.2f  MRR %.4f  Hits@1 %.3f  Hits@10 %.3f" % (name, mr, mrr, hits[1], hits[10]))
        return mrr

    print("%s on %s: |E| = %d, R2 = %d, %d train records" % (args.variant, loader.dataset_name, loader.num_ent, loader.num_rel,
                                                            len(loader.train_samples()["e1"])))
    t0 = time.perf_counter()
    for step in range(1, args.steps + 1):
        loss = model.train_op(next(train))                         # session.run((model.train_op, model.loss)), run_cpg.py:211-219
        if step % args.eval_every == 0 or step == args.steps:
            torch.cuda.synchronize()
            print("step %d  loss %.4f  (%.2f ms/step)" % (step, float(loss.cpu()[0]), (time.perf_counter() - t0) / step * 1e3))
            evaluate(model, dev, "dev")
    final = evaluate(model, test, "test")

    # checkpoint in the reference's format (run_cpg.py:252) and restore into a fresh model (run_cpg.py:206)
    prefix = os.path.join(work, "checkpoints", "model_weights.ckpt")
    slots, powers = model.optimizer_state()
    weights.save_tf_checkpoint(prefix, {k: v.cpu().numpy() for k, v in model._tensors.items()}, slots, powers)
    restored = ConvE(md, device="cuda:0", score_mode="bf16x3").load_parameters(weights.load_tf_checkpoint(prefix))
    again = evaluate(restored, test, "test*")
    assert again == final, (again, final)
    print("checkpoint %s.{index,data-00000-of-00001} restored: same test MRR" % prefix)
    model.close()
    restored.close()
    return final


if __name__ == "__main__":
    main()
