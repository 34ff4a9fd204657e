#!/usr/bin/env python
"""End-to-end parity budget of the HIP path against the CPU oracle at a given image size (default: the BASELINE
400x600 shape).  Test-side tool (imports oracle/): prints, and writes to gpurun_out/parity_probe_<h>x<w>.json,

  * per-stage relative L2 errors with every stage fed the ORACLE's inputs (what tests/test_gpu_graph.py bounds),
  * the accumulated end-to-end error: relative L2, codebook-index agreement, PSNR(ours, oracle) after the harness's
    post-processing, and the PSNR delta against a ground truth CORRELATED with the output (oracle output + noise at
    a chosen PSNR) -- the form of "output PSNR within 0.05 dB of the reference" that can actually fail,
  * the budget: the same figures with the product's stages swapped in one at a time from the back.

    python tools/parity_probe.py [h w] [seed] [precision: bf16|fp16] [regime: adversarial|representative]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from glare_amd import modules as M  # noqa: E402
from glare_amd import ops  # noqa: E402
from glare_amd.synthetic import representative_init_, seeded_init_, synthetic_lowlight, synthetic_pair  # noqa: E402
from oracle import torch_ref as O  # noqa: E402


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def nhwc(x, bf16=True):
    return ops.nchw_to_nhwc(x.cuda(), bf16=bf16)


def nchw(x):
    return ops.nhwc_to_nchw(x).cpu()


def psnr_pair(out, ref, h, gt_db=(27.0, 30.0), seed=5):
    """PSNR(ours, oracle) on the post-processed images and |PSNR(ours,GT) - PSNR(oracle,GT)| for GT = oracle + noise."""
    a = O.postprocess(out, h)
    b = O.postprocess(ref, h)
    res = {"psnr_ours_vs_oracle": float(O.psnr(a, b))}
    rng = np.random.default_rng(seed)
    for db in gt_db:
        sigma = 10 ** (-db / 20)
        gt = np.clip(b + rng.normal(0, sigma, b.shape), 0, 1)
        gt_u8 = np.round(gt * 255).astype(np.uint8)
        pa = O.psnr(gt_u8 / 255, O.postprocess(out, h, gt_u8))
        pb = O.psnr(gt_u8 / 255, O.postprocess(ref, h, gt_u8))
        res["gt%.0fdB" % db] = {"ours": float(pa), "oracle": float(pb), "delta": float(abs(pa - pb))}
    return res


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 2 else 400
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
    precision = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    regime = sys.argv[5] if len(sys.argv) > 5 else "adversarial"
    torch.manual_seed(0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    if regime == "representative":
        og, ov = representative_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), O.VQModel().eval(), 0)
    else:
        og = seeded_init_(O.VQLLFLOWDeformable(per_sample_mean=True).eval(), 0)
        ov = seeded_init_(O.VQModel().eval(), 1)
    ops.use_precision(precision).__enter__()
    pg, pv = M.VQLLFLOWDeformable().eval(), M.VQModel().eval()
    pg.load_state_dict(og.state_dict(), strict=True)
    pv.load_state_dict(ov.state_dict(), strict=True)
    pg.cuda()
    pv.cuda()
    lr = O.preprocess(synthetic_pair(1, h, w, seed=seed)[0][0] if regime == "representative" else synthetic_lowlight(1, h, w, seed=seed)[0])
    with torch.no_grad():
        ref = og.stages(ov, lr)
    R = {"h": h, "w": w, "seed": seed, "precision": precision, "regime": regime}
    print("== %dx%d precision %s, %s regime" % (h, w, precision, regime), flush=True)
    with torch.no_grad():
        # ---- stage-isolated (oracle inputs) --------------------------------------------------
        enc = pg.RRDB.forward_nhwc(lr.cuda())
        S = {"A.cond_feat": rel(nchw(enc["cond_feat"]), ref["enc"]["cond_feat"]),
             "A.color_map": rel(nchw(enc["color_map"]), ref["enc"]["color_map"]),
             "A.mid_feat0": rel(nchw(enc["mid_feat"][0]), ref["enc"]["mid_feat"][0]),
             "A.mid_feat1": rel(nchw(enc["mid_feat"][1]), ref["enc"]["mid_feat"][1])}
        o_cm, o_cf = nhwc(ref["enc"]["color_map"], bf16=False), nhwc(ref["enc"]["cond_feat"])
        o_lat = nhwc(ref["latent"], bf16=False)
        o_mid = [nhwc(f) for f in ref["enc"]["mid_feat"]]
        o_code = [nhwc(f) for f in ref["code_feats"]]
        z = pg.flowUpsamplerNet.decode_nhwc(o_cm, o_cf)
        S["B.latent"] = rel(nchw(z), ref["latent"])
        idx, img, feats = pv.decode_nhwc(o_lat, want_image=True)
        S["C.indices_equal"] = bool(torch.equal(idx.cpu(), ref["indices"]))
        S["D.code_feat0"] = rel(nchw(feats[0]), ref["code_feats"][0])
        S["D.code_feat1"] = rel(nchw(feats[1]), ref["code_feats"][1])
        S["D.vq_rec"] = rel(img.cpu(), ref["vq_rec"])
        out_e = pg.deformable_decoder.forward_nhwc(o_lat, o_code, o_mid)
        S["E.out"] = rel(out_e.cpu(), ref["out"])
        R["stage_isolated"] = S

        # ---- budget: swap the product's stages in from the back -------------------------------
        def tail(latent, code_feats, mid):
            return pg.deformable_decoder.forward_nhwc(latent, code_feats, mid).cpu()

        def entry(name, out, idx=None):
            e = {"rel": rel(out, ref["out"])}
            e.update(psnr_pair(out, ref["out"], h))
            if idx is not None:
                e["index_agreement"] = float((idx.cpu() == ref["indices"]).float().mean())
            R.setdefault("budget", {})[name] = e
            print("%-44s rel %.5f  PSNR(ours,oracle) %.2f dB  delta@27dB %.4f  @30dB %.4f%s"
                  % (name, e["rel"], e["psnr_ours_vs_oracle"], e["gt27dB"]["delta"], e["gt30dB"]["delta"],
                     "" if idx is None else "  idx agree %.5f" % e["index_agreement"]), flush=True)

        entry("E only (oracle latent, code feats, enc feats)", out_e.cpu())
        entry("D+E (oracle latent, enc feats)", tail(o_lat, feats, o_mid), idx)
        entry("D+E + our enc mid feats", tail(o_lat, feats, enc["mid_feat"]), idx)
        idx_b, _, feats_b = pv.decode_nhwc(z, want_image=False)
        entry("B+D+E (oracle cond_feat / color_map)", tail(z, feats_b, o_mid), idx_b)
        lat = pg.flowUpsamplerNet.decode_nhwc(enc["color_map"], enc["cond_feat"])
        R["latent_rel_e2e"] = rel(nchw(lat), ref["latent"])
        idx_f, _, feats_f = pv.decode_nhwc(lat, want_image=False)
        entry("A+B+D+E = full path", tail(lat, feats_f, enc["mid_feat"]), idx_f)
        # the flipped tokens' effect alone: our latent for the trunk, the oracle's indices for the VQ decoder
        _, _, feats_i = pv.decode_nhwc(o_lat, want_image=False)
        entry("full path with the ORACLE's indices", tail(lat, feats_i, enc["mid_feat"]))
        # distance margin of the flipped tokens (how close to a tie were they?)
        cb = ov.quantize.embedding.weight.detach()
        tok = ref["latent"].permute(0, 2, 3, 1).reshape(-1, 3)
        flip = (idx_f.cpu() != ref["indices"]).nonzero().flatten()
        if flip.numel():
            d = torch.cdist(tok[flip[:4096]], cb)
            top2 = d.topk(2, dim=1, largest=False).values
            R["flipped_tokens"] = {"count": int(flip.numel()), "median_margin": float((top2[:, 1] - top2[:, 0]).median()),
                                   "median_nearest_dist": float(top2[:, 0].median())}
        r = pg.reverse_flow_nhwc(pv, lr.cuda())
        R["fused_equals_staged"] = bool(torch.equal(r["out"].cpu(), tail(lat, feats_f, enc["mid_feat"])))
    print(json.dumps(R["stage_isolated"], indent=1))
    print({k: R[k] for k in ("latent_rel_e2e", "flipped_tokens", "fused_equals_staged") if k in R})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_probe_%dx%d_%s_%s.json" % (h, w, precision, regime)), "w") as f:
        json.dump(R, f, indent=1)


if __name__ == "__main__":
    main()
