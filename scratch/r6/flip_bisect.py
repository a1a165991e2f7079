"""r6 diagnostic: WHICH discrete decisions carry the raw gradient error of a path?

One full-size clip (32 x 224^2).  The engine's decisions (every ReLU sign pattern, every max-pool selection, the RoI arg-max
bins) are compared site by site with the fp64 oracle's own, and the oracle is re-evaluated with the engine's decisions applied
to ONE group of sites at a time (the rest keep the oracle's own): the group whose substitution moves the parameter-gradient
table from the raw numbers to the identical-decisions numbers is the one that carries the difference.

usage (GPU box): python scratch/r6/flip_bisect.py <preset> <dtype> <out.txt>
"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "video-long-term-feature-banks_amd", "lib"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (d if d > 0 else 1.0))


def stats(d):
    e = np.sort(list(d.values()))
    return "median %.3e  p90 %.3e  max %.3e" % (float(np.median(e)), float(e[int(0.9 * (len(e) - 1))]), float(e[-1]))


def stage_of(name):
    for s in ("res2", "res3", "res4", "res5"):
        if name.startswith(s + "_"):
            return s
    if name.startswith("nonlocal_conv3"):
        return "res3"
    if name.startswith("nonlocal_conv4"):
        return "res4"
    if name.startswith(("res_conv1", "conv1", "pool1")):
        return "stem"
    return "head"


def main(preset, dtype, out_path):
    import test_model_gpu as tm
    from oracle import model as om
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg, model, eng, inputs, params, seed_fn = tm.build(preset, dtype, tm.FULL)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    dec_e = eng.discrete_decisions()
    dec_o = {"_record": True}
    blobs, g_own = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=dec_o)
    gmax = max(float(g.norm()) for g in g_own.values())
    names = [n for n in eng.trainable if float(g_own[n].norm()) > 1e-9 * gmax]
    got = {n: eng.fetch_grad(n) for n in names}
    L = ["%s %s, 1 clip 32x224x224: the engine's discrete decisions against the fp64 oracle's own (scratch/r6/flip_bisect.py)" % (preset, dtype)]
    # ---- site-by-site differences
    relu_flip = {n: int((dec_e["relu"][n] != dec_o["relu"][n]).sum()) for n in dec_o["relu"] if n in dec_e["relu"]}
    relu_total = sum(int(np.prod(dec_o["relu"][n].shape)) for n in relu_flip)
    missing = [n for n in dec_o["relu"] if n not in dec_e["relu"]]
    pool_flip = {n: int((dec_e["pool"][n] != dec_o["pool"][n]).sum()) for n in dec_o["pool"] if n in dec_e["pool"]}
    # a max-pool selection that differs between two window elements of EQUAL value (ties among zeros behind a ReLU) routes the
    # gradient to an element whose ReLU mask is zero either way: count the selections that differ AND select a positive value
    roi_flip = int((dec_e["roi_bin"] != dec_o["roi_bin"]).sum()) if dec_o.get("roi_bin") is not None and dec_e.get("roi_bin") is not None else 0
    by_stage = collections.OrderedDict()
    for n, c in relu_flip.items():
        by_stage[stage_of(n)] = by_stage.get(stage_of(n), 0) + c
    L.append("ReLU units decided differently: %d of %d  (%s)%s" % (sum(relu_flip.values()), relu_total, ", ".join("%s: %d" % kv for kv in by_stage.items()),
                                                                   ("; sites the engine does not export: %s" % missing) if missing else ""))
    L.append("   per blob: " + (", ".join("%s: %d" % (n, c) for n, c in relu_flip.items() if c) or "none"))
    L.append("max-pool selections that differ: %s" % (", ".join("%s: %d of %d" % (n, c, dec_o["pool"][n].size) for n, c in pool_flip.items()) or "none"))
    L.append("RoI arg-max bins that differ: %d" % roi_flip)
    # ---- the gradient tables
    raw = {n: rel(got[n], g_own[n].numpy()) for n in names}
    L.append("parameter gradients (%d tensors), engine vs oracle:" % len(names))
    L.append("   oracle on its OWN decisions (raw):                    %s" % stats(raw))
    full = dict(dec_e)
    _, g_all = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=full)
    same = {n: rel(got[n], g_all[n].numpy()) for n in names}
    L.append("   oracle on ALL of the engine's decisions:              %s" % stats(same))
    groups = collections.OrderedDict()
    for st in ("stem", "res2", "res3", "res4", "res5", "head"):
        groups["ReLU " + st] = ("relu", [n for n in dec_o["relu"] if stage_of(n) == st and n in dec_e["relu"]])
    groups["max pools"] = ("pool", list(pool_flip))
    groups["RoI bins"] = ("roi", None)
    results = {}
    for gname, (kind, sites) in groups.items():
        d = {"relu": dict(dec_o["relu"]), "pool": dict(dec_o["pool"]), "roi_bin": dec_o.get("roi_bin")}
        if kind == "roi":
            if dec_e.get("roi_bin") is None:
                continue
            d["roi_bin"] = dec_e["roi_bin"]
        else:
            if not sites:
                continue
            for n in sites:
                d[kind][n] = dec_e[kind][n]
        _, g = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=d)
        e = {n: rel(got[n], g[n].numpy()) for n in names}
        L.append("   oracle's own decisions, the engine's for %-10s  %s" % (gname + ":", stats(e)))
        results[gname] = float(np.median(list(e.values())))
    # ---- inside the decisive ReLU stage: blob by blob, then the flipped units themselves
    relu_groups = {k: v for k, v in results.items() if k.startswith("ReLU")}
    best = min(relu_groups, key=relu_groups.get)
    st = best.split()[1]
    L.append("inside %s (the stage whose decisions carry the difference), blob by blob:" % best)
    flipped_blobs = [n for n in dec_o["relu"] if stage_of(n) == st and relu_flip.get(n, 0)]
    dec_p = {"_record": True, "_want_pre": set(flipped_blobs)}
    om.run(cfg, params, inputs, "train", torch.float64, False, seed_fn, decisions=dec_p)
    for n in flipped_blobs:
        d = {"relu": dict(dec_o["relu"]), "pool": dict(dec_o["pool"]), "roi_bin": dec_o.get("roi_bin")}
        d["relu"][n] = dec_e["relu"][n]
        _, g = om.run(cfg, params, inputs, "train", torch.float64, True, seed_fn, decisions=d)
        e = {k: rel(got[k], g[k].numpy()) for k in names}
        pre = dec_p["pre"][n]
        rms = float(np.sqrt(np.mean(pre ** 2)))
        idx = np.argwhere(dec_e["relu"][n] != dec_o["relu"][n])
        val_e = eng.fetch(n)
        units = ", ".join("%s: oracle pre-activation %+.2e (%.1e of the blob's rms), engine value %.2e" % (
            tuple(int(i) for i in ix), pre[tuple(ix)], abs(pre[tuple(ix)]) / rms, val_e[tuple(ix)]) for ix in idx)
        L.append("   the engine's decisions for %-22s %s   <- %d unit(s): %s" % (n + ":", stats(e), len(idx), units))
    open(out_path, "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
