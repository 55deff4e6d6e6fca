"""Predicted per-rank work of the multi-GPU path build from the per-unit records of an N = 1 run (no hardware curve).

    python profiles/predict_scaling.py gpurun_out/r03a/unit_steps_athaliana.npz > profiles/r03_predicted_scaling.json

Input: per-unit steps / evals of the four builders over the WHOLE config (profiles/probes/unit_steps_dump.py).  Work of a
rank = the evals of its units (the walk kernels run at a steady evals/s; steps are reported beside them).
Partitions compared for N = 2 / 4 / 8 ranks:
  contiguous  : rank r walks anchors [r * ceil(A / N), (r + 1) * ceil(A / N))   -- tlm_dist_anchor_range, what the product does;
                paths_ order = rank order, so the exchange places whole (rank, builder) blocks
  round_robin : chunk (anchor a, quarter q of its trials) -> rank (4 a + q) mod N; DET units of anchor a -> rank a mod N
  lpt         : whole anchors, heaviest first, each to the least loaded rank (what a balancing scheduler could reach with
                anchor granularity)
For each: per-rank work, max / mean (1.0 = perfect), and the parallel efficiency bound mean / max."""
import json
import sys

import numpy as np

z = np.load(sys.argv[1])
rb = int(z["rebuild_attempts"][0])
det_off = z["det_off"].astype(np.int64)
A = det_off.shape[0] - 1
out = {"source": sys.argv[1], "anchors": A, "rebuild_attempts": rb, "work_unit": "evals (steps beside them)", "N": {}}
for key in ("evals", "steps"):
    mc = (z["mc_es_" + key].astype(np.float64) + z["mc_os_" + key].astype(np.float64)).reshape(A, rb)
    det_u = z["det_es_" + key].astype(np.float64) + z["det_os_" + key].astype(np.float64)
    det = np.add.reduceat(det_u, det_off[:-1]) * (np.diff(det_off) > 0)
    per_anchor = mc.sum(1) + det
    quarters = mc.reshape(A, 4, rb // 4).sum(2)
    if key == "evals":
        out["per_anchor_evals"] = {"min": float(per_anchor.min()), "mean": float(per_anchor.mean()), "max": float(per_anchor.max()),
                                   "cv": float(per_anchor.std() / per_anchor.mean())}
    for N in (2, 4, 8):
        per = (A + N - 1) // N
        cont = np.array([per_anchor[r * per:(r + 1) * per].sum() for r in range(N)])
        rr = np.zeros(N)
        for a in range(A):
            for q in range(4):
                rr[(4 * a + q) % N] += quarters[a, q]
            rr[a % N] += det[a]
        lpt = np.zeros(N)
        for a in np.argsort(-per_anchor):
            lpt[np.argmin(lpt)] += per_anchor[a]
        e = out["N"].setdefault(str(N), {})
        for name, w in (("contiguous", cont), ("round_robin", rr), ("lpt", lpt)):
            d = e.setdefault(name, {})
            d[key + "_per_rank"] = [float(x) for x in w]
            d[key + "_max_over_mean"] = float(w.max() / w.mean())
for N, e in out["N"].items():
    best = min(e, key=lambda k: e[k]["evals_max_over_mean"])
    e["smallest_ratio"] = best
print(json.dumps(out, indent=1))
