"""Oracle: CSI / POD / FAR / HSS contingency scores (TEST INFRASTRUCTURE).

Follows metrics.py:45-47 (scale 90, thresholds 20/30/35/40), :105-114
(per-frame TP/FN/FP/TN counting on clip(x,0,1)*scale as uint16) and :220-288
(pooled scores).
"""
import numpy as np

THRESHOLDS = (20, 30, 35, 40)
SCALE = 90


def contingency(pred, target, threshold, scale=SCALE):
    p = (np.clip(pred, 0, 1) * scale).astype(np.uint16) >= threshold
    t = (np.clip(target, 0, 1) * scale).astype(np.uint16) >= threshold
    tp = int(np.sum(p & t)); fn = int(np.sum(~p & t)); fp = int(np.sum(p & ~t)); tn = int(np.sum(~p & ~t))
    return tp, fn, fp, tn


def scores(pred, target, thresholds=THRESHOLDS):
    out = {}
    for th in thresholds:
        tp, fn, fp, tn = contingency(pred, target, th)
        div = lambda a, b: float(a) / b if b else float("nan")   # reference divides unguarded
        out[th] = {
            "csi": div(tp, tp + fp + fn),
            "pod": div(tp, tp + fn),
            "far": div(fp, tp + fp),
            # metrics.py:262-264 form of the Heidke skill score
            "hss": div(2 * (tp * tn - fp * fn), fp ** 2 + fn ** 2 + 2 * tp * tn + (fp + fn) * (tp + tn)),
        }
    return out
