"""Oracle (test infrastructure): CPU restatement of weighted boxes fusion (WBF).

The reference's ``wbf.py:68`` calls ``ensemble_boxes.weighted_boxes_fusion`` with
``iou_thr=0.67, skip_box_thr=0.01, weights=1`` (wbf.py:34-35,62).  That package
(ensemble-boxes==1.0.9, requirements.txt:37) is not vendored under /root/reference, so this
restates its published algorithm (Solovyev et al., "Weighted boxes fusion", 2021) and is
**parity unpinned**: pinned only by this repo's own golden vectors.

Algorithm (conf_type='avg', allows_overflow=False, the defaults wbf.py uses):
 1. per model t, drop boxes with score < skip_box_thr; normalise corners (swap if reversed, clip to
    [0,1]); drop zero-area boxes; score *= weight[t]; bucket by label.
 2. per label, visit boxes by descending score.  Each box is matched to the existing *fused* box with
    the highest IoU if that IoU is > iou_thr, else it starts a new cluster.
 3. a cluster's fused box = score-weighted mean of member corners; fused score = mean member score.
 4. finally score *= min(n_members, sum(weights)) / sum(weights); all labels concatenated and sorted
    by descending score.
Ties in the two sorts are broken deterministically here (stable sort, see ``_desc``); the published
package uses numpy's default unstable argsort, so tie order is unspecified there.
The running fused box is held in float32 exactly as the package does (its accumulator is a float32
array), member boxes in float64.
"""
import numpy as np


def _desc(v):
    """Indices for descending order; ties keep the later element first (argsort(stable)[::-1])."""
    return np.argsort(v, kind='stable')[::-1]


def _prefilter(boxes_list, scores_list, labels_list, weights, thr):
    buckets = {}
    for t in range(len(boxes_list)):
        if len(boxes_list[t]) != len(scores_list[t]) or len(boxes_list[t]) != len(labels_list[t]):
            raise ValueError('boxes / scores / labels length mismatch')
        for j in range(len(boxes_list[t])):
            score = scores_list[t][j]
            if score < thr:
                continue
            label = int(labels_list[t][j])
            x1, y1, x2, y2 = (float(v) for v in boxes_list[t][j][:4])
            if x2 < x1:
                x1, x2 = x2, x1
            if y2 < y1:
                y1, y2 = y2, y1
            x1, y1, x2, y2 = (min(max(v, 0.0), 1.0) for v in (x1, y1, x2, y2))
            if (x2 - x1) * (y2 - y1) == 0.0:
                continue
            # row = label, score*w, w, model, x1, y1, x2, y2
            buckets.setdefault(label, []).append([label, float(score) * weights[t], weights[t], t, x1, y1, x2, y2])
    for k in buckets:
        arr = np.array(buckets[k])
        buckets[k] = arr[_desc(arr[:, 1])]
    return buckets


def _fuse(members):
    """Score-weighted corner mean; score = mean of member scores (float32 accumulator)."""
    box = np.zeros(8, dtype=np.float32)
    conf = 0
    w = 0
    for b in members:
        box[4:] += (b[1] * b[4:])
        conf += b[1]
        w += b[2]
    box[0] = members[0][0]
    box[1] = conf / len(members)
    box[2] = w
    box[3] = -1
    box[4:] /= conf
    return box


def _best_match(fused, new_box, thr):
    if fused.shape[0] == 0:
        return -1
    b = fused[:, 4:]
    xa = np.maximum(b[:, 0], new_box[4])
    ya = np.maximum(b[:, 1], new_box[5])
    xb = np.minimum(b[:, 2], new_box[6])
    yb = np.minimum(b[:, 3], new_box[7])
    inter = np.maximum(xb - xa, 0) * np.maximum(yb - ya, 0)
    area_a = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    area_b = (new_box[6] - new_box[4]) * (new_box[7] - new_box[5])
    iou = inter / (area_a + area_b - inter)
    iou[fused[:, 0] != new_box[0]] = -1
    k = int(np.argmax(iou))
    return k if iou[k] > thr else -1


def weighted_boxes_fusion(boxes_list, scores_list, labels_list, weights=None, iou_thr=0.55, skip_box_thr=0.0):
    """Returns (boxes (n,4) xyxy in [0,1], scores (n,), labels (n,)) sorted by descending score."""
    if weights is None:
        weights = np.ones(len(boxes_list))
    if len(weights) != len(boxes_list):
        weights = np.ones(len(boxes_list))
    weights = np.array(weights)
    buckets = _prefilter(boxes_list, scores_list, labels_list, weights, skip_box_thr)
    if len(buckets) == 0:
        return np.zeros((0, 4)), np.zeros((0,)), np.zeros((0,))
    per_label = []
    for label in buckets:
        rows = buckets[label]
        clusters = []
        fused = np.empty((0, 8))
        for j in range(len(rows)):
            k = _best_match(fused, rows[j], iou_thr)
            if k != -1:
                clusters[k].append(rows[j])
                fused[k] = _fuse(clusters[k])
            else:
                clusters.append([rows[j].copy()])
                fused = np.vstack((fused, rows[j].copy()))
        wsum = weights.sum()
        for i, members in enumerate(clusters):
            fused[i, 1] = fused[i, 1] * min(len(members), wsum) / wsum
        per_label.append(fused)
    allb = np.concatenate(per_label, axis=0)
    allb = allb[_desc(allb[:, 1])]
    return allb[:, 4:], allb[:, 1], allb[:, 0]
