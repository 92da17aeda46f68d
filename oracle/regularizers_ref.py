"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatements of the third-party ops the reference's
regularisers call and that are absent from /root/reference -- PARITY UNPINNED, they follow the packages' documented
behaviour:
  pytorch3d.ops.ball_query               (first K points within the radius, index order)   utils/deform_utils.py:123
  pytorch3d.ops.sample_farthest_points   (start at 0, squared distances, first maximum)    main_train_dimo.py:513
  chamferdist.ChamferDistance forward    (sum of squared NN distances, batch mean)         main_train_dimo.py:299
"""
import numpy as np


def ball_query_ref(p1, p2, K, radius):
    T, N1 = p1.shape[:2]
    idx = -np.ones((T, N1, K), np.int64)
    dist = np.zeros((T, N1, K), np.float32)
    for t in range(T):
        d2 = ((p1[t][:, None, :] - p2[t][None, :, :]) ** 2).sum(-1)
        for i in range(N1):
            hit = np.nonzero(d2[i] < radius * radius)[0][:K]
            idx[t, i, :len(hit)] = hit
            dist[t, i, :len(hit)] = d2[i, hit]
    return dist, idx


def farthest_point_sample_ref(xyz, K):
    xyz = np.asarray(xyz, np.float32)
    n = xyz.shape[0]
    out = np.zeros(K, np.int64)
    min_d = np.full(n, np.inf, np.float32)
    sel = 0
    for k in range(1, K):
        d = ((xyz - xyz[sel]) ** 2).astype(np.float32)
        d = (d[:, 0] + d[:, 1]) + d[:, 2]
        min_d = np.minimum(min_d, d)
        sel = int(np.argmax(min_d))  # first maximum
        out[k] = sel
    return out


def chamfer_forward_ref(source, target):
    d2 = ((source[:, :, None, :].astype(np.float64) - target[:, None, :, :]) ** 2).sum(-1)
    return d2.min(axis=2).sum(axis=1).mean()
