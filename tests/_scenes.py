"""Seeded synthetic two-view geometry shared by tests/golden/make_golden_metrics.py and the evaluation tests."""
import numpy as np


def random_rotation(rng, max_deg):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(-max_deg, max_deg))
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def make_scene(seed, counts, noise_px=0.5, outlier_frac=0.2, hw=(480, 640)):
    """-> dict of float32/int64 arrays: K0, K1 [N,3,3], T_0to1 [N,4,4], mkpts0_f, mkpts1_f [M,2], m_bids [M]."""
    rng = np.random.default_rng(seed)
    N = len(counts)
    K0 = np.zeros((N, 3, 3)); K1 = np.zeros((N, 3, 3)); T = np.zeros((N, 4, 4))
    p0s, p1s, bids = [], [], []
    for b, m in enumerate(counts):
        for K in (K0, K1):
            f = rng.uniform(450, 650)
            K[b] = [[f, 0, hw[1] / 2 + rng.uniform(-20, 20)], [0, f * rng.uniform(0.98, 1.02), hw[0] / 2 + rng.uniform(-20, 20)], [0, 0, 1]]
        R = random_rotation(rng, 25)
        t = rng.normal(size=3) * 0.4
        T[b] = np.eye(4); T[b, :3, :3] = R; T[b, :3, 3] = t
        X = np.stack([rng.uniform(-2, 2, m), rng.uniform(-1.5, 1.5, m), rng.uniform(2, 8, m)], 1)
        x0 = (K0[b] @ X.T).T; x0 = x0[:, :2] / x0[:, 2:]
        X1 = (R @ X.T).T + t
        x1 = (K1[b] @ X1.T).T; x1 = x1[:, :2] / x1[:, 2:]
        x1 = x1 + rng.normal(size=x1.shape) * noise_px
        out = rng.random(m) < outlier_frac
        x1[out] = rng.uniform([0, 0], [hw[1], hw[0]], size=(int(out.sum()), 2))
        p0s.append(x0); p1s.append(x1); bids.append(np.full(m, b))
    return dict(K0=K0.astype(np.float32), K1=K1.astype(np.float32), T_0to1=T.astype(np.float32),
                mkpts0_f=np.concatenate(p0s).astype(np.float32), mkpts1_f=np.concatenate(p1s).astype(np.float32),
                m_bids=np.concatenate(bids).astype(np.int64))
