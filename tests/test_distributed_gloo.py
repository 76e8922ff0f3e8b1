"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: shard bounds, count all-gather,
global rebasing, padded match all-gather.  The HIP kernels are not involved (no GPU here); each
rank fabricates the rank-local batch dict a forward() would have produced."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from loftr_amd.distributed import shard_bounds, all_gather_match_counts, globalize, all_gather_matches


def test_shard_bounds_cover():
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_local(rank, world, n_global, seed=0):
    """Deterministic fake per-pair match lists for the pairs this rank owns."""
    lo, hi = shard_bounds(n_global, rank, world)
    rng = np.random.default_rng(seed)
    counts_all = rng.integers(0, 6, n_global)
    b, rows = [], []
    for g in range(lo, hi):
        for k in range(counts_all[g]):
            b.append(g - lo)
            rows.append([g, k, g + 0.5, k + 0.25, 0.1 * (k + 1)])
    rows = np.asarray(rows, np.float32).reshape(-1, 5)
    data = {"b_ids": torch.tensor(b, dtype=torch.int64),
            "mkpts0_f": torch.from_numpy(rows[:, 0:2].copy()), "mkpts1_f": torch.from_numpy(rows[:, 2:4].copy()),
            "mconf": torch.from_numpy(rows[:, 4].copy()),
            "_match_counts": torch.tensor([len(b)] + list(counts_all[lo:hi]), dtype=torch.int32)}
    return data, counts_all


def _worker(rank, world, port, n_global, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data, counts_all = _fake_local(rank, world, n_global)
        counts = all_gather_match_counts(data["_match_counts"][1:], n_global)
        assert counts.tolist() == counts_all.tolist()
        globalize(data, n_global)
        lo, hi = shard_bounds(n_global, rank, world)
        assert data["match_offset"] == int(counts_all[:lo].sum())
        assert (data["b_ids_global"] >= lo).all() and (data["b_ids_global"] < max(hi, lo + 1)).all()
        rows, gb = all_gather_matches(data)
        assert rows.shape[0] == int(counts_all.sum())
        assert (np.diff(gb.numpy()) >= 0).all()                      # ascending global pair index
        assert np.array_equal(rows[:, 0].numpy(), gb.numpy().astype(np.float32))   # mkpts0_f.x encodes g
        assert np.array_equal(np.bincount(gb.numpy(), minlength=n_global), counts_all)
        q.put((rank, "ok"))
    except Exception as e:      # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [2, 5, 16])
def test_gloo_world2(n_global):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_global) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _fake_outputs(rank, world, n_pairs=12, seed=3):
    """Per-rank `test_step` outputs the way a DistributedSampler would deal them: pair i goes to rank i % world and the
    sampler pads the last rank with a repeat of pair 0 (same identifier: must be counted once)."""
    rng = np.random.default_rng(seed)
    R = rng.gamma(1.5, 4.0, n_pairs); t = rng.gamma(1.5, 6.0, n_pairs)
    epi = [rng.gamma(0.6, 4e-4, int(rng.integers(1, 50))).astype(np.float32) for _ in range(n_pairs)]
    mine = [i for i in range(n_pairs) if i % world == rank]
    if world > 1 and rank == world - 1:
        mine.append(0)
    outs = []
    for i in mine:
        outs.append({"metrics": {"identifiers": [f"pair{i}"], "epi_errs": [epi[i]], "R_errs": [R[i]], "t_errs": [t[i]],
                                 "inliers": [np.zeros(0, bool)]},
                     "dumps": [{"identifier": f"pair{i}", "epi_errs": epi[i]}]})
    return outs


def _eval_worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from loftr_amd import evaluation
        res = evaluation.test_epoch_end(_fake_outputs(rank, world), dump_dir=tmp)
        q.put((rank, res))
    except Exception as e:      # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_evaluation_gather(tmp_path):
    """test_epoch_end over two ranks == the single-process aggregation of all pairs; one dump file, duplicates kept in
    the dump (as the reference) but counted once in the metrics."""
    from loftr_amd import evaluation
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    single = evaluation.test_epoch_end(_fake_outputs(0, 1))
    assert isinstance(res[0], dict), res
    assert res[0] == res[1]
    for k, v in single.items():
        assert abs(res[0][k] - v) < 1e-12, (k, res[0][k], v)
    dumped = np.load(os.path.join(str(tmp_path), "LoFTR_pred_eval.npy"), allow_pickle=True)
    assert len(dumped) == 13 and sum(d["identifier"] == "pair0" for d in dumped) == 2
