"""CPU, world_size 2, gloo: the N>1 path (scan sharding + pose gather) is correct by construction.
The per-rank registration call is stood in for by the CPU oracle here (tests may use it); on the
GPU box the same plumbing wraps capi.Handle.match_scan2map_batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from msf_loam_amd import dist as mdist
from tests import common


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _case(orc, n=5):
    _, mc, ms = common.small_world(20000)
    cs, ss, co, so, guesses = [], [], [0], [0], []
    for pts, ring, truth, guess in common.scans(n, 20000):
        _, corner, surf = common.features_from_oracle(orc, pts, ring)
        cs.append(corner); ss.append(surf); co.append(co[-1] + len(corner)); so.append(so[-1] + len(surf)); guesses.append(guess)
    return mc, ms, np.concatenate(cs), np.array(co, np.int32), np.concatenate(ss), np.array(so, np.int32), np.stack(guesses)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    mc, ms, C, co, S, so, G = _case(orc)
    # map replicated from rank 0 (other ranks start without it)
    bc, bs = mdist.broadcast_map(mc if rank == 0 else np.zeros((0, 4), np.float32), ms if rank == 0 else np.zeros((0, 4), np.float32))
    bc, bs = (np.asarray(t) if not torch.is_tensor(t) else t.numpy() for t in (bc, bs))
    assert np.array_equal(bc, mc) and np.array_equal(bs, ms)
    reg = lambda c, c_off, s, s_off, g: orc.match_scan2map_batch(bc, bs, c, c_off, s, s_off, g)
    poses, status = mdist.register_sharded(reg, C, co, S, so, G)
    # equal-B gather path (bench.py): every rank contributes its own block
    lo, hi = mdist.shard_bounds(4, rank, world)
    pg = mdist.PoseGather(hi - lo, torch.device("cpu"))
    ap, ast = pg.all_gather(torch.from_numpy(poses[lo:hi].copy()), torch.from_numpy(status[lo:hi].copy()))
    if rank == 0:
        ret["poses"] = poses; ret["status"] = status; ret["eq"] = ap.reshape(-1, 7).numpy().copy()
    dist.destroy_process_group()


def test_sharded_registration_equals_single_process(oracle):
    mc, ms, C, co, S, so, G = _case(oracle)
    want, st = oracle.match_scan2map_batch(mc, ms, C, co, S, so, G)
    mgr = mp.Manager(); ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert np.array_equal(ret["poses"], want), "sharded + gathered poses must equal the unsharded run bit for bit"
    assert np.array_equal(ret["status"], st)
    assert np.array_equal(ret["eq"], want[:4])


def _pairs_case(orc, P=5):
    """P pairs with different (thinned) maps of the small world."""
    rng = np.random.default_rng(3)
    _, mc, ms = common.small_world(20000)
    mcs, mss, cs, ss, gs = [], [], [], [], []
    for k, (pts, ring, truth, guess) in enumerate(common.scans(P, 20000)):
        _, corner, surf = common.features_from_oracle(orc, pts, ring)
        mcs.append(mc[rng.uniform(size=len(mc)) < 0.9]); mss.append(ms[rng.uniform(size=len(ms)) < 0.7 + 0.05 * k])
        cs.append(corner); ss.append(surf); gs.append(guess)
    cat = lambda ls: (np.concatenate(ls), np.cumsum([0] + [len(a) for a in ls]).astype(np.int32))   # noqa: E731
    return cat(mcs), cat(mss), cat(cs), cat(ss), np.stack(gs)


def _oracle_pairs(orc):
    def fn(mc, mco, ms, mso, c, co, s, so, g):
        poses, status = [], []
        for p in range(len(g)):
            rc, pose, _ = orc.match_scan2map(mc[mco[p]:mco[p + 1]], ms[mso[p]:mso[p + 1]], c[co[p]:co[p + 1]], s[so[p]:so[p + 1]], g[p])
            poses.append(pose); status.append(rc)
        return np.array(poses).reshape(-1, 7), np.array(status, np.int32)
    return fn


def _pairs_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    (mc, mco), (ms, mso), (c, co), (s, so), g = _pairs_case(orc)
    poses, status = mdist.register_pairs_sharded(_oracle_pairs(orc), mc, mco, ms, mso, c, co, s, so, g)
    if rank == 0:
        ret["poses"] = poses; ret["status"] = status
    dist.destroy_process_group()


def test_sharded_pairs_equal_single_process(oracle):
    """"Many map-submap pairs" sharded over two ranks (5 pairs: 3 + 2): every rank registers its own pairs against its own maps,
    nothing is broadcast, the gathered poses equal the unsharded run bit for bit."""
    (mc, mco), (ms, mso), (c, co), (s, so), g = _pairs_case(oracle)
    want, st = _oracle_pairs(oracle)(mc, mco, ms, mso, c, co, s, so, g)
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_pairs_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert np.array_equal(ret["poses"], want) and np.array_equal(ret["status"], st)


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 8, 1024, 1031):
        for w in (1, 2, 3, 8):
            spans = [mdist.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    off = np.array([0, 3, 7, 7, 12])
    o, (a, b) = mdist.shard_offsets(off, 1, 3)
    assert list(o) == [0, 4, 4] and (a, b) == (3, 7)
