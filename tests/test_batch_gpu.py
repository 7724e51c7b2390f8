"""Throughput mode (pats_amd.batch.forward_pairs: a batch of pairs, every stage one launch, NO host read) against
 (a) the reference's own chained functions (tests/golden/pipeline_*.npz, tools/make_golden.py::gen_pipeline) and
 (b) pats_amd.pipeline.forward_path run pair by pair (itself held to those goldens): bit-identical matches.
The network callbacks of these tests place synthetic per-chunk tensors by reading the row table back - the PATH makes
no host read, its stand-in networks may."""
import numpy as np
import pytest
import torch

from conftest import golden
from pats_amd import synth

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


class _BatchNets:
    """synth.SynthNets (numpy, one per pair) as the callbacks of pats_amd.batch: chunk c of pair p gets
    nets[p].fine(c, B) / nets[p].third(c, P) - the same tensors pipeline.forward_path hands its per-chunk calls."""

    def __init__(self, nets):
        self.nets = nets

    def coarse(self, lefts, rights):
        c = [n.coarse() for n in self.nets]
        return (cu(np.concatenate([x["d0"] for x in c])), cu(np.concatenate([x["d1"] for x in c])),
                cu(np.concatenate([x["ns"] for x in c])), float(c[0]["alpha"]))

    def fine(self, rows, new_left, new_right):
        cell = rows.row_cell.cpu().numpy()
        base = rows.chunk_base.cpu().numpy()
        N = rows.h * rows.w
        f0 = np.zeros((rows.rows_cap, 264, 145), np.float32)
        f1 = np.zeros_like(f0)
        sx = np.ones((rows.rows_cap, 1, 144), np.float32)
        sy = np.ones_like(sx)
        self.blocks = []                                  # (chunk, pair, first row, rows)
        for c in range(rows.Cmax):
            r0, r1 = int(base[c]), int(base[c + 1])
            pair = cell[r0:r1] // N
            for p in np.unique(pair):
                idx = np.nonzero(pair == p)[0] + r0
                assert (np.diff(idx) == 1).all()          # a (chunk, pair) block is contiguous
                f = self.nets[p].fine(c, len(idx))
                f0[idx], f1[idx], sx[idx], sy[idx] = f["d0"], f["d1"], f["scale_x"], f["scale_y"]
                self.blocks.append((c, int(p), int(idx[0]), len(idx)))
        return cu(f0), cu(f1), cu(sx), cu(sy)

    def third(self, rows, mk0, mk1, b_ids, P_dev):
        P = int(P_dev.item())
        cap = mk0.shape[0]
        assert P <= cap
        b = b_ids[:P].cpu().numpy()
        d0 = np.zeros((cap, 128, 65), np.float32)
        d1 = np.zeros_like(d0)
        sc = np.ones((cap, 1, 64), np.float32)
        for c, p, r0, n in self.blocks:
            idx = np.nonzero((b >= r0) & (b < r0 + n))[0]
            if len(idx) == 0:
                continue
            assert (np.diff(idx) == 1).all()
            t = self.nets[p].third(c, len(idx))
            d0[idx], d1[idx], sc[idx] = t["d0"], t["d1"], t["scale"]
        return cu(d0), cu(d1), cu(sc)


@pytest.mark.parametrize("name", ["pipeline_outdoor.npz", "pipeline_indoor.npz", "pipeline_640x480_outdoor.npz",
                                  "pipeline_640x480_indoor.npz"])
def test_one_pair_against_the_reference_chain(name):
    from pats_amd import batch
    g = golden(name)
    nets = synth.SynthNets(seed=int(g["seed"]), h=int(g["h"]), w=int(g["w"]))
    left, right = [cu(x) for x in nets.images()]
    cap = batch.Capacities(1, int(g["h"]), int(g["w"]), if_local=bool(g["if_local"]))
    out = batch.forward_pairs(left, right, _BatchNets([nets]), cap, if_outdoor=bool(g["if_outdoor"]),
                              merge_new=bool(g["merge_new"]))
    (ml, mr), = batch.split_by_pair(out, cap)
    ml, mr = ml.cpu().numpy(), mr.cpu().numpy()
    assert ml.shape == g["matches_l"].shape and ml.shape[0] > 500
    np.testing.assert_allclose(ml, g["matches_l"], atol=1e-4, rtol=1e-6)
    np.testing.assert_allclose(mr, g["matches_r"], atol=6e-3, rtol=1e-6)       # gate of test_pipeline_chain
    chunks = g["chunks"]
    base = out["rows"].chunk_base.cpu().numpy()
    assert np.diff(base)[:len(chunks)].tolist() == chunks[:, 0].tolist() and int(base[-1]) == int(chunks[:, 0].sum())
    assert int(out["P"].item()) == int(chunks[:, 1].sum()) and int(out["M"].item()) == int(chunks[:, 2].sum())


@pytest.mark.parametrize("if_local,outdoor,new", [(True, True, True), (False, False, False)])
def test_three_different_pairs_equal_the_per_pair_path(if_local, outdoor, new):
    """Pairs with different match counts, chunk counts and crops in ONE batch: every pair's matches are bit-identical
    to pipeline.forward_path on that pair alone (same kernels, same per-problem inputs - nothing may leak between
    pairs, chunk blocks or padding rows)."""
    from pats_amd import batch, pipeline
    from test_gpu_parity import _CudaNets
    h, w = 15, 20
    seeds = [synth.SEED + 40, synth.SEED + 1040, synth.SEED + 2040]
    nets = [synth.SynthNets(seed=s, h=h, w=w) for s in seeds]
    imgs = [n.images() for n in nets]
    lefts = cu(np.concatenate([i[0] for i in imgs]))
    rights = cu(np.concatenate([i[1] for i in imgs]))
    cap = batch.Capacities(3, h, w, if_local=if_local)
    out = batch.forward_pairs(lefts, rights, _BatchNets(nets), cap, if_outdoor=outdoor, merge_new=new)
    per_pair = batch.split_by_pair(out, cap)
    total = 0
    for p, n in enumerate(nets):
        ref = pipeline.forward_path(lefts[p:p + 1], rights[p:p + 1], _CudaNets(n), if_local=if_local, if_outdoor=outdoor,
                                    merge_new=new)
        assert ref["matches_l"].shape[0] > 500
        assert torch.equal(per_pair[p][0], ref["matches_l"]) and torch.equal(per_pair[p][1], ref["matches_r"]), p
        total += ref["matches_l"].shape[0]
    assert total == int(out["M"].item())
    assert len({pp[0].shape[0] for pp in per_pair}) == 3           # the pairs really differ


def test_capacity_overflow_is_reported_not_silent():
    from pats_amd import batch
    nets = synth.SynthNets(seed=synth.SEED + 40, h=15, w=20)
    left, right = [cu(x) for x in nets.images()]
    cap = batch.Capacities(1, 15, 20, if_local=True, p_cap_per_pair=64)

    class Small(_BatchNets):
        def third(self, rows, mk0, mk1, b_ids, P_dev):
            cap_ = mk0.shape[0]
            return (torch.zeros((cap_, 128, 65), device="cuda"), torch.zeros((cap_, 128, 65), device="cuda"),
                    torch.ones((cap_, 1, 64), device="cuda"))
    out = batch.forward_pairs(left, right, Small([nets]), cap)
    assert int(out["P"].item()) > cap.P_cap
    with pytest.raises(RuntimeError, match="P_cap"):
        batch.split_by_pair(out, cap)


def test_row_table_against_the_host_planner_on_random_grids():
    """ops.chunk_rows (csrc/batch.hip) against a numpy restatement built on the HOST planner ops.split_patches (itself held
    to the reference's golden plans): cumulative counts, plans, chunk masks (first_layer.py:137-138), the rows in (chunk,
    pair, cell) order, the tail rows of pats.py:38-39, crop indices - on random grids, chunk caps and match densities,
    including pairs without a match and fully matched ones."""
    from pats_amd import ops
    rng = np.random.default_rng(404)
    for trial in range(40):
        h, w = int(rng.integers(2, 25)), int(rng.integers(2, 33))
        N = h * w
        cap = int(rng.choice([2 * w, 512, max(2, w // 2), 7]))
        pairs = int(rng.integers(1, 6))
        dens = rng.choice([0.0, 0.05, 0.5, 0.9, 1.0], size=pairs)
        ifn1 = np.stack([rng.random(N) >= d for d in dens])                 # True = no match
        Cmax = ops.max_chunks(h, w, cap)
        rows = ops.chunk_rows(cu(ifn1), h, w, cap)
        assert rows.Cmax == Cmax and int(rows.status.item()) == 0
        sc = np.cumsum(~ifn1, axis=1).astype(np.int32)
        assert np.array_equal(rows.sum_cycle.cpu().numpy(), sc)
        want_cell, want_forced, want_crop, base = [], [], [], [0]
        masks = np.ones((Cmax, pairs, N), bool)
        plans = [ops.split_patches(sc[p], h, w, cap) for p in range(pairs)]
        crop_base = np.concatenate([[0], np.cumsum(sc[:, -1])])
        for c in range(Cmax):
            for p in range(pairs):
                n, second, third = plans[p]
                if c >= n:
                    continue
                lo, hi = second[c]
                m = ifn1[p] | (sc[p] <= lo) | (sc[p] > hi)
                masks[c, p] = m
                cells = np.nonzero(~m)[0]
                tail = third[c][1]
                for rank, q in enumerate(cells):
                    want_cell.append(p * N + q)
                    want_forced.append(bool(rank >= len(cells) - tail) if tail > 0 else (bool(rank >= -tail) if tail < 0 else False))
                    want_crop.append(crop_base[p] + sc[p, q] - 1)
            base.append(len(want_cell))
        total = len(want_cell)
        assert rows.chunk_base.cpu().numpy().tolist() == base, (trial, h, w, cap)
        assert np.array_equal(rows.masks.cpu().numpy(), masks)
        assert np.array_equal(rows.row_cell.cpu().numpy()[:total], np.array(want_cell, np.int32).reshape(-1))
        assert np.array_equal(rows.row_forced.cpu().numpy()[:total].astype(bool), np.array(want_forced, bool).reshape(-1))
        assert np.array_equal(rows.row_crop.cpu().numpy()[:total], np.array(want_crop, np.int32).reshape(-1))
        assert (rows.row_cell.cpu().numpy()[total:] == -1).all() and (rows.row_forced.cpu().numpy()[total:] == 1).all()
        for p in range(pairs):
            n, second, third = plans[p]
            assert int(rows.cycle_num[p].item()) == n
            assert rows.second[p, :n].cpu().numpy().tolist() == [list(x) for x in second]
            assert rows.third[p, :n].cpu().numpy().tolist() == [list(x) for x in third]
    # capacities that do not hold the batch are reported, not silently truncated
    ifn1 = np.zeros((2, 300), bool)
    small = ops.chunk_rows(cu(ifn1), 15, 20, 40, rows_cap=100)
    assert int(small.status.item()) & 2
    few = ops.chunk_rows(cu(ifn1), 15, 20, 40, Cmax=2)
    assert int(few.status.item()) & 1


@pytest.mark.parametrize("new", [True, False])
def test_batched_merge_and_result_against_the_single_chunk_ops(new):
    """ops.merge_patches_batch / ops.get_result_chunks on random trust scores and flags against the per-chunk ops the
    reference's loop would call (ops.merge_patches_new / _old chunk after chunk with the scores_back hand-over and the
    pats.py:38-39 tail rows; ops.get_result with the expanded (chunk, pair) batch): bit-identical."""
    from pats_amd import ops
    rng = np.random.default_rng(99 + int(new))
    pairs, h, w, cap = 3, 6, 7, 14
    N, H, W = h * w, 32 * h, 32 * w
    ifn1 = np.stack([rng.random(N) >= d for d in (0.9, 0.5, 0.97)])
    rows = ops.chunk_rows(cu(ifn1), h, w, cap)
    total = int(rows.chunk_base[-1].item())
    R = rows.rows_cap
    trust = (rng.random((R, 144)) * 1.2).astype(np.float32)
    ifn2 = rng.random((R, 144)) < 0.3
    t_b, f_b = cu(trust.copy()), cu(ifn2.copy())
    merged = ops.merge_patches_batch(new, rows, t_b, (H, W), f_b)
    # the reference's order: per pair, chunk after chunk
    cell = rows.row_cell.cpu().numpy()
    base = rows.chunk_base.cpu().numpy()
    masks = rows.masks.cpu().numpy()
    third = rows.third.cpu().numpy()
    want = np.ones((R, 144), bool)
    want_t, want_f = trust.copy(), ifn2.copy()           # the in-place updates of the reference (second_layer.py:194-201)
    merge = ops.merge_patches_new if new else ops.merge_patches_old
    for p in range(pairs):
        sb = torch.zeros((1, N, 16, 9), dtype=torch.float64, device="cuda")
        for c in range(rows.Cmax):
            idx = np.nonzero(cell[base[c]:base[c + 1]] // N == p)[0] + int(base[c])
            if len(idx) == 0:
                continue
            t_in, f_in = cu(trust[idx].copy()), cu(ifn2[idx].copy())
            out, sb = merge(len(idx), t_in, (H, W), cu(masks[c, p:p + 1]), f_in, sb)
            want_t[idx], want_f[idx] = t_in.cpu().numpy(), f_in.cpu().numpy()
            out = out.cpu().numpy()
            tail = int(third[p, c, 1])
            if tail != 0:
                out[-tail:, :] = True
            want[idx] = out
    assert np.array_equal(merged.cpu().numpy(), want)
    assert merged[total:].all()
    assert np.array_equal(t_b.cpu().numpy()[:total], want_t[:total]) and np.array_equal(f_b.cpu().numpy()[:total], want_f[:total])
    # the same table walked chunk by chunk on each chunk's own tensors (pats_merge_patches_chunks: pipeline.forward_chunks_device),
    # scores_back handed from call to call
    sbw = torch.empty((pairs, N, 16, 9), dtype=torch.float64, device="cuda")
    first = True
    for c in range(rows.Cmax):
        lo, hi = int(base[c]), int(base[c + 1])
        if hi <= lo:
            continue
        t_c, f_c = cu(trust[lo:hi].copy()), cu(ifn2[lo:hi].copy())
        out_c = ops.merge_patches_chunk(new, rows, c, lo, t_c, (H, W), f_c, sbw, first=first)
        first = False
        assert np.array_equal(out_c.cpu().numpy(), want[lo:hi]), "chunk %d" % c
        assert np.array_equal(t_c.cpu().numpy(), want_t[lo:hi]) and np.array_equal(f_c.cpu().numpy(), want_f[lo:hi])
    # get_result for all chunks in one call against the expanded batch
    f16 = rng.random((R, 2304)) < 0.7
    f16[total:] = True
    pts16 = (rng.random((R, 2304, 2)) * 96).astype(np.float32)
    avn = (rng.random((pairs, N, 2)) * 600).astype(np.float32)
    xsn = (0.2 + rng.random((pairs, N, 2))).astype(np.float32)
    ml, mr, mrow, M = ops.get_result_chunks(rows, cu(f16), cu(avn), cu(pts16), cu(xsn))
    C = rows.Cmax
    masks_flat = cu(masks.reshape(C * pairs, N))
    av_c = cu(np.tile(avn, (C, 1, 1)))
    xs_c = cu(np.tile(xsn, (C, 1, 1)))
    sc_rows = xs_c.reshape(-1, 2)[torch.nonzero(~masks_flat.reshape(-1)).flatten()]
    wl, wr = ops.get_result(C * pairs, [masks_flat, cu(f16[:total])], [av_c.flip(dims=[2]) / 32.0, cu(pts16[:total]).flip(dims=[2]) / 2.0],
                            [xs_c, sc_rows], [[32, h, w], [2, 48, 48]],
                            [torch.ones(C * pairs, dtype=torch.bool, device="cuda"), torch.ones(total, dtype=torch.bool, device="cuda")])
    m = int(M.item())
    assert m == wl.shape[0] > 1000
    assert torch.equal(ml[:m], wl) and torch.equal(mr[:m], wr)
    assert bool((mrow[:m] >= 0).all()) and bool((mrow[:m] < total).all())


@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
def test_counted_fine_level_launches_skip_the_padding_rows(layout):
    """The fine level over a CAPACITY with the row count on the device (pats_cost_ot_flags_counted_f32,
    pats_iterative_expand_counted_f32, pats_fine_descriptors_counted_f32): the first `count` rows equal the plain launch
    over exactly those rows bit for bit; rows past the count keep what the buffers held (nothing is computed for them -
    NaN-filled padding descriptors trip no guard and cause no log-domain redo)."""
    from pats_amd import ops
    cap, live = 37, 23
    gen = torch.Generator(device="cuda")
    gen.manual_seed(synth.SEED + 77)
    maps = [torch.randn(sh, device="cuda", generator=gen) for sh in ((2 * cap, 64, 48, 48), (2 * cap, 64, 24, 24), (2 * cap, 128, 12, 12))]
    if layout == "channels_last":
        maps = [m.contiguous(memory_format=torch.channels_last) for m in maps]
    title, rub = torch.randn((cap, 8), device="cuda", generator=gen), torch.randn((cap, 264), device="cuda", generator=gen)
    cnt = torch.tensor([live], dtype=torch.int64, device="cuda")
    full = ops.fine_descriptors(maps, title, rub)
    out = torch.full((2, cap, 264, 145), float("nan"), device="cuda")
    ops.fine_descriptors(maps, title, rub, out=out, count=cnt)
    assert torch.equal(out[:, :live], full[:, :live]) and bool(torch.isnan(out[:, live:]).all())

    # descriptors with structure (a match per row), padding rows NaN
    base = torch.randn((cap, 264, 145), device="cuda", generator=gen)
    d0 = (3.0 * (base + 0.3 * torch.randn((cap, 264, 145), device="cuda", generator=gen))).contiguous()
    d1 = (3.0 * (base + 0.3 * torch.randn((cap, 264, 145), device="cuda", generator=gen))).contiguous()
    d0[live:], d1[live:] = float("nan"), float("nan")
    sx = torch.exp(0.3 * torch.randn((cap, 1, 144), device="cuda", generator=gen))
    sy = torch.exp(0.3 * torch.randn((cap, 1, 144), device="cuda", generator=gen))
    ns = (sx * sy).contiguous()
    one = torch.tensor(1.0, device="cuda")
    ops.sinkhorn_fallbacks(reset=True)
    Zr, fr = ops.cost_ot(d0[:live].contiguous(), d1[:live].contiguous(), 2, one, ns[:live].contiguous(), 100, bias_k=2.0, return_flags=True)
    assert ops.sinkhorn_fallbacks(reset=True) == 0
    Zc, fc = ops.cost_ot(d0, d1, 2, one, ns, 100, bias_k=2.0, return_flags=True, count=cnt)
    assert ops.sinkhorn_fallbacks(reset=True) == 0, "a padding row reached the log-domain redo"
    assert torch.equal(Zc[:live], Zr) and torch.equal(fc[:live], fr)
    er = ops.est_position_second(Zr, sx[:live].contiguous(), sy[:live].contiguous(), [96, 96], 8, col_nomatch=fr)
    Zc[live:] = float("nan")
    ec = ops.est_position_second(Zc, sx, sy, [96, 96], 8, col_nomatch=fc, count=cnt)
    for a, b in zip(ec, er):
        assert torch.equal(a[:live], b)
    # count = 0 and count > capacity are both legal
    ops.cost_ot(d0, d1, 2, one, ns, 100, bias_k=2.0, return_flags=True, count=torch.zeros(1, dtype=torch.int64, device="cuda"))
    Zb, _ = ops.cost_ot(d0[:live].contiguous(), d1[:live].contiguous(), 2, one, ns[:live].contiguous(), 100, bias_k=2.0,
                        return_flags=True, count=torch.tensor([10 ** 6], dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    assert torch.equal(Zb, Zr)


def test_matches_grouped_by_pair_on_the_device_equal_the_stable_argsort():
    """pats_matches_by_pair_f32 (run boundaries, offsets, copy - no host read) against the round-3 hand-over: stable argsort of
    the matches' pair index + bincount, on a batch of three different pairs with a pair that matches nothing in one chunk."""
    from pats_amd import batch, ops
    h, w = 15, 20
    nets = [synth.SynthNets(seed=s, h=h, w=w) for s in (synth.SEED + 40, synth.SEED + 1040, synth.SEED + 2040)]
    imgs = [n.images() for n in nets]
    lefts = cu(np.concatenate([i[0] for i in imgs]))
    rights = cu(np.concatenate([i[1] for i in imgs]))
    cap = batch.Capacities(3, h, w, if_local=True)
    out = batch.forward_pairs(lefts, rights, _BatchNets(nets), cap, if_outdoor=True, merge_new=True)
    M = int(out["M"].item())
    rows = out["rows"]
    pair = torch.div(rows.row_cell[out["match_row"][:M].long()], cap.N, rounding_mode="floor")
    order = torch.argsort(pair, stable=True)
    want_l, want_r = out["matches_l"][:M][order], out["matches_r"][:M][order]
    counts = torch.bincount(pair, minlength=cap.pairs)
    ml, mr, off = batch.group_by_pair(out, cap)
    torch.cuda.synchronize()
    assert off.cpu().tolist() == [0] + torch.cumsum(counts, 0).cpu().tolist()
    assert M > 0 and torch.equal(ml[:M], want_l) and torch.equal(mr[:M], want_r)
    # the step's summary behind the offsets (pats_matches_by_pair_summary_f32): M, P, table status - one copy hands a step over
    assert out["summary"].cpu().tolist() == off.cpu().tolist() + [M, int(out["P"].item()), int(out["status"].item())]
    ml2, mr2, off2 = ops.matches_by_pair(rows, out["matches_l"], out["matches_r"], out["match_row"], out["M"])    # without the summary
    assert torch.equal(off2, off) and torch.equal(ml2[:M], ml[:M]) and torch.equal(mr2[:M], mr[:M])
    per_pair = batch.split_by_pair(out, cap)
    assert [int(a.shape[0]) for a, _ in per_pair] == counts.cpu().tolist()
