"""CPU, world_size = 2, gloo: the data-parallel exchange pattern of disvae_amd.parallel.

(1) the Comm collectives (product code) deliver what fused_step assumes (rank-ordered
    gathers, summed column gradients sliced to the local rows);
(2) the sharding scheme of the btcvae step -- local rows of the GLOBAL B x B estimator,
    all-gathered latents, all-reduced column gradients, all-reduced flat gradient arena and
    packed loss sums -- reproduces the single-process global-batch gradients and loss.  The
    compute is done by the oracle here (there is no GPU in this test); the sequence of
    exchanges is the one of models/losses.py::_SingleOptimizerLoss.fused_step.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import disvae_oracle as O

HP = dict(n_data=202599, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)
IMG, B_LOCAL, WORLD = (1, 32, 32), 4, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from disvae_amd import parallel
    parallel.init_process_group_from_env("gloo")
    comm = parallel.Comm()
    try:
        # ---- (1) collectives ------------------------------------------------------------
        D = 10
        z = torch.full((B_LOCAL, D), float(rank)); mu = z + 10; lv = z + 20
        zg, mug, lvg = comm.all_gather_latents(z, mu, lv)
        assert zg.shape == (world * B_LOCAL, D)
        for r in range(world):
            sl = slice(r * B_LOCAL, (r + 1) * B_LOCAL)
            assert (zg[sl] == r).all() and (mug[sl] == r + 10).all() and (lvg[sl] == r + 20).all()
        rows = comm.all_gather_rows(torch.arange(B_LOCAL * D, dtype=torch.float32).view(B_LOCAL, D) + 1000 * rank)
        assert rows.shape == (world * B_LOCAL, D) and rows[B_LOCAL, 0] == 1000
        a = torch.arange(world * B_LOCAL * D, dtype=torch.float32).view(-1, D) * (rank + 1)
        da, db = comm.reduce_scatter_cols(a, 2 * a)
        full = torch.arange(world * B_LOCAL * D, dtype=torch.float32).view(-1, D) * sum(range(1, world + 1))
        sl = slice(rank * B_LOCAL, (rank + 1) * B_LOCAL)
        assert torch.equal(da, full[sl]) and torch.equal(db, 2 * full[sl])
        t = comm.all_reduce(torch.ones(5) * (rank + 1))
        assert (t == 3).all()

        # ---- (2) sharded btcvae step == global-batch step ----------------------------------
        torch.manual_seed(1234)
        params0 = O.init_vae_params(IMG, 10)
        gen = torch.Generator().manual_seed(99)
        Bg = world * B_LOCAL
        data_g = torch.rand((Bg,) + IMG, generator=gen, dtype=torch.float64)
        eps_g = torch.randn(Bg, 10, generator=gen, dtype=torch.float64)
        anneal, alpha, beta, gamma = 0.25, HP["btcvae_A"], HP["btcvae_B"], HP["btcvae_G"]
        # single-process reference at the global batch
        pr = O.clone_params(params0, dtype=torch.float64, requires_grad=True)
        recon, (m_, l_), z_ = O.vae_forward(pr, data_g, eps_g)
        mi, tc, dw = O.btcvae_terms(z_, m_, l_, HP["n_data"], True)
        ref_loss = O.reconstruction_loss(data_g, recon) + alpha * mi + beta * tc + anneal * gamma * dw
        ref_grads = torch.autograd.grad(ref_loss, list(pr.values()))
        # this rank's shard
        p = O.clone_params(params0, dtype=torch.float64, requires_grad=True)
        x, eps = data_g[sl], eps_g[sl]
        recon, (mu, lv), z = O.vae_forward(p, x, eps)
        rec_sum = torch.nn.functional.binary_cross_entropy(recon, x, reduction="sum")           # local sum
        zg, mug, lvg = comm.all_gather_latents(z.detach(), mu.detach(), lv.detach())
        zg, mug, lvg = (t.clone().requires_grad_(True) for t in (zg, mug, lvg))
        log_pz, log_qz, log_prod, log_qzcx = O.btcvae_log_densities(zg, mug, lvg, HP["n_data"], True)
        rows_obj = (alpha * (log_qzcx - log_qz) + beta * (log_qz - log_prod) + anneal * gamma * (log_prod - log_pz))[sl].sum() / Bg
        gz, gmu, glv = torch.autograd.grad(rows_obj, (zg, mug, lvg))
        # rows of other ranks contribute only through the COLUMN role of (mu, logvar): row-role
        # gradients live on the owning rank.  z only has a row role.
        dz_local = gz[sl]
        # split mu/logvar gradient into the row-role part (local, via log_q_zCx and the i = j cell)
        # and the column-role part: the product's bwd kernel returns exactly gmu/glv summed over local rows
        dmu_local, dlv_local = comm.reduce_scatter_cols(gmu, glv)
        packed = torch.stack((rec_sum.detach(), log_pz[sl].sum().detach(), log_qz[sl].sum().detach(),
                              log_prod[sl].sum().detach(), log_qzcx[sl].sum().detach()))
        comm.all_reduce(packed)
        mi_g = (packed[4] - packed[2]) / Bg; tc_g = (packed[2] - packed[3]) / Bg; dw_g = (packed[3] - packed[1]) / Bg
        loss_g = packed[0] / Bg + alpha * mi_g + beta * tc_g + anneal * gamma * dw_g
        # local backward: reconstruction (scaled by 1/Bg) + injected latent gradients
        obj = rec_sum / Bg + (z * dz_local).sum() + (mu * dmu_local).sum() + (lv * dlv_local).sum()
        grads = torch.autograd.grad(obj, list(p.values()))
        flat = torch.cat([g.reshape(-1) for g in grads])
        comm.all_reduce(flat)                                      # the flat-arena gradient all-reduce
        ref_flat = torch.cat([g.reshape(-1) for g in ref_grads])
        err = (flat - ref_flat).abs().max().item() / ref_flat.abs().max().item()
        assert err < 1e-10, err
        assert abs(loss_g.item() - ref_loss.item()) < 1e-9 * abs(ref_loss.item())
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in range(WORLD)]
    for p_ in procs:
        p_.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_mirrored(port, q):
    """MirroredWorldComm over a ONE-rank gloo group: what the absent peers of an identical-shard world would contribute."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.set_num_threads(1)
    from disvae_amd import parallel
    parallel.init_process_group_from_env("gloo")
    try:
        W, B, D = 4, 3, 5
        for rank in (0, 2):
            comm = parallel.MirroredWorldComm(parallel.Comm(), W, rank)
            assert (comm.world_size, comm.rank) == (W, rank)
            z = torch.arange(B * D, dtype=torch.float32).view(B, D)
            zg, mug, lvg = comm.all_gather_latents(z, z + 100, z + 200)
            assert zg.shape == (W * B, D)
            assert torch.equal(zg, z.repeat(W, 1)) and torch.equal(mug, (z + 100).repeat(W, 1)) and torch.equal(lvg, (z + 200).repeat(W, 1))
            rows = comm.all_gather_rows(z + 7)
            assert torch.equal(rows, (z + 7).repeat(W, 1))
            a = torch.arange(W * B * D, dtype=torch.float32).view(-1, D)
            da, db = comm.reduce_scatter_cols(a, 2 * a)
            sl = slice(rank * B, (rank + 1) * B)
            nx = slice(((rank + 1) % W) * B, ((rank + 1) % W + 1) * B)      # what this rank sends to a peer = what a peer sends here
            assert torch.equal(da, a[sl] + (W - 1) * a[nx]) and torch.equal(db, 2 * (a[sl] + (W - 1) * a[nx]))
            t = comm.all_reduce(torch.ones(6))
            assert (t == W).all()
            # the estimator's merged exchange: [dmu of all W*B columns | dlogvar | tail of loss sums] in ONE all-reduce
            n, tail = B * D, 4
            x0 = torch.arange(2 * W * n + tail, dtype=torch.float32)
            x = comm.all_reduce_cols_sums(x0.clone(), B, D, tail)
            o = (rank + 1) % W
            for slab in (0, 1):
                base = slab * W * n
                own, peer = x0[base + rank * n:base + (rank + 1) * n], x0[base + o * n:base + (o + 1) * n]
                assert torch.equal(x[base + rank * n:base + (rank + 1) * n], own + (W - 1) * peer)
            assert torch.equal(x[2 * W * n:], W * x0[2 * W * n:])
            u = torch.full((5,), 3.0)
            h = comm.all_reduce_async(u)
            h.wait()
            assert (u == 3.0 * W).all()
        with pytest.raises(ValueError):
            parallel.MirroredWorldComm(parallel.Comm(), 4, 4)
        q.put((0, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_mirrored_world_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_worker_mirrored, args=(_free_port(), q))
    p_.start()
    rank, msg = q.get(timeout=240)
    p_.join(timeout=60)
    assert msg == "ok", msg
