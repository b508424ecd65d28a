"""-m gpu: the PRODUCT data-parallel path (fused_step / call_optimize with a communicator) with
world_size = 2.  Both ranks share the single GPU of the test box, so the transport is gloo (RCCL
refuses two ranks on one device); the sharding logic, kernels and collectives' call sites are the
ones used with nccl on 2/4/8 GPUs.  Reference = the same engine run single-process on the global
batch: sharded gradients / losses / updated weights must agree to fp32 rounding."""
import os
import socket
from collections import defaultdict

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HP = dict(rec_dist="bernoulli", reg_anneal=10000, betaH_B=4, betaB_initC=0, betaB_finC=25, betaB_G=1000,
          factor_G=6.4, latent_dim=10, lr_disc=1e-4, btcvae_A=1, btcvae_B=6.4, btcvae_G=1)
IMG = (3, 64, 64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(loss, lr, D=10):
    from disvae_amd.models.vae import init_specific_model
    from disvae_amd.models.losses import get_loss_f
    torch.manual_seed(1234)
    model = init_specific_model("Burgess", IMG, D)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    loss_f = get_loss_f(loss, n_data=202599, device=torch.device("cuda"), **dict(HP, latent_dim=D))
    model.to("cuda").train()
    return model, opt, loss_f


def _worker(rank, world, port, loss, q, backend="gloo", transport="torch", device=0, replay=None):
    """replay="plan": three iterations on the same inputs -- the first one records the sharded step's launch plan (collectives
    included), the next two replay it; every iteration is compared with the single-process eager step on the global batch."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "disentangling-vae_amd"), os.path.join(root, "tests")):
            sys.path.insert(0, p)
        from disvae_amd import parallel
        torch.cuda.set_device(device)
        parallel.init_process_group_from_env(backend)
        lr = 1e-4 if loss == "factor" else 5e-4
        Bl = 12
        gen = torch.Generator().manual_seed(5)
        data_g = torch.rand((world * Bl,) + IMG, generator=gen)
        D = 10
        m0, o0, l0 = _make(loss, lr)             # single-process reference on the global batch (every rank computes it)
        l0.replay = None
        m1, o1, l1 = _make(loss, lr)             # the sharded run
        l1.replay = replay
        if backend == "gloo":
            from ddp_util import HostStagedComm
            comm = parallel.data_parallel(m1, l1, comm=HostStagedComm())
        else:
            comm = parallel.data_parallel(m1, l1, transport=transport)
        assert comm.world_size == world
        if loss == "factor":
            half = world * Bl // 2
            hl = Bl // 2
            sl = slice(rank * hl, (rank + 1) * hl)
            eps1, eps2 = torch.randn(half, D, generator=gen), torch.randn(half, D, generator=gen)
            perms = torch.stack([torch.randperm(half, generator=gen) for _ in range(D)])
            # global tensor = [data1 of all ranks ; data2 of all ranks]: rank r owns rows r*h..(r+1)*h of each half
            g_in = (data_g.cuda(), (eps1.cuda(), eps2.cuda(), perms))
            l_in = (torch.cat((data_g[:half][sl], data_g[half:][sl])).cuda(), (eps1[sl].cuda(), eps2[sl].cuda(), perms))
        else:
            sl = slice(rank * Bl, (rank + 1) * Bl)
            eps = torch.randn(world * Bl, D, generator=gen)
            g_in = (data_g.cuda(), eps.cuda())
            l_in = (data_g[sl].cuda(), eps[sl].cuda())
        for it in range(3 if replay else 1):
            st = defaultdict(list)
            if loss == "factor":
                out0 = l0.call_optimize(g_in[0], m0, o0, defaultdict(list), noise=g_in[1])
                out1 = l1.call_optimize(l_in[0], m1, o1, st, noise=l_in[1])
                dref, d1 = l0.discriminator.arena, l1.discriminator.arena
                derr = ((d1.grad - dref.grad).abs().max() / dref.grad.abs().max()).item()
                assert derr < 2e-5, "iteration %d: disc grad err %.3e" % (it, derr)
                assert (d1.flat - dref.flat).abs().max().item() <= 2.5 * HP["lr_disc"] * (it + 1)
            else:
                out0 = l0.fused_step(g_in[0], m0, o0, defaultdict(list), eps=g_in[1])
                out1 = l1.fused_step(l_in[0], m1, o1, st, eps=l_in[1])
            ref_loss = out0.item()
            err = ((m1.arena.grad - m0.arena.grad).abs().max() / m0.arena.grad.abs().max()).item()
            assert err < 2e-5 * (it + 1), "iteration %d: grad err %.3e" % (it, err)
            assert abs(out1.item() - ref_loss) <= 2e-6 * (it + 1) * abs(ref_loss), (it, out1.item(), ref_loss)
            assert (m1.arena.flat - m0.arena.flat).abs().max().item() <= 2.5 * lr * (it + 1)
            if it == 0:
                assert st["loss"] and abs(st["loss"][0] - ref_loss) <= 2e-6 * abs(ref_loss)
        if replay:
            # the first iteration allocates (workspace, gather buffers), which invalidates the plan it recorded: iteration 2
            # records again, iteration 3 (at least) is a replay
            assert l1._graphs.replays >= 1, l1._graphs.replays
        comm.close()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        try:
            torch.distributed.destroy_process_group()
        except Exception:
            pass


@pytest.mark.parametrize("loss", ["btcvae", "VAE", "factor"])
def test_sharded_step_matches_global_batch(loss):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, loss, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


@pytest.mark.parametrize("replay", [None, "plan"])
@pytest.mark.parametrize("transport", ["torch", "rccl"])
@pytest.mark.parametrize("loss", ["btcvae", "factor"])
def test_sharded_step_over_rccl_on_all_visible_gpus(loss, transport, replay):
    """The real thing: one rank per GPU over RCCL (backend nccl), both transports (torch.distributed collectives and
    the C-ABI's dvae_comm_*), sharded step == single-process step on the global batch -- issued eagerly (one iteration) and
    from the recorded launch plan, which is what "auto" selects at these sizes (three iterations: record, record again after the
    first iteration's allocations, replay; every one compared with the eager single-process step).  Needs >= 2 GPUs: skipped on
    the 1-GPU test boxes, runs on the multi-GPU node of the scaling tier."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs >= 2 visible GPUs (%d here)" % n)
    world = n                                   # every visible GPU: an 8-GPU node runs 8 ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, loss, q, "nccl", transport, r, replay)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


@pytest.mark.parametrize("loss", ["btcvae", "factor"])
def test_sharded_step_replays_from_a_recorded_plan(loss):
    """Two ranks (gloo, one GPU): the launch plan of the SHARDED step -- kernels, stream forks / joins and the collectives --
    is recorded in iteration 1 and replayed in iterations 2 and 3; every iteration equals the single-process eager step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, loss, q, "gloo", "torch", 0, "plan")) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


@pytest.mark.parametrize("replay", [None, "plan"])
@pytest.mark.parametrize("transport", ["torch", "rccl"])
@pytest.mark.parametrize("loss", ["btcvae", "factor"])
def test_rccl_call_sites_single_rank(loss, transport, replay):
    """backend "nccl" (= RCCL) with ONE rank on the one GPU of the test box: the same collectives'
    call sites as on 8 GPUs (broadcast, list all_gather, sum all_reduce, async bucket all_reduce under
    the side-stream context) run through RCCL itself; results must equal the communicator-free step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_worker, args=(0, 1, _free_port(), loss, q, "nccl", transport, 0, replay))
    p_.start()
    rank, msg = q.get(timeout=280)
    p_.join(timeout=60)
    assert msg == "ok", msg


def _worker_local(rank, world, port, loss, q):
    """estimator="local": the sharded step must equal the rank-AVERAGE of independent single-process steps on
    the shards (loss = mean of the shard losses, gradient = mean of the shard gradients)."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "disentangling-vae_amd"), os.path.join(root, "tests")):
            sys.path.insert(0, p)
        from disvae_amd import parallel
        torch.cuda.set_device(0)
        parallel.init_process_group_from_env("gloo")
        lr = 1e-4 if loss == "factor" else 5e-4
        Bl, D = 12, 10
        gen = torch.Generator().manual_seed(5)
        shards, noises = [], []
        for r in range(world):
            shards.append(torch.rand((Bl,) + IMG, generator=gen))
            if loss == "factor":
                noises.append((torch.randn(Bl // 2, D, generator=gen), torch.randn(Bl // 2, D, generator=gen),
                               torch.stack([torch.randperm(Bl // 2, generator=gen) for _ in range(D)])))
            else:
                noises.append(torch.randn(Bl, D, generator=gen))

        def step(m, o, l, r):
            if loss == "factor":
                e1, e2, pm = noises[r]
                return l.call_optimize(shards[r].cuda(), m, o, defaultdict(list), noise=(e1.cuda(), e2.cuda(), pm))
            return l.fused_step(shards[r].cuda(), m, o, defaultdict(list), eps=noises[r].cuda())

        ref_loss, ref_grad, ref_dgrad = 0.0, 0.0, 0.0
        for r in range(world):
            m0, o0, l0 = _make(loss, lr)
            ref_loss += step(m0, o0, l0, r).item() / world
            ref_grad = ref_grad + m0.arena.grad.clone() / world
            if loss == "factor":
                ref_dgrad = ref_dgrad + l0.discriminator.arena.grad.clone() / world
        m1, o1, l1 = _make(loss, lr)
        from ddp_util import HostStagedComm
        parallel.data_parallel(m1, l1, estimator="local", comm=HostStagedComm())
        out1 = step(m1, o1, l1, rank)
        err = ((m1.arena.grad - ref_grad).abs().max() / ref_grad.abs().max()).item()
        assert err < 2e-5, "grad err %.3e" % err
        assert abs(out1.item() - ref_loss) <= 2e-6 * abs(ref_loss), (out1.item(), ref_loss)
        if loss == "factor":
            dg = l1.discriminator.arena.grad
            derr = ((dg - ref_dgrad).abs().max() / ref_dgrad.abs().max()).item()
            assert derr < 2e-5, "disc grad err %.3e" % derr
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        try:
            torch.distributed.destroy_process_group()
        except Exception:
            pass


@pytest.mark.parametrize("loss", ["btcvae", "factor"])
def test_local_estimator_is_rank_average(loss):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_local, args=(r, world, port, loss, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _worker_mirrored(port, loss, q, transport, replay, world_emulated, Bl, D=10):
    """ONE rank (RCCL, backend nccl) standing in for rank 0 of `world_emulated` ranks with identical shards
    (parallel.MirroredWorldComm): the whole sharded code path at world > 1 -- packed gather / scatter, loss-sum all-reduce, the
    gradient spans all-reduced asynchronously under the backward pass, through the product transports -- must equal the
    single-process eager step on the shard tiled world_emulated times.  Exact for terms that are symmetric in the ranks: the
    beta-TCVAE estimator with uniform weights (is_mss=False; the stratified weights' one exception cell, math.py:72, lives in
    the last rank's row block only) and the VAE / betaH losses."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "disentangling-vae_amd"), os.path.join(root, "tests")):
            sys.path.insert(0, p)
        from disvae_amd import parallel
        torch.cuda.set_device(0)
        parallel.init_process_group_from_env("nccl")
        W, lr = world_emulated, 5e-4
        gen = torch.Generator().manual_seed(11)
        shard = torch.rand((Bl,) + IMG, generator=gen)
        eps = torch.randn(Bl, D, generator=gen)
        m0, o0, l0 = _make(loss, lr, D)          # single process, eager, on the tiled global batch
        l0.replay = None
        m1, o1, l1 = _make(loss, lr, D)          # the mirrored shard
        l1.replay = replay
        if loss == "btcvae":
            l0.is_mss = l1.is_mss = False
        inner = parallel.RcclComm() if transport == "rccl" else parallel.Comm()
        comm = parallel.data_parallel(m1, l1, comm=parallel.MirroredWorldComm(inner, W, 0))
        assert comm.world_size == W and l1._world() == (W, 0)
        g_data, g_eps = shard.repeat(W, 1, 1, 1).cuda(), eps.repeat(W, 1).cuda()
        l_data, l_eps = shard.cuda(), eps.cuda()
        if loss == "factor":
            # FactorVAE's global permutation is not symmetric in the ranks: no tiled single-process twin.  The sharded
            # two-optimizer step (z2 all-gather, discriminator arena all-reduced under the VAE backward, late epilogue) must
            # run, replay and stay finite and deterministic: two mirrored models fed the same noise agree bit for bit
            m2, o2, l2 = _make(loss, 1e-4, D)
            l2.replay = None
            m1, o1, l1 = _make(loss, 1e-4, D)
            l1.replay = replay
            comm2 = parallel.data_parallel(m2, l2, comm=parallel.MirroredWorldComm(inner, W, 0))
            comm = parallel.data_parallel(m1, l1, comm=parallel.MirroredWorldComm(inner, W, 0))
            h = Bl // 2
            noise = (torch.randn(h, D, generator=gen).cuda(), torch.randn(h, D, generator=gen).cuda(),
                     torch.stack([torch.randperm(h * W, generator=gen) for _ in range(D)]))
            for it in range(4 if replay else 1):
                out2 = l2.call_optimize(l_data, m2, o2, defaultdict(list), noise=noise)
                out1 = l1.call_optimize(l_data, m1, o1, defaultdict(list), noise=noise)
                assert torch.isfinite(out1).all() and torch.isfinite(m1.arena.grad).all()
                assert out1.item() == out2.item(), (it, out1.item(), out2.item())
                assert torch.equal(m1.arena.grad, m2.arena.grad) and torch.equal(l1.discriminator.arena.grad, l2.discriminator.arena.grad)
            if replay:
                assert l1._graphs.replays >= 2, l1._graphs.replays
            comm.close()
            q.put((0, "ok"))
            return
        for it in range(4 if replay else 1):
            st = defaultdict(list)
            # every iteration starts from the twin's weights: the comparison stays a one-step one (two Adam trajectories drift
            # apart at 1e-4 of max|g| within three steps, which would hide a replay that is off by that much)
            m1.arena.flat.copy_(m0.arena.flat)
            out0 = l0.fused_step(g_data, m0, o0, defaultdict(list), eps=g_eps)
            out1 = l1.fused_step(l_data, m1, o1, st, eps=l_eps)
            ref_loss = out0.item()
            err = ((m1.arena.grad - m0.arena.grad).abs().max() / m0.arena.grad.abs().max()).item()
            assert err < 2e-5, "iteration %d: grad err %.3e" % (it, err)
            assert abs(out1.item() - ref_loss) <= 2e-6 * abs(ref_loss), (it, out1.item(), ref_loss)
            if it == 0:
                assert (m1.arena.flat - m0.arena.flat).abs().max().item() <= 2.5 * lr
                assert st["loss"] and abs(st["loss"][0] - ref_loss) <= 2e-6 * abs(ref_loss)
        if replay:
            assert l1._graphs.replays >= 2, l1._graphs.replays
        comm.close()
        q.put((0, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((0, traceback.format_exc()))
    finally:
        try:
            torch.distributed.destroy_process_group()
        except Exception:
            pass


@pytest.mark.parametrize("replay", [None, "plan"])
@pytest.mark.parametrize("transport", ["torch", "rccl"])
@pytest.mark.parametrize("loss,world_emulated,Bl", [("btcvae", 8, 16), ("VAE", 4, 12), ("factor", 4, 16)])
def test_mirrored_world_runs_the_sharded_path_on_one_gpu(loss, world_emulated, Bl, transport, replay):
    """The data-parallel step at world_size > 1 through RCCL on the ONE GPU of the test box (see _worker_mirrored); with
    replay="plan" iterations 3 and 4 are replays of the recorded sharded launch plan, NCCL collectives included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_worker_mirrored, args=(_free_port(), loss, q, transport, replay, world_emulated, Bl))
    p_.start()
    rank, msg = q.get(timeout=280)
    p_.join(timeout=60)
    assert msg == "ok", msg


@pytest.mark.parametrize("loss,world_emulated,Bl", [("btcvae", 4, 12), ("factor", 2, 8)])
def test_mirrored_world_above_16_latents(loss, world_emulated, Bl):
    """The sharded step with latent_dim 20 (run-time-D estimator over the global batch, wide packed sums -- 32 + D floats -- behind
    the column gradients in the ONE all-reduce, FC layers one launch each): C-ABI RCCL transport, replayed from the recorded plan,
    against the single-process step on the tiled batch (btcvae) / a second mirrored model bit for bit (factor)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_worker_mirrored, args=(_free_port(), loss, q, "rccl", "plan", world_emulated, Bl, 20))
    p_.start()
    rank, msg = q.get(timeout=280)
    p_.join(timeout=60)
    assert msg == "ok", msg
