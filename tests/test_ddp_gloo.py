"""N>1 path on CPU: two processes, gloo backend (the GPU run uses the same code with RCCL).
Covers the outer step's data-parallel wiring (main.py:351-359, 889-890, 961-997): the batch is
sharded across ranks, the attack runs locally with NO collective, the only exchange is DDP's
gradient all-reduce, and the result equals single-process training on the whole batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import revisiting_at_amd as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _CutNet(nn.Module):
    """A model that names a cut point (``ddp_cut``, as architecture.ConvNeXt does): FlatGradSync then splits the backward there and
    sends the late parameters' gradients out before the early part of the backward runs."""

    def __init__(self):
        super().__init__()
        self.a = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.GELU())
        self.b = nn.Sequential(nn.Conv2d(8, 8, 3, padding=1), nn.GELU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 5))

    def forward(self, x):
        return self.b(self.a(x))

    def ddp_cut(self):
        return self.a, [self.b]


def _model(kind="seq"):
    torch.manual_seed(0)
    if kind == "cut":
        return _CutNet()
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.GELU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 5))


def _sign_attack(model, x, y, eps=0.03):
    """Stand-in for apgd_train on CPU (the HIP attack has no CPU path): one signed-gradient step.
    Counts collectives issued while it runs (must stay zero: SURVEY.md §2a 'inside the attack')."""
    assert not model.training
    x = x.clone().requires_grad_()
    loss = nn.functional.cross_entropy(model(x), y, reduction='sum')
    (g,) = torch.autograd.grad(loss, [x])
    return (x + eps * g.sign()).clamp(0, 1).detach(), None, None, None


def _data(n):
    g = torch.Generator().manual_seed(42)
    return torch.rand(n, 3, 8, 8, generator=g), torch.randint(0, 5, (n,), generator=g)


def _worker(rank, world, port, q, grad_sync="ddp", kind="seq"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, l, w = R.setup_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # Count every exchange and where it happens: DDP's bucket reductions go through a comm hook (the C++ reducer does not
    # call the Python-level collectives), the Python-level collectives are wrapped.  `in_attack` is raised by the perturb
    # callable, i.e. exactly for the span WrappedModel spends inside the attack (main.py:283).
    st = {"in_attack": False, "hook_in_attack": 0, "hook_total": 0, "py_in_attack": 0}

    def attack(model, x, y):
        st["in_attack"] = True
        try:
            return _sign_attack(model, x, y)
        finally:
            st["in_attack"] = False

    for name in ("all_reduce", "broadcast", "all_gather", "reduce_scatter", "barrier", "all_to_all", "reduce"):
        real = getattr(dist, name)

        def counted(*a, _real=real, **kw):
            if st["in_attack"]:
                st["py_in_attack"] += 1
            return _real(*a, **kw)
        setattr(dist, name, counted)

    tr = R.ATTrainStep(_model(kind), "toy", R.AdvConfig(), "cpu", lr=1e-2, distributed=True, channels_last=False,
                       amp_dtype=None, ema=True, perturb=attack, grad_sync=grad_sync)

    def hook(_, bucket):
        st["hook_total"] += 1
        st["hook_in_attack"] += int(st["in_attack"])
        fut = dist.all_reduce(bucket.buffer().div_(world), async_op=True).get_future()
        return fut.then(lambda f: f.value()[0])
    if grad_sync == "ddp":
        tr.model.register_comm_hook(None, hook)
    else:
        assert tr.sync is not None and tr.sync.world == world and not isinstance(tr.model, nn.parallel.DistributedDataParallel)
        assert (tr.sync.cut is not None) == (kind == "cut") and bool(tr.sync.late) == (kind == "cut")
    x, y = _data(8)
    xs, ys = x[rank::world], y[rank::world]                 # DistributedSampler-style disjoint shards (main.py:567)
    losses = [float(tr.step(xs, ys)) for _ in range(3)]
    if grad_sync == "flat":                                 # the flat path's exchanges are Python-level all-reduces (counted above)
        st["hook_total"] = tr.sync.reduces
        assert all(p.grad.data_ptr() == v.data_ptr() for g, vs in zip((tr.sync.late, tr.sync.early), tr.sync.views)
                   for p, v in zip(g, vs)), "a parameter's .grad left the flat buffers"
    flat = torch.cat([p.detach().flatten() for p in tr.inner.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ema0 = tr.ema.ema[0].clone()
    if rank == 0:
        q.put(dict(params=flat, same=bool(all(torch.equal(gathered[0], t) for t in gathered)), losses=losses,
                   ema_moved=bool(not torch.equal(ema0, tr.inner.state_dict()[list(tr.inner.state_dict())[0]])),
                   training=tr.model.training, keys=list(tr.model.state_dict())[:2], ckpt_keys=list(tr.state_dict())[:2],
                   coll=dict(st)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("grad_sync,kind", [("ddp", "seq"), ("flat", "seq"), ("flat", "cut")])
def test_ddp_world2_matches_single_process(grad_sync, kind):
    """``grad_sync="ddp"``: DistributedDataParallel as ``main.py:889-890``; ``"flat"`` (the default of ATTrainStep(distributed=True)):
    ``FlatGradSync`` - flat gradient buffers and one / two asynchronous all-reduces per step between the segments of the training
    pass (with a model that names a cut point: the late group goes out before the early backward runs)."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, grad_sync, kind)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["same"], "ranks diverged: gradients were not all-reduced"
    assert res["training"]
    assert res["keys"][0].startswith("module.base_model." if grad_sync == "ddp" else "base_model.")   # DDP(WrappedModel(model)) / WrappedModel(model)
    # what a checkpoint writer gets (ATTrainStep.state_dict): the reference's on-disk keys, module.base_model.*, on both paths
    assert all(k.startswith("module.base_model.") for k in res["ckpt_keys"]), res["ckpt_keys"]
    assert res["ema_moved"]
    # no inter-rank traffic inside the attack; the only exchange is the gradient all-reduce of the
    # outer backward (>= 1 bucket per step, 3 steps)
    assert res["coll"]["hook_in_attack"] == 0 and res["coll"]["py_in_attack"] == 0, res["coll"]
    assert res["coll"]["hook_total"] >= (6 if kind == "cut" else 3), res["coll"]
    # single process, whole batch: DDP averages per-rank mean losses == mean over the full batch (equal shards)
    torch.set_num_threads(1)
    tr = R.ATTrainStep(_model(kind), "toy", R.AdvConfig(), "cpu", lr=1e-2, distributed=False, channels_last=False,
                       amp_dtype=None, ema=False, perturb=_sign_attack)
    x, y = _data(8)
    for _ in range(3):
        tr.step(x, y)
    flat = torch.cat([p.detach().flatten() for p in tr.inner.parameters()])
    torch.testing.assert_close(res["params"], flat, rtol=1e-4, atol=1e-5)


class _RecordedPass:
    """CPU stand-in for train_step._TrainPassGraph (hipGraph capture needs the device): "captures" by recording nothing and "replays" by
    running the recorded pass - what matters here is ATTrainStep's control flow around it under a process group: warm-up count, capture,
    the fallback of a rank whose capture fails, and that both kinds of rank issue the same collectives."""
    fail = False

    def __init__(self, step, x, target, x_is_static=False):
        if _RecordedPass.fail:
            raise RuntimeError("capture refused (test)")
        self.step, self.n_graphs, self.steps = step, 3, []
        self.replays = 0

    def __call__(self, x, target):
        self.replays += 1
        return self.step._train_pass(x, target)


def _mixed_capture_worker(rank, world, port, q, fail_rank):
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    R.setup_distributed()
    _RecordedPass.fail = rank == fail_rank
    tr = R.ATTrainStep(_model("cut"), "toy", R.AdvConfig(), "cpu", lr=1e-2, distributed=True, channels_last=False, amp_dtype=None,
                       ema=True, perturb=_sign_attack, grad_sync="flat")
    tr._graph_cls = _RecordedPass
    tr.graph_train = True                                   # (the constructor only switches it on for a device model)
    x, y = _data(8)
    xs, ys = x[rank::world], y[rank::world]
    n_steps = R.train_step.TRAIN_GRAPH_WARMUP + 3
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        for _ in range(n_steps):
            tr.step(xs, ys)
    progs = list(tr._tg.values())
    mine = dict(rank=rank, reduces=tr.sync.reduces, captured=sum(v is not None for v in progs), failed=sum(v is None for v in progs),
                replays=sum(v.replays for v in progs if v is not None),
                warned=any("capture failed" in str(w.message) for w in wl))
    alls = [None] * world
    dist.all_gather_object(alls, mine)
    flat = torch.cat([p.detach().flatten() for p in tr.inner.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        q.put(dict(params=flat, same=bool(all(torch.equal(gathered[0], t) for t in gathered)), ranks=alls, n_steps=n_steps))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [1, 0, None])
def test_a_rank_whose_training_pass_capture_failed_stays_in_step_with_the_ranks_that_replay(fail_rank):
    """under a process group the ranks need not agree on replay-versus-eager for the training pass - a rank
    whose capture failed runs ``_train_pass`` from Python, the others replay their three segments, and BOTH issue the flat path's two
    all-reduces per step in the same order (the collectives live between the segments, never inside one).  Two gloo ranks, one of them
    (or none) refused its capture: no hang, identical parameters on both, the same trajectory as single-process training."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_capture_worker, args=(r, 2, port, q, fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["same"], "ranks diverged"
    for st in res["ranks"]:
        failed = st["rank"] == fail_rank
        assert st["reduces"] == 2 * res["n_steps"], st            # two exchanges per step on every rank, replaying or not
        assert (st["captured"], st["failed"], st["warned"]) == ((0, 1, True) if failed else (1, 0, False)), st
        assert st["replays"] == (0 if failed else 3), st
    torch.set_num_threads(1)
    tr = R.ATTrainStep(_model("cut"), "toy", R.AdvConfig(), "cpu", lr=1e-2, distributed=False, channels_last=False,
                       amp_dtype=None, ema=False, perturb=_sign_attack)
    x, y = _data(8)
    for _ in range(res["n_steps"]):
        tr.step(x, y)
    flat = torch.cat([p.detach().flatten() for p in tr.inner.parameters()])
    torch.testing.assert_close(res["params"], flat, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind", ["seq", "cut"])
def test_flat_gradient_path_on_one_rank_equals_the_plain_backward(kind):
    """``grad_sync="flat"`` without a process group (``bench.py --ddp-path 1``): the two-call backward into the flat buffers is the
    same training step as ``loss.backward()`` - identical parameters after three steps, bit for bit on the CPU."""
    torch.set_num_threads(1)
    x, y = _data(8)
    out = []
    for gs in (None, "flat"):
        tr = R.ATTrainStep(_model(kind), "toy", R.AdvConfig(), "cpu", lr=1e-2, distributed=False, channels_last=False,
                           amp_dtype=None, ema=True, perturb=_sign_attack, grad_sync=gs)
        losses = [float(tr.step(x, y)) for _ in range(3)]
        out.append((losses, torch.cat([p.detach().flatten() for p in tr.inner.parameters()]), tr.ema.ema[0].clone()))
        assert (tr.sync is not None) == (gs == "flat")
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


class _BadCutNet(_CutNet):
    """``ddp_cut`` forgets a module that lies behind the cut (a later edit adding e.g. an ``fc_norm``), and shares one parameter
    across the cut: the split backward would silently zero the first and halve the second."""

    def __init__(self):
        super().__init__()
        self.extra = nn.Linear(5, 5)
        self.shared = nn.Parameter(torch.ones(1))

    def forward(self, x):
        return self.extra(self.b(self.a(x) * self.shared)) * self.shared

    def ddp_cut(self):
        return self.a, [self.b]                              # `extra` is downstream of the cut but not listed


def test_flat_gradient_path_checks_the_models_cut_point():
    """Round-4 advice: FlatGradSync trusted ``ddp_cut()``.  The first eager pass now compares the two-call backward with the
    whole one; a partition that would drop (or split) a parameter's gradient loses its cut point with a warning and the step
    equals ``loss.backward()`` again."""
    torch.set_num_threads(1)
    x, y = _data(8)
    out = []
    for gs in (None, "flat"):
        torch.manual_seed(0)
        tr = R.ATTrainStep(_BadCutNet(), "toy", R.AdvConfig(), "cpu", lr=1e-2, distributed=False, channels_last=False,
                           amp_dtype=None, ema=False, perturb=_sign_attack, grad_sync=gs)
        if gs == "flat":
            assert tr.sync.cut is not None and not tr.sync.verified
            with pytest.warns(UserWarning, match="does not partition"):
                losses = [float(tr.step(x, y))]
            assert tr.sync.cut is None and tr.sync.verified
        else:
            losses = [float(tr.step(x, y))]
        losses += [float(tr.step(x, y)) for _ in range(2)]
        out.append((losses, torch.cat([p.detach().flatten() for p in tr.inner.parameters()])))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
    # a sound partition passes the check and keeps its cut
    tr = R.ATTrainStep(_model("cut"), "toy", R.AdvConfig(), "cpu", lr=1e-2, distributed=False, channels_last=False,
                       amp_dtype=None, ema=False, perturb=_sign_attack, grad_sync="flat")
    tr.step(x, y)
    assert tr.sync.verified and tr.sync.cut is not None


def test_flat_gradient_views_keep_the_parameters_layout():
    """The flat buffers hand every parameter a ``.grad`` view with the parameter's OWN strides (channels_last convolution weights
    included): the fused AdamW refuses params / grads of different layouts, which is how the first GPU run of the flat path failed."""
    from revisiting_at_amd.train_step import FlatGradSync
    m = _CutNet().to(memory_format=torch.channels_last)
    assert any(not p.is_contiguous() for p in m.parameters())
    fs = FlatGradSync(m, "cpu")
    n = 0
    for group, views, buf in zip((fs.late, fs.early), fs.views, fs.flat):
        for p, v in zip(group, views):
            assert p.grad is v and v.shape == p.shape and v.stride() == p.stride()
            assert buf.data_ptr() <= v.data_ptr() < buf.data_ptr() + 4 * buf.numel()
            n += p.numel()
    assert n == sum(b.numel() for b in fs.flat) == sum(p.numel() for p in m.parameters())
    # the views tile the buffers without overlap: writing one parameter's gradient leaves the others alone
    for v in fs.views[0] + fs.views[1]:
        v.zero_()
    fs.views[0][0].fill_(1.0)
    assert float(sum(b.sum() for b in fs.flat)) == fs.views[0][0].numel()
    assert fs.bytes_per_step == 4 * n and len(fs.late) == len(list(m.b.parameters()))


def test_optimizer_groups_follow_reference_rules():
    m = R.get_new_model('convnext_tiny', pretrained=False, not_original=True)
    opt = R.create_optimizer(m, 'convnext_tiny', 0.05)
    nd, d = opt.param_groups
    names = {id(p): n for n, p in m.named_parameters()}
    nd_names = {names[id(p)] for p in nd['params']}
    d_names = {names[id(p)] for p in d['params']}
    assert nd['weight_decay'] == 0 and d['weight_decay'] == 0.05 and opt.defaults['betas'] == (0.9, 0.95)
    assert all(n.endswith('.bias') for n in nd_names)                       # main.py:403 ('bn', '.bias')
    assert 'stages.0.blocks.0.gamma' in d_names and 'stages.0.blocks.0.norm.weight' in d_names   # LN weights ARE decayed
    v = R.get_new_model('vit_s', pretrained=False, not_original=True)
    opt = R.create_optimizer(v, 'vit_s', 0.05)
    names = {id(p): n for n, p in v.named_parameters()}
    assert {'cls_token', 'pos_embed'} <= {names[id(p)] for p in opt.param_groups[1]['params']}   # ndim 3 -> decayed (main.py:440)
    assert 'blocks.0.norm1.weight' in {names[id(p)] for p in opt.param_groups[0]['params']}


def _robust_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import revisiting_at_amd as R
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stats = {"n": 10 + rank, "clean_correct": 8 + rank, "robust": 3 + 2 * rank}
    q.put((rank, R.aa_eval.robust_accuracy(stats)))
    dist.destroy_process_group()


def test_eval_counts_are_summed_over_ranks():
    """The evaluation's only exchange (aa_eval.robust_accuracy): one sum of three counts, world_size 2 over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29621
    ps = [ctx.Process(target=_robust_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):
        ca, ra = res[r]
        assert abs(ca - 17 / 21) < 1e-12 and abs(ra - 8 / 21) < 1e-12


def test_training_graph_is_keyed_on_the_attack_program_whose_weight_copies_it_reads():
    """Round 6: a training-pass graph that reads the derived weight copies of attack program P may only replay behind a replay of P.
    ATTrainStep._graph_step on CPU with stand-ins for the attack (which "replays" a chosen program or runs "eagerly") and for the
    graph class: the first program a batch shape is captured behind is the one it shares with; behind another program or an eager
    attack a second, self-contained graph runs (attack_prog=None); with ops.SHARE_DERIVED off nothing is ever shared."""
    from revisiting_at_amd import graphed, ops

    class Prog:                                             # stands for graphed._Program
        def __init__(self):
            self.derived = {}

    made = []

    class Pass(_RecordedPass):
        def __init__(self, step, x, target, x_is_static=False, attack_prog=None):
            super().__init__(step, x, target, x_is_static)
            self.attack_prog, self.x_is_static = attack_prog, x_is_static
            made.append(self)

    state = {"prog": None}

    def attack(model, x, y):                                # what graphed.run does around a replay / an eager call
        out = _sign_attack(model, x, y)
        if state["prog"] is not None:
            graphed.STATS["replays"] += 1
            graphed.LAST = state["prog"]
        return out

    def trainer():
        tr = R.ATTrainStep(_model("seq"), "toy", R.AdvConfig(), "cpu", lr=1e-2, channels_last=False, amp_dtype=None, ema=False, perturb=attack)
        tr._graph_cls, tr.graph_train = Pass, True
        return tr
    x, y = _data(8)
    _RecordedPass.fail = False
    P, Q = Prog(), Prog()
    tr = trainer()
    for _ in range(R.train_step.TRAIN_GRAPH_WARMUP):         # eager warm-up steps: nothing captured
        tr.step(x, y)
    assert not made
    seq = [P, P, None, Q, P, None, Q]
    for prog in seq:
        state["prog"] = prog
        tr.step(x, y)
    # one graph shared with P (captured behind its first replay, static input), one self-contained graph for everything else
    assert [(m.attack_prog is P, m.x_is_static) for m in made] == [(True, True), (False, False)], [(m.attack_prog, m.x_is_static) for m in made]
    assert made[1].attack_prog is None
    assert (made[0].replays, made[1].replays) == (3, 4)
    # captured behind an EAGER attack first: the self-contained graph; the first program that replays later gets the shared one
    made.clear()
    tr = trainer()
    for prog in [None] * R.train_step.TRAIN_GRAPH_WARMUP + [None, Q, Q, P]:
        state["prog"] = prog
        tr.step(x, y)
    assert [m.attack_prog for m in made] == [None, Q] and (made[0].replays, made[1].replays) == (2, 2)
    # switched off: one self-contained graph whatever attack ran
    made.clear()
    prev = ops.SHARE_DERIVED
    ops.SHARE_DERIVED = False
    try:
        tr = trainer()
        for prog in [None] * R.train_step.TRAIN_GRAPH_WARMUP + [P, Q, None]:
            state["prog"] = prog
            tr.step(x, y)
    finally:
        ops.SHARE_DERIVED = prev
    assert [m.attack_prog for m in made] == [None] and made[0].replays == 3
    graphed.LAST = None
