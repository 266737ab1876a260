"""Data-parallel gradient exchange (stcat_amd/dist.py) over gloo, world_size 2, on CPU.
No kernels involved: this covers the bucket layout, the readiness hooks, asynchronous launch,
averaging, the statically excluded dead parameters and the dummy (RoBERTa-sized) tail bucket."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from stcat_amd.dist import GradBucketReducer, live_trainable


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(8, 16, 3, bias=False)
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.lin = nn.Linear(16, 4)
        self.ground_encoder = nn.Module()
        self.ground_encoder.fusion = nn.Linear(4, 4)  # dead in the reference: must stay out of the buckets
        self.frozen = nn.Linear(4, 4)
        for p in self.frozen.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        return self.lin(self.conv(x).mean((2, 3)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = Toy()
    red = GradBucketReducer(model, bucket_mb=0.002, extra_numel=100)  # ~2 KB buckets -> several buckets
    assert len(red.buckets) >= 2
    names = [n for b in red.buckets for n, _ in b["params"]]
    assert not any("fusion" in n or "frozen" in n for n in names)
    res = []
    for step in range(2):  # second step checks re-arming of the hooks / zeroing
        red.zero_grad()
        x = torch.full((2, 8, 5, 5), float(rank + 1 + step))
        model(x).square().sum().backward()
        red.finish()
        res.append({n: p.grad.clone().numpy() for n, p in live_trainable(model.named_parameters())})
        assert model.conv.weight.grad.data_ptr() != 0
        assert model.conv.weight.grad.stride() == model.conv.weight.stride()  # channels_last view kept
    q.put((rank, res, red.message_bytes))
    dist.barrier()
    dist.destroy_process_group()


def _single(step):
    torch.manual_seed(0)
    model = Toy()
    grads = []
    for rank in range(2):
        model.zero_grad()
        x = torch.full((2, 8, 5, 5), float(rank + 1 + step))
        model(x).square().sum().backward()
        grads.append({n: p.grad.clone() for n, p in live_trainable(model.named_parameters())})
    return {n: (grads[0][n] + grads[1][n]) / 2 for n in grads[0]}


def test_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda t: t[0])
    for step in range(2):
        want = _single(step)
        for rank, res, nbytes in out:
            for n, g in res[step].items():
                assert torch.allclose(torch.from_numpy(g), want[n], rtol=1e-5, atol=1e-6), (rank, step, n)
    assert out[0][2] == out[1][2] and out[0][2] > 400


def test_reducer_single_process_is_identity():
    model = Toy()
    red = GradBucketReducer(model)
    red.zero_grad()
    model(torch.ones(1, 8, 5, 5)).sum().backward()
    red.finish()
    ref = Toy()
    ref.load_state_dict(model.state_dict())
    ref(torch.ones(1, 8, 5, 5)).sum().backward()
    for (n, p), (_, q) in zip(live_trainable(model.named_parameters()), live_trainable(ref.named_parameters())):
        assert torch.allclose(p.grad, q.grad), n


# ---- early delivery: a backward node hands parameter gradients to the reducer itself (ops.GRAD_SINK) -----------------
class _EarlyConv(torch.autograd.Function):
    """conv whose backward delivers dW through the gradient sink and returns None for it — what the backbone node does
    block by block (stcat_amd/backbone.py)"""

    @staticmethod
    def forward(ctx, x, w, owner):
        ctx.save_for_backward(x)
        ctx.owner = owner
        return torch.nn.functional.conv2d(x, w)

    @staticmethod
    def backward(ctx, g):
        from stcat_amd import ops
        (x,) = ctx.saved_tensors
        w = ctx.owner.conv.weight
        dw = torch.nn.grad.conv2d_weight(x, w.shape, g)
        took = ops.GRAD_SINK is not None and ops.GRAD_SINK.early([w], [dw])
        ctx.owner.took.append(bool(took))
        return None, (None if took else dw), None


class ToyEarly(Toy):
    def __init__(self):
        super().__init__()
        self.took = []

    def forward(self, x):
        return self.lin(_EarlyConv.apply(x, self.conv.weight, self).mean((2, 3)))


def _worker_early(rank, world, port, q):
    from stcat_amd import ops
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = ToyEarly()
    red = GradBucketReducer(model, bucket_mb=0.002)
    assert ops.GRAD_SINK is red
    res = []
    for step in range(2):
        red.zero_grad()
        x = torch.full((2, 8, 5, 5), float(rank + 1 + step))
        model(x).square().sum().backward()
        red.finish()
        res.append({n: p.grad.clone().numpy() for n, p in live_trainable(model.named_parameters())})
    assert model.took == [True, True]
    red.close()
    assert ops.GRAD_SINK is None
    q.put((rank, res, red.message_bytes))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_early_delivery_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_early, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for step in range(2):
        want = _single(step)
        for rank, res, nbytes in out:
            for n, g in res[step].items():
                assert torch.allclose(torch.from_numpy(g), want[n], rtol=1e-5, atol=1e-6), (rank, step, n)


def test_bucket_layout_has_a_small_tail():
    """readiness-ordered buckets of at most bucket_mb with a last bucket of at most tail_mb (what completes last is the
    exposed part of the exchange)"""
    m = nn.Sequential(*[nn.Linear(64, 64, bias=False) for _ in range(10)])   # 10 x 16 KB, registered first = ready last
    red = GradBucketReducer(m, bucket_mb=0.0625, tail_mb=0.02)                # 64 KB buckets, 20 KB tail
    sizes = [b["numel"] * 4 for b in red.buckets]
    assert sum(sizes) == 10 * 64 * 64 * 4
    assert sizes[-1] <= 20 * 1024 and max(sizes) <= 64 * 1024
    assert [n for n, _ in red.buckets[-1]["params"]] == ["0.weight"]          # the first layer's gradient is ready last


def _rccl_single_rank(collective, q):
    """subprocess body: the reducer over RCCL at world size 1 (what a 1-GPU box can run of the N > 1 path)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), STCAT_DP_COLLECTIVE=collective)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from stcat_amd.dist import init_rccl_process_group
    init_rccl_process_group(dev, rank=0, world_size=1)
    torch.manual_seed(0)
    model = Toy().to(dev)
    red = GradBucketReducer(model, bucket_mb=0.002, extra_numel=1000, force_comm=True)
    assert red.collective == collective and all(b["padded"] % 64 == 0 for b in red.buckets)
    out = []
    for step in range(2):
        red.zero_grad()
        x = torch.full((2, 8, 5, 5), float(1 + step), device=dev)
        model(x).square().sum().backward()
        red.finish()
        out.append({n: p.grad.detach().cpu().clone() for n, p in live_trainable(model.named_parameters())})
    q.put(out)
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpu_reducer_rs_ag_equals_allreduce_single_rank():
    """STCAT_DP_COLLECTIVE=rs_ag (reduce-scatter + all-gather per padded bucket, round 4) against the default all-reduce,
    both over RCCL with one rank: the mean over one rank is the identity, so both must return the plain gradients — what
    is exercised is the real RCCL call sequence (async reduce_scatter_tensor -> all_gather_into_tensor on one stream, AVG
    inside the collective, padded flat buckets, the dummy message launched behind its trigger bucket)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    res = {}
    for coll in ("allreduce", "rs_ag"):
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_single_rank, args=(coll, q))
        p.start()
        res[coll] = q.get(timeout=300)
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    model = Toy()
    for step in range(2):
        for prm in model.parameters():
            prm.grad = None
        model(torch.full((2, 8, 5, 5), float(1 + step))).square().sum().backward()
        for n, prm in live_trainable(model.named_parameters()):
            for coll in ("allreduce", "rs_ag"):
                got = res[coll][step][n]
                assert torch.allclose(got, prm.grad, rtol=1e-4, atol=1e-5), (coll, step, n)
