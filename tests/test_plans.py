"""Launch plans (stcat_amd/plans.py, csrc/launch_plan.h): a step replayed from the recorded C++ launch sequences gives
the SAME outputs, losses and parameter gradients as the eager Python path — on clips it was not recorded on, in eval
mode (deterministic) and in train mode (dropout masks regenerated from the step's counter range), and the guard rails
hold (no retained .grad aliasing a static buffer, a second forward before the backward falls back to eager)."""
import pytest
import torch

from stcat_amd import _lib, ops, plans, synth
from stcat_amd.misc import BoxList, NestedTensor
from stcat_amd.pipeline import SyntheticText, build_model
from tests.backends import use_emu


def _build(dev, L=5, train=False):
    from stcat_amd import backbone
    saved = backbone.BLOCKS
    if dev.type == "cpu":
        # the emulator runs every wave as a fiber: a ResNet of 1+1+2+1 bottlenecks (same node code, same stream forks,
        # a frozen stem / layer1 and trainable layer2-4) keeps the CPU suite in minutes; eager-vs-plan only, no oracle
        backbone.BLOCKS = (1, 1, 2, 1)
    try:
        model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
    finally:
        backbone.BLOCKS = saved
    model.train(train)
    synth.fill_module_(model)
    model.to(dev)
    return model, criterion, wd


def _clip(dev, T, res, k):
    g = torch.Generator().manual_seed(100 + k)
    frames = (synth.synth_frames(T, res) + 0.25 * torch.randn(T, 3, res, res, generator=g)).to(dev)
    mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
    if k % 2 == 1:                       # a padded right / bottom margin on every other clip
        mask[:, :, res - res // 4:] = True
        mask[:, res - res // 8:, :] = True
    return NestedTensor(frames, mask, [T])


def _step(model, criterion, wd, clip, T, res, dev):
    for p in model.parameters():
        p.grad = None
    out = model(clip, ["synthetic"])
    act, tb = synth.synth_targets(T)
    losses = criterion(out, [{"actioness": act.to(dev), "boxs": BoxList(tb, (res, res)).to(dev)}], [T])
    total = sum(losses[k] * wd[k] for k in losses)
    total.backward()
    outs = {k: out[k].detach().cpu().clone() for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights")}
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    return outs, total.item(), grads


def _run(dev, T, res, steps, use_plans, train=False, mma="f32"):
    _lib.set_mma_mode(mma)
    plans.clear()
    plans.enable(use_plans)
    plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0)
    try:
        ops.manual_seed(7)
        model, criterion, wd = _build(dev, train=train)
        res_ = []
        for k in range(steps):
            if train:
                ops.dropout_begin_step(dev)
            res_.append(_step(model, criterion, wd, _clip(dev, T, res, k), T, res, dev))
        return res_, dict(plans.STATS)
    finally:
        plans.enable(False)
        plans.clear()
        _lib.set_mma_mode("f32")


def _same(a, b, what, tol):
    assert a.shape == b.shape, what
    err = (a.double() - b.double()).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{what}: {err:.3e} vs scale {scale:.3g}"


def _rel_l2(a, b, what, tol):
    assert a.shape == b.shape, what
    # absolute floor of 2e-6 per element: key-side attention biases (and layer 0's query-content bias, whose input is the
    # zero state) have structurally zero gradients — what is left there is rounding noise
    err = (a.double() - b.double()).norm().item() / (b.double().norm().item() + 2e-6 * b.numel() ** 0.5 / tol)
    assert err <= tol, f"{what}: rel-L2 {err:.3e}"


def _check_equal(ref, got, tol, grad_l2=None):
    """grad_l2: compare gradients by per-tensor relative L2 error instead of max-abs (GPU: two EAGER runs already
    differ by single ReLU-kink flips and the order of the atomically summed split-K partials — tests/test_dp_model.py)"""
    for k, ((o1, t1, g1), (o2, t2, g2)) in enumerate(zip(ref, got)):
        for n in o1:
            _same(o2[n], o1[n], f"step {k} {n}", tol)
        assert abs(t1 - t2) <= tol * max(1.0, abs(t1)), (k, t1, t2)
        assert set(g1) == set(g2), (k, set(g1) ^ set(g2))
        for n in g1:
            if grad_l2:
                _rel_l2(g2[n], g1[n], f"step {k} grad {n}", grad_l2)
            else:
                _same(g2[n], g1[n], f"step {k} grad {n}", tol)


def _guard_rails(dev, model, criterion, wd, T, res):
    """on a model whose plans exist (>= 3 steps ran)"""
    # (1) a second forward while the first one's backward is outstanding runs eagerly: static buffers stay intact
    for p in model.parameters():
        p.grad = None
    before = plans.STATS["eager"]
    out1 = model(_clip(dev, T, res, 0), ["synthetic"])
    keep = out1["pred_boxes"].detach().clone()
    out2 = model(_clip(dev, T, res, 1), ["synthetic"])
    assert plans.STATS["eager"] > before
    assert torch.equal(out1["pred_boxes"].detach(), keep)
    assert not torch.equal(out2["pred_boxes"].detach(), keep)
    del out1, out2
    # (2) gradients kept across steps alias the plans' static buffers: refused, not silently doubled
    _step(model, criterion, wd, _clip(dev, T, res, 2), T, res, dev)
    out = model(_clip(dev, T, res, 3), ["synthetic"])
    act, tb = synth.synth_targets(T)
    losses = criterion(out, [{"actioness": act.to(dev), "boxs": BoxList(tb, (res, res)).to(dev)}], [T])
    with pytest.raises((_lib.StcatHipError, RuntimeError), match="zero_grad"):
        sum(losses[k] * wd[k] for k in losses).backward()


def test_emu_plans_replay_equals_eager_and_guard_rails():
    """3 steps on 3 different clips (one padded): step 0 eager, step 1 recorded, step 2 replayed — equal to the eager
    run; then the guard rails on the same model (one model build: the emulator is slow)"""
    dev = use_emu()
    T, res = 2, 32
    ref, _ = _run(dev, T, res, 3, False)
    plans.clear()
    plans.enable(True)
    plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0)
    try:
        ops.manual_seed(7)
        model, criterion, wd = _build(dev)
        got = [_step(model, criterion, wd, _clip(dev, T, res, k), T, res, dev) for k in range(3)]
        assert plans.STATS["recorded"] >= 8 and plans.STATS["replayed"] >= plans.STATS["recorded"], plans.STATS
        _check_equal(ref, got, 2e-5)
        _guard_rails(dev, model, criterion, wd, T, res)
    finally:
        plans.enable(False)
        plans.clear()


@pytest.mark.gpu
def test_gpu_plans_replay_equals_eager():
    """4 steps on 4 different clips (two of them padded): step 0 eager, step 1 recorded, steps 2-3 replayed"""
    from tests.backends import use_hip
    dev = use_hip()
    ref, _ = _run(dev, 8, 224, 4, False, mma="bf16x6p")
    got, stats = _run(dev, 8, 224, 4, True, mma="bf16x6p")
    assert stats["recorded"] >= 8 and stats["replayed"] >= 2 * stats["recorded"], stats
    # the weight gradients are sums of atomically ordered split-K partials: run-to-run differences of a few ulp
    _check_equal(ref, got, 2e-4, grad_l2=3e-3)


@pytest.mark.gpu
def test_gpu_plans_train_mode_dropout():
    """train mode: a replayed step draws the masks of ITS counter range — equal to the eager run with the same seed
    (GPU only: the emulator variant of this test passed during development and costs the CPU suite three minutes)"""
    from tests.backends import use_hip
    dev, big = use_hip(), True
    T, res = (8, 224) if big else (2, 32)
    ref, _ = _run(dev, T, res, 4, False, train=True)
    got, stats = _run(dev, T, res, 4, True, train=True)
    assert stats["replayed"] > 0, stats
    _check_equal(ref, got, 2e-4 if big else 2e-5, grad_l2=3e-3 if big else None)
    # and the masks differ from step to step (same clip shape, different losses even on the same clip is not tested
    # here; the counter base advanced: the host offset restarts while the device base moved on)
    assert ref[2][1] != ref[3][1]


def test_plan_table_covers_every_launch_entry_point():
    """every stream-ordered entry point of the C ABI can be recorded (the emulator build exports the same table)"""
    use_emu()
    lib = _lib.load()
    for name, sig in _lib.SIGNATURES.items():
        if sig.endswith("s") and "P" not in sig:
            fn = lib.stcat_plan_fn_index(name.encode())
            assert fn >= 0, name
            assert lib.stcat_plan_fn_nargs(fn) == len(sig), (name, lib.stcat_plan_fn_nargs(fn), len(sig))
    assert lib.stcat_plan_fn_index(b"stcat_set_mma_mode") == -1


class _NodeWithForeignKernel(torch.autograd.Function):
    """a composite-style node whose body runs an aten kernel on a strided input (x.t().contiguous()): replaying only OUR
    launches would silently drop that copy"""

    @staticmethod
    def forward(ctx, x, w):
        xt = x.t().contiguous()                    # foreign: an aten copy kernel
        y = ops.ew(_lib.EW_MUL, xt, w)
        ctx.save_for_backward(xt, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        xt, w = ctx.saved_tensors
        return ops.ew(_lib.EW_MUL, dy.contiguous(), w).t(), ops.ew(_lib.EW_MUL, dy.contiguous(), xt)


def test_emu_recording_refuses_a_node_that_runs_foreign_kernels():
    """the dispatch watch is on for EVERY recording; a node body that executes a kernel that is not
    ours is refused as a plan — that call's results are still right, later calls stay eager and right, nothing is replayed"""
    import warnings
    dev = use_emu()
    plans.clear()
    plans.enable(True)
    plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0, refused=0)
    try:
        w = torch.nn.Parameter(torch.arange(12, dtype=torch.float32).reshape(4, 3) + 1.0)
        outs = []
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for k in range(4):
                x0 = (torch.arange(12, dtype=torch.float32).reshape(3, 4) * (k + 1)).requires_grad_(True)
                x = x0 * 1.0                       # (a leaf that requires grad would count as a parameter: matched by identity)
                w.grad = None
                y = plans.apply(_NodeWithForeignKernel, x, w)
                y.sum().backward()
                assert torch.equal(y.detach(), x.detach().t() * w.detach()), k       # every call, replayed or not, is right
                assert torch.equal(w.grad, x.detach().t())
                outs.append(y)
        assert plans.STATS["refused"] >= 1 and plans.STATS["replayed"] == 0 and plans.STATS["recorded"] == 0, plans.STATS
        assert any("launch plan refused" in str(c.message) for c in caught)
    finally:
        plans.enable(False)
        plans.clear()


class _NodeWithHostCall(torch.autograd.Function):
    """a node whose backward hands its weight gradient to the host in mid-sequence (ops.host_call: what the backbone's
    backward does for the data-parallel reducer) — the host action runs aten kernels (a bucket copy, an all-reduce)"""
    LOG = []
    BUCKET = None

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.ew(_lib.EW_MUL, x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        gw = ops.ew(_lib.EW_MUL, dy, x)

        def deliver():
            _NodeWithHostCall.BUCKET.copy_(gw)            # an aten kernel, on purpose
            _NodeWithHostCall.LOG.append(float(_NodeWithHostCall.BUCKET.sum()))
            return True
        ops.host_call(deliver)
        return ops.ew(_lib.EW_MUL, dy, w), gw


def test_emu_host_call_inside_a_recording_is_not_a_foreign_kernel():
    """round 6: with a live process group the reducer's early hand-over (a plan YIELD) launches the all-reduce of a
    completed bucket — an aten / c10d op — from inside the backbone's backward.  The dispatch watch counted it as a
    foreign kernel of the node body and refused the plan: the backbone stayed eager in every N > 1 run.  A host call's
    kernels are the host's: the node is recorded, and the action runs again at that point of every replay."""
    dev = use_emu()
    plans.clear()
    plans.enable(True)
    plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0, refused=0)
    _NodeWithHostCall.LOG.clear()
    _NodeWithHostCall.BUCKET = torch.zeros(3, 4)
    try:
        w = torch.nn.Parameter(torch.arange(12, dtype=torch.float32).reshape(3, 4) + 1.0)
        for k in range(4):
            x = (torch.arange(12, dtype=torch.float32).reshape(3, 4) * (k + 1)).requires_grad_(True) * 1.0
            w.grad = None
            y = plans.apply(_NodeWithHostCall, x, w)
            y.sum().backward()
            assert torch.equal(w.grad, x.detach()), k
            assert torch.equal(_NodeWithHostCall.BUCKET, x.detach()), k          # delivered at every call, replays included
        assert len(_NodeWithHostCall.LOG) == 4
        assert not plans.STATS.get("refused") and plans.STATS["recorded"] == 2 and plans.STATS["replayed"] == 4, plans.STATS
    finally:
        plans.enable(False)
        plans.clear()


class _LinearNode(torch.autograd.Function):
    """y = x w^T + b with the composite layers' own backward (`composite._lin_b`): rows > 256 take the data gradient
    through the CACHED transposed weight (ops.LinearTransposes.get)"""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return ops.linear_fwd_raw(x, w, b)

    @staticmethod
    def backward(ctx, g):
        from stcat_amd import composite
        x, w = ctx.saved_tensors
        dx, dw, db, _ = composite._lin_b(g, x, w)
        return dx, dw, db


def _cached_transpose_case(dev):
    """a plan recorded while LinearTransposes.get() HIT its cache holds no transpose launch; with a
    loss that is not StgLossFn nobody refreshes W^T before the replay.  The weights move between the steps (an optimizer
    update): every step's dx must be g . W of THAT step's W."""
    _lib.set_mma_mode("f32" if dev.type == "cpu" else "bf16x6")
    plans.clear()
    plans.enable(True)
    plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0)
    try:
        gen = torch.Generator().manual_seed(3)
        M, K, N = 320, 64, 64
        w = torch.nn.Parameter(torch.randn(N, K, generator=gen).to(dev))
        b = torch.nn.Parameter(torch.zeros(N, device=dev))
        for k in range(5):
            x0 = torch.randn(M, K, generator=gen).to(dev).requires_grad_(True)
            x = x0 * 1.0
            w.grad = b.grad = None
            y = plans.apply(_LinearNode, x, w, b)
            gy = torch.randn(M, N, generator=gen).to(dev)
            y.backward(gy)
            want = gy.double().cpu() @ w.detach().double().cpu()
            err = (x0.grad.double().cpu() - want).abs().max().item() / want.abs().max().item()
            assert err < 2e-5, (k, err, plans.STATS)
            if k >= 1:        # (no update between the eager call and the recording: the recording HITS the cache)
                with torch.no_grad():
                    w.add_(0.5 * torch.randn(N, K, generator=gen).to(dev))     # the "optimizer step": version bump, same storage
        assert plans.STATS["replayed"] >= 4, plans.STATS      # forward + backward of steps 3.. came out of the plans
    finally:
        plans.enable(False)
        plans.clear()


def test_emu_replay_checks_the_cached_weight_transposes():
    _cached_transpose_case(use_emu())


@pytest.mark.gpu
def test_gpu_replay_checks_the_cached_weight_transposes():
    from tests.backends import use_hip
    _cached_transpose_case(use_hip())


@pytest.mark.gpu
def test_gpu_plans_follow_load_state_dict_and_mode_switch():
    """(medium) replays skip FrozenBatchNorm2d.folded() and the weight-table key checks, so whatever rewrites
    buffers / parameters behind a plan must invalidate it: load_state_dict with other FrozenBN statistics is followed by
    the next steps (eager -> record -> replay again) and the replayed step equals an eager step of the modified model.
    (high) switching the plane mode on a LIVE model (bf16x3p -> bf16x6p: two -> three planes per weight) rebuilds the weight
    planes instead of writing a third plane past the two-plane buffers."""
    from tests.backends import use_hip
    dev = use_hip()
    T, res = 8, 224
    _lib.set_mma_mode("bf16x3p")
    plans.clear()
    plans.enable(True)
    try:
        ops.manual_seed(7)
        model, criterion, wd = _build(dev)
        clip = _clip(dev, T, res, 0)
        for _ in range(3):
            before = _step(model, criterion, wd, clip, T, res, dev)
        replayed0 = plans.STATS["replayed"]
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        key = "vis_encoder.0.body.layer3.1.bn2.bias"
        sd[key] = sd[key] + 0.5
        epoch = plans.STATIC_EPOCH
        model.load_state_dict(sd)
        assert plans.STATIC_EPOCH > epoch
        after = [_step(model, criterion, wd, clip, T, res, dev) for _ in range(3)]
        assert plans.STATS["replayed"] > replayed0                     # the third step after the load is a replay again
        plans.enable(False)
        eager = _step(model, criterion, wd, clip, T, res, dev)
        for k in eager[0]:
            _same(after[0][0][k], eager[0][k], "first step after load_state_dict " + k, 2e-4)
            _same(after[2][0][k], eager[0][k], "replayed step after load_state_dict " + k, 2e-4)
        assert (eager[0]["pred_sted"] - before[0]["pred_sted"]).abs().max() > 1e-4     # the new statistics do matter
        # ---- plane-mode switch on the same live model
        _lib.set_mma_mode("bf16x6p")
        six = _step(model, criterion, wd, clip, T, res, dev)
        torch.cuda.synchronize()
        for k in eager[0]:
            _same(six[0][k], eager[0][k], "bf16x6p after bf16x3p on one model " + k, 2e-3)
        body = model.vis_encoder[0].body
        assert all(p.t.shape[0] == 3 for p in body._wpl_cache.fwd.values())
        _lib.set_mma_mode("bf16x3p")
        two = _step(model, criterion, wd, clip, T, res, dev)
        for k in eager[0]:
            _same(two[0][k], eager[0][k], "back to bf16x3p " + k, 2e-4)
    finally:
        plans.enable(False)
        plans.clear()
        _lib.set_mma_mode("f32")


# ------------------------------------------------------------------------------------------------------------------
# round 6: the next clip's frozen prefix (stem + max-pool + layer1) computed under the current step's grounding section
# ------------------------------------------------------------------------------------------------------------------
def _prefix_pipeline_case(dev, T, res, steps, train, tol, grad_l2):
    """Two DIFFERENT clips alternate (A, B, A, B, ...).  Reference: every step un-pipelined (the prefix computed inside the
    step), eager launches.  Pipelined: step k declares step k + 1's frames (Backbone.stage_next), launch plans on.  Every
    step's outputs, loss and gradients equal the un-pipelined step ON THE SAME CLIP — a stale prefix (clip A's handed to
    a step on clip B) or one cached across steps (the frames are rewritten in place between two visits) would show."""
    _lib.set_mma_mode("bf16x6p")
    try:
        runs = {}
        for mode in ("plain", "pipelined"):
            plans.clear()
            plans.enable(mode == "pipelined")
            plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0)
            ops.manual_seed(7)
            model, criterion, wd = _build(dev, train=train)
            bb = model.vis_encoder[0]
            clips = [_clip(dev, T, res, 0), _clip(dev, T, res, 2)]          # (both unpadded: one position table)
            base = [c.tensors.clone() for c in clips]
            res_ = []
            for k in range(steps):
                if train:
                    ops.dropout_begin_step(dev)
                cur, nxt = clips[k % 2], clips[(k + 1) % 2]
                if mode == "pipelined":
                    bb.stage_next(nxt.tensors)
                res_.append(_step(model, criterion, wd, cur, T, res, dev))
                # after the step: the NEXT clip's buffer is rewritten in place at k == 2 (fresh pixels for step 3).  The
                # prefix staged for it is stale now — the version check must send step 3 down the in-place path
                if k == 2:
                    nxt.tensors.mul_(0.5)
            runs[mode] = res_
            if mode == "pipelined":
                st = bb.prefix_stats
                # step 0 computes in place (nothing staged before it), step 3 too (its frames changed after staging)
                n_inline = 2 if steps > 3 else 1       # (step 0; step 3: its frames changed after they were staged)
                assert st["inline"] == n_inline and st["taken"] == steps - n_inline, st
                assert plans.STATS["replayed"] > 0, plans.STATS
            for c, b in zip(clips, base):
                c.tensors.copy_(b)
        _check_equal(runs["plain"], runs["pipelined"], tol, grad_l2=grad_l2)
        # the clips do differ, and the rewritten visit differs from the first visit of the same clip
        assert runs["plain"][0][1] != runs["plain"][1][1]
        assert steps <= 3 or runs["plain"][1][1] != runs["plain"][3][1]
    finally:
        plans.enable(False)
        plans.clear()
        _lib.set_mma_mode("f32")


def test_emu_prefix_pipeline_equals_unpipelined_steps():
    """the whole model on the emulator (the wiring: stage_next -> the deferred fill at the query decoder's entry -> the
    staged prefix handed to the backbone node, eager and under a recording): clip A computed in place, clip B with its
    staged prefix, clip A again with ITS staged prefix — the third step equals the first (same model, eval mode, same
    clip).  Gradients by rel-L2: the emulator's threads order the split-K atomics of the skinny forwards differently from
    run to run (1e-6 of noise between two eager steps on one clip), and one FFN pre-activation of this tiny clip sits
    within that noise of its ReLU kink — a single flipped mask element moves one row of linear1's gradient by 1e-2 (seen:
    bias gradient equal in 2047 of 2048 columns).  The backbone-level test below is the bit-exact one; the two-run
    comparison against un-pipelined steps (alternating clips, a rewritten buffer, train mode) runs on the GPU."""
    dev = use_emu()
    T, res = 2, 32
    _lib.set_mma_mode("bf16x6p")
    plans.clear()
    plans.enable(True)
    plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0)
    try:
        ops.manual_seed(7)
        model, criterion, wd = _build(dev)
        bb = model.vis_encoder[0]
        clips = [_clip(dev, T, res, 0), _clip(dev, T, res, 2)]
        got = []
        for k in range(3):
            bb.stage_next(clips[(k + 1) % 2].tensors)
            got.append(_step(model, criterion, wd, clips[k % 2], T, res, dev))
        assert bb.prefix_stats["inline"] == 1 and bb.prefix_stats["taken"] == 2, bb.prefix_stats
        assert plans.STATS["recorded"] >= 8, plans.STATS
        _check_equal([got[0]], [got[2]], 2e-5, grad_l2=3e-3)
        assert got[0][1] != got[1][1]
    finally:
        plans.enable(False)
        plans.clear()
        _lib.set_mma_mode("f32")


def _prefix_pipeline_backbone_bit_exact(dev, T, res, blocks):
    """the visual encoder alone (no atomics on this path: the plane weight gradients are ordered sums): features and every
    weight gradient of a pipelined step — eager, recorded and REPLAYED — are bit-identical to the un-pipelined step on the
    same clip; two clips alternate, one is rewritten in place between two visits"""
    from stcat_amd import backbone
    _lib.set_mma_mode("bf16x6p")
    saved = backbone.BLOCKS
    if blocks is not None:
        backbone.BLOCKS = blocks
    try:
        enc = backbone.build_vis_encoder(None)
    finally:
        backbone.BLOCKS = saved
    try:
        synth.fill_module_(enc)
        enc.to(dev)
        bb = enc[0]
        g = torch.Generator().manual_seed(1)
        clips = [torch.randn(T, 3, res, res, generator=g).to(dev) for _ in range(2)]
        gy = torch.randn(T, res // 32, res // 32, 2048, generator=g).to(dev)

        def step(frames):
            for p in bb.parameters():
                p.grad = None
            f = bb.features_nhwc(frames)
            ops.run_deferred()              # (what the query decoder's entry does in the full model)
            f.backward(gy)
            return f.detach().clone(), {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None}

        plans.clear()
        plans.enable(False)
        ref = [step(c) for c in clips]
        plans.enable(True)
        plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0)
        for k in range(4):                  # in place | staged, eager | staged, recorded | staged, REPLAYED
            cur = clips[k % 2]
            bb.stage_next(clips[(k + 1) % 2])
            f, gr = step(cur)
            rf, rg = ref[k % 2]
            assert torch.equal(f, rf), k
            assert set(gr) == set(rg) and all(torch.equal(gr[n], rg[n]) for n in rg), k
        assert bb.prefix_stats["inline"] == 3 and bb.prefix_stats["taken"] == 3, bb.prefix_stats
        assert plans.STATS["replayed"] >= 2, plans.STATS
        # a clip rewritten AFTER it was declared: the staged prefix is stale, the step computes in place and is still right
        bb.stage_next(clips[1])
        f, _ = step(clips[1])               # (the prefix staged at k = 3 belongs to clip 0: not taken; clip 1 is staged now)
        clips[1].mul_(0.5)
        plans.enable(False)
        inline_before = bb.prefix_stats["inline"]
        f1, _ = step(clips[1])
        assert bb.prefix_stats["inline"] == inline_before + 1
        bb.stage_next(None)
        f2, _ = step(clips[1])
        assert torch.equal(f1, f2) and not torch.equal(f1, ref[1][0])
    finally:
        plans.enable(False)
        plans.clear()
        _lib.set_mma_mode("f32")


def test_emu_prefix_pipeline_backbone_bit_exact():
    _prefix_pipeline_backbone_bit_exact(use_emu(), 2, 32, (1, 1, 2, 1))


def _prefix_pipeline_two_forwards_before_backward(dev, T, res, blocks):
    """gradient accumulation over two clips: forward A, forward B, then both backward passes.  B's forward must NOT stage a
    prefix (it would fill the resident buffer A's backward still reads — layer2.0's weight gradients); every gradient equals
    the one-forward-one-backward reference, bit for bit."""
    from stcat_amd import backbone
    _lib.set_mma_mode("bf16x6p")
    saved = backbone.BLOCKS
    if blocks is not None:
        backbone.BLOCKS = blocks
    try:
        enc = backbone.build_vis_encoder(None)
    finally:
        backbone.BLOCKS = saved
    try:
        synth.fill_module_(enc)
        enc.to(dev)
        bb = enc[0]
        g = torch.Generator().manual_seed(5)
        A, B, C = (torch.randn(T, 3, res, res, generator=g).to(dev) for _ in range(3))
        gy = torch.randn(T, res // 32, res // 32, 2048, generator=g).to(dev)
        plans.clear()
        plans.enable(False)

        def grads_of(frames):
            for p in bb.parameters():
                p.grad = None
            f = bb.features_nhwc(frames)
            ops.run_deferred()
            f.backward(gy)
            return {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None}

        ref_a, ref_b = grads_of(A), grads_of(B)
        # pipelined: a warm step on C stages A; then A (takes its prefix, stages B), B (takes its prefix; staging C is refused:
        # A's backward is outstanding), backward of B, backward of A
        bb.stage_next(A)
        grads_of(C)
        for p in bb.parameters():
            p.grad = None
        bb.stage_next(B)
        fa = bb.features_nhwc(A)
        ops.run_deferred()
        bb.stage_next(C)
        fb = bb.features_nhwc(B)
        ops.run_deferred()
        assert bb.prefix_stats.get("skipped_busy") == 1 and bb._prefix is None, bb.prefix_stats
        fb.backward(gy)
        gb = {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None}
        for p in bb.parameters():
            p.grad = None
        fa.backward(gy)
        ga = {n: p.grad.clone() for n, p in bb.named_parameters() if p.grad is not None}
        assert all(torch.equal(ga[n], ref_a[n]) for n in ref_a) and all(torch.equal(gb[n], ref_b[n]) for n in ref_b)
        # and the step after that stages again
        bb.stage_next(A)
        grads_of(C)
        assert bb._prefix is not None
    finally:
        plans.enable(False)
        plans.clear()
        _lib.set_mma_mode("f32")


def test_emu_prefix_pipeline_two_forwards_before_backward():
    _prefix_pipeline_two_forwards_before_backward(use_emu(), 2, 32, (1, 1, 2, 1))


@pytest.mark.gpu
def test_gpu_prefix_pipeline_two_forwards_before_backward():
    from tests.backends import use_hip
    _prefix_pipeline_two_forwards_before_backward(use_hip(), 8, 224, None)


@pytest.mark.gpu
def test_gpu_prefix_pipeline_backbone_bit_exact():
    """the full ResNet-101 at T = 8, 224 x 224: the staged prefix issues each conv once per frame range of the forward chains,
    one range after the other on its side stream; the in-step path runs the ranges on two streams — the same launches, so
    bit-identical features and weight gradients (a whole-clip launch would pick another tile / kernel variant: 1 ulp off)"""
    from tests.backends import use_hip
    _prefix_pipeline_backbone_bit_exact(use_hip(), 8, 224, None)


@pytest.mark.gpu
def test_gpu_prefix_pipeline_equals_unpipelined_steps():
    from tests.backends import use_hip
    _prefix_pipeline_case(use_hip(), 8, 224, 6, False, 2e-4, 3e-3)


@pytest.mark.gpu
def test_gpu_prefix_pipeline_train_mode():
    from tests.backends import use_hip
    # (train mode at T = 8 in bf16x6p: a gradient is a sum over few samples and two runs of ONE schedule already differ by
    #  single ReLU-kink flips behind the atomically ordered split-K sums of the grounding model — 3.0e-3 .. 6.2e-3 rel-L2 on
    #  single backbone tensors seen between runs; that the pipelined prefix itself is bit-identical is what
    #  test_gpu_prefix_pipeline_backbone_bit_exact checks, the eval-mode case above keeps the 3e-3 bar)
    _prefix_pipeline_case(use_hip(), 8, 224, 6, True, 2e-4, 1.5e-2)
