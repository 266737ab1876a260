"""Optimizer tail (clip_grad_norm_ + AdamW + EMA, scripts/train_net.py:134-143) through the C ABI against
torch.optim.AdamW / torch.nn.utils.clip_grad_norm_ / the reference's update_ema arithmetic on CPU, plus the
learning-rate schedule of engine/lr_scheduler.py:212-252 against known values."""
import copy
from types import SimpleNamespace as NS

import pytest
import torch
from torch import nn

from stcat_amd import optim
from tests.backends import both, close


class _Toy(nn.Module):
    """parameter names hit all four groups of engine/optimizer.py:26-43; sizes cover the vector path, ragged tails,
    several chunks and a frozen tensor"""

    def __init__(self):
        super().__init__()
        self.vis_encoder = nn.Linear(300, 301)                      # 90 300 + 301 elements: > 1 chunk, ragged
        self.text_encoder = nn.Linear(7, 5)
        self.ground_decoder = nn.Module()
        self.ground_decoder.temp_decoder = nn.Linear(16, 3)
        self.rest = nn.Parameter(torch.randn(2, 65536 + 5))        # chunk boundary + tail
        # a 3x3 conv weight in channels_last memory (physically OHWI), the layout of every backbone conv weight
        # AND of its gradient (stcat_amd/backbone.py) — ADVICE r01: the standalone clip scrambled these
        self.vis_encoder_conv = nn.Parameter(torch.randn(64, 64, 3, 3).contiguous(memory_format=torch.channels_last))
        self.frozen = nn.Parameter(torch.randn(9), requires_grad=False)
        self.unused = nn.Parameter(torch.randn(4))                  # never gets a gradient


def _cfg():
    return NS(SOLVER=NS(OPTIMIZER="adamw", BASE_LR=3e-4, VIS_BACKBONE_LR=2e-5, TEXT_LR=5e-5, TEMP_LR=1e-4,
                        WEIGHT_DECAY=1e-4, MAX_GRAD_NORM=0.1, WARMUP_PROP=0.01, MAX_EPOCH=10,
                        SCHEDULE=NS(TYPE="multistep_with_warmup", DROP_STEP=[8])),
              MODEL=NS(EMA_DECAY=0.9998))


def _fake_grads(model, step):
    g = torch.Generator().manual_seed(100 + step)
    out = {}
    for n, p in model.named_parameters():
        if p.requires_grad and n != "unused":
            out[n] = torch.randn(p.shape, generator=g) * (10.0 if step == 1 else 0.01)   # clipped / not clipped
    return out


def _reference_run(cfg, steps, decay):
    torch.manual_seed(0)
    model = _Toy()
    ema = copy.deepcopy(model)
    named = dict(model.named_parameters())
    groups = [{"params": [named["rest"], named["unused"]]},
              {"params": [named["vis_encoder.weight"], named["vis_encoder.bias"], named["vis_encoder_conv"]],
               "lr": cfg.SOLVER.VIS_BACKBONE_LR},
              {"params": [named["text_encoder.weight"], named["text_encoder.bias"]], "lr": cfg.SOLVER.TEXT_LR},
              {"params": [named["ground_decoder.temp_decoder.weight"], named["ground_decoder.temp_decoder.bias"]],
               "lr": cfg.SOLVER.TEMP_LR}]
    opt = torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    norms = []
    for s in range(steps):
        opt.zero_grad()
        for n, g in _fake_grads(model, s).items():
            named[n].grad = g.clone()
        norms.append(torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.MAX_GRAD_NORM).item())
        opt.step()
        optim.adjust_learning_rate(cfg, opt, s, 1000)
        with torch.no_grad():
            msd = model.state_dict()
            for k, v in ema.state_dict().items():
                v.copy_(v * decay + (1.0 - decay) * msd[k])
    return model, ema, norms


@both
def _fused_tail(dev, big):
    cfg = _cfg()
    decay = 0.9
    steps = 4
    ref_model, ref_ema, ref_norms = _reference_run(cfg, steps, decay)
    torch.manual_seed(0)
    model = _Toy().to(dev)
    ema = copy.deepcopy(model)
    opt = optim.make_optimizer(cfg, model)
    assert [len(g["params"]) for g in opt.param_groups] == [2, 3, 2, 2]
    named = dict(model.named_parameters())
    for s in range(steps):
        opt.zero_grad()
        for n, g in _fake_grads(model, s).items():
            # gradients arrive in the parameter's own memory layout (channels_last for conv weights)
            named[n].grad = torch.empty_like(named[n]).copy_(g.to(dev))
        sq = opt.step(max_grad_norm=cfg.SOLVER.MAX_GRAD_NORM, model_ema=ema, ema_decay=decay, model=model)
        assert abs(sq.sqrt().item() - ref_norms[s]) <= 1e-5 * ref_norms[s]
        optim.adjust_learning_rate(cfg, opt, s, 1000)
    for (n, p), (_, r) in zip(model.named_parameters(), ref_model.named_parameters()):
        close(p, r, 2e-6, "parameter " + n)
    for (n, p), (_, r) in zip(ema.named_parameters(), ref_ema.named_parameters()):
        close(p, r, 2e-6, "ema " + n)
    assert torch.equal(model.frozen.cpu(), ref_model.frozen) and torch.equal(model.unused.cpu(), ref_model.unused)
    # checkpoint round trip of the moments, in torch.optim.AdamW's own layout (utils/checkpoint.py saves
    # optimizer.state_dict()): ours loads into torch's optimizer and torch's loads into ours
    sd = opt.state_dict()
    opt2 = optim.make_optimizer(cfg, model)
    opt2.load_state_dict(sd)
    assert opt2.step_count == steps and len(opt2.state) == len(opt.state)
    cpu_model = _Toy()
    topt = torch.optim.AdamW([{"params": g["params"]} for g in optim.make_optimizer(cfg, cpu_model).param_groups],
                             lr=cfg.SOLVER.BASE_LR)
    topt.load_state_dict({"state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                          "param_groups": sd["param_groups"]})
    tsd = topt.state_dict()
    assert all(int(v["step"]) == steps for v in tsd["state"].values())
    opt3 = optim.make_optimizer(cfg, model)
    opt3.load_state_dict(tsd)
    assert opt3.step_count == steps
    conv = dict(model.named_parameters())["vis_encoder_conv"]
    close(opt3.state[conv]["exp_avg"], opt.state[conv]["exp_avg"], 1e-7, "reloaded channels_last moment")
    assert opt3.state[conv]["exp_avg"].stride() == conv.stride()


@both
def _standalone_clip_and_ema(dev, big):
    torch.manual_seed(1)
    model = _Toy()
    ref = copy.deepcopy(model)
    for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        if p.requires_grad and n != "unused":
            r.grad = torch.randn_like(r) * 3
    model = model.to(dev)
    for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        if r.grad is not None:
            p.grad = torch.empty_like(p).copy_(r.grad.to(dev))     # parameter's layout (channels_last conv grads)
            assert p.grad.stride() == p.stride()
    n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
    n_hip = optim.clip_grad_norm_(model.parameters(), 0.1)
    assert abs(n_hip.item() - n_ref.item()) <= 1e-5 * n_ref.item()
    for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        if r.grad is not None:
            close(p.grad, r.grad, 2e-6, "clipped grad " + n)
    ema, ema_ref = copy.deepcopy(model), copy.deepcopy(ref)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
        for p in ref.parameters():
            p.add_(1.0)
        for k, v in ema_ref.state_dict().items():
            v.copy_(v * 0.75 + 0.25 * ref.state_dict()[k])
    optim.update_ema(model, ema, 0.75)
    for (n, p), (_, r) in zip(ema.named_parameters(), ema_ref.named_parameters()):
        close(p, r, 1e-6, "standalone ema " + n)


def test_lr_schedule_values():
    cfg = _cfg()
    opt = NS(param_groups=[{"lr": 0.0} for _ in range(4)])
    total = 1000                                  # warmup = 10 steps, 100 steps per epoch
    optim.adjust_learning_rate(cfg, opt, 5, total)
    lrs = [g["lr"] for g in opt.param_groups]
    assert lrs[0] == cfg.SOLVER.BASE_LR and lrs[1] == cfg.SOLVER.VIS_BACKBONE_LR
    assert abs(lrs[2] - cfg.SOLVER.TEXT_LR * 0.5) < 1e-12 and abs(lrs[3] - cfg.SOLVER.TEMP_LR * 0.5) < 1e-12
    optim.adjust_learning_rate(cfg, opt, 850, total)          # epoch 8 >= DROP_STEP -> x0.1; linear decay side
    lrs = [g["lr"] for g in opt.param_groups]
    assert abs(lrs[0] - cfg.SOLVER.BASE_LR * 0.1) < 1e-12
    assert abs(lrs[2] - cfg.SOLVER.TEXT_LR * (150 / 990)) < 1e-12
    cfg.SOLVER.SCHEDULE.TYPE = "multistep_with_warmup_all"
    optim.adjust_learning_rate(cfg, opt, 2, total)
    assert all(abs(g["lr"] - b * 0.2) < 1e-12 for g, b in zip(opt.param_groups, [3e-4, 2e-5, 5e-5, 1e-4]))


@pytest.mark.gpu
def test_gpu_fp16_plane_mode_overflow_skips_the_step_and_backs_the_scale_off():
    """mode f16x3p with an absurd gradient scale (2^28: every gradient plane overflows fp16): the global gradient norm is
    non-finite, AdamW.step() leaves the weights untouched (the clip alone would have made them NaN), and PlaneLossScale
    halves the scale until a step goes through — then the weights move."""
    import torch
    from stcat_amd import _lib, optim, synth
    from stcat_amd.misc import BoxList, NestedTensor
    from stcat_amd.pipeline import SyntheticText, build_model
    from tests.backends import use_hip
    dev = use_hip()
    T, res, L_ = 4, 128, 6
    _lib.set_mma_mode("f16x3p")
    try:
        model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L_)))
        model.eval()
        synth.fill_module_(model)
        model.to(dev)
        opt = optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
        scaler = optim.PlaneLossScale(init_log2=28, growth_interval=1000)
        frames = synth.synth_frames(T, res).to(dev)
        mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
        act, tb = synth.synth_targets(T)
        tgt = [{"actioness": act.to(dev), "boxs": BoxList(tb, (res, res)).to(dev)}]
        probe = model.vis_encoder[0].body.layer3[0].conv2.weight
        w0 = probe.detach().clone()
        applied = []
        for k in range(24):
            opt.zero_grad()
            out = model(NestedTensor(frames, mask, [T]), ["q"])
            l = criterion(out, tgt, [T])
            sum(l[n] * wd[n] for n in l).backward()
            ok = scaler.update(opt.step(max_grad_norm=0.1))
            applied.append(ok)
            if not ok:
                assert torch.equal(probe.detach(), w0), "a skipped step must not touch the weights"
            else:
                break
        assert applied[0] is False and applied[-1] is True, applied          # overflowed first, recovered by backing off
        assert scaler.skipped >= 1 and scaler.log2 < 28
        assert torch.isfinite(probe).all() and not torch.equal(probe.detach(), w0)
    finally:
        _lib.call("stcat_set_f16_scales", 6, 16)
        _lib.set_mma_mode("f32")
