"""Pins the CPU oracle (oracle/stcat_oracle.py) against golden vectors produced by
the imported reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import stcat_oracle as O
from stcat_amd import synth

from tests.golden.make_golden import sub  # same deterministic sub-sampling


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _close(a, b, tol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3g}) > {tol}"


@pytest.fixture(scope="module")
def c1(golden_dir):
    g = _load(golden_dir, "C1.npz")
    T, res, L = [int(v) for v in g["meta/config"]]
    sd = synth.synth_state_dict()
    for v in sd.values():
        v.requires_grad_(True)
    frames = synth.synth_frames(T, res)
    mask = torch.zeros(T, res, res, dtype=torch.bool)
    text = synth.synth_text(L)
    out = O.stcat_forward(sd, frames, mask, text, return_stages=True)
    return g, sd, out, (T, res, L)


def test_state_dict_keys_match_reference(c1):
    g, sd, _, _ = c1
    ref_keys = set(str(k) for k in g["meta/state_dict_keys"])
    assert set(sd.keys()) == ref_keys


def test_forward_stages(c1):
    g, _, out, _ = c1
    st = out["_stages"]
    for k in ("layer1", "layer2", "layer3", "layer4", "vis_pos", "input_proj", "encoded_memory"):
        _close(sub(st[k]), g[f"stage/{k}"], 2e-5, k)
    for k in ("frames_cls", "videos_cls", "pos_query", "hs", "ref", "time_hs", "weights"):
        _close(st[k].detach().numpy(), g[f"stage/{k}"], 2e-5, k)


def test_forward_outputs(c1):
    g, _, out, _ = c1
    for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
        _close(out[k].detach().numpy(), g[f"out/{k}"], 2e-5, k)
        for i, aux in enumerate(out["aux_outputs"]):
            _close(aux[k].detach().numpy(), g[f"out/aux{i}/{k}"], 2e-5, f"aux{i}/{k}")


def test_post_process(c1):
    g, _, out, (T, res, L) = c1
    sizes = torch.tensor([[float(res), float(res)]]).repeat(T, 1)
    boxes, sted, _ = O.post_process(out["pred_sted"].detach(), out["pred_boxes"].detach(), sizes,
                                    list(range(100, 100 + T)), T)
    _close(boxes.numpy(), g["post/boxes"], 2e-5, "post boxes")
    assert sted == [int(v) for v in g["post/sted"][0]]  # bit-exact span


def test_loss_and_grads(c1):
    g, sd, out, (T, res, L) = c1
    act, boxes = synth.synth_targets(T)
    losses = O.criterion(out, act, boxes)
    keys = [str(k) for k in g["loss/keys"]]
    assert sorted(losses.keys()) == keys
    vals = np.asarray([losses[k].item() for k in keys])
    _close(vals, g["loss/values"], 2e-5, "loss values")
    wd = O.weight_dict()
    _close(np.asarray([wd[k] for k in keys]), g["loss/weights"], 0, "weight dict")
    total = O.total_loss(losses, wd)
    _close(total.item(), g["loss/total"], 2e-5, "total")
    total.backward()
    unused = set(str(k) for k in g["grad/unused"])
    names = [str(n) for n in g["grad/names"]]
    norms = []
    for n in names:
        gr = sd[synth.canonical_name(n)].grad
        assert gr is not None, n
        norms.append(gr.norm().item())
    _close(np.asarray(norms), g["grad/norms"], 1e-4, "grad norms")
    for n in unused:
        assert sd[n].grad is None or float(sd[n].grad.abs().max()) == 0.0, n
    for k in g.files:
        if k.startswith("grad/full/"):
            n = k[len("grad/full/"):]
            _close(sub(sd[synth.canonical_name(n)].grad), g[k], 1e-4, k)


def test_op_vectors(golden_dir):
    g = _load(golden_dir, "ops.npz")
    _close(O.gen_sineembed(torch.from_numpy(g["sine/anchors"])).numpy(), g["sine/embed"], 1e-6, "sine")
    _close(O.inverse_sigmoid(torch.from_numpy(g["invsig/x"])).numpy(), g["invsig/y"], 1e-6, "invsig")
    _close(O.pos_sine_2d(torch.from_numpy(g["pos2d/mask"])).numpy(), g["pos2d/pos"], 1e-6, "pos2d")
    _close(synth.time_sine_table(301)[:10], g["seqsine/te"], 1e-6, "seq sine")
    sd = {"x.out_proj.weight": torch.from_numpy(synth.synth_value("op/dab/out_proj.weight", (256, 256)).copy()),
          "x.out_proj.bias": torch.from_numpy(synth.synth_value("op/dab/out_proj.bias", (256,)).copy())}
    q = torch.from_numpy(synth.hash_normal("op/dab/q", 3 * 512).reshape(1, 3, 512))
    k = torch.from_numpy(synth.hash_normal("op/dab/k", 11 * 3 * 512).reshape(11, 3, 512))
    v = torch.from_numpy(synth.hash_normal("op/dab/v", 11 * 3 * 256).reshape(11, 3, 256))
    o = O.dab_mha(sd, "x.", q, k, v, torch.from_numpy(g["dab/kpm"]))
    _close(o.numpy(), g["dab/out"], 1e-5, "dab mha")
    sted = torch.from_numpy(g["post/in_sted"])
    boxes = torch.from_numpy(g["post/in_boxes"])
    T = sted.shape[1]
    sizes = torch.tensor([[240.0, 320.0]]).repeat(T, 1)
    for dur in (12, 9, 6):
        pb, st, _ = O.post_process(sted, boxes, sizes, list(range(50, 50 + T)), dur)
        assert st == [int(v) for v in g[f"post/dur{dur}/sted"][0]], dur
    _close(pb.numpy(), g["post/boxes"], 1e-6, "post boxes")


def test_linear_interp_against_reference_golden(golden_dir):
    """engine/evaluate.py:11-35: the oracle restatement AND the product's tensor form against the reference's output"""
    import numpy as np
    import torch
    from oracle import stcat_oracle as O
    from stcat_amd.pipeline import linear_interp
    g = np.load(os.path.join(golden_dir, "eval.npz"))
    ids, boxes = g["interp/ids"], g["interp/boxes"]
    d = {int(f): [[float(v) for v in b]] for f, b in zip(ids, boxes)}
    out = O.linear_interp(dict(d))
    assert sorted(out) == g["interp/out_ids"].tolist()
    assert np.array_equal(np.array([out[f][0] for f in sorted(out)]), g["interp/out_boxes"])      # bit-exact (fp64)
    perm = torch.randperm(len(ids), generator=torch.Generator().manual_seed(0))                  # any input order
    full, bx = linear_interp(ids[perm.numpy()].tolist(), torch.from_numpy(boxes)[perm])
    assert full == g["interp/out_ids"].tolist()
    assert np.array_equal(bx.numpy(), g["interp/out_boxes"])
    full1, bx1 = linear_interp([7], torch.tensor([[1.0, 2.0, 3.0, 4.0]]))
    assert full1 == g["interp/single_ids"].tolist() and bx1.shape == (1, 4)


def _oracle_case(name, dtype):
    T, res, L, pad, bwd = synth.MODEL_CASES[name]
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        sd = {k: v.to(dtype).requires_grad_(True) for k, v in synth.synth_state_dict().items()}
        frames, mask, H, W = synth.synth_clip(T, res, pad)
        (tm, tmem, _), tcls = synth.synth_text(L)
        out = O.stcat_forward(sd, frames.to(dtype), mask, ((tm, tmem.to(dtype), None), tcls.to(dtype)))
        grads = None
        if bwd:
            act, tb = synth.synth_targets(T)
            O.total_loss(O.criterion(out, act, tb.to(dtype))).backward()
            grads = {k: v.grad for k, v in sd.items() if v.grad is not None}
        return out, grads
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("name", ["C1"] + (["SQ8_ragged", "NS8_ragged", "C3"] if os.environ.get("STCAT_SLOW") else []))
def test_oracle_against_model_fixture(golden_dir, name):
    """The round-4 model fixtures (tests/golden/model_<case>.npz: the imported reference in fp32 AND fp64) pin the oracle
    in both precisions: outputs, and every gradient tensor at the fixture's sample positions.  C1 runs in the CPU suite;
    STCAT_SLOW=1 adds the padded square clip, the padded 405 x 720 clip and C3 (minutes, ~40 GB).  Measured (C1 / padded
    square): fp64 3.3e-8 / 3.5e-8 worst tensor (= the fixture's fp32 storage), fp32 median 3e-7, worst 4.8e-3 / 7.2e-3."""
    g = _load(golden_dir, f"model_{name}.npz")
    names = [str(n) for n in g["grad/names"]]
    offs = g["grad/offsets"]
    for dtype, okey, skey, tol in ((torch.float32, "out/", "grad/sample32", 2e-5), (torch.float64, "out64/", "grad/sample64", 2e-6)):
        out, grads = _oracle_case(name, dtype)
        for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
            _close(out[k].detach().numpy(), g[okey + k], tol, f"{name} {dtype} {k}")
        errs = []
        for i, n in enumerate(names):
            gr = grads[synth.canonical_name(n)].reshape(-1)
            idx = torch.from_numpy(synth.sample_indices(n, gr.numel()))
            ref = g[skey][offs[i]:offs[i + 1]].astype(np.float64)
            got = gr[idx].double().numpy()
            # (absolute floor: key-side attention biases have a structurally zero gradient — softmax shift invariance — and
            #  what is left there is fp32 noise of 1e-7 that depends on the thread partition)
            errs.append(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-5 * ref.size ** 0.5))
        errs = np.sort(np.asarray(errs))
        print(f"{name} {dtype}: oracle vs reference gradient samples, rel-L2 median {errs[len(errs) // 2]:.2e} max {errs[-1]:.2e}")
        if dtype == torch.float64:
            # exact arithmetic on both sides (the fixture stores fp64 values to fp32 precision)
            assert errs[-1] <= 1e-6, (name, errs[-5:])
        else:
            # two fp32 evaluations of this chain (module tree vs functional restatement: other summation orders, other
            # thread partitions) differ by ReLU-kink flips: as far apart as the fp32 reference is from its fp64 run
            # (C1: layer2 / layer3 tensors 1e-2 .. 3e-2) — the fp64 comparison above is the sharp one
            assert errs[len(errs) // 2] <= 1e-5 and errs[-1] <= 2e-1, (name, errs[len(errs) // 2], errs[-5:])   # (8 threads: 4.8e-3; 2 threads: 7.8e-2)


def test_train_mode_fixture_is_the_oracle_with_the_recorded_dropout_stream(golden_dir):
    """Round 6: tests/golden/model_C1_train.npz (and its C3 sibling, same recipe) holds what the ORACLE gives for the
    benchmark's train-mode step when every dropout site draws the masks of the stream recorded on the GPU — seed, device
    base and (offset, size) per site, stored inside the fixture.  Re-derived here from that stream alone: the host twin of
    csrc/stcat_rng.h rebuilds the 124 masks, the oracle runs with them in fp32, and outputs, span and the 30 losses
    reproduce the stored ones; the eval-mode fixture of the same clip is far away (the masks matter)."""
    from tests import test_model_parity as P
    g = _load(golden_dir, "model_C1_train.npz")
    T, H, W, L = (int(v) for v in g["meta/config"])
    assert (T, H, W, L) == (8, 224, 224, 10) and str(g["meta/source"]).startswith("oracle")
    sites = [(int(o), int(n)) for o, n in g["dropout/sites"]]
    assert len(sites) == 12 * 4 + 6 * 6 + 6 * 6 + 4
    seed, base = int(g["dropout/seed"]), int(g["dropout/base"])
    out, boxes, sted, losses, _ = P._run_oracle(T, H, L, True, torch.float32, None,
                                                lambda: P._HipMasks(sites, seed, base), trainable_only=True)
    for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
        _close(out[k].detach().numpy(), g[f"out/{k}"], 2e-6, "train fixture " + k)
    assert [sted] == g["post/sted"].tolist()
    for k, v in zip(g["loss/keys"], g["loss/values"]):
        assert abs(losses[str(k)] - float(v)) <= 2e-5 * max(1.0, abs(float(v))), (k, losses[str(k)], float(v))
    ev = _load(golden_dir, "model_C1.npz")
    assert np.abs(ev["out/pred_sted"] - g["out/pred_sted"]).max() > 1e-2
