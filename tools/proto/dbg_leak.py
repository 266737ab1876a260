import gc, sys, torch
sys.path.insert(0, '.')
from stcat_amd import _lib as L, ops, synth
from stcat_amd.misc import NestedTensor
from stcat_amd.pipeline import SyntheticText, build_model
L.load(); L.set_mma_mode("bf16x6p")
dev = torch.device("cuda:0")
def mem(tag):
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    print(f"{tag:50s} {torch.cuda.memory_allocated()/2**30:7.2f} GiB live", flush=True)
T, res, Lt = 16, 448, 10
for rep in range(2):
    model, crit, wd = build_model(None, SyntheticText(synth.synth_text(Lt)))
    model.eval(); synth.fill_module_(model); model.to(dev)
    frames = synth.synth_frames(T, res).to(dev); mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
    mem("model built")
    out = model(NestedTensor(frames, mask, [T]), ["q"])
    mem("after forward (graph alive)")
    keys = list(out.keys())
    del out
    mem("out deleted")
    del model, crit
    mem("model deleted")
    ops.LINEAR_WT.entries.clear()
    mem("LINEAR_WT cleared")
    # which GPU tensors are still alive and who refers to them
    live = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda and o.numel() * o.element_size() > (64 << 20)]
    print("   big live tensors visible to gc:", [(tuple(t.shape), t.dtype) for t in live][:12])
    del live
    if rep == 0:
        nodes = [o for o in gc.get_objects() if type(o).__name__.endswith("Backward") and hasattr(o, "__dict__")]
        print("alive custom nodes:", sorted(set(type(o).__name__ for o in nodes)), len(nodes))
        def tensors_in(x, depth=0):
            if torch.is_tensor(x):
                yield x
            elif isinstance(x, (list, tuple)) and depth < 4:
                for y in x:
                    yield from tensors_in(y, depth + 1)
            elif isinstance(x, dict) and depth < 4:
                for y in x.values():
                    yield from tensors_in(y, depth + 1)
            elif hasattr(x, "t") and torch.is_tensor(getattr(x, "t", None)):
                yield x.t
            elif hasattr(x, "__dict__") and depth < 3 and type(x).__name__ in ("_ShimCtx",):
                yield from tensors_in(vars(x), depth + 1)
        for nd in nodes:
            for k, v in vars(nd).items():
                for t in tensors_in(v):
                    b = t._base
                    if b is not None and b.grad_fn is nd:
                        print("   SELF-REFERENCE through a view's base:", type(nd).__name__, "attr", k, tuple(t.shape), tuple(b.shape))
                    if t.grad_fn is nd:
                        print("   SELF-REFERENCE:", type(nd).__name__, "attr", k, tuple(t.shape))
                    elif t.grad_fn is not None:
                        print("   holds a graph tensor:", type(nd).__name__, "attr", k, tuple(t.shape), type(t.grad_fn).__name__)
