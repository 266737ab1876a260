"""Copy-stream frame prefetcher (stcat_amd/loader.py): every clip arrives intact and in order while the
consumer keeps a kernel queue running on the previous one; two resident buffers alternate."""
import pytest
import torch


@pytest.mark.gpu
def test_gpu_frame_prefetcher_delivers_clips_in_order():
    from stcat_amd.loader import DeviceFramePrefetcher
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    clips = [torch.randint(0, 256, (8, 64, 64, 3), dtype=torch.uint8, generator=g) for _ in range(6)]
    clips[1] = clips[1].pin_memory()          # pinned and pageable sources both
    seen, ptrs = [], []
    acc = torch.zeros(8, 64, 64, 3, device=dev)
    for fr in DeviceFramePrefetcher(iter(clips), dev):
        ptrs.append(fr.data_ptr())
        for _ in range(20):                   # a queue of consumer work that reads the buffer after the hand-over
            acc = acc * 0.5 + fr.float()
        seen.append(fr.clone())
    torch.cuda.synchronize()
    assert len(seen) == len(clips)
    for a, b in zip(seen, clips):
        assert torch.equal(a.cpu(), b)
    assert len(set(ptrs)) == 2 and ptrs[0] == ptrs[2] and ptrs[1] == ptrs[3] and ptrs[0] != ptrs[1]
    want = torch.zeros(8, 64, 64, 3)
    for c in clips:
        for _ in range(20):
            want = want * 0.5 + c.float()
    assert torch.allclose(acc.cpu(), want, rtol=1e-5, atol=1e-3)


@pytest.mark.gpu
def test_gpu_prefetcher_declares_the_next_clip_to_the_backbone():
    """round 6: DeviceFramePrefetcher(stage=Backbone.stage_next) — uint8 decoder frames arrive on the copy stream, every
    hand-over declares the NEXT clip (with its copy event) and that clip's frozen prefix runs under the current step; the
    features of every clip equal those of a plain pass over the same frames, bit for bit"""
    from stcat_amd import _lib, backbone, ops, synth
    from stcat_amd.loader import DeviceFramePrefetcher
    dev = torch.device("cuda:0")
    _lib.load()
    _lib.set_mma_mode("bf16x6p")
    try:
        enc = backbone.build_vis_encoder(None)
        synth.fill_module_(enc)
        enc.to(dev)
        bb = enc[0]
        g = torch.Generator().manual_seed(3)
        clips = [torch.randint(0, 256, (8, 224, 224, 3), dtype=torch.uint8, generator=g) for _ in range(5)]
        with torch.no_grad():
            want = [bb.features_nhwc(c.to(dev)).clone() for c in clips]
            got = []
            taken0 = bb.prefix_stats["taken"]
            for fr in DeviceFramePrefetcher(iter(clips), dev, stage=bb.stage_next):
                got.append(bb.features_nhwc(fr).clone())
                ops.run_deferred()              # (the query decoder's entry in the full model)
        torch.cuda.synchronize()
        assert bb.prefix_stats["taken"] - taken0 == len(clips) - 1, bb.prefix_stats
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    finally:
        _lib.set_mma_mode("f32")
