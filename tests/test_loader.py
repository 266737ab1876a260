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
