"""ss_dice_sums_static, the entry point that allocates nothing (partial sums in a file-scope device array), through the C ABI
against ss_dice_sums (caller workspace): bit-identical sums on two calls in a row, the second with a smaller input, so stale
partials of the first call would show; sizes beyond the static array are refused with a message."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(probs, labels):
    from semseg_amd import ops
    from semseg_amd._lib import lib, check
    B, C, H, W = probs.shape
    want = ops.dice_sums(probs, labels)                                    # existing entry point (torch-allocated workspace)
    got = torch.full((B, C, 3), float('nan'), dtype=torch.float32, device=probs.device)
    sn, sc, sh, sw = probs.stride()
    check(lib().ss_dice_sums_static(ops.dt(probs), probs.data_ptr(), sn, sc, sh, sw, labels.data_ptr(), B, C, H, W, got.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream))
    return got, want


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_static_scratch_entry_matches_workspace_entry_twice(dtype):
    torch.manual_seed(5)
    big = torch.randn(4, 6, 300, 520, device='cuda').softmax(1).to(dtype)              # 64 workgroups per image
    big_lab = torch.randint(0, 6, (4, 300, 520), device='cuda')
    small = torch.randn(2, 6, 40, 56, device='cuda').softmax(1).to(dtype)               # 1 workgroup per image, fewer images
    small = small.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)                  # channels-last strides
    small_lab = torch.randint(0, 6, (2, 40, 56), device='cuda')
    for probs, lab in ((big, big_lab), (small, small_lab), (big, big_lab)):
        got, want = _both(probs, lab)
        assert torch.equal(got, want)
        ref_cnt = torch.stack([torch.bincount(l.reshape(-1), minlength=6) for l in lab]).float()
        assert torch.equal(got[..., 2], ref_cnt)                                        # and the counts are right, not merely equal


def test_static_scratch_entry_rejects_oversize():
    from semseg_amd import ops
    from semseg_amd._lib import lib
    limit = lib().ss_dice_sums_static_max_floats()
    B, C, H, W = 16, 100, 1024, 1024                                                    # 16 * 64 * 3 * 100 = 307200 > 262144
    assert B * 64 * 3 * C > limit
    probs = torch.zeros(1, device='cuda').expand(B, C, H, W)                            # stride-0 view: nothing that size is allocated
    lab = torch.zeros((B, H, W), dtype=torch.int64, device='cuda')
    out = torch.zeros((B, C, 3), device='cuda')
    rc = lib().ss_dice_sums_static(0, probs.data_ptr(), 0, 0, 0, 0, lab.data_ptr(), B, C, H, W, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b'exceed the static scratch' in lib().ss_last_error()
    assert float(out.abs().max()) == 0.0                                                # nothing was launched
