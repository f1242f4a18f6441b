"""GPU: the device input pipeline (mono_depth_estimation_amd/augment.py, csrc/augment.hip) against PIL itself — every
operation and the whole of base_module.py's train_preprocess / val_preprocess BIT-exact (integer work: equality, no
tolerance)."""
import numpy as np
import pytest
import torch

from oracle import augment as OA

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def _img(a):
    return Image.fromarray(a[:, :, 0], "L") if a.shape[2] == 1 else Image.fromarray(a, "RGB")


def test_resize_and_rotate_match_pil_bit_for_bit():
    from mono_depth_estimation_amd import augment
    rng = np.random.RandomState(0)
    for trial in range(30):
        H, W, C = rng.randint(20, 300), rng.randint(20, 400), int(rng.choice([1, 3]))
        a = rng.randint(0, 256, (H, W, C)).astype(np.uint8)
        t = torch.from_numpy(a).cuda()
        ow, oh = (W if trial % 3 == 0 else rng.randint(8, 500)), (H if trial % 5 == 0 else rng.randint(8, 500))
        ref = np.array(_img(a).resize((ow, oh), Image.BILINEAR)).reshape(oh, ow, C)
        got = augment.resize_u8(t, ow, oh).cpu().numpy()
        assert np.array_equal(got, ref), ("resize", H, W, C, oh, ow, int(np.abs(got.astype(int) - ref.astype(int)).max()))
        ang = float(rng.uniform(-30, 30)) if trial % 4 else [0.0, 180.0, 90.0, -5.0][trial // 4 % 4]
        ref = np.array(_img(a).rotate(ang, Image.NEAREST, False, None)).reshape(H, W, C)
        assert np.array_equal(augment.rotate_u8(t, ang).cpu().numpy(), ref), ("rotate", H, W, C, ang)
    # a multi-channel "image" (the D depth layers of a sample) equals its layers processed one by one
    a = rng.randint(0, 256, (90, 130, 5)).astype(np.uint8)
    got = augment.rotate_u8(augment.resize_u8(torch.from_numpy(a).cuda(), 77, 61), 3.3).cpu().numpy()
    for c in range(5):
        ref = np.array(Image.fromarray(a[:, :, c], "L").resize((77, 61), Image.BILINEAR).rotate(3.3, Image.NEAREST, False, None))
        assert np.array_equal(got[:, :, c], ref)


@pytest.mark.parametrize("size,resize_to,out", [((480, 640), 250, (240, 320)), ((427, 561), 250, (240, 320)), ((120, 160), 64, (56, 72))])
def test_train_and_val_preprocess_match_the_reference_pipeline(size, resize_to, out):
    from mono_depth_estimation_amd import augment
    rng = np.random.RandomState(7)
    H, W = size
    rgb = torch.from_numpy(rng.rand(3, H, W).astype(np.float32))
    depth = [torch.from_numpy(rng.rand(1, H, W).astype(np.float32)) for _ in range(3)]
    for seed in range(6):
        np.random.seed(seed)
        r_ref, d_ref = OA.train_preprocess(rgb, depth, resize_to, out)            # draws (s, angle, flip) as the reference does
        np.random.seed(seed)
        r, d = augment.train_preprocess(rgb.cuda(), [x.cuda() for x in depth], resize_to, out)
        assert r.shape == r_ref.shape and d.shape == d_ref.shape and r.dtype == torch.float32
        assert torch.equal(r.cpu(), r_ref), ("rgb", seed, float((r.cpu() - r_ref).abs().max()))
        assert torch.equal(d.cpu(), d_ref), ("depth", seed, float((d.cpu() - d_ref).abs().max()))
    r_ref, d_ref = OA.val_preprocess(rgb, depth, resize_to, out)
    r, d = augment.val_preprocess(rgb.cuda(), [x.cuda() for x in depth], resize_to, out)
    assert torch.equal(r.cpu(), r_ref) and torch.equal(d.cpu(), d_ref)
    # uint8 H x W x 3 input (what ToPILImage takes from an ndarray) goes through unchanged
    u8 = (rgb.mul(255).byte()).permute(1, 2, 0).contiguous()
    r2, _ = augment.val_preprocess(u8.cuda(), [x.cuda() for x in depth], resize_to, out)
    assert torch.equal(r2.cpu(), r_ref)


def test_pipeline_against_the_committed_pil_vectors(golden):
    """The same check against vectors minted by PIL in the build container (tests/golden/gen_golden.py augment): does not need
    Pillow on the box."""
    from mono_depth_estimation_amd import augment
    g = golden("augment")
    rgb, depth = torch.from_numpy(g["rgb"]).cuda(), [torch.from_numpy(g["depth"][i:i + 1]).cuda() for i in range(2)]
    lut = torch.from_numpy(np.arange(256, dtype=np.float32) / 255.0)
    for seed in range(4):
        np.random.seed(seed)
        r, d = augment.train_preprocess(rgb, depth, 64, (56, 72))
        assert torch.equal(r.cpu(), lut[torch.from_numpy(g["train%d_rgb" % seed]).long()]), seed
        assert torch.equal(d.cpu(), lut[torch.from_numpy(g["train%d_depth" % seed]).long()]), seed
    r, d = augment.val_preprocess(rgb, depth, 64, (56, 72))
    assert torch.equal(r.cpu(), lut[torch.from_numpy(g["val_rgb"]).long()]) and torch.equal(d.cpu(), lut[torch.from_numpy(g["val_depth"]).long()])


@pytest.mark.parametrize("size,out", [((480, 640), (416, 544)), ((600, 600), (512, 512))])
def test_bts_train_preprocess_matches_the_reference_pipeline(size, out):
    """modules/bts.py:154-199 (margin crop with PIL's box rounding, RandomRotation / RandomCrop draws from torch's generator,
    np.random.choice resize): bit-exact against the pipeline composed over PIL."""
    from mono_depth_estimation_amd import augment
    rng = np.random.RandomState(9)
    H, W = size
    rgb = torch.from_numpy(rng.rand(3, H, W).astype(np.float32))
    depth = [torch.from_numpy(rng.rand(1, H, W).astype(np.float32)) for _ in range(2)]
    for seed in range(5):
        np.random.seed(seed)
        torch.manual_seed(seed)
        r_ref, d_ref = OA.bts_train_preprocess(rgb, depth, out)
        np.random.seed(seed)
        torch.manual_seed(seed)
        r, d = augment.bts_train_preprocess(rgb.cuda(), [x.cuda() for x in depth], out)
        assert r.shape == (3, *out) and d.shape == (2, *out)
        assert torch.equal(r.cpu(), r_ref) and torch.equal(d.cpu(), d_ref), seed


@pytest.mark.parametrize("size", [(480, 640), (427, 561), (600, 450)])
def test_midas_train_and_val_preprocess_match_the_reference_pipeline(size):
    """modules/midas.py:107-150: random Resize / RandomCrop(384) / flip (draws from numpy's and torch's global generators in the
    reference's order), then the hub transform on the colour image -- bit-exact against the pipeline composed over PIL and the
    transform's published definition evaluated by numpy (oracle/augment.py)."""
    from mono_depth_estimation_amd import augment
    rng = np.random.RandomState(11)
    H, W = size
    rgb = torch.from_numpy(rng.rand(3, H, W).astype(np.float32))
    depth = [torch.from_numpy(rng.rand(1, H, W).astype(np.float32)) for _ in range(2)]
    for seed in range(5):
        np.random.seed(seed)
        torch.manual_seed(seed)
        r_ref, d_ref = OA.midas_train_preprocess(rgb, depth)
        np.random.seed(seed)
        torch.manual_seed(seed)
        r, d = augment.midas_train_preprocess(rgb.cuda(), [x.cuda() for x in depth])
        assert r.shape == (3, 384, 384) and d.shape == (2, 384, 384) and r.dtype == torch.float32
        assert torch.equal(r.cpu(), r_ref), ("rgb", seed, float((r.cpu() - r_ref).abs().max()))
        assert torch.equal(d.cpu(), d_ref), ("depth", seed)
    r_ref, d_ref = OA.midas_val_preprocess(rgb, depth)
    r, d = augment.midas_val_preprocess(rgb.cuda(), [x.cuda() for x in depth])
    assert torch.equal(r.cpu(), r_ref) and torch.equal(d.cpu(), d_ref)
    with pytest.raises(NotImplementedError, match="cv2"):
        augment.midas_test_preprocess(rgb.cuda(), [x.cuda() for x in depth])


def test_vnl_flip_pad_crop_matches_numpy():
    """modules/vnl.py:32-78 up to its cv2.resize: the draws of set_flip_pad_reshape_crop and np.flip / np.pad / the crop, for the
    uint8 image (pad value 128) and the float32 depth map (pad value -1), 'train' and 'val' phases, with and without padding."""
    from mono_depth_estimation_amd import augment
    rng = np.random.RandomState(13)
    for trial, (H, W) in enumerate([(512, 683), (400, 640), (512, 512), (448, 600)]):
        A = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        B = rng.rand(H, W).astype(np.float32) * 10
        for phase in ("train", "val"):
            np.random.seed(trial)
            flip, crop, pad, ratio = OA.vnl_draw_params(phase, (H, W))
            np.random.seed(trial)
            assert augment.vnl_draw_params(phase, (H, W)) == (flip, crop, pad, ratio)
            a_ref, b_ref = OA.vnl_flip_pad_crop(A, flip, crop, pad, 128), OA.vnl_flip_pad_crop(B, flip, crop, pad, -1)
            a = augment.vnl_flip_pad_crop(torch.from_numpy(A).cuda(), flip, crop, pad, 128).cpu().numpy()
            b = augment.vnl_flip_pad_crop(torch.from_numpy(B).cuda(), flip, crop, pad, -1).cpu().numpy()
            assert a.shape == a_ref.shape and np.array_equal(a, a_ref), (trial, phase)
            assert b.shape == b_ref.shape and np.array_equal(b, b_ref), (trial, phase)
    with pytest.raises(NotImplementedError, match="cv2"):
        augment.vnl_preprocess(torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), "train")
