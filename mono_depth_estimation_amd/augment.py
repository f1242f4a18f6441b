"""The reference's input pipeline on the device: `train_preprocess` / `val_preprocess` of
modules/base_module.py:234-284 — ToPILImage, Resize, ±5° rotation, Resize, CenterCrop, horizontal flip, to_tensor — for a sample
that is already in HBM, with Pillow's 8-bit arithmetic reproduced bit for bit by the kernels of csrc/augment.hip (PIL's
BILINEAR resize = two separable passes of 22-bit fixed-point triangle filters with antialiasing, each rounded to uint8;
PIL's rotate(NEAREST) = a 16.16 fixed-point affine walk).  The reference runs this per sample on CPU workers; at ~1 000
images/s per GPU eight workers no longer keep up.

The random draws stay on the host and in the reference's order (`draw_train_params`: np.random.uniform three times), so a seeded
run augments every sample exactly as the reference does.  No CPU fallback: tensors must live on the GPU.

    rgb, depth = augment.train_preprocess(rgb, depth, resize_to=250, output_size=(240, 320))        # FCRNModule's sizes
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream, check

PRECISION_BITS = 32 - 8 - 2
_coeff_cache, _lut_cache = {}, {}


def _coeffs(in_size, out_size, device):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the bilinear filter, in double on the
    host, cached per (in, out, device) -> (bounds int32 [out][2], kk int32 [out][ksize], ksize, first row, row count)."""
    key = (in_size, out_size, str(device))
    hit = _coeff_cache.get(key)
    if hit is not None:
        return hit
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - v if v < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + k[x] * (1 << PRECISION_BITS)) if k[x] < 0 else int(0.5 + k[x] * (1 << PRECISION_BITS))
    out = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize, int(bounds[0, 0]), int(bounds[-1, 0] + bounds[-1, 1] - bounds[0, 0]))
    _coeff_cache[key] = out
    return out


def _lut(device):
    t = _lut_cache.get(str(device))
    if t is None:
        t = torch.from_numpy(np.arange(256, dtype=np.float32) / 255.0).to(device)       # np.array(img, float32) / 255.0, entry by entry
        _lut_cache[str(device)] = t
    return t


def resized_size(w, h, size):
    """torchvision functional.resize with an int: the shorter edge becomes `size`."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_long = int(size * long_ / short)
    return (size, new_long) if w <= h else (new_long, size)


def to_u8(t, divisor=1.0):
    """C x H x W float (or H x W x C uint8, returned as is) -> H x W x C uint8 as transforms.ToPILImage does."""
    if t.dtype == torch.uint8:
        return t.contiguous()
    Cc, H, W = t.shape
    out = torch.empty(H, W, Cc, dtype=torch.uint8, device=t.device)
    check(_lib.load().mde_aug_to_u8(_p(t.contiguous().float()), Cc, H, W, float(np.float32(divisor)), _p(out), _stream()), "mde_aug_to_u8")
    return out


def resize_u8(img, out_w, out_h):
    """PIL Image.resize((out_w, out_h), BILINEAR) of an H x W x C uint8 image."""
    H, W, Cc = img.shape
    if (W, H) == (out_w, out_h):
        return img
    hb = hk = vb = vk = None
    hks = vks = 0
    y0, rows = 0, H
    if out_w != W:
        hb, hk, hks, _, _ = _coeffs(W, out_w, img.device)
    if out_h != H:
        vb, vk, vks, y0v, rowsv = _coeffs(H, out_h, img.device)
        if hb is not None:
            y0, rows = y0v, rowsv
    tmp = torch.empty(rows, out_w, Cc, dtype=torch.uint8, device=img.device) if (hb is not None and vb is not None) else None
    out = torch.empty(out_h, out_w, Cc, dtype=torch.uint8, device=img.device)
    check(_lib.load().mde_aug_resample_u8(_p(img), H, W, Cc, _p(hb), _p(hk), hks, out_w, _p(vb), _p(vk), vks, out_h, y0, rows, _p(tmp), _p(out),
                                          _stream()), "mde_aug_resample_u8")
    return out


def rotate_u8(img, angle):
    """PIL Image.rotate(angle, NEAREST, expand=False) of an H x W x C uint8 image."""
    H, W, Cc = img.shape
    ang = angle % 360.0
    if ang == 0:                                         # Image.rotate's shortcuts
        return img
    if ang == 180:
        return img.flip(0, 1).contiguous()
    if ang in (90, 270) and W == H:
        return torch.rot90(img, 1 if ang == 90 else 3, (0, 1)).contiguous()
    cx, cy = W / 2.0, H / 2.0
    rad = -math.radians(ang)
    m = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    coef = (C.c_int32 * 6)(fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))
    out = torch.empty_like(img)
    check(_lib.load().mde_aug_affine_nearest_u8(_p(img), H, W, Cc, C.cast(coef, C.c_void_p), _p(out), _stream()), "mde_aug_affine_nearest_u8")
    return out


def crop_flip_to_float(img, output_size, flip, top=None, left=None, lut=None):
    """CenterCrop(output_size) (or the crop at (top, left)) -> hflip (if flip) -> np.array(img, float32) / 255.0 -> to_tensor:
    C x oh x ow float32.  lut: a [C][256] table of the output values instead of v / 255 (midas_lut)."""
    H, W, Cc = img.shape
    oh, ow = output_size
    if oh > H or ow > W:
        raise ValueError("crop %s of a %dx%d image (torchvision would pad a center crop / refuse a random one; the reference's sizes "
                         "never need it)" % (output_size, H, W))
    if top is None:
        top, left = int(round((H - oh) / 2.0)), int(round((W - ow) / 2.0))
    out = torch.empty(Cc, oh, ow, device=img.device)
    if lut is not None:
        assert lut.shape == (Cc, 256) and lut.dtype == torch.float32 and lut.is_contiguous()
        check(_lib.load().mde_aug_crop_flip_to_float_c(_p(img), H, W, Cc, top, left, oh, ow, int(bool(flip)), _p(lut), 256, _p(out), _stream()),
              "mde_aug_crop_flip_to_float_c")
        return out
    check(_lib.load().mde_aug_crop_flip_to_float(_p(img), H, W, Cc, top, left, oh, ow, int(bool(flip)), _p(_lut(img.device)), _p(out), _stream()),
          "mde_aug_crop_flip_to_float")
    return out


def draw_train_params():
    """The three draws of train_preprocess, in its order (base_module.py:235,247,259), from the global numpy RNG."""
    s = np.random.uniform(1, 1.5)
    angle = np.random.uniform(-5, 5)
    flip = np.random.uniform(0, 1) > 0.5
    return s, angle, flip


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("mono_depth_estimation_amd.augment runs on MI355X only (tensor on %s); no CPU fallback" % t.device)


def _stack_depth(depth, divisor):
    layers = [d.reshape(1, d.shape[-2], d.shape[-1]) for d in depth]
    return to_u8(torch.cat(layers, 0) if len(layers) > 1 else layers[0], divisor)


def train_preprocess(rgb, depth, resize_to, output_size, params=None):
    """base_module.py:234-265 for one sample on the device.  rgb: 3 x H x W float in [0, 1] (or H x W x 3 uint8); depth: a sequence
    of 1 x H x W float layers.  -> (3 x oh x ow, D x oh x ow) float32.  params = (s, angle, flip) or None to draw them."""
    _need_gpu(rgb)
    s, angle, flip = params if params is not None else draw_train_params()
    outs = []
    for img in (to_u8(rgb), _stack_depth(list(depth), s)):        # the D depth layers travel as one D-channel image
        H, W, _ = img.shape
        w1, h1 = resized_size(W, H, resize_to)
        img = rotate_u8(resize_u8(img, w1, h1), angle)
        w2, h2 = resized_size(w1, h1, int(resize_to * s))
        outs.append(crop_flip_to_float(resize_u8(img, w2, h2), output_size, flip))
    return outs[0], outs[1]


def val_preprocess(rgb, depth, resize_to, output_size):
    """base_module.py:267-281 for one sample on the device."""
    _need_gpu(rgb)
    outs = []
    for img in (to_u8(rgb), _stack_depth(list(depth), 1.0)):
        H, W, _ = img.shape
        w1, h1 = resized_size(W, H, resize_to)
        outs.append(crop_flip_to_float(resize_u8(img, w1, h1), output_size, False))
    return outs[0], outs[1]


# ---------------------------------------------------------------------------------------------- modules/bts.py:154-217
def bts_draw_train_params(w, h, output_size):
    """The draws of BtsModule.train_preprocess in its order: RandomRotation.get_params (torch's global generator), np.random.choice
    of the resize target, RandomCrop.get_params on the resized image (two torch.randint draws) and the flip (np.random.uniform)."""
    cw = int(round(w * (1.0 - 0.05))) - int(round(w * 0.05))
    ch = int(round(h * (1.0 - 0.05))) - int(round(h * 0.05))
    angle = float(torch.empty(1).uniform_(-2.5, 2.5).item())
    size = int(np.random.choice([512, 518, 550, 600, 650, 720]))
    rw, rh = resized_size(cw, ch, size)
    th, tw = output_size
    if rh < th or rw < tw:
        raise ValueError("Required crop size %s is larger than input image size %s" % ((th, tw), (rh, rw)))
    if (rw, rh) == (tw, th):
        i = j = 0
    else:
        i = int(torch.randint(0, rh - th + 1, size=(1,)).item())
        j = int(torch.randint(0, rw - tw + 1, size=(1,)).item())
    flip = np.random.uniform(0, 1) > 0.5
    return angle, size, i, j, flip


def bts_train_preprocess(rgb, depth, output_size, params=None):
    """modules/bts.py:154-199 for one sample on the device: 5 % margin crop (PIL rounds the float box half-to-even), ±2.5° rotation,
    random resize, random crop, flip, / 255.  params = (angle, size, top, left, flip) or None to draw them."""
    _need_gpu(rgb)
    imgs = [to_u8(rgb), _stack_depth(list(depth), 1.0)]
    H, W, _ = imgs[0].shape
    l, t, r, b = (int(round(v)) for v in (W * 0.05, H * 0.05, W * (1.0 - 0.05), H * (1.0 - 0.05)))      # Image.crop: map(int, map(round, box))
    angle, size, i, j, flip = params if params is not None else bts_draw_train_params(W, H, output_size)
    outs = []
    for img in imgs:
        img = rotate_u8(img[t:b, l:r].contiguous(), angle)
        h1, w1, _ = img.shape
        w2, h2 = resized_size(w1, h1, size)
        outs.append(crop_flip_to_float(resize_u8(img, w2, h2), output_size, flip, top=i, left=j))
    return outs[0], outs[1]


bts_val_preprocess = val_preprocess          # modules/bts.py:202-217: the same operations as the base module's (Resize, CenterCrop, / 255)


# ---------------------------------------------------------------------------------------------- modules/midas.py:107-150
MIDAS_SIZE = 384
MIDAS_MEAN, MIDAS_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
_midas_lut_cache = {}


def midas_lut(device):
    """The hub's `default_transform` (modules/midas.py:12: torch.hub 'intel-isl/MiDaS' transforms) on an image that already has
    its 384 x 384 size -- what train_preprocess / val_preprocess hand it -- as a [3][256] table.  Its published definition:
    img / 255.0 (float64), Resize(384, 384, keep_aspect_ratio, multiple of 32, cv2.INTER_CUBIC) -- the identity at that size (cv2.resize
    copies when source and destination sizes agree) --, NormalizeImage(mean, std) in float64, PrepareForNet (HWC -> CHW,
    float32).  cv2 and the hub repository are absent from the image: the table is this restatement, pinned by definition."""
    t = _midas_lut_cache.get(str(device))
    if t is None:
        v = np.arange(256, dtype=np.uint8)
        rows = [(((v / 255.0) - m) / s).astype(np.float32) for m, s in zip(MIDAS_MEAN, MIDAS_STD)]
        t = _midas_lut_cache[str(device)] = torch.from_numpy(np.stack(rows)).to(device).contiguous()
    return t


def midas_draw_train_params(w, h):
    """The draws of MidasModule.train_preprocess in its order: np.random.randint(384, 720) (the resize target),
    transforms.RandomCrop.get_params on the RESIZED image (two torch.randint draws, none if it is 384 x 384 already), the flip
    (np.random.uniform)."""
    size = int(np.random.randint(384, 720))
    rw, rh = resized_size(w, h, size)
    th = tw = MIDAS_SIZE
    if rh + 1 < th or rw + 1 < tw:
        raise ValueError("Required crop size %s is larger then input image size %s" % ((th, tw), (rh, rw)))
    if (rw, rh) == (tw, th):
        i = j = 0
    else:
        i = int(torch.randint(0, rh - th + 1, size=(1,)).item())
        j = int(torch.randint(0, rw - tw + 1, size=(1,)).item())
    flip = np.random.uniform(0, 1) > 0.5
    return size, i, j, flip


def midas_train_preprocess(rgb, depth, params=None):
    """modules/midas.py:107-130 for one sample on the device: random Resize (shorter edge 384 ... 719), RandomCrop(384, 384), flip,
    then the hub transform on the colour image (midas_lut) and / 255 on the depth layers.  params = (size, top, left, flip)."""
    _need_gpu(rgb)
    imgs = [to_u8(rgb), _stack_depth(list(depth), 1.0)]
    H, W, _ = imgs[0].shape
    size, i, j, flip = params if params is not None else midas_draw_train_params(W, H)
    w1, h1 = resized_size(W, H, size)
    out = (MIDAS_SIZE, MIDAS_SIZE)
    return (crop_flip_to_float(resize_u8(imgs[0], w1, h1), out, flip, top=i, left=j, lut=midas_lut(rgb.device)),
            crop_flip_to_float(resize_u8(imgs[1], w1, h1), out, flip, top=i, left=j))


def midas_val_preprocess(rgb, depth):
    """modules/midas.py:132-150: Resize(384), CenterCrop((384, 384)), the hub transform / the division by 255."""
    _need_gpu(rgb)
    imgs = [to_u8(rgb), _stack_depth(list(depth), 1.0)]
    H, W, _ = imgs[0].shape
    w1, h1 = resized_size(W, H, MIDAS_SIZE)
    out = (MIDAS_SIZE, MIDAS_SIZE)
    return (crop_flip_to_float(resize_u8(imgs[0], w1, h1), out, False, lut=midas_lut(rgb.device)),
            crop_flip_to_float(resize_u8(imgs[1], w1, h1), out, False))


def midas_test_preprocess(rgb, depth):
    """modules/midas.py:152-184 pads with cv2.copyMakeBorder and resizes 640 x 640 -> 384 x 384 with cv2.resize (INTER_LINEAR, OpenCV's
    own 11-bit fixed-point weights for 8-bit images): cv2 is absent from this image, so there is nothing to pin that arithmetic
    to, and an unpinned restatement of an integer resampler is not shipped."""
    raise NotImplementedError("MidasModule.test_preprocess resizes with cv2.resize; OpenCV is not available to pin a device "
                              "version against (mono_depth_estimation_amd.augment covers train_preprocess / val_preprocess)")


# ---------------------------------------------------------------------------------------------- modules/vnl.py:32-138
VNL_CROP_SIZE = (385, 385)        # modules/vnl.py CROP_SIZE


def vnl_draw_params(phase, uniform_size, crop_size=VNL_CROP_SIZE):
    """set_flip_pad_reshape_crop (modules/vnl.py:32-57), draw for draw: np.random.uniform (flip), np.random.randint (crop size index,
    'train' only), np.random.randint (start_x), np.random.randint (start_y, only without padding).
    -> (flip, [start_x, start_y, crop_height, crop_width], [pad_up, 0, 0, 0], resize_ratio)."""
    flip_prob = np.random.uniform(0.0, 1.0)
    flip = bool(flip_prob > 0.5 and 'train' in phase)
    raw_size = np.array([crop_size[1], 416, 448, 480, 512])
    idx = np.random.randint(0, len(raw_size)) if 'train' in phase else len(raw_size) - 1
    pad_height = int(raw_size[idx] - uniform_size[0]) if raw_size[idx] > uniform_size[0] else 0
    ch = cw = int(raw_size[idx])
    start_x = int(np.random.randint(0, int(uniform_size[1] - cw) + 1))
    start_y = 0 if pad_height != 0 else int(np.random.randint(0, int(uniform_size[0] - ch) + 1))
    return flip, [start_x, start_y, ch, cw], [pad_height, 0, 0, 0], float(crop_size[1] / cw)


def vnl_flip_pad_crop(img, flip, crop, pad, pad_value=0):
    """flip_pad_reshape_crop (modules/vnl.py:59-78) UP TO its last line: np.flip(img, axis=1), np.pad(..., 'constant', pad_value),
    the crop [start_y : start_y + crop[3], start_x : start_x + crop[2]] -- exact data movement, on the device.  img: H x W x C uint8
    or H x W float32.  The last line, cv2.resize(img_crop, CROP_SIZE, INTER_LINEAR), is NOT here: OpenCV is absent from the image
    and its 8-bit resampler (11-bit fixed-point weights) has nothing to be pinned to; vnl_preprocess says so when called."""
    _need_gpu(img)
    squeeze = img.dim() == 2
    a = (img.unsqueeze(-1) if squeeze else img).contiguous()
    H, W, Cc = a.shape
    assert a.dtype in (torch.uint8, torch.float32), a.dtype
    oh, ow = int(crop[3]), int(crop[2])
    out = torch.empty(oh, ow, Cc, dtype=a.dtype, device=a.device)
    fill = (C.c_uint8(int(pad_value)) if a.dtype == torch.uint8 else C.c_float(float(pad_value)))
    check(_lib.load().mde_aug_flip_pad_crop(_p(a), a.element_size(), H, W, Cc, int(bool(flip)), int(pad[0]), int(pad[2]), int(crop[1]), int(crop[0]),
                                            oh, ow, C.cast(C.pointer(fill), C.c_void_p), _p(out), _stream()), "mde_aug_flip_pad_crop")
    return out.squeeze(-1) if squeeze else out


def vnl_preprocess(A, B, phase):
    """modules/vnl.py:86-111 `preprocess`: every path ends in cv2.resize (INTER_LINEAR to CROP_SIZE; a further cv2.resize to height
    512 in front when the depth map is not 512 rows tall).  The numpy half is vnl_draw_params + vnl_flip_pad_crop; the resize has
    no pinnable counterpart here (OpenCV absent), so the composed pipeline is not offered."""
    raise NotImplementedError("VNL's preprocess resizes with cv2.resize (INTER_LINEAR); OpenCV is not available to pin a device version "
                              "against -- vnl_draw_params / vnl_flip_pad_crop cover its draws, flip, padding and crop")
