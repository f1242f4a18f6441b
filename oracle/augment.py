"""CPU oracle of the reference's input pipeline (modules/base_module.py:234-284: train_preprocess / val_preprocess) — TEST
INFRASTRUCTURE ONLY.

The reference composes torchvision transforms over PIL images.  torchvision is absent from this image (and from
/root/reference); Pillow — the library that does the arithmetic — is here (12.2.0), so this oracle calls PIL itself for every
image operation and restates only torchvision's thin wrappers from their public definitions, each cited below:
  transforms.ToPILImage   (functional.to_pil_image: a float tensor is `pic.mul(255).byte()`, C x H x W -> H x W x C, mode L / RGB)
  transforms.Resize(int)  (functional.resize / _compute_resized_output_size: the SHORTER edge becomes `size`, the longer one
                           int(size * long / short); unchanged if it already matches; PIL `resize(..., BILINEAR)`)
  TF.rotate               (PIL `rotate(angle, NEAREST, expand=False, center=None, fillcolor=None)`)
  transforms.CenterCrop   (functional.center_crop: top = int(round((h - th) / 2.0)), left likewise; PIL `crop`)
  TF.hflip                (PIL `transpose(FLIP_LEFT_RIGHT)`)
  TF.to_tensor            (an H x W[x C] float32 ndarray -> C x H x W tensor, no rescaling)
Parity is therefore pinned on PIL's own output; the wrapper semantics are restatements ("parity unpinned" for those five
one-liners only).

Also here: `pil_coeffs` / `resample_u8` / `affine_nearest_u8`, plain-numpy restatements of Pillow's 8-bit resampling and
nearest-neighbour affine transform (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc; Geometry.c: affine_fixed; Image.rotate's matrix) — the arithmetic the HIP
kernels implement, checked against PIL itself in tests/test_augment_cpu.py.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


# ---------------------------------------------------------------------------------------------- torchvision wrappers over PIL
def to_pil(pic):
    from PIL import Image
    if isinstance(pic, torch.Tensor):
        if pic.is_floating_point():
            pic = pic.mul(255).byte()
        arr = np.transpose(pic.cpu().numpy(), (1, 2, 0))
    else:
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
    assert arr.dtype == np.uint8, "this oracle covers the 8-bit modes the reference's pipeline produces"
    return Image.fromarray(arr[:, :, 0], mode="L") if arr.shape[2] == 1 else Image.fromarray(arr, mode="RGB")


def resized_size(w, h, size):
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def resize(img, size):
    from PIL import Image
    w, h = img.size
    nw, nh = resized_size(w, h, size)
    return img if (w, h) == (nw, nh) else img.resize((nw, nh), Image.BILINEAR)


def rotate(img, angle):
    from PIL import Image
    return img.rotate(angle, Image.NEAREST, False, None)


def center_crop(img, out_hw):
    w, h = img.size
    th, tw = out_hw
    assert th <= h and tw <= w, "center_crop pads smaller images in torchvision; the reference's sizes never need it"
    top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    return img.crop((left, top, left + tw, top + th))


def hflip(img):
    from PIL import Image
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def to_tensor_div255(img):
    a = np.array(img, dtype=np.float32) / 255.0
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose((2, 0, 1))))


def draw_train_params():
    """The three draws of train_preprocess, in its order (base_module.py:235,247,259)."""
    s = np.random.uniform(1, 1.5)
    angle = np.random.uniform(-5, 5)
    flip = np.random.uniform(0, 1) > 0.5
    return s, angle, flip


def train_preprocess(rgb, depth, resize_to, output_size, params=None):
    """base_module.py:234-265.  rgb: C x H x W float tensor in [0, 1] or H x W x 3 uint8; depth: an iterable of 1 x H x W float
    tensors.  params: (s, angle, flip) or None to draw them from the global numpy RNG as the reference does."""
    s, angle, flip = params if params is not None else draw_train_params()
    depth = [d / s for d in depth]
    imgs = [to_pil(rgb)] + [to_pil(d) for d in depth]
    imgs = [resize(i, resize_to) for i in imgs]
    imgs = [rotate(i, angle) for i in imgs]
    imgs = [resize(i, int(resize_to * s)) for i in imgs]
    imgs = [center_crop(i, output_size) for i in imgs]
    if flip:
        imgs = [hflip(i) for i in imgs]
    out = [to_tensor_div255(i) for i in imgs]
    return out[0], torch.cat(out[1:], dim=0)


def val_preprocess(rgb, depth, resize_to, output_size):
    """base_module.py:267-281."""
    imgs = [to_pil(rgb)] + [to_pil(d) for d in depth]
    imgs = [center_crop(resize(i, resize_to), output_size) for i in imgs]
    out = [to_tensor_div255(i) for i in imgs]
    return out[0], torch.cat(out[1:], dim=0)


# ---------------------------------------------------------------------------------------------- modules/bts.py:154-217
def bts_draw_train_params(w, h, output_size):
    """The draws of BtsModule.train_preprocess in its order: transforms.RandomRotation.get_params([-2.5, 2.5]) (torch's global
    generator: `torch.empty(1).uniform_(a, b).item()`), np.random.choice of the resize target, transforms.RandomCrop.get_params
    on the RESIZED image (two torch.randint draws, none if the sizes already match) and the flip (np.random.uniform).
    (w, h): the sample's size before the 5 % margin crop."""
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
    """modules/bts.py:154-199: 5 % margin crop (PIL rounds the float box), rotation, random resize, random crop, flip, / 255."""
    imgs = [to_pil(rgb)] + [to_pil(d) for d in depth]
    w, h = imgs[0].size
    box = (w * 0.05, h * 0.05, w * (1.0 - 0.05), h * (1.0 - 0.05))
    angle, size, i, j, flip = params if params is not None else bts_draw_train_params(w, h, output_size)
    imgs = [im.crop(box) for im in imgs]
    imgs = [rotate(im, angle) for im in imgs]
    imgs = [resize(im, size) for im in imgs]
    th, tw = output_size
    imgs = [im.crop((j, i, j + tw, i + th)) for im in imgs]
    if flip:
        imgs = [hflip(im) for im in imgs]
    out = [to_tensor_div255(im) for im in imgs]
    return out[0], torch.cat(out[1:], dim=0)


# ---------------------------------------------------------------------------------------------- Pillow's arithmetic, restated
def pil_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear ("triangle", support 1) filter over the whole axis.
    -> (bounds int32 [out][2] = (first input index, count), kk int32 [out][ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0.0 else v
            w = 1.0 - v if v < 1.0 else 0.0
            k[x] = w
            ww += w
        if ww != 0.0:
            k[:xmax] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + k[x] * (1 << PRECISION_BITS)) if k[x] < 0 else int(0.5 + k[x] * (1 << PRECISION_BITS))
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_u8(a, out_w, out_h):
    """ImagingResample for 8-bit images: the horizontal pass first (only over the rows the vertical pass will read), each pass
    rounding to uint8.  a: H x W x C uint8."""
    H, W, C = a.shape
    need_h, need_v = out_w != W, out_h != H
    bh, kh = pil_coeffs(W, out_w) if need_h else (None, None)
    bv, kv = pil_coeffs(H, out_h) if need_v else (None, None)
    y0 = 0
    if need_h:
        if need_v:
            y0, y1 = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
        else:
            y0, y1 = 0, H
        src = a[y0:y1].astype(np.int64)
        tmp = np.zeros((y1 - y0, out_w, C), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = bh[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], kh[xx, :n].astype(np.int64), axes=([1], [0]))
            tmp[:, xx, :] = _clip8(acc)
        a = tmp
    if need_v:
        src = a.astype(np.int64)
        out = np.zeros((out_h, a.shape[1], C), dtype=np.uint8)
        for yy in range(out_h):
            r0, n = bv[yy]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kv[yy, :n].astype(np.int64), src[r0 - y0:r0 - y0 + n], axes=([0], [0]))
            out[yy] = _clip8(acc)
        a = out
    return a


def rotate_matrix(w, h, angle):
    """Image.rotate's affine matrix (output -> input coordinates) for expand=False, center=None."""
    angle = angle % 360.0
    cx, cy = w / 2.0, h / 2.0
    rad = -math.radians(angle)
    m = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    return m


def affine_fixed_coeffs(m):
    """Geometry.c affine_fixed: 16.16 fixed-point coefficients, the half-pixel offset folded into the constant terms."""
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def affine_nearest_u8(a, m):
    """ImagingTransformAffine, nearest filter, fill 0.  a: H x W x C uint8; m: the 6 affine coefficients."""
    H, W, C = a.shape
    a0, a1, a2, a3, a4, a5 = affine_fixed_coeffs(m)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.int64)
    xin = (a2 + a1 * ys + a0 * xs) >> 16
    yin = (a5 + a4 * ys + a3 * xs) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.zeros_like(a)
    out[ok] = a[yin[ok], xin[ok]]
    return out


def rotate_u8(a, angle):
    """Image.rotate(angle, NEAREST, expand=False) including its shortcuts for multiples of 90 degrees."""
    ang = angle % 360.0
    H, W, _ = a.shape
    if ang == 0:
        return a.copy()
    if ang == 180:
        return a[::-1, ::-1].copy()
    if ang in (90, 270) and W == H:
        return np.rot90(a, 1 if ang == 90 else 3).copy()
    return affine_nearest_u8(a, rotate_matrix(W, H, angle))


# ---------------------------------------------------------------------------------------------- modules/midas.py:107-150
MIDAS_MEAN, MIDAS_STD = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])


def midas_default_transform(img_u8):
    """The hub transform MidasModule applies last (modules/midas.py:12,125,145: torch.hub 'intel-isl/MiDaS' transforms
    .default_transform), restated from its published definition -- the hub repository and cv2 are absent from the image, so this
    step is pinned by definition only: {"image": img / 255.0} -> Resize(384, 384, keep_aspect_ratio=True, ensure_multiple_of=32,
    resize_method="upper_bound", cv2.INTER_CUBIC) -> NormalizeImage(mean, std) -> PrepareForNet (transpose to CHW, float32) ->
    torch.from_numpy(...).unsqueeze(0).  On the 384 x 384 crops train_preprocess / val_preprocess produce, Resize computes a scale
    of exactly 1 and cv2.resize to the source's own size is a copy; any other size is refused here."""
    a = np.asarray(img_u8)
    assert a.dtype == np.uint8 and a.shape[:2] == (384, 384), "this restatement covers the 384 x 384 crops of train / val preprocess"
    x = a / 255.0
    x = (x - MIDAS_MEAN) / MIDAS_STD
    x = np.ascontiguousarray(np.transpose(x, (2, 0, 1))).astype(np.float32)
    return torch.from_numpy(x).unsqueeze(0)


def midas_draw_train_params(w, h):
    size = int(np.random.randint(384, 720))
    rw, rh = resized_size(w, h, size)
    if (rw, rh) == (384, 384):
        i = j = 0
    else:
        i = int(torch.randint(0, rh - 384 + 1, size=(1,)).item())
        j = int(torch.randint(0, rw - 384 + 1, size=(1,)).item())
    flip = np.random.uniform(0, 1) > 0.5
    return size, i, j, flip


def midas_train_preprocess(rgb, depth, params=None):
    """modules/midas.py:107-130 over PIL."""
    imgs = [to_pil(rgb)] + [to_pil(d) for d in depth]
    w, h = imgs[0].size
    size, i, j, flip = params if params is not None else midas_draw_train_params(w, h)
    imgs = [resize(im, size) for im in imgs]
    imgs = [im.crop((j, i, j + 384, i + 384)) for im in imgs]
    if flip:
        imgs = [hflip(im) for im in imgs]
    return midas_default_transform(np.array(imgs[0], dtype=np.uint8)).squeeze(0), torch.cat([to_tensor_div255(im) for im in imgs[1:]], dim=0)


def midas_val_preprocess(rgb, depth):
    """modules/midas.py:132-150 over PIL."""
    imgs = [center_crop(resize(im, 384), (384, 384)) for im in [to_pil(rgb)] + [to_pil(d) for d in depth]]
    return midas_default_transform(np.array(imgs[0], dtype=np.uint8)).squeeze(0), torch.cat([to_tensor_div255(im) for im in imgs[1:]], dim=0)


# ---------------------------------------------------------------------------------------------- modules/vnl.py:32-78 (the numpy half)
def vnl_draw_params(phase, uniform_size, crop_size=(385, 385)):
    """set_flip_pad_reshape_crop, modules/vnl.py:32-57."""
    flip_prob = np.random.uniform(0.0, 1.0)
    flip_flg = True if flip_prob > 0.5 and 'train' in phase else False
    raw_size = np.array([crop_size[1], 416, 448, 480, 512])
    size_index = np.random.randint(0, len(raw_size)) if 'train' in phase else len(raw_size) - 1
    pad_height = raw_size[size_index] - uniform_size[0] if raw_size[size_index] > uniform_size[0] else 0
    pad = [int(pad_height), 0, 0, 0]
    crop_height = crop_width = int(raw_size[size_index])
    start_x = np.random.randint(0, int(uniform_size[1] - crop_width) + 1)
    start_y = 0 if pad_height != 0 else np.random.randint(0, int(uniform_size[0] - crop_height) + 1)
    return flip_flg, [int(start_x), int(start_y), crop_height, crop_width], pad, float(crop_size[1] / crop_width)


def vnl_flip_pad_crop(img, flip, crop_size, pad, pad_value=0):
    """flip_pad_reshape_crop, modules/vnl.py:59-75, up to (not including) its cv2.resize."""
    if flip:
        img = np.flip(img, axis=1)
    if len(img.shape) == 3:
        img_pad = np.pad(img, ((pad[0], pad[1]), (pad[2], pad[3]), (0, 0)), 'constant', constant_values=(pad_value, pad_value))
    else:
        img_pad = np.pad(img, ((pad[0], pad[1]), (pad[2], pad[3])), 'constant', constant_values=(pad_value, pad_value))
    return img_pad[crop_size[1]:crop_size[1] + crop_size[3], crop_size[0]:crop_size[0] + crop_size[2]]
