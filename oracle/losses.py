"""CPU fp32 oracle of the depth losses — TEST INFRASTRUCTURE ONLY.

Functional restatements of reference criteria.py.  Each function cites the lines it
follows; latent defects of the reference that change the numbers are reproduced on
purpose and flagged, because parity is with the reference as it runs.
"""
import torch


# ---------------------------------------------------------------- B1  criteria.py:724-732
def silog(est, gt, variance_focus=0.85):
    """10*sqrt(mean(d^2) - lambda*mean(d)^2), d = log est - log gt over gt > 1e-2.
    Global over the whole batch; no guard for an empty mask (NaN, like the reference)."""
    m = gt > 1e-2
    d = torch.log(est[m]) - torch.log(gt[m])
    return 10.0 * torch.sqrt((d * d).mean() - variance_focus * d.mean() ** 2)


# ---------------------------------------------------------------- B6  criteria.py:17-64
def masked_depth(pred, target):
    """Eigen linear-space scale-invariant term + masked forward-difference gradient MSE."""
    n = target.shape[0]
    mask = (target > 0).to(torch.float32)
    d = ((pred - target) * mask).reshape(n, -1)
    nv = mask.reshape(n, -1).sum(1)
    data = ((nv * (d * d).sum(1)).sum() - 0.5 * (d.sum(1) ** 2).sum()) / (nv * nv).sum()
    if pred.ndim == 4:
        pred, target, mask = pred[:, 0], target[:, 0], mask[:, 0]
    gi = (pred[:, 1:] - pred[:, :-1]) - (target[:, 1:] - target[:, :-1])
    gj = (pred[:, :, 1:] - pred[:, :, :-1]) - (target[:, :, 1:] - target[:, :, :-1])
    mi = (mask[:, 1:] * mask[:, :-1])
    mj = (mask[:, :, 1:] * mask[:, :, :-1])
    return data + (mi * gi * gi).sum() / mi.sum() + (mj * gj * gj).sum() / mj.sum()


# ---------------------------------------------------------------- B9  criteria.py:67-133
def masked_mse(pred, target):
    v = target > 0
    return ((target - pred)[v] ** 2).mean()


def masked_l1(pred, target):
    v = target > 0
    return (target - pred)[v].abs().mean()


def berhu(pred, target):
    """Reverse Huber as the reference writes it: c = 0.2*max(pred-target) over ALL
    pixels (valid or not), loss = mean(cat(|d|, |d|[|d|>c]^2)) (criteria.py:118-131)."""
    c = 0.2 * (pred - target).max()
    d = (target - pred)[target > 0].abs()
    return torch.cat((d, d[d > c] ** 2)).mean()


# ---------------------------------------------------------------- B4  criteria.py:154-176
def scale_and_shift(pred, target, mask):
    """Per-image closed-form least squares for (scale, shift); zero where det == 0."""
    a00 = (mask * pred * pred).sum((1, 2))
    a01 = (mask * pred).sum((1, 2))
    a11 = mask.sum((1, 2))
    b0 = (mask * pred * target).sum((1, 2))
    b1 = (mask * target).sum((1, 2))
    det = a00 * a11 - a01 * a01
    ok = det != 0
    safe = torch.where(ok, det, torch.ones_like(det))
    scale = torch.where(ok, (a11 * b0 - a01 * b1) / safe, torch.zeros_like(det))
    shift = torch.where(ok, (-a01 * b0 + a00 * b1) / safe, torch.zeros_like(det))
    return scale, shift


# ---------------------------------------------------------------- criteria.py:179-199
def _reduce(image_loss, M, batch_based):
    if batch_based:
        div = M.sum()
        return image_loss.sum() / div if div != 0 else image_loss.sum() * 0
    ok = M != 0
    per = torch.where(ok, image_loss / torch.where(ok, M, torch.ones_like(M)), image_loss)
    return per.mean()


# ---------------------------------------------------------------- criteria.py:201-224
def data_mse(pred, target, mask, batch_based=True):
    """NB reference quirk: the per-pixel map (not per-image sums) is handed to the
    reduction together with 2*M, so 'image-based' would index a 3-D map with a 1-D mask
    (shape error in the reference); only batch-based is defined and is what the modules
    use (midas.py:29-37)."""
    if not batch_based:
        raise NotImplementedError("reference mse_loss is only well-formed with batch-based reduction")
    M = mask.sum((1, 2))
    r = pred - target
    return _reduce(mask * r * r, 2 * M, True)


def data_l1(pred, target, mask):
    M = mask.sum((1, 2))
    return _reduce((target - pred)[mask.bool()].abs(), 2 * M, True)


def data_trimmed_mae(pred, target, mask, trim=0.2):
    """Reference defect reproduced (criteria.py:214-216): ``torch.sort(...)[:k]`` slices the
    (values, indices) TUPLE, not the values -> nothing is trimmed; result == L1/(2M)."""
    M = mask.sum((1, 2))
    res = (pred - target)[mask.bool()].abs()
    vals, _ = torch.sort(res.view(-1))
    return _reduce(vals, 2 * M, True)


# ---------------------------------------------------------------- B3  criteria.py:227-244,283-303
def gradient_term(pred, target, mask, batch_based=True):
    M = mask.sum((1, 2))
    diff = mask * (pred - target)
    gx = (diff[:, :, 1:] - diff[:, :, :-1]).abs() * (mask[:, :, 1:] * mask[:, :, :-1])
    gy = (diff[:, 1:, :] - diff[:, :-1, :]).abs() * (mask[:, 1:, :] * mask[:, :-1, :])
    return _reduce(gx.sum((1, 2)) + gy.sum((1, 2)), M, batch_based)


def gradient_multiscale(pred, target, mask, scales=4, batch_based=True):
    total = 0
    for s in range(scales):
        k = 2 ** s
        total = total + gradient_term(pred[:, ::k, ::k], target[:, ::k, ::k], mask[:, ::k, ::k], batch_based)
    return total


# ---------------------------------------------------------------- criteria.py:306-332
def midas_loss(pred, target, alpha=0.5, scales=4, loss="ssimse", batch_based=True):
    if pred.ndim == 4:
        pred = pred.squeeze(1)
    if target.ndim == 4:
        target = target.squeeze(1)
    mask = (target > 0).to(torch.float32)
    if "ssi" in loss:
        s, t = scale_and_shift(pred, target, mask)
        pred = s.view(-1, 1, 1) * pred + t.view(-1, 1, 1)
    if "trim" in loss:
        total = data_trimmed_mae(pred, target, mask)
    elif "mse" in loss:
        total = data_mse(pred, target, mask, batch_based)
    elif "l1" in loss:
        total = data_l1(pred, target, mask)
    else:
        raise ValueError(loss)
    if alpha > 0:
        total = total + alpha * gradient_multiscale(pred, target, mask, scales, batch_based)
    return total


# ---------------------------------------------------------------- B5  criteria.py:135-152,335-363
def normalize_robust(x, mask):
    """Per-image: subtract the median of (mask*x) (torch.median = lower median, zeros of
    masked-out pixels included, as in the reference), divide by mean |.| over valid."""
    n = x.shape[0]
    cnt = mask.sum((1, 2))
    ok = cnt > 0
    med = torch.zeros_like(cnt)
    if ok.any():
        med[ok] = torch.median((mask[ok] * x[ok]).view(int(ok.sum()), -1), dim=1).values
    x = x - med.view(n, 1, 1)
    sq = (mask * x.abs()).sum((1, 2))
    s = torch.ones_like(cnt)
    s[ok] = torch.clamp(sq[ok] / cnt[ok], min=1e-6)
    return x / s.view(n, 1, 1)


def trimmed_procrustes(pred, target, alpha=0.5, scales=4, batch_based=True):
    if pred.ndim == 4:
        pred = pred.squeeze(1)
    if target.ndim == 4:
        target = target.squeeze(1)
    mask = (target > 0).to(torch.float32)
    p = normalize_robust(pred, mask)
    t = normalize_robust(target, mask)
    total = data_trimmed_mae(p, t, mask)
    if alpha > 0:
        total = total + alpha * gradient_multiscale(p, t, mask, scales, batch_based)
    return total


# ---------------------------------------------------------------- B8  criteria.py:839-863
def wcel(pred_logit, gt_bins, gt, weight):
    """Weighted cross-entropy over depth bins.  `weight` is the [C][C] matrix AFTER the
    row normalisation the reference's constructor applies (criteria.py:846-848); a bin
    label outside [0, C) selects an all-zero row (the one-hot comparison matches nothing),
    and the divisor counts gt > 0 over the depth map, not the labels."""
    C = pred_logit.shape[1]
    logp = torch.log_softmax(pred_logit, 1).permute(0, 2, 3, 1).reshape(-1, C)
    b = gt_bins.reshape(-1).to(torch.int64)
    inside = (b >= 0) & (b < C)
    rows = weight.to(torch.float32)[b.clamp(0, C - 1)] * inside.unsqueeze(1)
    return -(rows * logp).sum() / (gt > 0).sum().to(torch.float32)


def wcel_weight(C):
    """modules/vnl.py:162 followed by criteria.py:846-847 (float64 on the host, like numpy)."""
    import numpy as np
    i = np.arange(C, dtype=np.float64)
    w = np.exp(-0.2 * (i[None, :] - i[:, None]) ** 2)
    return torch.from_numpy(w / w.sum(1, keepdims=True))


# ---------------------------------------------------------------- B7  criteria.py:866-1045
def vnl_back_project(depth, fx, fy):
    """criteria.py:877-910: principal point = size // 2, x = (u - u0) * |d| / fx, z = d.
    depth [B,1,H,W] -> points [B,H,W,3]."""
    _, _, H, W = depth.shape
    u = (torch.arange(W, dtype=torch.float32) - float(W // 2)).view(1, 1, 1, W)
    v = (torch.arange(H, dtype=torch.float32) - float(H // 2)).view(1, 1, H, 1)
    x = u * depth.abs() / torch.tensor([fx], dtype=torch.float32)
    y = v * depth.abs() / torch.tensor([fy], dtype=torch.float32)
    return torch.cat([x, y, depth], 1).permute(0, 2, 3, 1)


def vnl_groups(pw, p123, W):
    """criteria.py:934-953: [B, n, 3 (xyz), 3 (point)] from three linear pixel index arrays."""
    pts = [pw[:, torch.as_tensor(p // W), torch.as_tensor(p % W), :] for p in p123]
    return torch.stack(pts, 3)


def vnl_filter(groups_gt, delta_cos=0.867, delta_diff=0.005, delta_z=0.0001):
    """criteria.py:955-988 with the thresholds select_points_groups actually passes
    (0.867 / 0.005, :996-1000 — the constructor's 0.01s are never used)."""
    d12 = groups_gt[..., 1] - groups_gt[..., 0]
    d13 = groups_gt[..., 2] - groups_gt[..., 0]
    d23 = groups_gt[..., 2] - groups_gt[..., 1]
    diff = torch.stack([d12, d13, d23], 3)                       # [B, n, xyz, 3 diffs]
    B, n = diff.shape[:2]
    q = diff.reshape(B * n, 3, 3).permute(0, 2, 1)               # [Bn, diff, xyz]
    qn = q.norm(2, dim=2)
    nm = qn.unsqueeze(2) * qn.unsqueeze(1)
    energy = torch.bmm(q, q.permute(0, 2, 1)) / (nm + 1e-8)
    energy = energy.reshape(B * n, 9)
    mask_cos = (((energy > delta_cos) | (energy < -delta_cos)).sum(1) > 3).view(B, n)
    mask_pad = (groups_gt[:, :, 2, :] > delta_z).sum(2) == 3
    near = [(diff[:, :, c, :].abs() < delta_diff).sum(2) > 0 for c in range(3)]
    ignore = (near[0] & near[1] & near[2]) | mask_cos
    return mask_pad & ~ignore


def vnl(gt_depth, pred_depth, p123, fx, fy, select=True):
    """Virtual-normal loss on GIVEN sample indices (the reference draws them from the global
    numpy RNG, criteria.py:912-932; p123 = three arrays of linear pixel indices).

    Reproduced on purpose: `pw_groups_pred[pw_groups_pred[:, :, 2, :] == 0] = 0.0001`
    (criteria.py:1004) indexes the first THREE dims [B, n, xyz] with a [B, n, point] mask, so a
    predicted point j with z == 0 overwrites coordinate j (x, y or z) of all three points."""
    W = gt_depth.shape[-1]
    g_gt = vnl_groups(vnl_back_project(gt_depth, fx, fy), p123, W)
    g_dt = vnl_groups(vnl_back_project(pred_depth, fx, fy), p123, W)
    keep = vnl_filter(g_gt)
    zero = g_dt[:, :, 2, :] == 0                                 # [B, n, point]
    g_dt = torch.where(zero.unsqueeze(3), torch.full_like(g_dt, 0.0001), g_dt)
    gt, dt = g_gt[keep], g_dt[keep]                              # [M, xyz, point]
    n_gt = torch.cross(gt[..., 1] - gt[..., 0], gt[..., 2] - gt[..., 0], dim=1)
    n_dt = torch.cross(dt[..., 1] - dt[..., 0], dt[..., 2] - dt[..., 0], dim=1)
    l_gt = n_gt.norm(2, dim=1, keepdim=True)
    l_dt = n_dt.norm(2, dim=1, keepdim=True)
    l_gt = l_gt + (l_gt == 0).to(torch.float32) * 0.01
    l_dt = l_dt + (l_dt == 0).to(torch.float32) * 0.01
    loss = (n_gt / l_gt - n_dt / l_dt).abs().sum(1)
    if select:
        loss = torch.sort(loss)[0][int(loss.numel() * 0.25):]
    return loss.mean()


def vnl_select_index(H, W, sample_ratio=0.15):
    """criteria.py:912-932: the exact sequence of draws from the GLOBAL numpy RNG."""
    import numpy as np
    num = H * W
    out = []
    for _ in range(3):
        p = np.random.choice(num, int(num * sample_ratio), replace=True)
        np.random.shuffle(p)
        out.append(p)
    return out


def model_loss(pred_depth, pred_logit, depth_bins, depth_gt, weight, p123, fx, fy, diff_loss_weight):
    """criteria.py:1047-1062."""
    return wcel(pred_logit, depth_bins, depth_gt, weight) + diff_loss_weight * vnl(depth_gt, pred_depth, p123, fx, fy)


# ---------------------------------------------------------------- B8 companion  modules/vnl.py:202-230
def bins_to_depth(depth_bin, border):
    """[b, c, h, w] probabilities -> [b, 1, h, w]: 10 ** sum_c p_c * border_c."""
    d = (depth_bin.permute(0, 2, 3, 1) * border.to(torch.float32)).sum(3, dtype=torch.float32, keepdim=True)
    return (10 ** d).permute(0, 3, 1, 2)


def depth_to_bins(depth, depth_min, depth_max, C):
    """Returns (bins int32, rewritten depth) — the reference mutates `depth` in place (clamp; invalid -> -1)."""
    import numpy as np
    depth = depth.clone()
    invalid = depth < 0.
    depth[depth < depth_min] = depth_min
    depth[depth > depth_max] = depth_max
    dmin_log = np.log10(depth_min)
    interval = (np.log10(depth_max) - dmin_log) / C
    bins = ((torch.log10(depth) - dmin_log) / interval).to(torch.int)
    bins[invalid] = C + 1
    bins[bins == C] = C - 1
    depth[invalid] = -1.0
    return bins, depth


def ord_loss(ord_labels, target):
    """ordLoss.forward (criteria.py:744-787): plane index k against the (float) SID label by type promotion."""
    N, C, H, W = ord_labels.shape
    K = torch.arange(C, dtype=torch.int32).view(1, C, 1, 1).expand(N, C, H, W)
    m0, m1 = (K <= target), (K > target)
    s = torch.sum(torch.log(torch.clamp(ord_labels[m0], min=1e-8, max=1e8))) + \
        torch.sum(torch.log(torch.clamp(1.0 - ord_labels[m1], min=1e-8, max=1e8)))
    return s / (-(N * H * W))


def sid_labels(depth, alpha, beta, ord_num):
    """DORNModule.depth_to_label (modules/dorn.py:102-107, SID): the FLOAT label map the module hands to ordLoss."""
    a, b, k = torch.tensor(float(alpha)), torch.tensor(float(beta)), torch.tensor(int(ord_num)).int()
    return k * torch.log(depth / a) / torch.log(b / a)


def sid_depth(label, alpha, beta, ord_num):
    """DORNModule.label_to_depth (modules/dorn.py:95-100, SID)."""
    a, b, k = torch.tensor(float(alpha)), torch.tensor(float(beta)), torch.tensor(int(ord_num)).int()
    return torch.exp(torch.log(a) + torch.log(b / a) * label / k)
