"""CPU fp32 oracle of the stdepth composite criterion — TEST INFRASTRUCTURE ONLY.

Restates reference modules/base_module.py:124-208 (`BaseModule.setup_criterion` -> `_loss`) and the
stdepth_utils.py helpers it calls (depth_sort :4-17, composite_layers :19-42, separable-Gaussian SSIM
:63-140).  Pinned by tests/golden/stdepth.npz, minted by running the reference's own function body.
"""
import torch
import torch.nn.functional as F

from . import losses as L


def depth_sort(layers):
    """stdepth_utils.py:4-17: [B, L, C, H, W] sorted along L by the LAST channel (stable, detached keys)."""
    _, idx = torch.sort(layers[:, :, -1].detach(), dim=1, stable=True)
    return torch.stack([layers[:, :, i].gather(1, idx) for i in range(layers.shape[2])], 2)


def composite_layers(layers):
    """stdepth_utils.py:19-42: front-to-back "over" of [B, L, 4+, H, W].  NB the first layer's colour enters
    un-premultiplied (acc_rgb = rgb_0, not a_0 * rgb_0), as the reference writes it."""
    rgb, a = layers[:, 0, :3], layers[:, 0, 3:4]
    for i in range(1, layers.shape[1]):
        rgb = rgb + (1.0 - a) * layers[:, i, 3:4] * layers[:, i, :3]
        a = a + (1.0 - a) * layers[:, i, 3:4]
    return torch.clamp(torch.cat([rgb, a], 1), 0.0, 1.0)


def _gauss(x, win):
    """filter_gaussian_separated, dim=2 (stdepth_utils.py:63-78): zero-padded, per channel, W then H."""
    g, p = x.shape[1], win.numel() // 2
    w = win.view(1, 1, 1, -1).expand(g, 1, 1, -1)
    out = F.conv2d(x, w, groups=g, padding=(0, p))
    return F.conv2d(out, w.transpose(2, 3), groups=g, padding=(p, 0))


def dssim2d_map(pred, targ, win_size=11, sigma=1.5):
    """1 - SSIM map (stdepth_utils.py:81-118,139): data_range 1, K = (0.01, 0.03), cs clamped at 0."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    c = torch.arange(win_size) - win_size // 2
    win = torch.exp(-c ** 2 / (2 * sigma ** 2))
    win = (win / win.sum()).to(pred.dtype)
    mu1, mu2 = _gauss(pred, win), _gauss(targ, win)
    s1 = _gauss(pred * pred, win) - mu1 * mu1
    s2 = _gauss(targ * targ, win) - mu2 * mu2
    s12 = _gauss(pred * targ, win) - mu1 * mu2
    cs = torch.relu((2 * s12 + C2) / (s1 + s2 + C2))
    return 1.0 - ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs


def stdepth_loss(pred, targ, rgba, loss, single_layer=True, variance_focus=0.85, depth_w=10.0, comp_w=2.0,
                 fbdiv_w=0.2, ssim_w=2.0):
    """base_module.py:132-206.  Returns (total, pred_full or None, dict of terms).  The term selection is by
    SUBSTRING of `loss`, exactly as the reference does."""
    mask1 = rgba[:, [3]] > 0.0
    mask4 = mask1.expand(-1, 4, -1, -1)
    mask8 = mask1.expand(-1, 8, -1, -1)
    maskN = mask1.expand(-1, targ.size(1), -1, -1)
    dsl = slice(8, 10) if single_layer else slice(16, 20)
    maskD = targ[:, dsl] > 0.0
    out = {}

    def silog(p, t):
        return torch.nan_to_num(L.silog(p, t, variance_focus))
    if single_layer:
        targ_full = rgba
        pred_full = composite_layers(torch.stack([pred[:, :4], pred[:, 4:8]], 1))
    else:
        targ_full = torch.cat([rgba, targ[:, [19]]], 1)
        ls = [torch.cat([pred[:, 4 * i:4 * i + 4], pred[:, [16 + i]]], 1) for i in range(3)]
        srt = depth_sort(torch.stack(ls, 1))[:, :, :4]
        pred_full = composite_layers(torch.cat([srt, pred[:, 12:16].unsqueeze(1)], 1))
    if 'silma' in loss:
        out['depth_silog'] = depth_w * torch.nan_to_num(silog(pred[:, dsl][maskD], targ[:, dsl][maskD]))
        out['color_mae'] = F.l1_loss(pred[:, :8][mask8], targ[:, :8][mask8])
    if 'silms' in loss:
        out['depth_silog'] = depth_w * torch.nan_to_num(silog(pred[:, dsl][maskD], targ[:, dsl][maskD]))
        out['color_mse'] = F.mse_loss(pred[:, :8][mask8], targ[:, :8][mask8])
    if 'mse' in loss:
        out['all_mse'] = F.mse_loss(pred[maskN], targ[maskN]) + depth_w * F.mse_loss(pred[:, dsl][maskD], targ[:, dsl][maskD])
    if 'mae' in loss:
        out['all_mae'] = F.l1_loss(pred[maskN], targ[maskN]) + depth_w * F.l1_loss(pred[:, dsl][maskD], targ[:, dsl][maskD])
    if 'allssim' in loss:
        out['all_ssim'] = ssim_w * dssim2d_map(pred.clamp(0, 1), targ.clamp(0, 1))[maskN].mean()
    if 'colorssim' in loss:
        out['front_ssim'] = ssim_w * dssim2d_map(pred[:, :4].clamp(0, 1), targ[:, :4].clamp(0, 1))[mask4].mean()
        out['back_ssim'] = ssim_w * dssim2d_map(pred[:, 4:8].clamp(0, 1), targ[:, 4:8].clamp(0, 1))[mask4].mean()
    if 'composite' in loss:
        # (pred_full of the multi-layer case has 4 channels, targ_full 5: the reference's mask4 indexing needs
        #  matching shapes, so 'composite' is only well-formed single-layer; restated as written)
        comp = comp_w * F.mse_loss(pred_full[mask4], targ_full[mask4], reduction='none')
        out['composite_mse'] = torch.mean(torch.nan_to_num(comp))
        if 'ssim' in loss:
            out['composite_ssim'] = ssim_w * comp_w * dssim2d_map(pred_full.clamp(0, 1), targ_full.clamp(0, 1))[mask4].mean()
    if 'fbdivergence' in loss:
        n = torch.linalg.vector_norm
        fpbg = n(pred[:, :3], dim=1, keepdim=True) * n(targ[:, 4:7], dim=1, keepdim=True) + 1e-3
        fgbp = n(pred[:, 4:7], dim=1, keepdim=True) * n(targ[:, :3], dim=1, keepdim=True) + 1e-3
        fb = ((pred[:, :3] * targ[:, 4:7] / fpbg).sum(1) + (pred[:, 4:7] * targ[:, :3] / fgbp).sum(1))[mask1.squeeze(1)]
        out['fb_divergence'] = fbdiv_w * fb.mean()
    return torch.stack(list(out.values())).sum(), pred_full, out
