"""End-to-end GPU parity of the HIP VNL network (mono_depth_estimation_amd.network.VNL.MetricDepthModel)
against the CPU oracle (oracle/nets.py: vnl_forward, pinned to the reference's own network/VNL.py by
tests/golden/vnl_net.npz) and against the reference's golden values directly.

Tolerances (bf16 MFMA path vs fp32): eval-mode depth (bins_to_depth of the softmax) within 2 % relative L2 and AbsRel of
that depth within 2e-3 of the reference's; train-mode ModelLoss within 2 %; parameter-gradient norms within 15 % for
90 % of the tensors (measured: 10th..90th percentile of the ratio 0.977..1.027), direction cosine >= 0.95 on decoder tensors
(measured 0.99-0.999) and >= 0.85 in the trunk (measured 0.93-0.97); running statistics
within 2 %."""
import numpy as np
import pytest
import torch

from oracle import losses as L
from oracle import nets
from oracle import weights as W

pytestmark = pytest.mark.gpu
SIZE = (64, 96)


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def setup():
    from mono_depth_estimation_amd.network import VNL
    params = nets.vnl_params()
    torch.manual_seed(0)
    net = VNL.MetricDepthModel(params)
    sd = W.vnl_fixture_state(net, 41)
    rgb, tgt = W.synthetic_batch(41, 2, *SIZE)
    P = nets.leaf_state(sd)
    with torch.no_grad():
        nets.vnl_forward(P, rgb, True, momentum=1.0)
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    return net.cuda(), params, P, rgb, tgt


def test_vnl_eval_against_oracle_and_reference(setup, golden):
    net, params, P, rgb, tgt = setup
    g = golden("vnl_net")
    border = torch.tensor(params.depth_bin_border, dtype=torch.float32)
    net.eval()
    with torch.no_grad():
        logit, prob = net(rgb.cuda())
        lo, po = nets.vnl_forward(P, rgb, False)
        lq, pq = nets.vnl_forward(P, rgb, False, q=nets.bf16_round)      # the oracle rounding to bf16 where the HIP path stores
    assert logit.shape == (2, 150, *SIZE) and prob.shape == logit.shape and logit.dtype == torch.float32
    assert torch.allclose(prob.sum(1), torch.ones(2, *SIZE, device="cuda"), atol=1e-5)
    depth, depth_o, depth_q = (L.bins_to_depth(p, border) for p in (prob.cpu(), po, pq))
    # Rounding the fp32 ORACLE's own activations to bf16 moves its logits by `noise` (1.4e-2 on this fixture: 70 layers,
    # res5 and the ASPP see 4 x 6 maps).  The HIP path must sit inside that noise, and close to the rounding oracle.
    noise = _rel(lq, lo)
    e_fp32, e_q = _rel(logit.cpu(), lo), _rel(logit.cpu(), lq)
    print("VNL eval: logits HIP vs fp32 oracle %.3e, vs bf16-rounding oracle %.3e, rounding noise of the oracle %.3e" % (e_fp32, e_q, noise))
    assert noise < 2e-2 and e_fp32 < 1.5 * noise + 5e-3, (e_fp32, noise)
    assert e_q < 1.2 * noise + 5e-3, (e_q, noise)
    assert _rel(depth, depth_o) < 1.5 * _rel(depth_q, depth_o) + 1e-2
    assert _rel(depth, torch.from_numpy(g["eval_depth"])) < 1.5 * _rel(depth_q, depth_o) + 1e-2
    t = tgt.clamp(min=0)
    m = t > 0
    absrel = lambda d: float(((d - t).abs() / t.clamp(min=1e-9))[m].mean())
    a_ref, a_hip, a_q = absrel(torch.from_numpy(g["eval_depth"])), absrel(depth), absrel(depth_q)
    print("VNL eval AbsRel: reference %.5f, HIP %.5f, bf16-rounding oracle %.5f" % (a_ref, a_hip, a_q))
    assert abs(a_hip - a_ref) < 2.0 * abs(a_q - a_ref) + 1e-3


def test_vnl_train_step_against_oracle_and_reference(setup, golden):
    from mono_depth_estimation_amd import criteria
    net, params, P0, rgb, tgt = setup
    g = golden("vnl_net")
    net.train()
    net.zero_grad(set_to_none=True)
    x = rgb.cuda()
    logit, prob = net(x)
    border = torch.tensor(params.depth_bin_border, dtype=torch.float32)
    gt, bins, p123 = torch.from_numpy(g["gt"]), torch.from_numpy(g["bins"]), torch.from_numpy(g["p123"]).long()
    # the loss through the ORACLE's loss code on the HIP outputs (fp32 tensors): isolates the network's parity
    lg, pr = logit.detach().cpu().requires_grad_(True), prob.detach().cpu().requires_grad_(True)
    loss = L.model_loss(L.bins_to_depth(pr, border), lg, bins, gt, L.wcel_weight(150), p123, 519.0, 519.0, 6)
    loss.backward()
    with torch.no_grad():      # what storage rounding alone does to the oracle's training-mode loss (batch statistics of 2 x 4 x 6 maps)
        lq, pq = nets.vnl_forward(nets.leaf_state(P0), rgb, True, q=nets.bf16_round)
        loss_q = float(L.model_loss(L.bins_to_depth(pq, border), lq, bins, gt, L.wcel_weight(150), p123, 519.0, 519.0, 6))
    ref_loss = float(g["train_loss"])
    print("VNL train ModelLoss: reference %.4f, HIP %.4f, bf16-rounding oracle %.4f" % (ref_loss, float(loss), loss_q))
    assert abs(float(loss) - ref_loss) < 2.0 * abs(loss_q - ref_loss) + 5e-3 * ref_loss, (float(loss), loss_q, ref_loss)
    torch.autograd.backward([logit, prob], [lg.grad.cuda(), pr.grad.cuda()])
    # oracle gradients from the same state
    P = nets.leaf_state(P0, requires_grad=True)
    lo, po = nets.vnl_forward(P, rgb, True)
    L.model_loss(L.bins_to_depth(po, border), lo, bins, gt, L.wcel_weight(150), p123, 519.0, 519.0, 6).backward()
    named = dict(net.named_parameters())
    ratios, cosines = [], {}
    for k, p in named.items():
        go, gh = P[k].grad, p.grad.detach().cpu()
        assert gh.shape == go.shape and torch.isfinite(gh).all(), k
        if float(go.norm()) > 1e-8:
            ratios.append(float(gh.norm() / go.norm()))
            cosines[k] = float((gh * go).sum() / (gh.norm() * go.norm() + 1e-30))
    ratios = np.array(ratios)
    print("VNL train gradient-norm ratios HIP / oracle, percentiles 1 10 50 90 99:", np.percentile(ratios, [1, 10, 50, 90, 99]))
    print("cosines:", {k.split("modules.")[-1]: round(v, 3) for k, v in cosines.items() if "conv" in k and ("predict" in k or "fcn5" in k or "res2.0" in k or "aspp" in k)})
    assert np.mean(np.abs(ratios - 1) < 0.15) >= 0.9, np.percentile(ratios, [1, 10, 50, 90, 99])
    d = "depth_model.decoder_modules."
    for k in (d + "topdown_predict.conv1.weight", d + "topdown_predict.conv1.bias", d + "topdown_fcn5.ftb.conv3.weight",
              d + "topdown_fcn4.afa_block.conv2.weight", d + "topdown_fcn1.ftb_block.conv1.weight", d + "top.0.weight"):
        assert cosines[k] >= 0.95, (k, cosines[k])
    e = "depth_model.encoder_modules."
    # (not asserted: bottomup_top.globalpool_conv1x1 -- its BatchNorm normalises over the 2 images of the batch, so its
    #  normalised output is +-1 whatever the conv computes and the weight gradient is numerically zero on both sides)
    for k in (e + "bottomup_top.aspp_conv3_2.weight", e + "bottomup_top.aspp_conv1x1.weight", e + "bottomup.res5.2.conv2.weight",
              e + "bottomup.res2.0.conv2.weight", e + "topdown_lateral_modules.3.lateral.conv2.weight"):
        assert cosines[k] >= 0.85, (k, cosines[k])
    # FTB's conv2 bias sits in front of a train-mode BatchNorm: its gradient is zero (up to fp32 noise in the oracle)
    kb = d + "topdown_fcn5.ftb.conv2.bias"
    assert float(named[kb].grad.abs().max()) == 0.0 and float(P[kb].grad.abs().max()) < 1e-4
    sd = net.state_dict()
    for key, ref in (("depth_model.encoder_modules.bottomup.res5.2.bn3.running_mean", g["rm_res5"]),
                     ("depth_model.encoder_modules.bottomup_top.globalpool_bn.running_var", g["rv_aspp"])):
        assert _rel(sd[key].cpu(), torch.from_numpy(ref)) < 2e-2, key


def test_vnl_module_path_with_the_hip_criteria_and_sgd(setup):
    """modules/vnl.py:252-260,289-326: ModelLoss on (bins_to_depth(softmax), logits) + SGD momentum 0.9, through the drop-in
    criteria and the fused flat-range SGD; the loss goes down over a few steps."""
    from mono_depth_estimation_amd import criteria
    net, params, _, rgb, tgt = setup
    params.crop_size = SIZE
    crit = criteria.ModelLoss(params)
    x, gt = rgb.cuda(), tgt.cuda().clone()
    bins = criteria.depth_to_bins(gt, params.depth_min, 1.1, params.dec_out_c)
    net.train()
    losses = []
    for _ in range(4):
        np.random.seed(5)                         # VNL_Loss draws its point triples from numpy's global stream: same triples each step
        net.zero_grad(set_to_none=True)
        logit, prob = net(x)
        loss = crit(criteria.bins_to_depth(prob, params.depth_bin_border), logit, bins, gt)
        loss.backward()
        net._store.sgd_step(1e-4, 1e-5, momentum=0.9, weight_decay=5e-4)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def _vnl_trajectories(weight, steps, with_rounding_oracle):
    """`steps` SGD steps (modules/vnl.py:289-326: momentum 0.9, weight decay 5e-4, encoder at a tenth of the decoder's rate) on one
    batch from the fixture state with ModelLoss = WCEL + weight x virtual-normal loss: the HIP path (drop-in ModelLoss + fused
    flat-range SGD), the fp32 functional oracle (its own loss code + torch.optim.SGD) and, optionally, the same oracle with
    its activations rounded to bf16 where the HIP path stores them.  numpy's stream is re-seeded before every draw of
    point triples, so all of them see the same triples."""
    import copy
    from mono_depth_estimation_amd import criteria
    from mono_depth_estimation_amd.network import VNL
    params = nets.vnl_params()
    params.crop_size, params.diff_loss_weight = SIZE, weight
    torch.manual_seed(0)
    net = VNL.MetricDepthModel(params)
    sd = W.vnl_fixture_state(net, 41)
    rgb, tgt = W.synthetic_batch(41, 2, *SIZE)
    P = nets.leaf_state(sd, requires_grad=True)
    with torch.no_grad():
        nets.vnl_forward(P, rgb, True, momentum=1.0)
    start = {k: v.detach().clone() for k, v in P.items()}
    net.load_state_dict(copy.deepcopy(start))
    net = net.cuda().train()
    lr_e, lr_d = 5e-5, 5e-4
    crit = criteria.ModelLoss(params)
    x, gt_h = rgb.cuda(), tgt.cuda().clone()
    bins_h = criteria.depth_to_bins(gt_h, params.depth_min, 1.1, params.dec_out_c)
    lh = []
    for _ in range(steps):
        np.random.seed(5)
        net.zero_grad(set_to_none=True)
        logit, prob = net(x)
        loss = crit(criteria.bins_to_depth(prob, params.depth_bin_border), logit, bins_h, gt_h)
        loss.backward()
        net._store.sgd_step(lr_e, lr_d, momentum=0.9, weight_decay=5e-4)
        lh.append(float(loss))
    border = torch.tensor(params.depth_bin_border, dtype=torch.float32)
    bins, gt = L.depth_to_bins(tgt.clone(), params.depth_min, 1.1, params.dec_out_c)

    def oracle_run(P, q):
        enc = [v for k, v in P.items() if v.requires_grad and ".encoder_modules." in k]
        dec = [v for k, v in P.items() if v.requires_grad and ".encoder_modules." not in k]
        opt = torch.optim.SGD([{"params": enc, "lr": lr_e}, {"params": dec, "lr": lr_d}], lr=lr_d, momentum=0.9, weight_decay=5e-4)
        out = []
        for _ in range(steps):
            np.random.seed(5)
            p123 = torch.from_numpy(np.stack(L.vnl_select_index(*SIZE))).long()
            opt.zero_grad()
            lg, pr = nets.vnl_forward(P, rgb, True, q=q)
            loss = L.model_loss(L.bins_to_depth(pr, border), lg, bins, gt, L.wcel_weight(150), p123, 519.0, 519.0, weight)
            loss.backward()
            opt.step()
            out.append(float(loss))
        return np.array(out)
    lo = oracle_run(P, None)
    lq = oracle_run(nets.leaf_state(start, requires_grad=True), nets.bf16_round) if with_rounding_oracle else None
    return net, P, rgb, tgt, x, border, np.array(lh), lo, lq


def test_vnl_loss_curves_agree_with_the_oracle():
    """Convergence parity for VNL, in two parts, because the virtual-normal loss is itself ill-conditioned under storage
    rounding: rounding the ORACLE's activations to bf16 changes d(VNL_Loss) / d(prob) by 36 % in norm (cosine 0.93) on this
    state -- unit normals of point triples on a nearly flat predicted surface, an |.| of their difference, a sort that drops
    the lowest quarter (measured for prediction scales x1 ... x30: cosine 0.87-0.93 throughout).
    (a) WCEL alone (ModelLoss with diff_loss_weight 0: same network, same optimiser, a smooth loss): 20 SGD steps, the HIP
        and the fp32 oracle's curves within 1 % at every step, both fall; on the state the ORACLE reached -- weights off the
        16-bit grid -- the AbsRel of the decoded depth agrees to 1e-4 + the activation-rounding shift of the oracle (the eval
        forward's two-term weight shadow; the one-term figure is printed beside it).
    (b) the configured loss (WCEL + 6 x VNL): 12 steps, three trajectories (HIP, fp32 oracle, the oracle with its activations
        rounded to bf16).  REPORTED, not bounded by a ratio: two runs of identical code gave a largest HIP gap to the fp32
        oracle of 21 % and of 39 % (float atomics feeding the ill-conditioned loss), the rounding oracle 23 % and 20 % (its
        host's thread count changes the summation order) -- the spread between runs is as large as the quantity.  Asserted:
        every trajectory is finite, falls (the rounding oracle's: gets below its start), and stays within 60 % of the fp32
        oracle's at every step."""
    net, P, rgb, tgt, x, border, lh, lo, _ = _vnl_trajectories(0, 20, False)
    print("WCEL only, HIP   :", np.round(lh[[0, 1, 2, 4, 9, 14, 19]], 4))
    print("WCEL only, oracle:", np.round(lo[[0, 1, 2, 4, 9, 14, 19]], 4))
    band = np.abs(lh - lo) / lo
    print("relative gap between the curves: max %.4f mean %.4f; fall HIP %.4f oracle %.4f" % (band.max(), band.mean(), lh[-1] / lh[0], lo[-1] / lo[0]))
    assert np.isfinite(lh).all() and lh[-1] < 0.98 * lh[0] and lo[-1] < 0.98 * lo[0]
    assert band.max() < 1e-2 and band.mean() < 4e-3
    trained = {k: v.detach().clone() for k, v in P.items()}
    net.load_state_dict(trained)
    net.eval()
    with torch.no_grad():
        dh = L.bins_to_depth(net(x)[1].cpu(), border)
        net._store.split_eval = False
        dh1 = L.bins_to_depth(net(x)[1].cpu(), border)          # the one-term weight shadow the training step uses
        net._store.split_eval = True
        do = L.bins_to_depth(nets.vnl_forward(trained, rgb, False)[1], border)
        # free-running SGD leaves weights that are not representable in 16 bits; a path that convolves with their rounded copies
        # carries an error that is the same for every pixel and does not average out of AbsRel the way activation rounding
        # does -- the oracles that stand for such a path round conv weights AND activations.  The HIP path's EVAL forward
        # contracts with a two-term shadow (FlatStore.ensure_split), so it answers to the activation-rounding oracles alone.
        rw = lambda f: {k: (f(v) if v.dtype.is_floating_point and v.dim() >= 2 else v) for k, v in trained.items()}
        d16 = L.bins_to_depth(nets.vnl_forward(rw(nets.fp16_round), rgb, False, q=nets.fp16_round)[1], border)
        dbf = L.bins_to_depth(nets.vnl_forward(rw(nets.bf16_round), rgb, False, q=nets.bf16_round)[1], border)
        from mono_depth_estimation_amd import _lib
        to = torch.float16 if _lib.ACT_NAME == "fp16" else torch.bfloat16
        dacts = [L.bins_to_depth(nets.vnl_forward(trained, rgb, False, q=nets.rounding_draw(k, to))[1], border) for k in range(3)]
    t = tgt.clamp(min=0)
    m = t > 0
    absrel = lambda d: float(((d - t).abs() / t.clamp(min=1e-9))[m].mean())
    s16, sbf, ship, ship1 = (abs(absrel(d) - absrel(do)) for d in (d16, dbf, dh, dh1))
    sact = max(abs(absrel(d) - absrel(do)) for d in dacts)
    print("trained state, eval AbsRel: HIP %.6f oracle %.6f" % (absrel(dh), absrel(do)))
    print("AbsRel shift of the ORACLE under 16-bit storage: fp16 weights + activations %.2e, bf16 weights + activations %.2e; %s "
          "activations alone (three realisations) up to %.2e.  The HIP path (%s): %.2e with the two-term eval shadow, %.2e with the one-term one"
          % (s16, sbf, _lib.ACT_NAME, sact, _lib.ACT_NAME, ship, ship1))
    # identical weights, off the 16-bit grid: within 1e-4 + what activation rounding alone does to the oracle (round 3, one-term
    # shadow: 7.8e-4 on the bf16 build, 1.07e-4 on the fp16 build)
    assert ship <= 1e-4 + sact, (ship, sact)
    assert sbf <= 2e-3 and s16 <= sbf + 1e-4
    # (b)
    _, _, _, _, _, _, lh, lo, lq = _vnl_trajectories(6, 12, True)
    print("WCEL + 6 VNL, HIP            :", np.round(lh[[0, 1, 2, 4, 7, 11]], 4))
    print("WCEL + 6 VNL, oracle         :", np.round(lo[[0, 1, 2, 4, 7, 11]], 4))
    print("WCEL + 6 VNL, rounding oracle:", np.round(lq[[0, 1, 2, 4, 7, 11]], 4))
    gap_h, gap_q = np.abs(lh - lo) / lo, np.abs(lq - lo) / lo
    print("gap to the fp32 oracle: HIP max %.3f mean %.3f; rounding oracle max %.3f mean %.3f" % (gap_h.max(), gap_h.mean(), gap_q.max(), gap_q.mean()))
    # (the rounding oracle is a yardstick, not the product: its curve has ended a step above its start on one run -- 18.41 after
    # 16.40, from 18.28 -- so it is held to "gets below its start", the two real trajectories to "end below it")
    assert np.isfinite(lh).all() and np.isfinite(lq).all() and lh[-1] < lh[0] and lo[-1] < lo[0] and lq.min() < lq[0]
    assert gap_h.max() < 0.6 and gap_q.max() < 0.6


def test_criterion_fused_head_equals_the_public_route(setup):
    """The private route between the head and ModelLoss (criteria._FusedDepthFunction / _FusedWcelFunction -> mde_vnl_head_*: the
    criterion reads the head's 16-bit input instead of the fp32 logits / softmax it was handed, its backward writes d(input) in
    one launch) against the public route through the same tensors (bins_to_depth, WCEL_Loss, softmax-head backward): the same loss
    to 1e-5, the same parameter gradients to the rounding of d(input) -- in deterministic order of everything else -- and a
    derived tensor (a clone) falls back to the public route by itself."""
    from mono_depth_estimation_amd import criteria
    net, params, P0, rgb, tgt = setup
    net.load_state_dict({k: v.clone() for k, v in P0.items()})
    net.train()
    x = rgb.cuda()
    gt = tgt.clone().cuda()
    gt[:, :, :, :4] = -1.0
    params.crop_size = SIZE
    crit = criteria.ModelLoss(params)
    bins = criteria.depth_to_bins(gt, params.depth_min, 1.1, params.dec_out_c)

    def step(fused, clone=False):
        criteria._FUSE_HEAD = fused
        try:
            np.random.seed(5)
            net.zero_grad(set_to_none=True)
            logit, prob = net(x)
            if clone:
                logit, prob = logit.clone(), prob.clone()
            loss = crit(criteria.bins_to_depth(prob, params.depth_bin_border), logit, bins, gt)
            loss.backward()
            return float(loss), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        finally:
            criteria._FUSE_HEAD = True
    step(True)                                      # (the first backward of a plan traces its fused BatchNorm sums)
    l_pub, g_pub = step(False)
    _, g_pub2 = step(False)                         # the public route again: what two passes of identical code differ by (float atomics)
    l_fus, g_fus = step(True)
    l_cln, g_cln = step(True, clone=True)
    print("ModelLoss: public route %.6f, private route %.6f, private route refused for clones %.6f" % (l_pub, l_fus, l_cln))
    assert abs(l_fus - l_pub) <= 2e-5 * abs(l_pub) and abs(l_cln - l_pub) <= 2e-5 * abs(l_pub)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    noise = {k: rel(g_pub2[k], g) for k, g in g_pub.items()}
    diff = {k: rel(g_fus[k], g) for k, g in g_pub.items()}
    # The kernels' arithmetic is pinned in tests/test_vnl_gpu.py::test_criterion_fused_head_kernels; here the two routes' d(input)
    # differ by one 16-bit rounding, which the network's backward amplifies tensor by tensor like any other rounding (and the
    # BatchNorm-bias gradients that are exact zeros in exact arithmetic -- a per-channel constant in front of another train-mode
    # BatchNorm -- are residues on either route): the bulk is compared, not the tail
    d = np.array([diff[k] for k in g_pub if float(g_pub[k].norm()) > 1e-9])
    print("parameter gradients, private vs public route: rel. difference median %.2e, 90 %% %.2e (two public passes: median %.2e)" % (
        float(np.median(d)), float(np.quantile(d, 0.9)), float(np.median(list(noise.values())))))
    assert float(np.median(d)) <= 3e-2 and float(np.quantile(d, 0.9)) <= 0.15
    k = "depth_model.decoder_modules.topdown_predict.conv1.bias"
    # (the private route takes the bias gradient as the column sums of the 16-bit d(input) it has just written, the public one sums
    #  the fp32 values before they are rounded: 12 K roundings of up to half an ulp each per channel, 0.6 % of the largest entry here)
    assert torch.allclose(g_fus[k], g_pub[k], rtol=1e-2, atol=1.5e-2 * float(g_pub[k].abs().max())), "head bias gradient"
