#!/usr/bin/env python3
"""Layer-by-layer comparison of the HIP FCRN engine with the CPU oracle (run on the GPU box):
    python tests/diag_fcrn.py [N H W]
Prints, per activation, the relative L2 error (bf16 path vs fp32 oracle) forward and backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fcrn as ofcrn  # noqa: E402
from oracle import losses as OL  # noqa: E402
from oracle import weights as W  # noqa: E402
from mono_depth_estimation_amd.network import FCRN  # noqa: E402
from mono_depth_estimation_amd import criteria  # noqa: E402


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def emulate_bf16(ora):
    """Round where the HIP path rounds: GEMM conv weights, every conv output, every ReLU /
    block output (and, through autograd of the casts, the matching gradients)."""
    rnd = lambda mod, inp, out: out.to(torch.bfloat16).float()
    for name, mod in ora.named_modules():
        if isinstance(mod, torch.nn.Conv2d):
            if name not in ("conv1", "conv3"):
                mod.weight.data = mod.weight.data.to(torch.bfloat16).float()
            if name != "conv3":
                mod.register_forward_hook(rnd)
        elif isinstance(mod, (torch.nn.ReLU, ofcrn.UpProjModule)) or name == "bn2":
            mod.register_forward_hook(rnd)


def main():
    emu = "--emulate" in sys.argv
    if emu:
        sys.argv.remove("--emulate")
    N, H, Wd = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 96, 128)
    torch.manual_seed(0)
    ora = ofcrn.FCRNOracle(50, (H, Wd), out_channels=1)
    sd = W.fcrn_fixture_state(ora, 5)
    rgb, tgt = W.synthetic_batch(5, N, H, Wd)
    hip = FCRN.ResNet(layers=50, output_size=(H, Wd), out_channels=1, pretrained=False)
    hip.load_state_dict(sd)
    hip = hip.cuda()
    if emu:
        emulate_bf16(ora)
    ora.train()
    hip.train()

    acts = {}

    def hook(name):
        def f(mod, inp, out):
            out.retain_grad()
            acts[name] = out
        return f
    ora.maxpool.register_forward_hook(hook("pool"))
    for li in (1, 2, 3, 4):
        for bi, blk in enumerate(getattr(ora, "layer%d" % li)):
            blk.register_forward_hook(hook("layer%d.%d" % (li, bi)))
    ora.bn2.register_forward_hook(hook("bn2"))
    for li in (1, 2, 3, 4):
        getattr(ora.upSample, "layer%d" % li).register_forward_hook(hook("up%d" % li))
    # ---------------- teacher-forced, layer-isolated comparison (immune to chaotic amplification)
    ins = {}

    def in_hook(name):
        def f(mod, inp):
            ins[name] = inp[0]
        return f
    for li in (1, 2, 3, 4):
        for bi, blk in enumerate(getattr(ora, "layer%d" % li)):
            blk.register_forward_pre_hook(in_hook("layer%d.%d" % (li, bi)))
    ora.conv2.register_forward_pre_hook(in_hook("bn2"))
    for li in (1, 2, 3, 4):
        getattr(ora.upSample, "layer%d" % li).register_forward_pre_hook(in_hook("up%d" % li))
    ora.zero_grad()
    y_o = ora(rgb)
    for t in ins.values():
        t.retain_grad()
    loss_o = OL.silog(y_o, tgt)
    loss_o.backward()
    with torch.no_grad():
        hip(rgb.cuda())
    eng = next(iter(hip._engines.values()))
    names = ["layer%d.%d" % (li, bi) for li in (1, 2, 3, 4) for bi in range(len(getattr(ora, "layer%d" % li)))]
    names += ["bn2", "up1", "up2", "up3", "up4"]
    omods = dict(ora.named_modules())
    hparams = dict(hip.named_parameters())

    def to_dev(t):
        return t.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    print("teacher-forced   %-10s %10s %10s %10s" % ("layer", "fwd", "dx", "worst dW"))
    for nm, L in zip(names, eng.layers):
        L.x.t.copy_(to_dev(ins[nm]))
        eng.reset_sums()               # (a single layer outside forward(): the fused-finalize launches leave their sums to the other pass)
        L.fwd(True)
        e_f = rel(nchw(L.out.t), acts[nm].detach())
        eng.store.G.zero_()
        L.reset_grad_flags()
        L.x.gw = False
        L.out.g.copy_(to_dev(acts[nm].grad))
        L.bwd()
        e_b = rel(nchw(L.x.g), ins[nm].grad)
        prefix = {"bn2": None}.get(nm, nm if nm.startswith("layer") else "upSample.layer" + nm[2:])
        worst = (0.0, "")
        for k, q in ora.named_parameters():
            if (prefix and k.startswith(prefix + ".")) or (nm == "bn2" and (k.startswith("conv2.") or k.startswith("bn2."))):
                worst = max(worst, (rel(hparams[k]._mde_grad.cpu(), q.grad), k))
        print("                 %-10s %10.3e %10.3e %10.3e %s" % (nm, e_f, e_b, worst[0], worst[1]))
    torch.cuda.synchronize()
    return
    y_h = hip(rgb.cuda())
    loss_h = criteria.silog_loss(0.85)(y_h, tgt.cuda())
    loss_h.backward()
    torch.cuda.synchronize()
    eng = next(iter(hip._engines.values()))
    names = ["layer%d.%d" % (li, bi) for li in (1, 2, 3, 4) for bi in range(len(getattr(ora, "layer%d" % li)))]
    names += ["bn2", "up1", "up2", "up3", "up4"]
    print("%-12s %10s %10s" % ("activation", "fwd relL2", "bwd relL2"))
    print("%-12s %10.3e %10.3e" % ("pool", rel(nchw(eng.pool.t), acts["pool"].detach()), rel(nchw(eng.pool.g), acts["pool"].grad)))
    for nm, L in zip(names, eng.layers):
        print("%-12s %10.3e %10.3e" % (nm, rel(nchw(L.out.t), acts[nm].detach()), rel(nchw(L.out.g), acts[nm].grad)))
    print("output  max|diff| %.3e  relL2 %.3e" % (float((y_h.cpu() - y_o).abs().max()), rel(y_h.detach().cpu(), y_o.detach())))
    print("silog   hip %.6f  oracle %.6f" % (float(loss_h), float(loss_o)))
    worst = []
    for (k, p), (_, q) in zip(hip.named_parameters(), ora.named_parameters()):
        worst.append((rel(p.grad.cpu(), q.grad), k))
    worst.sort(reverse=True)
    print("param grads: median relL2 %.3e; worst:" % worst[len(worst) // 2][0])
    for e, k in worst[:8]:
        print("   %-50s %.3e" % (k, e))
    for (k, p), (_, q) in zip(hip.state_dict().items(), ora.state_dict().items()):
        if k in ("bn1.running_mean", "bn1.running_var", "layer4.2.bn3.running_var", "upSample.layer4.bottom_branch.batchnorm.running_var"):
            print("   %-50s %.3e" % (k, rel(p.cpu().float(), q.float())))


if __name__ == "__main__":
    main()
